// kbench.cpp -- per-kernel timing of the fused decode kernels in a captured graph (what the engine replays).
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/kbench.cpp -Luzu_amd/lib -luzu_hip -Wl,-rpath,$PWD/uzu_amd/lib -o tools/kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <array>
#include <vector>
#include <algorithm>
#include "../uzu_amd/csrc/kernels.h"
#include "../uzu_amd/csrc/kernels_decode.h"
using namespace uzu::k;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
extern "C" const char* uzu_hip_last_error(void);
static hipStream_t s;
template <class T> T* dalloc(size_t n, int fill = 0x11) { T* p; CK(hipMalloc(&p, n * sizeof(T) + 256)); CK(hipMemset(p, fill, n * sizeof(T))); return p; }
static double time_graph(const char* name, size_t bytes, int reps, const std::function<uzu_status(int)>& launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; ++i) if (launch(i) != UZU_OK) { printf("%s: launch failed: %s\n", name, uzu_hip_last_error()); exit(1); }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s)); for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (5.0 * reps);
    printf("%-34s %8.2f us/launch  %8.1f GB/s\n", name, us, bytes / us / 1e3);
    return us;
}
int main(int argc, char** argv) {
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int NBUF0 = getenv("KB_NBUF") ? atoi(getenv("KB_NBUF")) : 8; const int reps = NBUF0 > 64 ? NBUF0 : 64; // rotate weight buffers so weights come from HBM, not L2/MALL
    auto bench_gemv = [&](const char* name, uint32_t n, uint32_t k, bool norm, bool act, bool amax, uint32_t n2 = 0, uint32_t debug = 0) {
        const uint32_t g = 128, groups = k / g;
        const uint32_t bits = getenv("KB_BITS") ? atoi(getenv("KB_BITS")) : 4;
        const int NBUF = (size_t)n * k > (64u << 20) ? (NBUF0 < 3 ? NBUF0 : 3) : NBUF0;
        std::vector<uint8_t*> w(NBUF); std::vector<uint16_t*> sc(NBUF), bi(NBUF);
        const uint32_t nt = n + n2;
        for (int i = 0; i < NBUF; ++i) { w[i] = dalloc<uint8_t>((size_t)nt * k * bits / 8, 0x53); sc[i] = dalloc<uint16_t>((size_t)nt * groups, 0x3c); bi[i] = dalloc<uint16_t>((size_t)nt * groups, 0x3c); }
        uint16_t* x = dalloc<uint16_t>(k, 0x3f); uint16_t* sh = dalloc<uint16_t>(k, 0x3f); uint16_t* sho = dalloc<uint16_t>(k);
        float* ns = dalloc<float>(k, 0); uint16_t* out = dalloc<uint16_t>(nt); float* pv = dalloc<float>(8192); uint32_t* pi = dalloc<uint32_t>(8192);
        const size_t bytes = (size_t)nt * k * bits / 8 + (size_t)nt * groups * 4 + nt * 2 + k * 2;
        time_graph(name, bytes, reps, [&](int i) {
            DecGemvParams p{};
            p.w[0] = w[i % NBUF], p.scales[0] = sc[i % NBUF], p.biases[0] = bi[i % NBUF], p.out[0] = out, p.n[0] = n;
            if (n2) { p.w[1] = w[i % NBUF] + (size_t)n * k * bits / 8, p.scales[1] = sc[i % NBUF] + (size_t)n * groups, p.biases[1] = bi[i % NBUF] + (size_t)n * groups, p.out[1] = out + n, p.n[1] = n2; }
            p.k = k, p.bits = bits, p.group_size = g, p.b_kind = UZU_MATMUL_B_SCALE_BIAS, p.x = x;
            if (norm) { p.norm_scales = ns, p.norm_eps = 1e-6f, p.norm_offset = 1.f, p.norm_full_layer = 1, p.residual_add = 1, p.shortcut_in = sh, p.shortcut_out = sho; }
            if (act) p.act_mul = 1;
            if (amax) p.part_val = pv, p.part_idx = pi;
            return gemv_dec(s, p, cus, nullptr);
        });
    };
    if (getenv("KB_GEMM_AB")) { // 256-thread form against the ping-pong form (UZU_GEMM_PP), same box, random data: time + byte comparison of the outputs
        struct Shape { uint32_t m, n, k, g, bits; int gated; };
        std::vector<Shape> shapes = {{2048, 8224, 1024, 128, 4, 0}, {2048, 7168, 1024, 128, 4, 1}, {2048, 1024, 3584, 128, 4, 0}, {2048, 1024, 2048, 128, 4, 0}, {2048, 5120, 1024, 128, 4, 0},
                                     {4096, 14336, 4096, 128, 4, 0}, {4096, 28672, 4096, 128, 4, 1}, {4096, 4096, 14336, 128, 4, 0}, {4096, 6144, 4096, 128, 4, 0}, {4096, 4096, 4096, 128, 4, 0},
                                     {300, 520, 2048, 64, 4, 0}, {1000, 520, 2048, 256, 4, 0}, {4096, 14336, 4096, 64, 8, 0}, {2048, 7168, 1024, 128, 8, 1}};
        if (getenv("KB_GEMM_AB_SHORT")) shapes = {{2048, 7168, 1024, 128, 4, 1}, {2048, 1024, 3584, 128, 4, 0}, {4096, 14336, 4096, 128, 4, 0}, {4096, 4096, 14336, 128, 4, 0}};
        if (getenv("KB_GEMM_AB_ONE")) shapes = {{4096, 14336, 4096, 128, 4, 0}};
        uint64_t rs = 0x9E3779B97F4A7C15ull;
        auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 32); };
        for (const Shape& sh : shapes) {
            const uint32_t m = sh.m, n = sh.n, k = sh.k, g = sh.g, G = k / g;
            const size_t wb = (size_t)n * k * sh.bits / 8;
            std::vector<uint8_t> hw(wb); for (auto& v : hw) v = (uint8_t)rnd();
            std::vector<uint16_t> hs((size_t)n * G), hb((size_t)n * G), hx((size_t)m * k);
            for (auto& v : hs) v = (uint16_t)(0x3c00 + (rnd() & 0xff));             // scales ~ 0.0078 .. 0.0156
            for (auto& v : hb) v = (uint16_t)(0xbd00 + (rnd() & 0xff) + ((rnd() & 1) << 15)); // biases ~ +-0.03 .. 0.06
            for (auto& v : hx) v = (uint16_t)(0x3e00 + (rnd() & 0x1ff) + ((rnd() & 1) << 15)); // activations ~ +-0.125 .. 1
            uint8_t* w = dalloc<uint8_t>(wb); uint16_t* sc = dalloc<uint16_t>(hs.size()); uint16_t* bi = dalloc<uint16_t>(hb.size()); uint16_t* x = dalloc<uint16_t>(hx.size());
            CK(hipMemcpy(w, hw.data(), wb, hipMemcpyHostToDevice)); CK(hipMemcpy(sc, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
            CK(hipMemcpy(bi, hb.data(), hb.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
            const size_t on = (size_t)m * (sh.gated ? n / 2 : n);
            const int alt = getenv("KB_GEMM_FORM") ? atoi(getenv("KB_GEMM_FORM")) : 1; // the form compared with the 256-thread one
            uint16_t* out[2] = {dalloc<uint16_t>(on, 0), dalloc<uint16_t>(on, 0)};
            MatmulParams p{}; p.a = x, p.b = w, p.scales = sc, p.biases = bi; p.w_dt = p.a_dt = p.d_dt = UZU_BF16; p.b_kind = UZU_MATMUL_B_SCALE_BIAS;
            p.bits = sh.bits, p.group_size = g, p.ab_scale = 1.f, p.m = m, p.n = n, p.k = k; p.act_mul = sh.gated; p.act_type = 0;
            if (!gemm_q_mfma128_supported(p, cus)) { printf("%ux%ux%u: not a large-tile shape\n", m, n, k); continue; }
            void* gemm_ws = dalloc<uint8_t>(gemm_q_mfma128_workspace_bytes(p, cus) + 65536);
            double us[2];
            for (int pp = 0; pp < 2; ++pp) {
                setenv("UZU_HIP_TUNE", pp ? (alt == 2 ? "gemm_form=2" : "gemm_form=1") : "gemm_form=0", 1);
                char name[96]; snprintf(name, sizeof name, "gemm %ux%ux%u g%u int%u%s %s", m, n, k, g, sh.bits, sh.gated ? " +act" : "", pp ? (alt == 2 ? "wave-spec" : "ping-pong") : "256-thread");
                p.d = out[pp];
                us[pp] = time_graph(name, wb + (size_t)m * k * 2 + on * 2, 8, [&](int) { return gemm_q_mfma128(s, p, cus, gemm_ws); });
            }
            std::vector<uint16_t> h0(on), h1(on);
            CK(hipMemcpy(h0.data(), out[0], on * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), out[1], on * 2, hipMemcpyDeviceToHost));
            size_t diff = 0, nz = 0; for (size_t i = 0; i < on; ++i) diff += h0[i] != h1[i], nz += (h0[i] & 0x7fff) != 0;
            printf("    -> %.1f vs %.1f TFLOP/s (x%.3f); outputs differing: %zu of %zu (%zu non-zero)\n", 2.0 * m * n * k / us[0] / 1e6, 2.0 * m * n * k / us[1] / 1e6, us[0] / us[1], diff, on, nz);
            if (getenv("KB_GEMM_PP_TIMING")) { // a library built with -DUZU_GEMM_PP_TIMING: mean shader cycles per k-step and wave in each phase
                const size_t slots = 65536 * 9;
                unsigned long long* d = dalloc<unsigned long long>(slots * 8, 0);
                uzu::k::g_gemm128_dbg = d; p.d = out[1];
                gemm_q_mfma128(s, p, cus, gemm_ws); CK(hipStreamSynchronize(s));
                uzu::k::g_gemm128_dbg = nullptr;
                std::vector<unsigned long long> h(slots * 8); CK(hipMemcpy(h.data(), d, slots * 64, hipMemcpyDeviceToHost));
                double sum[2][8] = {}; double steps[2] = {0, 0};
                for (size_t i = 65536; i < slots; ++i) { const unsigned long long* o = &h[i * 8]; if (!o[4]) continue; const int hf = (int)o[5] & 1; steps[hf] += (double)o[4]; for (int q = 0; q < 4; ++q) sum[hf][q] += (double)o[q]; sum[hf][4] += (double)(o[5] >> 8), sum[hf][5] += (double)o[6], sum[hf][6] += (double)(o[7] & 0xffffffffull), sum[hf][7] += (double)(o[7] >> 32); }
                for (int hf = 0; hf < 2; ++hf) if (steps[hf] > 0)
                    printf(alt == 2 ? "    timing %d (0 = consumers: [2] work [3] barrier; 1 = producers: [0] work [1] barrier): per k-step cycles  %.0f  %.0f  %.0f  %.0f\n" : "    timing half %d: per k-step cycles  convert %.0f  barrier %.0f  mfma %.0f  barrier %.0f\n", hf, sum[hf][0] / steps[hf], sum[hf][1] / steps[hf], sum[hf][2] / steps[hf], sum[hf][3] / steps[hf]),
                    printf("        convert phase: fold: entry + hazard wait %.0f  FMAs %.0f  scales %.0f;  weight loads %.0f  convert %.0f\n", sum[hf][6] / steps[hf], sum[hf][7] / steps[hf], sum[hf][4] / steps[hf], sum[hf][5] / steps[hf], sum[hf][0] / steps[hf]);
                CK(hipFree(d));
            }
            for (void* q : {(void*)w, (void*)sc, (void*)bi, (void*)x, (void*)out[0], (void*)out[1], gemm_ws}) CK(hipFree(q));
        }
        return 0;
    }
    if (getenv("KB_GEMM")) { // prefill GEMM on the matrix cores (k_gemm.hip)
        std::vector<std::array<uint32_t, 3>> gemm_shapes{{1024, 8224, 1024}, {1024, 7168, 1024}, {1024, 1024, 3584}, {1024, 1024, 2048}, {1024, 3072, 1024}, {4096, 14336, 4096}};
        if (const char* e = getenv("KB_GEMM_SHAPES")) { // "m,n,k;m,n,k;..."
            gemm_shapes.clear();
            for (const char* q = e; *q;) { unsigned a = 0, b = 0, c = 0; int used = 0; if (sscanf(q, "%u,%u,%u%n", &a, &b, &c, &used) != 3) break; gemm_shapes.push_back({a, b, c}); q += used; if (*q == ';') ++q; }
        }
        for (auto sh : gemm_shapes) {
            const uint32_t m = sh[0], n = sh[1], k = sh[2], g = 128;
            uint8_t* w = dalloc<uint8_t>((size_t)n * k / 2, 0x53); uint16_t* sc = dalloc<uint16_t>((size_t)n * k / g, 0x3c); uint16_t* bi = dalloc<uint16_t>((size_t)n * k / g, 0x3c);
            uint16_t* x = dalloc<uint16_t>((size_t)m * k, 0x3f); uint16_t* out = dalloc<uint16_t>((size_t)m * n);
            void* gemm_ws = dalloc<uint8_t>((size_t)m * 2048 * 2 + (size_t)n * (k / g) * 4 + (size_t)4 * m * n * 4 + 65536);
            char name[64]; snprintf(name, sizeof name, "gemm_q_mfma %ux%ux%u", m, n, k);
            const double us = time_graph(name, (size_t)n * k / 2 + (size_t)m * k * 2 + (size_t)m * n * 2, 8, [&](int) {
                MatmulParams p{}; p.a = x, p.b = w, p.scales = sc, p.biases = bi, p.d = out; p.w_dt = p.a_dt = p.d_dt = UZU_BF16; p.b_kind = UZU_MATMUL_B_SCALE_BIAS;
                p.bits = 4, p.group_size = g, p.ab_scale = 1.f, p.m = m, p.n = n, p.k = k;
                if (!getenv("KB_GEMM64") && gemm_q_mfma128_supported(p, cus)) return gemm_q_mfma128(s, p, cus, gemm_ws); // preallocated: no pool allocation while capturing
                return matmul(s, p, cus); });
            printf("    -> %.1f TFLOP/s\n", 2.0 * m * n * k / us / 1e6);
            if (getenv("KB_GEMM_DBG")) { // one instrumented launch: phase durations per workgroup
                const size_t slots = 65536;
                unsigned long long* d = dalloc<unsigned long long>(slots * 8, 0);
                uzu::k::g_gemm128_dbg = d;
                MatmulParams p{}; p.a = x, p.b = w, p.scales = sc, p.biases = bi, p.d = out; p.w_dt = p.a_dt = p.d_dt = UZU_BF16; p.b_kind = UZU_MATMUL_B_SCALE_BIAS;
                p.bits = 4, p.group_size = g, p.ab_scale = 1.f, p.m = m, p.n = n, p.k = k;
                gemm_q_mfma128(s, p, cus, gemm_ws); CK(hipStreamSynchronize(s));
                uzu::k::g_gemm128_dbg = nullptr;
                std::vector<unsigned long long> h(slots * 8); CK(hipMemcpy(h.data(), d, slots * 64, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, t1 = 0; double ph[3] = {0, 0, 0}; int cnt = 0;
                for (size_t i = 0; i < slots; ++i) { const unsigned long long* o = &h[i * 8]; if (!o[0]) continue; ++cnt; t0 = o[0] < t0 ? o[0] : t0; t1 = o[3] > t1 ? o[3] : t1; for (int q = 0; q < 3; ++q) ph[q] += (double)(o[q + 1] - o[q]); }
                printf("    dbg: %d workgroups, span %.1f us; mean us: prologue %.2f  main loop %.2f  offset term + epilogue %.2f\n", cnt, (t1 - t0) * 0.01,
                       ph[0] / cnt * 0.01, ph[1] / cnt * 0.01, ph[2] / cnt * 0.01);
                // start-time histogram: how many rounds
                std::vector<double> st; for (size_t i = 0; i < slots; ++i) if (h[i * 8]) st.push_back((h[i * 8] - t0) * 0.01);
                std::sort(st.begin(), st.end()); printf("    dbg: start times us: p10 %.1f p50 %.1f p90 %.1f max %.1f\n", st[st.size() / 10], st[st.size() / 2], st[st.size() * 9 / 10], st.back());
            }
        }
        return 0;
    }
    if (getenv("KB_LLAMA")) { // Llama-3-8B decode shapes (bandwidth regime)
        bench_gemv("L8 qkv 6144x4096 +norm", 6144, 4096, true, false, false);
        bench_gemv("L8 out 4096x4096", 4096, 4096, false, false, false);
        bench_gemv("L8 up+act 28672x4096 +norm", 28672, 4096, true, true, false);
        bench_gemv("L8 down 4096x14336", 4096, 14336, false, false, false);
        bench_gemv("L8 readout 128256x4096 +norm", 128256, 4096, true, false, true);
        bench_gemv("Q readout 248320x1024 +norm", 248320, 1024, true, false, true);
        return 0;
    }
    bench_gemv("gemv_dec in_proj 8224x1024 +norm", 8224, 1024, true, false, false);
    bench_gemv("gemv_dec in_proj 8224x1024", 8224, 1024, false, false, false);
    bench_gemv("gemv_dec out_proj 1024x2048", 1024, 2048, false, false, false);
    bench_gemv("gemv_dec up+act 7168x1024 +norm", 7168, 1024, true, true, false);
    bench_gemv("gemv_dec down 1024x3584", 1024, 3584, false, false, false);
    bench_gemv("gemv_dec qkv+gate 5120x1024 +norm", 3072, 1024, true, false, false, 2048);
    bench_gemv("gemv_dec readout 248320x1024 +norm", 248320, 1024, true, false, true);
    {   // unfused reference: plain GEMV kernel through k::matmul
        const uint32_t n = 248320, k = 1024, g = 128;
        uint8_t* w = dalloc<uint8_t>((size_t)n * k / 2, 0x53); uint16_t* sc = dalloc<uint16_t>((size_t)n * 8, 0x3c); uint16_t* bi = dalloc<uint16_t>((size_t)n * 8, 0x3c);
        uint16_t* x = dalloc<uint16_t>(k, 0x3f); uint16_t* out = dalloc<uint16_t>(n);
        time_graph("gemv_q (unfused) 248320x1024", (size_t)n * k / 2 + n * 32, 16, [&](int) {
            MatmulParams p{}; p.a = x, p.b = w, p.scales = sc, p.biases = bi, p.d = out; p.w_dt = p.a_dt = p.d_dt = UZU_BF16; p.b_kind = UZU_MATMUL_B_SCALE_BIAS;
            p.bits = 4, p.group_size = g, p.ab_scale = 1.f, p.m = 1, p.n = n, p.k = k; return matmul(s, p, cus); });
    }
    {   // delta_dec, Qwen3.5-0.8B shape
        const int NBUF = NBUF0;
        const uint32_t Hv = 16, Dk = 128, Dv = 128, key_dim = 2048, value_dim = 2048, conv_dim = 6144, total = conv_dim + value_dim + 32;
        uint16_t* in_proj = dalloc<uint16_t>(total, 0x3c);
        float* al = dalloc<float>(Hv, 0); float* dt = dalloc<float>(Hv, 0); float* o = dalloc<float>(value_dim); float* sz = dalloc<float>(value_dim);
        std::vector<float*> st(NBUF); for (auto& p : st) p = dalloc<float>((size_t)Hv * Dv * Dk, 0);
        time_graph("delta_dec 16x128x128", (size_t)2 * Hv * Dv * Dk * 4, reps, [&](int i) {
            DeltaDecParams p{}; p.in_proj = in_proj, p.a_log = al, p.dt_bias = dt, p.state = st[i % NBUF], p.o = o, p.sz = sz;
            p.num_v_heads = Hv, p.num_k_heads = 16, p.head_v_dim = Dv, p.key_dim = key_dim, p.value_dim = value_dim; return delta_dec(s, p); });
        // in-proj with the conv epilogue and out-proj with the norm-gate prologue
        const uint32_t g = 128;
        uint8_t* w = dalloc<uint8_t>((size_t)8224 * 1024 / 2, 0x53); uint16_t* sc = dalloc<uint16_t>((size_t)8224 * 8, 0x3c); uint16_t* bi = dalloc<uint16_t>((size_t)8224 * 8, 0x3c);
        uint16_t* x = dalloc<uint16_t>(2048, 0x3f); uint16_t* sh = dalloc<uint16_t>(1024, 0x3f); uint16_t* sho = dalloc<uint16_t>(1024); float* ns = dalloc<float>(1024, 0);
        float* cw = dalloc<float>(conv_dim * 4, 0); float* cs = dalloc<float>(conv_dim * 3, 0); uint16_t* out = dalloc<uint16_t>(8224);
        time_graph("gemv_dec in_proj +norm +conv", (size_t)8224 * 1024 / 2, reps, [&](int) {
            DecGemvParams p{}; p.w[0] = w, p.scales[0] = sc, p.biases[0] = bi, p.out[0] = out, p.n[0] = 8224; p.k = 1024, p.bits = 4, p.group_size = g, p.b_kind = UZU_MATMUL_B_SCALE_BIAS, p.x = x;
            p.norm_scales = ns, p.norm_eps = 1e-6f, p.norm_offset = 1.f, p.norm_full_layer = 1, p.residual_add = 1, p.shortcut_in = sh, p.shortcut_out = sho;
            p.conv_w = cw, p.conv_state = cs, p.conv_dim = conv_dim, p.conv_ks = 4; return gemv_dec(s, p, cus, nullptr); });
        float* nw = dalloc<float>(Dv, 0);
        time_graph("gemv_dec out_proj +norm-gate", (size_t)1024 * 2048 / 2, reps, [&](int) {
            DecGemvParams p{}; p.w[0] = w, p.scales[0] = sc, p.biases[0] = bi, p.out[0] = out, p.n[0] = 1024; p.k = 2048, p.bits = 4, p.group_size = g, p.b_kind = UZU_MATMUL_B_SCALE_BIAS, p.x = x;
            p.dg_o = o, p.dg_sz = sz, p.dg_w = nw, p.dg_dv = Dv, p.dg_eps = 1e-6f; return gemv_dec(s, p, cus, nullptr); });
    }
    {   // attn_dec + merge, Qwen3.5-0.8B shape at ctx 2048
        const int NBUF = NBUF0;
        const uint32_t nq = 8, nkv = 2, hd = 256, ctx = 2048, splits = 128;
        uint16_t* qkv = dalloc<uint16_t>((nq + 2 * nkv) * hd, 0x3c); float* cosr = dalloc<float>((size_t)(ctx + 8) * 64, 0); float* sinr = dalloc<float>((size_t)(ctx + 8) * 64, 0);
        uint32_t* len = dalloc<uint32_t>(1, 0); uint32_t h = ctx; CK(hipMemcpy(len, &h, 4, hipMemcpyHostToDevice));
        float* qs = dalloc<float>(hd, 0); float* parts = dalloc<float>((size_t)nq * splits * hd); float* sums = dalloc<float>(nq * splits); float* maxs = dalloc<float>(nq * splits);
        uint16_t* gate = dalloc<uint16_t>(nq * hd, 0x3c); uint16_t* out = dalloc<uint16_t>(nq * hd);
        std::vector<uint16_t*> K(NBUF), V(NBUF); for (int i = 0; i < NBUF; ++i) { K[i] = dalloc<uint16_t>((size_t)(ctx + 8) * nkv * hd, 0x3c); V[i] = dalloc<uint16_t>((size_t)(ctx + 8) * nkv * hd, 0x3c); }
        for (uint32_t sp : {32u, 64u, 128u})
            time_graph(sp == 32 ? "attn_dec ctx2048 S=32" : sp == 64 ? "attn_dec ctx2048 S=64" : "attn_dec ctx2048 S=128", (size_t)2 * ctx * nkv * hd * 2, reps, [&](int i) {
                AttnDecParams p{}; p.qkv = qkv, p.keys = K[i % NBUF], p.values = V[i % NBUF], p.cosines = cosr, p.sines = sinr, p.ctx_len = len;
                p.q_norm = {1, 1, 1e-6f, 1.f, qs}; p.k_norm = {1, 1, 1e-6f, 1.f, qs}; p.num_heads = nq, p.gqa_factor = 4, p.head_dim = hd, p.rope_dim = 64, p.scale = 0.0625f;
                p.partials = parts, p.sums = sums, p.maxs = maxs, p.cache_rows = ctx + 8; return attn_dec(s, p, sp); });
        time_graph("attn_merge S=64", (size_t)nq * 64 * hd * 4, reps, [&](int) { return attn_merge(s, parts, sums, maxs, gate, out, nq, hd, 64); });
    }
    return 0;
}
