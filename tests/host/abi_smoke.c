/* abi_smoke.c -- the FFI's view of include/uzu_hip.h: a plain C program (gcc -std=c99, no Python, no C++) that includes the
 * header, links libuzu_hip.so and drives  Context::new -> create_buffer -> XxxKernel::new -> command buffer
 * (start_encoding -> encode -> end_encoding -> submit -> wait_until_completed) -> download  for MatmulKernel (int4 ScaleBias)
 * and Normalization (RMS norm + residual protocol), against constants whose expected bits are computed here in integer /
 * exactly representable arithmetic -- every partial sum is a multiple of 0.5 below 2^11, so any summation order gives the same
 * f32 and the comparison is bit-exact.  What a Rust `backends/hip` shim would call through bindgen.
 *
 *   gcc -std=c99 -Iinclude tests/host/abi_smoke.c -Luzu_amd/lib -luzu_hip -Wl,-rpath,$PWD/uzu_amd/lib -lm -o abi_smoke
 *   ./abi_smoke                  (exit 0 = "abi_smoke ok"; needs a GPU)
 *   ./abi_smoke --no-gpu         (exit 0 when context creation FAILS loudly with a message: the no-GPU contract)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "uzu_hip.h"

#define CHECK(expr)                                                                                   \
    do {                                                                                              \
        uzu_status st_ = (expr);                                                                      \
        if (st_ != UZU_OK) {                                                                          \
            fprintf(stderr, "%s:%d: %s -> status %d: %s\n", __FILE__, __LINE__, #expr, (int)st_, uzu_hip_last_error()); \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

static uint16_t bf16_from_f32(float f) { /* half 2.7 bf16::from_f32: round to nearest even */
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float f32_from_bf16(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

enum { M = 3, N = 64, K = 128, GROUP = 64, G = K / GROUP, D = 256 };

static int run_matmul(uzu_hip_context* ctx) {
    /* A[m][k] = (m + 1) * (k % 2 ? -1 : 2)      (bf16-exact small integers)
     * code[n][k] = (n + 3 k) % 16, scale[n][g] = 0.5 * (1 + g), bias[n][g] = -(float)(n % 4)
     * D[m][n] = sum_k A[m][k] * (scale * code + bias)  + out_bias[n],   out_bias[n] = n % 8 */
    static uint16_t a[M * K], scales[N * G], biases[N * G], out_bias[N], got[M * N];
    static uint8_t codes[N * K / 2];
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) a[m * K + k] = bf16_from_f32((float)((m + 1) * ((k & 1) ? -1 : 2)));
    for (int n = 0; n < N; ++n) {
        for (int k = 0; k < K; k += 2) {
            const unsigned lo = (unsigned)(n + 3 * k) % 16u, hi = (unsigned)(n + 3 * (k + 1)) % 16u; /* low nibble = even k (kernel.rs:236-242) */
            codes[n * (K / 2) + k / 2] = (uint8_t)(lo | (hi << 4));
        }
        for (int g = 0; g < G; ++g) {
            scales[n * G + g] = bf16_from_f32(0.5f * (float)(1 + g));
            biases[n * G + g] = bf16_from_f32(-(float)(n % 4));
        }
        out_bias[n] = bf16_from_f32((float)(n % 8));
    }
    uzu_hip_buffer *ba, *bw, *bs, *bb, *bo, *bd;
    CHECK(uzu_hip_buffer_create(ctx, sizeof a, &ba));
    CHECK(uzu_hip_buffer_create(ctx, sizeof codes, &bw));
    CHECK(uzu_hip_buffer_create(ctx, sizeof scales, &bs));
    CHECK(uzu_hip_buffer_create(ctx, sizeof biases, &bb));
    CHECK(uzu_hip_buffer_create(ctx, sizeof out_bias, &bo));
    CHECK(uzu_hip_buffer_create(ctx, sizeof got, &bd));
    CHECK(uzu_hip_buffer_upload(ba, 0, a, sizeof a));
    CHECK(uzu_hip_buffer_upload(bw, 0, codes, sizeof codes));
    CHECK(uzu_hip_buffer_upload(bs, 0, scales, sizeof scales));
    CHECK(uzu_hip_buffer_upload(bb, 0, biases, sizeof biases));
    CHECK(uzu_hip_buffer_upload(bo, 0, out_bias, sizeof out_bias));

    uzu_hip_kernel* mk;
    CHECK(uzu_hip_matmul_create(ctx, UZU_BF16, UZU_BF16, UZU_BF16, &mk));
    uzu_matmul_arguments args;
    memset(&args, 0, sizeof args);
    args.a.buffer = ba;
    args.b_kind = UZU_MATMUL_B_SCALE_BIAS;
    args.b.buffer = bw, args.scales.buffer = bs, args.biases.buffer = bb;
    args.mode = UZU_QMODE_U4, args.group_size = GROUP, args.b_transpose = 1;
    args.d.buffer = bd, args.ab_scale = 1.0f, args.bias.buffer = bo;
    args.m = M, args.n = N, args.k = K;

    uzu_hip_cmdbuf* cb;
    CHECK(uzu_hip_cmdbuf_create(ctx, "abi_smoke.matmul", UZU_CMDBUF_EAGER, &cb));
    /* typestate: encode before start_encoding must be refused (command_buffer.rs: Initial has no encode) */
    if (uzu_hip_matmul_encode(mk, cb, &args) != UZU_ERR_STATE) {
        fprintf(stderr, "matmul_encode on an Initial command buffer was not refused with UZU_ERR_STATE\n");
        return 1;
    }
    CHECK(uzu_hip_cmdbuf_start_encoding(cb));
    CHECK(uzu_hip_cmdbuf_push_debug_group(cb, "matmul"));
    CHECK(uzu_hip_matmul_encode(mk, cb, &args));
    CHECK(uzu_hip_cmdbuf_pop_debug_group(cb));
    CHECK(uzu_hip_cmdbuf_end_encoding(cb));
    CHECK(uzu_hip_cmdbuf_submit(cb));
    CHECK(uzu_hip_cmdbuf_wait_until_completed(cb));
    uint64_t ns = 0;
    CHECK(uzu_hip_cmdbuf_gpu_execution_time_ns(cb, &ns));
    CHECK(uzu_hip_buffer_download(bd, 0, got, sizeof got));

    int bad = 0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double acc = 0.0; /* exact: multiples of 0.25 below 2^13 */
            for (int k = 0; k < K; ++k) {
                const int g = k / GROUP;
                const double w = 0.5 * (1 + g) * (double)((n + 3 * k) % 16) - (double)(n % 4);
                acc += (double)((m + 1) * ((k & 1) ? -1 : 2)) * w;
            }
            /* kernel.rs:281-292: value = ab_scale * acc (+ bias) in f32, one rounding to bf16 at the store */
            const uint16_t want = bf16_from_f32((float)acc + (float)(n % 8));
            if (want != got[m * N + n]) {
                if (bad++ < 5) fprintf(stderr, "matmul[%d][%d]: got %g (0x%04x), want %g (0x%04x)\n", m, n, f32_from_bf16(got[m * N + n]), got[m * N + n], f32_from_bf16(want), want);
            }
        }
    uzu_hip_cmdbuf_destroy(cb);
    uzu_hip_kernel_destroy(mk);
    uzu_hip_buffer_destroy(ba), uzu_hip_buffer_destroy(bw), uzu_hip_buffer_destroy(bs), uzu_hip_buffer_destroy(bb), uzu_hip_buffer_destroy(bo), uzu_hip_buffer_destroy(bd);
    if (bad) {
        fprintf(stderr, "matmul: %d of %d outputs differ\n", bad, M * N);
        return 1;
    }
    printf("matmul int4 ScaleBias %dx%dx%d: bit-exact (%llu ns)\n", M, N, K, (unsigned long long)ns);
    return 0;
}

static int run_normalization(uzu_hip_context* ctx) {
    /* residual_add + copy_to_shortcut RMS norm (normalization.rs:56-125): x = input + shortcut (bf16), shortcut = x,
     * out = x * rms_inv * (scale + offset).  input = +-1, shortcut = +-3 (same sign) => x = +-4, mean(x^2) = 16, eps = 0 =>
     * rms_inv = 0.25 exactly; scales[i] = (i % 4) * 0.5, offset = 1 => out = +-(1 + 0.5 (i % 4)): every step exact. */
    enum { ROWS = 2 };
    static uint16_t in[ROWS * D], sc[ROWS * D], out[ROWS * D], sc_out[ROWS * D];
    static float scales[D];
    for (int r = 0; r < ROWS; ++r)
        for (int i = 0; i < D; ++i) {
            const float sgn = ((i + r) % 3 == 0) ? -1.0f : 1.0f;
            in[r * D + i] = bf16_from_f32(sgn), sc[r * D + i] = bf16_from_f32(3.0f * sgn);
        }
    for (int i = 0; i < D; ++i) scales[i] = 0.5f * (float)(i % 4);
    uzu_hip_buffer *bi, *bs, *bo, *bsc;
    CHECK(uzu_hip_buffer_create(ctx, sizeof in, &bi));
    CHECK(uzu_hip_buffer_create(ctx, sizeof scales, &bs));
    CHECK(uzu_hip_buffer_create(ctx, sizeof out, &bo));
    CHECK(uzu_hip_buffer_create(ctx, sizeof sc, &bsc));
    CHECK(uzu_hip_buffer_upload(bi, 0, in, sizeof in));
    CHECK(uzu_hip_buffer_upload(bs, 0, scales, sizeof scales));
    CHECK(uzu_hip_buffer_upload(bsc, 0, sc, sizeof sc));
    uzu_hip_kernel* nk;
    /* new(InputT, AffineT, OutputT, AccumT, in_place, subtract_mean, full_layer, copy_to_shortcut, residual_add, use_hadamard,
     *     scale_residual_sum, scale_output, has_biases, has_scales) */
    CHECK(uzu_hip_normalization_create(ctx, UZU_BF16, UZU_F32, UZU_BF16, UZU_F32, 0, 0, 1, 1, 1, 0, 0, 0, 0, 1, &nk));
    uzu_hip_cmdbuf* cb;
    CHECK(uzu_hip_cmdbuf_create(ctx, "abi_smoke.normalization", UZU_CMDBUF_GRAPH, &cb)); /* captured into a hipGraph, submitted twice */
    CHECK(uzu_hip_cmdbuf_start_encoding(cb));
    uzu_buf none = {NULL, 0}, b_in = {bi, 0}, b_s = {bs, 0}, b_o = {bo, 0}, b_sc = {bsc, 0};
    CHECK(uzu_hip_normalization_encode(nk, cb, b_in, b_s, none, b_o, b_sc, none, ROWS, D, 0.0f, 1.0f, 1.0f));
    CHECK(uzu_hip_cmdbuf_end_encoding(cb));
    CHECK(uzu_hip_cmdbuf_submit(cb));
    CHECK(uzu_hip_cmdbuf_wait_until_completed(cb));
    CHECK(uzu_hip_buffer_download(bo, 0, out, sizeof out));
    CHECK(uzu_hip_buffer_download(bsc, 0, sc_out, sizeof sc_out));
    int bad = 0;
    for (int r = 0; r < ROWS; ++r)
        for (int i = 0; i < D; ++i) {
            const float sgn = ((i + r) % 3 == 0) ? -1.0f : 1.0f;
            const uint16_t want = bf16_from_f32(sgn * (1.0f + 0.5f * (float)(i % 4))), want_sc = bf16_from_f32(4.0f * sgn);
            if (out[r * D + i] != want || sc_out[r * D + i] != want_sc) {
                if (bad++ < 5) fprintf(stderr, "normalization[%d][%d]: out 0x%04x (want 0x%04x), shortcut 0x%04x (want 0x%04x)\n", r, i, out[r * D + i], want, sc_out[r * D + i], want_sc);
            }
        }
    uzu_hip_cmdbuf_destroy(cb);
    uzu_hip_kernel_destroy(nk);
    uzu_hip_buffer_destroy(bi), uzu_hip_buffer_destroy(bs), uzu_hip_buffer_destroy(bo), uzu_hip_buffer_destroy(bsc);
    if (bad) {
        fprintf(stderr, "normalization: %d of %d outputs differ\n", bad, ROWS * D);
        return 1;
    }
    printf("normalization (residual add + shortcut copy, %d x %d): bit-exact\n", ROWS, D);
    return 0;
}

int main(int argc, char** argv) {
    uzu_hip_context* ctx = NULL;
    const uzu_status st = uzu_hip_context_create(0, &ctx);
    if (argc > 1 && strcmp(argv[1], "--no-gpu") == 0) {
        /* the product must fail loudly without a GPU: an error status AND a message, never a silent CPU path */
        if (st == UZU_OK) {
            printf("abi_smoke --no-gpu: a GPU is present (context created); nothing to check\n");
            uzu_hip_context_destroy(ctx);
            return 0;
        }
        const char* msg = uzu_hip_last_error();
        if (!msg || !msg[0]) {
            fprintf(stderr, "context_create failed with status %d but no message\n", (int)st);
            return 1;
        }
        printf("abi_smoke --no-gpu: context_create refused (status %d): %s\n", (int)st, msg);
        return 0;
    }
    if (st != UZU_OK) {
        fprintf(stderr, "uzu_hip_context_create: status %d: %s\n", (int)st, uzu_hip_last_error());
        return 1;
    }
    char name[128];
    CHECK(uzu_hip_context_device_name(ctx, name, sizeof name));
    uint32_t caps = 0;
    CHECK(uzu_hip_context_device_capabilities(ctx, &caps));
    printf("device: %s (capabilities 0x%x)\n", name, caps);
    if (run_matmul(ctx) || run_normalization(ctx)) return 1;
    size_t peak = 0;
    CHECK(uzu_hip_context_peak_memory_usage(ctx, &peak));
    uzu_hip_context_destroy(ctx);
    printf("abi_smoke ok (peak device memory %zu bytes)\n", peak);
    return 0;
}
