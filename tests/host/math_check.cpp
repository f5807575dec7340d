// Host-side validation of uzu_amd/csrc/uzu_math.h against the system libm (glibc).
// usage: math_check <stride>   -> prints mismatch counts for expf / logf over every `stride`-th float
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "../../uzu_amd/csrc/uzu_math.h"
int main(int argc, char** argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 97;
    unsigned long long n_exp = 0, bad_exp = 0, n_log = 0, bad_log = 0;
    for (uint64_t u = 0; u <= 0xFFFFFFFFull; u += stride) {
        const float x = uzu::bits_to_f32((uint32_t)u);
        if (x != x) continue;
        {
            const float a = expf(x), b = uzu::expf_glibc(x);
            ++n_exp;
            if (uzu::f32_to_bits(a) != uzu::f32_to_bits(b)) { if (bad_exp < 5) printf("exp mismatch x=%a sys=%a ours=%a\n", x, a, b); ++bad_exp; }
        }
        if (x > 0) {
            const float a = logf(x), b = uzu::logf_glibc(x);
            ++n_log;
            if (uzu::f32_to_bits(a) != uzu::f32_to_bits(b)) { if (bad_log < 5) printf("log mismatch x=%a sys=%a ours=%a\n", x, a, b); ++bad_log; }
        }
    }
    printf("expf: %llu checked, %llu mismatches\nlogf: %llu checked, %llu mismatches\n", n_exp, bad_exp, n_log, bad_log);
    return 0;
}
