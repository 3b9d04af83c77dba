// div_exhaustive.hip -- proof by exhaustion for the division inside fast_tanh / fast_atanh (ft8_lib ldpc.c).
//
// Both rational functions are a / b with a = a(x), b = b(x) computed from ONE float x by a fixed sequence of
// IEEE operations, so "which rcp/fma chain returns the correctly rounded quotient" is a question about 2^32
// inputs, not 2^64: this program evaluates every float bit pattern x, forms a(x), b(x) exactly as decode.hip
// does (-ffp-contract=off), takes the compiler's IEEE division as the truth and counts, per candidate chain,
// the inputs of the fast path's domain on which the chain returns other bits.
//
// Domain of the fast path (decode.hip: guard_key): fast_tanh sees x == 0 or |x| >= 2^-82, and its result is
// overridden by the clamp for |x| > 4.97; fast_atanh sees P == 0 or 2^-59 <= |P| <= 1.0073^6 < 1.05.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o div_exhaustive div_exhaustive.hip && ./div_exhaustive
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

enum { kVariants = 6, kExamples = 16 };
static const char *kNames[kVariants] = {
    "v7_nr_2corr (round 2 chain: NR on the reciprocal, two residual corrections)",
    "v5a_nr_1corr (NR on the reciprocal, one residual correction)",
    "v5b_raw_2corr (raw v_rcp_f32, two residual corrections)",
    "v3_raw_1corr (raw v_rcp_f32, one residual correction)",
    "v4_q0fix_1corr (q0 = a*r0 refined by e0, one residual correction with raw r0)",
    "v1_mul_only (a * NR(r0), no correction)",
};

struct Result {
    unsigned long long in_domain[2];
    unsigned long long mismatch[2][kVariants];         // [function][variant], bitwise (a zero of either sign counts as equal)
    unsigned long long mismatch_signed_zero[2][kVariants];
    unsigned int nexamples[2][kVariants];
    uint32_t examples[2][kVariants][kExamples];
};

__device__ __forceinline__ float chain(int v, float a, float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    if (v == 0) {
        const float e0 = __builtin_fmaf(-b, r0, 1.0f);
        const float r1 = __builtin_fmaf(e0, r0, r0);
        const float q0 = a * r1;
        const float e1 = __builtin_fmaf(-b, q0, a);
        const float q1 = __builtin_fmaf(e1, r1, q0);
        const float e2 = __builtin_fmaf(-b, q1, a);
        return __builtin_fmaf(e2, r1, q1);
    } else if (v == 1) {
        const float e0 = __builtin_fmaf(-b, r0, 1.0f);
        const float r1 = __builtin_fmaf(e0, r0, r0);
        const float q0 = a * r1;
        const float e1 = __builtin_fmaf(-b, q0, a);
        return __builtin_fmaf(e1, r1, q0);
    } else if (v == 2) {
        const float q0 = a * r0;
        const float e1 = __builtin_fmaf(-b, q0, a);
        const float q1 = __builtin_fmaf(e1, r0, q0);
        const float e2 = __builtin_fmaf(-b, q1, a);
        return __builtin_fmaf(e2, r0, q1);
    } else if (v == 3) {
        const float q0 = a * r0;
        const float e1 = __builtin_fmaf(-b, q0, a);
        return __builtin_fmaf(e1, r0, q0);
    } else if (v == 4) {
        const float e0 = __builtin_fmaf(-b, r0, 1.0f);
        const float q0 = a * r0;
        const float q0r = __builtin_fmaf(q0, e0, q0);
        const float e1 = __builtin_fmaf(-b, q0r, a);
        return __builtin_fmaf(e1, r0, q0r);
    } else {
        const float e0 = __builtin_fmaf(-b, r0, 1.0f);
        const float r1 = __builtin_fmaf(e0, r0, r0);
        return a * r1;
    }
}

__global__ __launch_bounds__(256) void probe(Result *res, uint32_t first, uint32_t count_log2) {
    unsigned long long dom[2] = { 0, 0 }, mm[2][kVariants] = {}, mz[2][kVariants] = {};
    const uint64_t n = 1ull << count_log2;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = first + (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float ax = __builtin_fabsf(x);
        if (!(ax == ax)) continue;                               // NaN
        const float x2 = x * x;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float a, b;
            bool in;
            if (f == 0) {                                        // fast_tanh
                a = x * (945.0f + x2 * (105.0f + x2));
                b = 945.0f + x2 * (420.0f + x2 * 15.0f);
                in = (ax == 0.0f || ax >= 0x1p-82f) && ax <= 4.97f;
            } else {                                             // fast_atanh
                a = x * (945.0f + x2 * (-735.0f + x2 * 64.0f));
                b = (945.0f + x2 * (-1050.0f + x2 * 225.0f));
                in = (ax == 0.0f || ax >= 0x1p-59f) && ax <= 1.05f;
            }
            if (!in) continue;
            dom[f]++;
            const float ref = __fdiv_rn(a, b);
            const uint32_t rb = __float_as_uint(ref);
#pragma unroll
            for (int v = 0; v < kVariants; ++v) {
                const uint32_t qb = __float_as_uint(chain(v, a, b));
                if (qb != rb) {
                    mz[f][v]++;
                    if (((qb | rb) << 1) != 0) {                 // not merely +0 against -0
                        if (mm[f][v]++ < 2) {                     // a few examples per thread at most
                            const unsigned int slot = atomicAdd(&res->nexamples[f][v], 1u);
                            if (slot < kExamples) res->examples[f][v][slot] = bits;
                        }
                    }
                }
            }
        }
    }
    for (int f = 0; f < 2; ++f) {
        if (dom[f]) atomicAdd(&res->in_domain[f], dom[f]);
        for (int v = 0; v < kVariants; ++v) {
            if (mm[f][v]) atomicAdd(&res->mismatch[f][v], mm[f][v]);
            if (mz[f][v]) atomicAdd(&res->mismatch_signed_zero[f][v], mz[f][v]);
        }
    }
}

int main() {
    Result *d, h;
    if (hipMalloc(&d, sizeof(Result)) != hipSuccess) { fprintf(stderr, "no GPU\n"); return 1; }
    hipMemset(d, 0, sizeof(Result));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int part = 0; part < 16; ++part)                        // 16 launches of 2^28 patterns: all 2^32
        hipLaunchKernelGGL(probe, dim3(256 * 32), dim3(256), 0, 0, d, (uint32_t)part << 28, 28u);
    hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *fn[2] = { "fast_tanh", "fast_atanh" };
    printf("{\"patterns\": 4294967296, \"ms\": %.1f", ms);
    for (int f = 0; f < 2; ++f) {
        printf(", \"%s\": {\"inputs_in_fast_path_domain\": %llu, \"variants\": {", fn[f], h.in_domain[f]);
        for (int v = 0; v < kVariants; ++v) {
            printf("%s\"%s\": {\"mismatches\": %llu, \"incl_zero_sign\": %llu, \"examples_hex\": [", v ? ", " : "", kNames[v],
                   h.mismatch[f][v], h.mismatch_signed_zero[f][v]);
            int shown = 0;
            for (int k = 0; k < kExamples; ++k)
                if (h.examples[f][v][k]) printf("%s\"%08x\"", shown++ ? ", " : "", h.examples[f][v][k]);
            printf("]}");
        }
        printf("}}");
    }
    printf("}\n");
    return 0;
}
