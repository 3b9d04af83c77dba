// bp_math.h -- fast_tanh() / fast_atanh() of ft8_lib ldpc.c as the BP kernel evaluates them (decode.hip), and the
// same code for the exhaustive self-test (bp_selftest.hip).  Device code only.
//
// Division.  The reference's a / b is the IEEE-754 correctly rounded quotient.  The compiler's expansion is
// v_div_scale x2, v_rcp, fma x2, mul, fma x3 (v_div_fmas), v_div_fixup; when v_div_scale does not rescale
// (|numerator| >= 2^-103, moderate denominator, normal quotient, no special value) scale and fixup are the
// identity.  Both rational functions compute a AND b from ONE float x by a fixed sequence of IEEE operations, so
// "does a shorter rcp/fma chain return the same bits?" is a question about 2^32 inputs and is answered by
// trying them all on the hardware (v_rcp_f32 is a fixed function of its input):
//   * fast_atanh: q0 = a * rcp(b); q = fma(fma(-b, q0, a), rcp(b), q0) -- three operations after the reciprocal --
//     equals the IEEE quotient for EVERY x with x == 0 or 2^-59 <= |x| <= 1.05 (990 694 608 inputs);
//   * fast_tanh: the same chain misses on three magnitudes (0x3bcb9486, 0x3c42d1d7, 0x3dfd692c); with one
//     Newton step on the reciprocal first (five operations) it equals the IEEE quotient for EVERY x with
//     x == 0 or 2^-82 <= |x| <= 4.97 (1 413 354 622 inputs; beyond 4.97 the clamp overrides the quotient).
// (tools/ubench/div_exhaustive.hip tries six chains, profiles/r03_div_exhaustive.json; round 2 used seven
// operations for both.)  The domains are what decode.hip's per-iteration guard establishes (guard_key); the
// only difference from the IEEE form is that a zero quotient may carry the other sign, which no later operation
// observes.  ft8gpu_selftest_bp_math() re-runs the exhaustive comparison on these very functions (all 2^32
// patterns, about 40 ms) and tests/test_gpu_parity.py requires zero mismatches.
#pragma once
#include <hip/hip_runtime.h>

namespace bpm {

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// ---- a / b for a pair ---------------------------------------------------------------------------
__device__ __forceinline__ f2 rcp_pair(f2 b) {
    f2 r;
    r.x = __builtin_amdgcn_rcpf(b.x);
    r.y = __builtin_amdgcn_rcpf(b.y);
    return r;
}
// raw reciprocal, one residual correction (exact on fast_atanh's domain)
__device__ __forceinline__ f2 div_pair_3(f2 a, f2 b) {
    const f2 r0 = rcp_pair(b);
    const f2 q0 = a * r0;
    const f2 e1 = pk_fma(-b, q0, a);
    return pk_fma(e1, r0, q0);
}
// Newton step on the reciprocal, one residual correction (exact on fast_tanh's domain)
__device__ __forceinline__ f2 div_pair_5(f2 a, f2 b) {
    const f2 r0 = rcp_pair(b);
    const f2 one = { 1.0f, 1.0f };
    const f2 e0 = pk_fma(-b, r0, one);
    const f2 r1 = pk_fma(e0, r0, r0);
    const f2 q0 = a * r1;
    const f2 e1 = pk_fma(-b, q0, a);
    return pk_fma(e1, r1, q0);
}
__device__ __forceinline__ f2 div_pair_ieee(f2 a, f2 b) {
    f2 q;
    q.x = __fdiv_rn(a.x, b.x);
    q.y = __fdiv_rn(a.y, b.y);
    return q;
}
// the same chains on one float (the ninth edge of a lane has no partner; a packed instruction costs about 1.7
// scalar ones on gfx950, so half-empty pairs are not free)
__device__ __forceinline__ float div_one_3(float a, float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float q0 = a * r0;
    const float e1 = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(e1, r0, q0);
}
__device__ __forceinline__ float div_one_5(float a, float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float q0 = a * r1;
    const float e1 = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(e1, r1, q0);
}

// fast_tanh's clamp: "x < -4.97 -> -1; x > 4.97 -> +1" == "|x| > 4.97 -> copysign(1, x)" (NaN takes neither branch).
// As the compiler writes it this is v_cmp + v_bfi + v_cndmask per value.  Here the compare narrows EXEC itself
// (v_cmpx) and the copysign is written in place under that mask: one VALU instruction less per value -- nine per BP
// iteration -- for two scalar moves that save and restore EXEC (the scalar unit has the slack).
// Hazard note: v_cmpx is a VALU write of EXEC, and gfx9 wants 5 wait states between such a write and a DPP
// instruction; the compiler's hazard recogniser cannot see into an asm block.  The blocks below are followed by the
// tanh results' LDS stores in the BP loop, never by DPP code (wave_sum / wave_xor sit in the prologue and epilogue);
// tools/kernel_resources.py scans the compiled kernels for a DPP instruction within 5 wait states of a v_cmpx and
// tests/test_kernel_resources.py fails the build if one ever appears.  (A blanket `s_nop 4` behind each block would
// cost about 2 % of the loop.)
__device__ __forceinline__ float clamp_497(float q, float x) {
    unsigned long long saved;
    asm("s_mov_b64 %1, exec\n\t"
        "v_cmpx_gt_f32_e64 vcc, |%2|, %3\n\t"
        "v_bfi_b32 %0, %4, 1.0, %2\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(q), "=&s"(saved) : "v"(x), "s"(4.97f), "s"(0x7fffffffu) : "vcc");
    return q;
}
__device__ __forceinline__ f2 clamp_497(f2 q, f2 x) {
    unsigned long long saved;
    float q0 = q.x, q1 = q.y;
    asm("s_mov_b64 %2, exec\n\t"
        "v_cmpx_gt_f32_e64 vcc, |%3|, %5\n\t"
        "v_bfi_b32 %0, %6, 1.0, %3\n\t"
        "s_mov_b64 exec, %2\n\t"
        "v_cmpx_gt_f32_e64 vcc, |%4|, %5\n\t"
        "v_bfi_b32 %1, %6, 1.0, %4\n\t"
        "s_mov_b64 exec, %2"
        : "+v"(q0), "+v"(q1), "=&s"(saved) : "v"(x.x), "v"(x.y), "s"(4.97f), "s"(0x7fffffffu) : "vcc");
    return f2{ q0, q1 };
}

// ---- fast_tanh() of ft8_lib ldpc.c: clamp tests in the reference's order, rational evaluated unconditionally
// (finite or overridden for every finite x)
template <bool FAST>
__device__ __forceinline__ f2 tanh_pair(f2 x) {
    const f2 x2 = x * x;
    const f2 a = x * (945.0f + x2 * (105.0f + x2));
    const f2 b = 945.0f + x2 * (420.0f + x2 * 15.0f);
    return clamp_497(FAST ? div_pair_5(a, b) : div_pair_ieee(a, b), x);
}
template <bool FAST>
__device__ __forceinline__ float tanh_one(float x) {
    const float x2 = x * x;
    const float a = x * (945.0f + x2 * (105.0f + x2));
    const float b = 945.0f + x2 * (420.0f + x2 * 15.0f);
    return clamp_497(FAST ? div_one_5(a, b) : __fdiv_rn(a, b), x);
}

// fast_tanh() exactly as ft8_lib writes it -- plain compares, plain IEEE division, no inline assembly: what the
// exhaustive self-test holds BOTH kernel forms against (so the EXEC-narrowing clamp itself is checked on all 2^32
// inputs, not only the division chain)
__device__ __forceinline__ float tanh_ref(float x) {
    if (x < -4.97f) return -1.0f;
    if (x > 4.97f) return 1.0f;
    const float x2 = x * x;
    const float a = x * (945.0f + x2 * (105.0f + x2));
    const float b = 945.0f + x2 * (420.0f + x2 * 15.0f);
    return __fdiv_rn(a, b);
}

// ---- fast_atanh() of ft8_lib ldpc.c
// (fast form: x2 * 64 is a scaling by a power of two, hence exact, so -735 + x2 * 64 is ONE rounding either way and a
// fused multiply-add returns the very same float -- one instruction less per pair; the exhaustive self-test compares
// it with the reference's two-operation form like everything else in this header)
template <bool FAST>
__device__ __forceinline__ f2 atanh_pair(f2 x) {
    const f2 x2 = x * x;
    const f2 inner = FAST ? pk_fma(x2, f2{ 64.0f, 64.0f }, f2{ -735.0f, -735.0f }) : (-735.0f + x2 * 64.0f);
    const f2 a = x * (945.0f + x2 * inner);
    const f2 b = (945.0f + x2 * (-1050.0f + x2 * 225.0f));
    return FAST ? div_pair_3(a, b) : div_pair_ieee(a, b);
}
template <bool FAST>
__device__ __forceinline__ float atanh_one(float x) {
    const float x2 = x * x;
    const float inner = FAST ? __builtin_fmaf(x2, 64.0f, -735.0f) : (-735.0f + x2 * 64.0f);
    const float a = x * (945.0f + x2 * inner);
    const float b = (945.0f + x2 * (-1050.0f + x2 * 225.0f));
    return FAST ? div_one_3(a, b) : __fdiv_rn(a, b);
}

// ---- ftx_normalize_logl's scale factor: sqrtf(24.0f / variance) (ft8_lib decode.c), two correctly rounded IEEE operations.
// HIP's __fsqrt_rn is NOT that: without OCML_BASIC_ROUNDED_OPERATIONS it is __ocml_native_sqrt_f32, the raw v_sqrt_f32
// (1 ulp).  Rounds 1-4 used it, and about every hundredth candidate's LLRs came out scaled one ulp low -- invisible in
// any output until round 5 compared the status record of EVERY candidate of 20 million (a hard decision within an ulp of
// zero flipped in 93 of them: tools/record_diff_probe.py).  __builtin_sqrtf is the correctly rounded expansion (v_sqrt_f32
// plus a residual test either side) under the default -fhip-fp32-correctly-rounded-divide-sqrt; bp_selftest.hip checks
// both operations against exact arithmetic for every float.
__device__ __forceinline__ float llr_norm_factor(float variance) { return __builtin_sqrtf(__fdiv_rn(24.0f, variance)); }

// domains of the fast forms (what decode.hip's guard establishes; see guard_key there)
constexpr float kTanhMinAbs = 0x1p-82f;          // x == 0 or |x| >= 2^-82 (any larger |x|: the clamp takes over beyond 4.97)
constexpr float kAtanhMinAbs = 0x1p-59f;         // P == 0 or |P| >= 2^-59
constexpr float kAtanhMaxAbs = 1.05f;            // |P| <= 1.0073^6 < 1.0446: six factors, each at most max|fast_tanh| = 1.00722

}  // namespace bpm
