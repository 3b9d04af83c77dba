// bp_selftest.hip -- proof by exhaustion that the short division chains of bp_math.h return the IEEE-754 quotient.
// fast_tanh / fast_atanh (ft8_lib ldpc.c) are functions of ONE float, so every input the BP kernel's fast path can
// see is tried: all 2^32 bit patterns, filtered to the domain the kernel's guard establishes, fast form against the
// compiler's IEEE division, scalar and packed forms alike.  About 40 ms on one MI355X.
#include "ft8gpu_internal.h"
#include "bp_math.h"

namespace {

struct Counts {
    unsigned long long tanh_inputs, tanh_mismatch, atanh_inputs, atanh_mismatch, pair_mismatch;
    unsigned int tanh_max_bits;         // bits of max |fast_tanh(x)| over the domain (bounds fast_atanh's inputs)
    unsigned int first_bad;             // one offending bit pattern (0 = none)
};

// equal bits, or both zero (the fast chains may return the other zero; decode.hip: file header)
__device__ __forceinline__ bool same(float a, float b) {
    const uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
    return x == y || ((x | y) << 1) == 0u || (a != a && b != b);
}

__global__ __launch_bounds__(256) void bp_math_exhaustive(Counts *out, uint32_t first, uint64_t n) {
    unsigned long long ti = 0, tm = 0, ai = 0, am = 0, pm = 0;
    unsigned int tmax = 0, bad = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = first + (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float ax = __builtin_fabsf(x);
        if (!(ax == ax)) continue;
        const bool in_t = ax == 0.0f || ax >= bpm::kTanhMinAbs;                     // beyond 4.97 the clamp decides: included
        const bool in_a = (ax == 0.0f || ax >= bpm::kAtanhMinAbs) && ax <= bpm::kAtanhMaxAbs;
        if (in_t) {
            ++ti;
            // r: the reference's own expression (compares + IEEE division, no assembly); f, g: the kernel's two forms,
            // both of which go through the EXEC-narrowing clamp
            const float f = bpm::tanh_one<true>(x), g = bpm::tanh_one<false>(x), r = bpm::tanh_ref(x);
            if (!same(f, r) || !same(g, r)) { ++tm; bad = bits; }
            tmax = max(tmax, __float_as_uint(__builtin_fabsf(r)));
        }
        if (in_a) {
            ++ai;
            if (!same(bpm::atanh_one<true>(x), bpm::atanh_one<false>(x))) { ++am; bad = bits; }
        }
        // packed forms: this x beside a partner from elsewhere in the domain (lane-mixing must not matter)
        const float y = __uint_as_float((bits * 2654435761u) & 0x7FFFFFFFu);
        const float ay = __builtin_fabsf(y);
        const bool y_t = (ay == ay) && (ay == 0.0f || ay >= bpm::kTanhMinAbs);
        const bool y_a = (ay == ay) && (ay == 0.0f || ay >= bpm::kAtanhMinAbs) && ay <= bpm::kAtanhMaxAbs;
        if (in_t && y_t) {
            const bpm::f2 f = bpm::tanh_pair<true>(bpm::f2{ x, y }), g = bpm::tanh_pair<false>(bpm::f2{ x, y });
            const float rx = bpm::tanh_ref(x), ry = bpm::tanh_ref(y);
            if (!same(f.x, rx) || !same(f.y, ry) || !same(g.x, rx) || !same(g.y, ry)) { ++pm; bad = bits; }
        }
        if (in_a && y_a) {
            const bpm::f2 f = bpm::atanh_pair<true>(bpm::f2{ y, x }), r = bpm::atanh_pair<false>(bpm::f2{ y, x });
            if (!same(f.x, r.x) || !same(f.y, r.y)) { ++pm; bad = bits; }
        }
    }
    if (ti) atomicAdd(&out->tanh_inputs, ti);
    if (tm) atomicAdd(&out->tanh_mismatch, tm);
    if (ai) atomicAdd(&out->atanh_inputs, ai);
    if (am) atomicAdd(&out->atanh_mismatch, am);
    if (pm) atomicAdd(&out->pair_mismatch, pm);
    atomicMax(&out->tanh_max_bits, tmax);
    if (bad) atomicMax(&out->first_bad, bad);
}

// ---- the LLR scale factor sqrtf(24.0f / variance): both operations against exact arithmetic -------------------------
// For every positive float v in [2^-60, 2^60]: q = 24.0f / v must be the correctly rounded quotient, s = sqrtf(q) the
// correctly rounded root, and bpm::llr_norm_factor(v) must be exactly s.  "Correctly rounded" is tested without trusting
// any other division or root: in double, q * v is exact (48 bits) and so is its distance from 24; (s +- half an ulp)^2 is
// exact (50 bits).  q is right iff |q v - 24| <= (ulp(q) / 2) v, s iff (s - h)^2 < q < (s + h)^2.
// And the premise of the proof above it: the exhaustive comparison of bp_math.h's chains is against "the compiler's
// division" -- which is the IEEE quotient only if THAT is correctly rounded.  Same exact test, for the numerator and
// denominator of fast_tanh on every x it can see and of fast_atanh on its domain: |q b - a| <= (ulp(q) / 2) |b|.
struct NormCounts { unsigned long long inputs, div_bad, sqrt_bad, compose_bad, rational_inputs, rational_div_bad; unsigned int first_bad; };

__device__ __forceinline__ bool quotient_is_rounded(float q, float a, float b) {
    const uint32_t e = (__float_as_uint(q) >> 23) & 0xFF;
    if (e == 0 || e == 0xFF) return true;                       // zero, subnormal, non-finite quotients: outside this test
    const double r = __builtin_fabs((double)q * (double)b - (double)a);
    return r <= __longlong_as_double((long long)((int)e - 127 - 24 + 1023) << 52) * __builtin_fabs((double)b);
}

__device__ __forceinline__ double half_ulp(float y) {          // half the distance to the next float above |y| (y normal)
    const int e = (int)((__float_as_uint(y) >> 23) & 0xFF) - 127;
    return __longlong_as_double((long long)(e - 24 + 1023) << 52);
}

__global__ __launch_bounds__(256) void norm_math_exhaustive(NormCounts *out, uint32_t first, uint64_t n) {
    unsigned long long ni = 0, db = 0, sb = 0, cb = 0, ri = 0, rb = 0;
    unsigned int bad = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = first + (uint32_t)i;
        const float v = __uint_as_float(bits);
        {   // the two rational functions' own divisions (x = v and x = -v give the same |q|: positive x suffices)
            const float x = v, x2 = x * x;
            if (x >= bpm::kTanhMinAbs && x <= 4.97f) {
                const float a = x * (945.0f + x2 * (105.0f + x2)), b = 945.0f + x2 * (420.0f + x2 * 15.0f);
                ++ri;
                if (!quotient_is_rounded(__fdiv_rn(a, b), a, b)) { ++rb; bad = bits; }
            }
            if (x >= bpm::kAtanhMinAbs && x <= bpm::kAtanhMaxAbs) {
                const float a = x * (945.0f + x2 * (-735.0f + x2 * 64.0f)), b = 945.0f + x2 * (-1050.0f + x2 * 225.0f);
                ++ri;
                if (!quotient_is_rounded(__fdiv_rn(a, b), a, b)) { ++rb; bad = bits; }
            }
        }
        if (!(v >= 0x1p-60f && v <= 0x1p60f)) continue;
        ++ni;
        const float q = __fdiv_rn(24.0f, v);
        const float s = __builtin_sqrtf(q);
        const double r = __builtin_fabs((double)q * (double)v - 24.0);
        if (r > half_ulp(q) * (double)v) { ++db; bad = bits; }
        // rounding boundaries of s: the midpoints to its two neighbours.  Below a power of two the spacing halves, so the lower
        // midpoint of s = 2^k is s - h/2, not s - h (q = pred(4^k): its root lies in (s - h, s - h/2) and must round to pred(s)).
        // lo and hi have at most 26 significant bits: their squares are exact in binary64.
        const double h = half_ulp(s), lo = (double)s - ((__float_as_uint(s) & 0x7FFFFFu) == 0 ? 0.5 * h : h), hi = (double)s + h;
        if (!(lo * lo < (double)q && (double)q < hi * hi)) { ++sb; bad = bits; }
        if (__float_as_uint(bpm::llr_norm_factor(v)) != __float_as_uint(s)) { ++cb; bad = bits; }
    }
    if (ni) atomicAdd(&out->inputs, ni);
    if (db) atomicAdd(&out->div_bad, db);
    if (sb) atomicAdd(&out->sqrt_bad, sb);
    if (cb) atomicAdd(&out->compose_bad, cb);
    if (ri) atomicAdd(&out->rational_inputs, ri);
    if (rb) atomicAdd(&out->rational_div_bad, rb);
    if (bad) atomicMax(&out->first_bad, bad);
}

}  // namespace

// out[0..6]: inputs, quotients 24/v not correctly rounded, roots not correctly rounded, llr_norm_factor != sqrtf(24/v), one
//            offending input (0 = none), divisions of fast_tanh / fast_atanh tested, of those not correctly rounded
hipError_t run_norm_math_selftest(uint64_t out[7], hipStream_t s) {
    NormCounts *d = nullptr, h;
    hipError_t e = hipMalloc(&d, sizeof(NormCounts));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d, 0, sizeof(NormCounts), s);
    for (uint32_t part = 0; part < 8 && e == hipSuccess; ++part) {          // positive floats only
        hipLaunchKernelGGL(norm_math_exhaustive, dim3(256 * 32), dim3(256), 0, s, d, part << 28, (uint64_t)1 << 28);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) return e;
    out[0] = h.inputs; out[1] = h.div_bad; out[2] = h.sqrt_bad; out[3] = h.compose_bad; out[4] = h.first_bad;
    out[5] = h.rational_inputs; out[6] = h.rational_div_bad;
    return hipSuccess;
}

// out[0..6]: tanh inputs, tanh mismatches, atanh inputs, atanh mismatches, packed-form mismatches,
//            bits of max |fast_tanh|, one offending input (0 = none)
hipError_t run_bp_math_selftest(uint64_t out[7], hipStream_t s) {
    Counts *d = nullptr, h;
    hipError_t e = hipMalloc(&d, sizeof(Counts));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d, 0, sizeof(Counts), s);
    for (uint32_t part = 0; part < 16 && e == hipSuccess; ++part) {
        hipLaunchKernelGGL(bp_math_exhaustive, dim3(256 * 32), dim3(256), 0, s, d, part << 28, (uint64_t)1 << 28);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) return e;
    out[0] = h.tanh_inputs; out[1] = h.tanh_mismatch; out[2] = h.atanh_inputs; out[3] = h.atanh_mismatch;
    out[4] = h.pair_mismatch; out[5] = h.tanh_max_bits; out[6] = h.first_bad;
    return hipSuccess;
}
