// rx.hip -- SURVEY.md section 8(f-1): the RX front end of the reference, rtlsdr_callback()
// (rtlsdr_ft8d.c:76-202), for whole raw captures resident in HBM:
//   unsigned 8-bit I/Q at 2.4 Msps -> fs/4 mixer (:129-140) -> CIC decimator, N = 2, comb delay 2,
//   effective ratio 751 because of the `<=` at :157 (:148-176) -> 57-tap compensation FIR (:178-192)
//   -> scale by 1/(32768*750) (:197-198) -> float I/Q at ~3200 sps, then the decoder thread's tail
//   zeroing and peak normalisation (:243-263).
//
// This stage moves 72 MB per 15 s capture against 384 KB out: it is the HBM-bound part of the chain.
//
// How a recursive filter becomes a streaming reduction.  With wrapping 32-bit arithmetic (what the
// reference's int32 integrators do), the two integrators after decimation block b (751 input pairs
// x_1..x_751) are
//     P1_b = P1_{b-1} + A_b,                       A_b = sum_i x_i
//     P2_b = P2_{b-1} + 751 * P1_{b-1} + W_b,      W_b = sum_i (752 - i) * x_i
// exactly (ring arithmetic mod 2^32 is associative), so
//   kernel 1 streams the raw bytes once and reduces every block to (A, W) for I and Q: a quarter wave
//            (16 lanes) per block, 16 B per lane per load, the fs/4 rotation and the int8 wrap of the
//            reference's in-place negation done four bytes at a time (SWAR), sums by v_dot4_i32_i8;
//            the 16 blocks of a workgroup are then pre-integrated locally;
//   kernel 2 scans the <= 3000 workgroup totals of a capture (one workgroup) into entry states;
//   kernel 3 rebuilds the integrator at every block, applies the two combs, the FIR in the
//            reference's summation order and the scaling, and tracks the capture's peak;
//   kernel 4 (optional) is the decoder thread's peak normalisation.
// All integer results are bit-identical to the sequential code; the float FIR repeats its order.
#include "ft8gpu_internal.h"

namespace {

constexpr int kR = 751;                       // input pairs per output sample (DOWNSAMPLING + 1, rtlsdr_ft8d.c:157)
constexpr int kFirTaps = 56;                  // FIR_TAPS, rtlsdr_ft8d.h:40 (57 coefficients)

__constant__ float c_zCoef[kFirTaps + 1] = {  // rtlsdr_ft8d.c:94-110
    -0.0025719973f,  0.0010118403f,  0.0009110571f, -0.0034940765f,
     0.0069713409f, -0.0114242790f,  0.0167023466f, -0.0223683056f,
     0.0276808966f, -0.0316243672f,  0.0329894230f, -0.0305042011f,
     0.0230074504f, -0.0096499429f, -0.0098950502f,  0.0352349632f,
    -0.0650990428f,  0.0972406918f, -0.1284211497f,  0.1544893973f,
    -0.1705667465f,  0.1713383321f, -0.1514501610f,  0.1060148823f,
    -0.0312560926f, -0.0745846391f,  0.2096088743f, -0.3638689868f,
     0.5000000000f,
    -0.3638689868f,  0.2096088743f, -0.0745846391f, -0.0312560926f,
     0.1060148823f, -0.1514501610f,  0.1713383321f, -0.1705667465f,
     0.1544893973f, -0.1284211497f,  0.0972406918f, -0.0650990428f,
     0.0352349632f, -0.0098950502f, -0.0096499429f,  0.0230074504f,
    -0.0305042011f,  0.0329894230f, -0.0316243672f,  0.0276808966f,
    -0.0223683056f,  0.0167023466f, -0.0114242790f,  0.0069713409f,
    -0.0034940765f,  0.0009110571f,  0.0010118403f, -0.0025719973f
};

// Four raw bytes -> four signed samples (x = raw ^ 0x80), with the bytes selected by `neg` (0xFF per
// byte) negated the way an int8 store does it: -(-128) wraps back to -128 (rtlsdr_ft8d.c:134-139).
__device__ __forceinline__ uint32_t mix4(uint32_t raw, uint32_t neg) {
    const uint32_t t = raw ^ (0x80808080u ^ neg);                     // x, or ~x where negated
    return ((t & 0x7F7F7F7Fu) + (neg & 0x01010101u)) ^ (t & 0x80808080u);   // ~x + 1 per selected byte
}

// One group of four I/Q pairs (8 bytes, pair index of the first = multiple of 4) with the fs/4
// rotation of :129-140: I stream = (x0, -x3, -x4, x7), Q stream = (x1, x2, -x5, -x6).
// wbase = CIC weight of the group's first pair (752 - position in block); keep = byte masks that
// clear pairs outside the block.
__device__ __forceinline__ void group_sums(uint32_t lo, uint32_t hi, uint32_t keep_lo, uint32_t keep_hi, int wbase,
                                           int &aI, int &wI, int &aQ, int &wQ) {
    const int ylo = (int)(mix4(lo, 0xFF000000u) & keep_lo);
    const int yhi = (int)(mix4(hi, 0x00FFFFFFu) & keep_hi);
    const int sI = __builtin_amdgcn_sdot4(yhi, 0x01000001, __builtin_amdgcn_sdot4(ylo, 0x01000001, 0, false), false);
    const int uI = __builtin_amdgcn_sdot4(yhi, 0x03000002, __builtin_amdgcn_sdot4(ylo, 0x01000000, 0, false), false);
    const int sQ = __builtin_amdgcn_sdot4(yhi, 0x00010100, __builtin_amdgcn_sdot4(ylo, 0x00010100, 0, false), false);
    const int uQ = __builtin_amdgcn_sdot4(yhi, 0x00030200, __builtin_amdgcn_sdot4(ylo, 0x00010000, 0, false), false);
    aI += sI;
    wI += wbase * sI - uI;
    aQ += sQ;
    wQ += wbase * sQ - uQ;
}

// byte mask keeping pairs [plo, phi) of a 4-pair group (2 bytes per pair)
__device__ __forceinline__ uint64_t pair_mask(int plo, int phi) {
    plo = plo < 0 ? 0 : (plo > 4 ? 4 : plo);
    phi = phi < 0 ? 0 : (phi > 4 ? 4 : phi);
    const uint64_t lo = plo >= 4 ? 0ull : (~0ull << (16 * plo));
    const uint64_t hi = phi >= 4 ? ~0ull : ~(~0ull << (16 * phi));
    return lo & hi;
}

__global__ __launch_bounds__(256)
void ft8_rx_block_kernel(const uint8_t *__restrict__ raw, size_t capture_bytes, int nblocks,
                         int4 *__restrict__ sums) {
    const int capture = blockIdx.y;
    const int quarter = threadIdx.x >> 4, ql = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + quarter;                 // decimation block of this quarter wave
    const uint8_t *base = raw + (size_t)capture * capture_bytes;
    int aI = 0, wI = 0, aQ = 0, wQ = 0;
    if (b < nblocks) {
        const int first_pair = kR * b, end_pair = first_pair + kR;
        const int u0 = (first_pair * 2) >> 4, u1 = (end_pair * 2 - 1) >> 4;      // 16-byte units touched
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int u = u0 + ql + 16 * t;
            if (u <= u1) {
                const uint4 v = *reinterpret_cast<const uint4 *>(base + (size_t)u * 16);
                const int n0 = 8 * u;                                            // first pair of the unit
                uint32_t k0 = ~0u, k1 = ~0u, k2 = ~0u, k3 = ~0u;
                if (n0 < first_pair || n0 + 8 > end_pair) {                      // boundary unit: clear outsiders
                    const uint64_t ma = pair_mask(first_pair - n0, end_pair - n0);
                    const uint64_t mb = pair_mask(first_pair - n0 - 4, end_pair - n0 - 4);
                    k0 = (uint32_t)ma; k1 = (uint32_t)(ma >> 32); k2 = (uint32_t)mb; k3 = (uint32_t)(mb >> 32);
                }
                group_sums(v.x, v.y, k0, k1, end_pair - n0, aI, wI, aQ, wQ);
                group_sums(v.z, v.w, k2, k3, end_pair - n0 - 4, aI, wI, aQ, wQ);
            }
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {                        // reduce over the 16 lanes of the quarter
        aI += __shfl_xor(aI, o, 16);
        wI += __shfl_xor(wI, o, 16);
        aQ += __shfl_xor(aQ, o, 16);
        wQ += __shfl_xor(wQ, o, 16);
    }
    // Local running integrators over the 16 blocks of this workgroup (starting from zero state):
    //   lp1_j = sum_{i<=j} A_i,   lp2_j = sum_{i<=j} (751 * lp1_{i-1} + W_i).
    // With the workgroup's true entry state (P1base, P2base) the integrators after block j are
    //   P1 = P1base + lp1_j,   P2 = P2base + 751 * (j+1) * P1base + lp2_j.
    __shared__ int4 s_blk[16];
    if (ql == 0) s_blk[quarter] = (b < nblocks) ? make_int4(aI, wI, aQ, wQ) : make_int4(0, 0, 0, 0);
    __syncthreads();
    if (threadIdx.x < 16) {
        uint32_t p1I = 0, p2I = 0, p1Q = 0, p2Q = 0;
        for (int i = 0; i <= (int)threadIdx.x; ++i) {
            const int4 v = s_blk[i];
            p2I += (uint32_t)kR * p1I + (uint32_t)v.y;
            p2Q += (uint32_t)kR * p1Q + (uint32_t)v.w;
            p1I += (uint32_t)v.x;
            p1Q += (uint32_t)v.z;
        }
        const int bb = blockIdx.x * 16 + threadIdx.x;
        if (bb < nblocks) sums[(size_t)capture * nblocks + bb] = make_int4((int)p1I, (int)p2I, (int)p1Q, (int)p2Q);
    }
}

// inclusive scan of one uint32 per thread over a 1024-thread workgroup
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t v, uint32_t *s_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(v, o, 64);
        if (lane >= o) v += n;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    uint32_t offs = 0;
    for (int w = 0; w < wave; ++w) offs += s_wave[w];
    return v + offs;
}

// Entry state (P1base, P2base) of every 16-block group of a capture, both channels:
// base[capture][ngroups] = (P1base_I, P2base_I, P1base_Q, P2base_Q).  One workgroup per capture scans
// the group totals (the local values of each group's last block).
__global__ __launch_bounds__(1024)
void ft8_rx_scan_kernel(const int4 *__restrict__ sums, int nblocks, int ngroups, int4 *__restrict__ base) {
    __shared__ uint32_t s_wave[16];
    const int capture = blockIdx.x, tid = threadIdx.x;
    const int4 *s = sums + (size_t)capture * nblocks;
    int4 *out = base + (size_t)capture * ngroups;
    const int per = (ngroups + 1023) / 1024;
    const int g0 = min(tid * per, ngroups), g1 = min(g0 + per, ngroups);
    auto total = [&](int g) { return s[min(16 * g + 15, nblocks - 1)]; };          // local values after the group's last block
    auto count = [&](int g) { return (uint32_t)(min(16 * g + 16, nblocks) - 16 * g); };
    uint32_t aI = 0, aQ = 0;
    for (int g = g0; g < g1; ++g) { const int4 t = total(g); aI += (uint32_t)t.x; aQ += (uint32_t)t.z; }
    uint32_t p1I = block_scan_incl(aI, s_wave) - aI;          // P1base of this thread's first group
    uint32_t p1Q = block_scan_incl(aQ, s_wave) - aQ;
    uint32_t tI = 0, tQ = 0;
    {
        uint32_t qI = p1I, qQ = p1Q;
        for (int g = g0; g < g1; ++g) {
            const int4 t = total(g);
            tI += (uint32_t)kR * count(g) * qI + (uint32_t)t.y;
            tQ += (uint32_t)kR * count(g) * qQ + (uint32_t)t.w;
            qI += (uint32_t)t.x;
            qQ += (uint32_t)t.z;
        }
    }
    uint32_t p2I = block_scan_incl(tI, s_wave) - tI;
    uint32_t p2Q = block_scan_incl(tQ, s_wave) - tQ;
    for (int g = g0; g < g1; ++g) {
        out[g] = make_int4((int)p1I, (int)p2I, (int)p1Q, (int)p2Q);
        const int4 t = total(g);
        p2I += (uint32_t)kR * count(g) * p1I + (uint32_t)t.y;
        p2Q += (uint32_t)kR * count(g) * p1Q + (uint32_t)t.w;
        p1I += (uint32_t)t.x;
        p1Q += (uint32_t)t.z;
    }
}

// second integrator after block k (k >= 0): group entry state + local running value
__device__ __forceinline__ uint32_t p2_at(const int4 *__restrict__ sums, const int4 *__restrict__ base, int chan, int k) {
    const int4 b = base[k >> 4], l = sums[k];
    const uint32_t p1b = (uint32_t)(chan ? b.z : b.x), p2b = (uint32_t)(chan ? b.w : b.y), lp2 = (uint32_t)(chan ? l.w : l.y);
    return p2b + (uint32_t)kR * (uint32_t)((k & 15) + 1) * p1b + lp2;
}

// combs (:162-176), FIR (:178-192), scaling (:197-198), tail zeroing (:243-246); out: [capture][2][48000];
// also one partial peak |sample| per workgroup for the normalisation that follows
__global__ __launch_bounds__(256)
void ft8_rx_fir_kernel(const int4 *__restrict__ sums, const int4 *__restrict__ base, int nblocks, int ngroups,
                       float *__restrict__ iq, float *__restrict__ peak) {
    __shared__ uint32_t s_p[256 + kFirTaps + 4];
    __shared__ float s_y[256 + kFirTaps];
    __shared__ float s_max[4];
    const int capture = blockIdx.z, chan = blockIdx.y, nout = nblocks;
    const int4 *cs = sums + (size_t)capture * nblocks, *cb = base + (size_t)capture * ngroups;
    float *out = iq + ((size_t)capture * 2 + chan) * kNSamples;
    const int k0 = blockIdx.x * 256;
    for (int i = threadIdx.x; i < 256 + kFirTaps + 4; i += 256) {
        const int k = k0 - kFirTaps - 4 + i;
        s_p[i] = (k >= 0 && k < nout) ? p2_at(cs, cb, chan, k) : 0u;      // integrator state before the capture is 0
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256 + kFirTaps; i += 256) {
        const int k = k0 - kFirTaps + i;                     // comb output index
        float y = 0.0f;                                      // FIR history starts at zero (:113-114)
        if (k >= 0 && k < nout) {
            const uint32_t a = s_p[i + 4], b = s_p[i + 2], c = s_p[i];
            y = (float)(int32_t)((a - b) - (b - c));         // Iy2 = (Ix2 - z^-2) - (... z^-2), wrapping
        }
        s_y[i] = y;
    }
    __syncthreads();
    const int k = k0 + threadIdx.x;
    float v = 0.0f;
    if (k < nout) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j <= kFirTaps; ++j) acc += s_y[threadIdx.x + j] * c_zCoef[j];   // :181-192, oldest first
        v = (float)((double)acc / (32768.0 * 750));
    }
    if (k < kNSamples) out[k] = v;
    float m = fabsf(v);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)                                    // one partial peak per workgroup (no atomics: 376 per capture)
        peak[((size_t)capture * 2 + chan) * gridDim.x + blockIdx.x] = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
}

// decoder thread, rtlsdr_ft8d.c:248-263: peak-normalise I and Q of a frame to 0.5
__global__ __launch_bounds__(256)
void ft8_rx_normalise_kernel(float *__restrict__ iq, const float *__restrict__ peak, int npartials) {
    __shared__ float s_max[4];
    const int capture = blockIdx.y;
    float pk = 0.0f;
    for (int i = threadIdx.x; i < npartials; i += 256) pk = fmaxf(pk, peak[(size_t)capture * npartials + i]);
    for (int o = 32; o > 0; o >>= 1) pk = fmaxf(pk, __shfl_xor(pk, o, 64));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = pk;
    __syncthreads();
    pk = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    float maxSig = 1e-24f;                                   // :249
    if (pk > maxSig) maxSig = pk;
    const float sc = (float)(0.5 / (double)maxSig);          // :259
    float4 *f = reinterpret_cast<float4 *>(iq + (size_t)capture * 2 * kNSamples);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * kNSamples / 4) {
        float4 v = f[i];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        f[i] = v;
    }
}

}  // namespace

// raw: [ncaptures][2*npairs] bytes on the device; scratch_sums: 16 B per block; scratch_base: 16 B per
// 16-block group followed by 376 partial peaks (float) per capture
hipError_t launch_rx(const uint8_t *raw, int ncaptures, size_t npairs, void *scratch_sums, void *scratch_base,
                     float *iq, int normalise, hipStream_t s) {
    if (ncaptures < 1) return hipSuccess;
    const size_t nb = npairs / kR;
    const int nblocks = (int)(nb > (size_t)kNSamples ? (size_t)kNSamples : nb);   // samples past 48000 are dropped (:196)
    const int ngroups = (nblocks + 15) / 16;
    int4 *base = (int4 *)scratch_base;
    float *peak = (float *)(base + (size_t)ncaptures * (ngroups > 0 ? ngroups : 1));
    constexpr int kFirGrid = (kNSamples + 255) / 256;        // 188 workgroups per channel
    if (nblocks > 0) {
        hipLaunchKernelGGL(ft8_rx_block_kernel, dim3(ngroups, ncaptures), dim3(256), 0, s,
                           raw, npairs * 2, nblocks, (int4 *)scratch_sums);
        hipLaunchKernelGGL(ft8_rx_scan_kernel, dim3(ncaptures), dim3(1024), 0, s,
                           (const int4 *)scratch_sums, nblocks, ngroups, base);
    }
    hipLaunchKernelGGL(ft8_rx_fir_kernel, dim3(kFirGrid, 2, ncaptures), dim3(256), 0, s,
                       (const int4 *)scratch_sums, (const int4 *)base, nblocks, ngroups, iq, peak);
    if (normalise)
        hipLaunchKernelGGL(ft8_rx_normalise_kernel, dim3((2 * kNSamples / 4 + 255) / 256, ncaptures), dim3(256), 0, s, iq, peak, 2 * kFirGrid);
    return hipGetLastError();
}
