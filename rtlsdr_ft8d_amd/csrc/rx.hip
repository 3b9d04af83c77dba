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
//            (16 lanes) per block, six 16-byte loads per lane all in flight, the bytes of each stream gathered by
//            v_perm_b32, the fs/4 rotation and the int8 wrap of the reference's in-place negation done four bytes
//            at a time (SWAR), sums by v_dot4_i32_i8 into running accumulators; the 16 blocks of a workgroup are
//            then pre-integrated locally (DPP row scans);
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

// One group of four I/Q pairs (8 bytes x0..x7, pair index of the first = multiple of 4) with the fs/4 rotation of :129-140:
// I stream = (x0, -x3, -x4, x7), Q stream = (x1, x2, -x5, -x6).  The four bytes of each stream are gathered into one
// dword (v_perm_b32), negated where the rotation says so, and summed with v_dot4_i32_i8 straight into the lane's running
// sums: a = sum of the stream, u = sum of (pair index within the 16-byte unit) * sample.  keep = byte mask of the pairs
// inside the block (byte p = pair p); uw = the group's pair indices as dot weights.
__device__ __forceinline__ void group_sums(uint32_t lo, uint32_t hi, uint32_t keep, int uw, int &aI, int &uI, int &aQ, int &uQ) {
    const int yi = (int)(mix4(__builtin_amdgcn_perm(hi, lo, 0x07040300u), 0x00FFFF00u) & keep);   // (x0, x3, x4, x7), middle two negated
    const int yq = (int)(mix4(__builtin_amdgcn_perm(hi, lo, 0x06050201u), 0xFFFF0000u) & keep);   // (x1, x2, x5, x6), upper two negated
    aI = __builtin_amdgcn_sdot4(yi, 0x01010101, aI, false);
    uI = __builtin_amdgcn_sdot4(yi, uw, uI, false);
    aQ = __builtin_amdgcn_sdot4(yq, 0x01010101, aQ, false);
    uQ = __builtin_amdgcn_sdot4(yq, uw, uQ, false);
}

// byte mask keeping pairs [plo, phi) of a 4-pair group (byte p = pair p)
__device__ __forceinline__ uint32_t pair_mask(int plo, int phi) {
    plo = plo < 0 ? 0 : (plo > 4 ? 4 : plo);
    phi = phi < 0 ? 0 : (phi > 4 ? 4 : phi);
    const uint32_t lo = plo >= 4 ? 0u : (~0u << (8 * plo));
    const uint32_t hi = phi >= 4 ? ~0u : ~(~0u << (8 * phi));
    return lo & hi;
}

// sum over the 16 lanes of a DPP row, result in every lane of the row (wrapping int32)
__device__ __forceinline__ int row_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
    return v;
}
// inclusive prefix sum over the 16 lanes of a DPP row (row_shr:n shifts in zeros: bound_ctrl)
__device__ __forceinline__ uint32_t row_scan(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
    return (uint32_t)v;
}

__global__ __launch_bounds__(256)
void ft8_rx_block_kernel(const uint8_t *__restrict__ raw, size_t capture_bytes, int nblocks,
                         int4 *__restrict__ sums, int4 *__restrict__ gtot) {
    const int capture = blockIdx.y;
    const int quarter = threadIdx.x >> 4, ql = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + quarter;                 // decimation block of this quarter wave
    const uint8_t *base = raw + (size_t)capture * capture_bytes;
    int aI = 0, wI = 0, aQ = 0, wQ = 0;
    if (b < nblocks) {
        const int first_pair = kR * b, end_pair = first_pair + kR;
        const int u0 = (first_pair * 2) >> 4, u1 = (end_pair * 2 - 1) >> 4;      // 16-byte units touched: u1 - u0 = 93 or 94
        // All six loads of the lane are issued before the first one is consumed (96 bytes in flight per lane: the
        // kernel is a pure stream, so memory-level parallelism is its throughput).  Units ql + 16 t, t < 5, lie inside
        // the block for every lane (ql + 64 <= 79 < 93); the sixth exists for ql <= 13 or 14 and is clamped to the
        // block's last unit otherwise (a line the neighbouring lanes fetch anyway) and not summed.
        uint4 v[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int u = min(u0 + ql + 16 * t, u1);
            v[t] = *reinterpret_cast<const uint4 *>(base + (size_t)u * 16);
        }
        // CIC weight of the pair at index j of unit t: (end_pair - n0) - j with n0 = 8 (u0 + ql) + 128 t, so the lane's
        //   W = sum_t [(wb0 - 128 t) s_t - U_t] = (wb0 - 768) S + 128 R - U,     wb0 = end_pair - 8 (u0 + ql),
        // with s_t the sum of unit t, S = sum_t s_t, U = sum of (index in unit) * sample and R = sum_k (s_0 + ... + s_k)
        // -- the running sum added up once per unit (sum_t t s_t = 6 S - R).  All of it wraps mod 2^32 like the reference.
        int uI = 0, uQ = 0, rI = 0, rQ = 0;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int u = u0 + ql + 16 * t;
            if (t < 5 || u <= u1) {
                const int n0 = 8 * u;                                            // first pair of the unit
                uint32_t k0 = ~0u, k1 = ~0u;
                if (n0 < first_pair || n0 + 8 > end_pair) {                      // boundary unit: clear outsiders
                    k0 = pair_mask(first_pair - n0, end_pair - n0);
                    k1 = pair_mask(first_pair - n0 - 4, end_pair - n0 - 4);
                }
                group_sums(v[t].x, v[t].y, k0, 0x03020100, aI, uI, aQ, uQ);
                group_sums(v[t].z, v[t].w, k1, 0x07060504, aI, uI, aQ, uQ);
            }
            rI += aI;
            rQ += aQ;
        }
        const int wb0 = end_pair - 8 * (u0 + ql);
        wI = (wb0 - 768) * aI + 128 * rI - uI;
        wQ = (wb0 - 768) * aQ + 128 * rQ - uQ;
    }
    // reduce over the 16 lanes of the quarter (one DPP row): every lane ends up with the block's sums
    aI = row_sum(aI);
    wI = row_sum(wI);
    aQ = row_sum(aQ);
    wQ = row_sum(wQ);
    // Local running integrators over the 16 blocks of this workgroup (starting from zero state):
    //   lp1_j = sum_{i<=j} A_i,   lp2_j = sum_{i<=j} (751 * lp1_{i-1} + W_i).
    // With the workgroup's true entry state (P1base, P2base) the integrators after block j are
    //   P1 = P1base + lp1_j,   P2 = P2base + 751 * (j+1) * P1base + lp2_j.
    // Both are prefix sums over the 16 blocks (wrapping arithmetic is associative): lanes 0..15 take one block each
    // and scan within their DPP row -- eight shift-and-add steps per channel instead of a 16-step serial loop that
    // kept the workgroup's other three waves' slots occupied for a microsecond.
    __shared__ int4 s_blk[16];
    if (ql == 0) s_blk[quarter] = (b < nblocks) ? make_int4(aI, wI, aQ, wQ) : make_int4(0, 0, 0, 0);
    __syncthreads();
    if (threadIdx.x < 64) {                                  // (whole first wave: DPP needs the row's lanes active; rows 1-3 are idle copies)
        const int4 v = s_blk[threadIdx.x & 15];
        const uint32_t p1I = row_scan((uint32_t)v.x), p1Q = row_scan((uint32_t)v.z);
        const uint32_t p2I = row_scan((uint32_t)kR * (p1I - (uint32_t)v.x) + (uint32_t)v.y);
        const uint32_t p2Q = row_scan((uint32_t)kR * (p1Q - (uint32_t)v.z) + (uint32_t)v.w);
        const int bb = blockIdx.x * 16 + threadIdx.x;
        if (threadIdx.x < 16 && bb < nblocks) {
            const int4 r = make_int4((int)p1I, (int)p2I, (int)p1Q, (int)p2Q);
            sums[(size_t)capture * nblocks + bb] = r;
            // the local values after the group's LAST block are the group's totals: a compact copy for the scan (16 B per
            // group, contiguous -- read out of `sums` they sit 256 B apart, one cache line each)
            if (threadIdx.x == 15 || bb == nblocks - 1) gtot[(size_t)capture * gridDim.x + blockIdx.x] = r;
        }
    }
}

// inclusive scan of a pair of uint32 per thread over a 1024-thread workgroup (wrapping sums; I and Q share the barriers)
__device__ __forceinline__ void block_scan_incl2(uint32_t &a, uint32_t &b, uint32_t (*s_wave)[2]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t na = __shfl_up(a, o, 64), nb = __shfl_up(b, o, 64);
        if (lane >= o) { a += na; b += nb; }
    }
    __syncthreads();                                          // the previous scan's readers are done with s_wave
    if (lane == 63) { s_wave[wave][0] = a; s_wave[wave][1] = b; }
    __syncthreads();
    uint32_t oa = 0, ob = 0;
    for (int w = 0; w < wave; ++w) { oa += s_wave[w][0]; ob += s_wave[w][1]; }
    a += oa;
    b += ob;
}

// Entry state (P1base, P2base) of every 16-block group of a capture, both channels:
// base[capture][ngroups] = (P1base_I, P2base_I, P1base_Q, P2base_Q).  One workgroup per capture scans the group
// totals gtot[capture][ngroups] (at most 3000 groups: three per thread, held in registers).
constexpr int kScanPer = 3;                                   // ceil(3000 / 1024)
__global__ __launch_bounds__(1024)
void ft8_rx_scan_kernel(const int4 *__restrict__ gtot, int nblocks, int ngroups, int4 *__restrict__ base) {
    __shared__ uint32_t s_wave[16][2];
    const int capture = blockIdx.x, tid = threadIdx.x;
    const int4 *s = gtot + (size_t)capture * ngroups;
    int4 *out = base + (size_t)capture * ngroups;
    const int g0 = min(tid * kScanPer, ngroups);
    int4 t[kScanPer];
    uint32_t w[kScanPer];                                     // 751 * (blocks in the group)
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) {
        const int g = g0 + i;
        t[i] = g < ngroups ? s[g] : make_int4(0, 0, 0, 0);
        w[i] = g < ngroups ? (uint32_t)kR * (uint32_t)(min(16 * g + 16, nblocks) - 16 * g) : 0u;
    }
    uint32_t aI = 0, aQ = 0;
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) { aI += (uint32_t)t[i].x; aQ += (uint32_t)t[i].z; }
    uint32_t p1I = aI, p1Q = aQ;
    block_scan_incl2(p1I, p1Q, s_wave);
    p1I -= aI;                                                // P1base of this thread's first group
    p1Q -= aQ;
    uint32_t tI = 0, tQ = 0;
    {
        uint32_t qI = p1I, qQ = p1Q;
#pragma unroll
        for (int i = 0; i < kScanPer; ++i) {
            tI += w[i] * qI + (uint32_t)t[i].y;
            tQ += w[i] * qQ + (uint32_t)t[i].w;
            qI += (uint32_t)t[i].x;
            qQ += (uint32_t)t[i].z;
        }
    }
    uint32_t p2I = tI, p2Q = tQ;
    block_scan_incl2(p2I, p2Q, s_wave);
    p2I -= tI;
    p2Q -= tQ;
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) {
        if (g0 + i < ngroups) out[g0 + i] = make_int4((int)p1I, (int)p2I, (int)p1Q, (int)p2Q);
        p2I += w[i] * p1I + (uint32_t)t[i].y;
        p2Q += w[i] * p1Q + (uint32_t)t[i].w;
        p1I += (uint32_t)t[i].x;
        p1Q += (uint32_t)t[i].z;
    }
}

// combs (:162-176), FIR (:178-192), scaling (:197-198), tail zeroing (:243-246); out: [capture][2][48000];
// also one partial peak |sample| per workgroup and channel for the normalisation that follows.
// One workgroup = 256 consecutive outputs of BOTH channels (the integrator records hold I and Q side by side, so one load
// serves both): 188 workgroups per capture, every thread runs two independent 57-tap chains, each in the reference's
// summation order.
constexpr int kFirHalves = 1;                               // 256-output halves per workgroup (2: measured 2 us slower -- four chains of straight-line code per thread)
constexpr int kFirTile = 256 * kFirHalves;
__global__ __launch_bounds__(256)
void ft8_rx_fir_kernel(const int4 *__restrict__ sums, const int4 *__restrict__ base, int nblocks, int ngroups,
                       float *__restrict__ iq, float *__restrict__ peak) {
    __shared__ uint32_t s_p[2][kFirTile + kFirTaps + 4];
    __shared__ float s_y[2][kFirTile + kFirTaps];
    __shared__ float s_max[2][4];
    const int capture = blockIdx.y, nout = nblocks;
    const int4 *cs = sums + (size_t)capture * nblocks, *cb = base + (size_t)capture * ngroups;
    const int k0 = blockIdx.x * kFirTile;
    for (int i = threadIdx.x; i < kFirTile + kFirTaps + 4; i += 256) {
        const int k = k0 - kFirTaps - 4 + i;
        uint32_t pI = 0u, pQ = 0u;                           // integrator state before the capture is 0
        if (k >= 0 && k < nout) {
            // second integrator after block k: group entry state + local running value
            const int4 b = cb[k >> 4], l = cs[k];
            const uint32_t w = (uint32_t)kR * (uint32_t)((k & 15) + 1);
            pI = (uint32_t)b.y + w * (uint32_t)b.x + (uint32_t)l.y;
            pQ = (uint32_t)b.w + w * (uint32_t)b.z + (uint32_t)l.w;
        }
        s_p[0][i] = pI;
        s_p[1][i] = pQ;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kFirTile + kFirTaps; i += 256) {
        const int k = k0 - kFirTaps + i;                     // comb output index
        const bool in = k >= 0 && k < nout;                  // FIR history starts at zero (:113-114)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const uint32_t a = s_p[ch][i + 4], b = s_p[ch][i + 2], c = s_p[ch][i];
            s_y[ch][i] = in ? (float)(int32_t)((a - b) - (b - c)) : 0.0f;   // Iy2 = (Ix2 - z^-2) - (... z^-2), wrapping
        }
    }
    __syncthreads();
    float acc[kFirHalves][2] = {};   // [half of the tile][channel]
#pragma unroll
    for (int j = 0; j <= kFirTaps; ++j) {                    // :181-192, oldest first
        const float cj = c_zCoef[j];
#pragma unroll
        for (int r = 0; r < kFirHalves; ++r)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) acc[r][ch] += s_y[ch][threadIdx.x + 256 * r + j] * cj;
    }
    float m[2] = { 0.0f, 0.0f };
#pragma unroll
    for (int r = 0; r < kFirHalves; ++r) {
        const int k = k0 + threadIdx.x + 256 * r;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const float v = k < nout ? (float)((double)acc[r][ch] / (32768.0 * 750)) : 0.0f;
            if (k < kNSamples) iq[((size_t)capture * 2 + ch) * kNSamples + k] = v;
            m[ch] = fmaxf(m[ch], fabsf(v));
        }
    }
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        for (int o = 32; o > 0; o >>= 1) m[ch] = fmaxf(m[ch], __shfl_xor(m[ch], o, 64));
        if ((threadIdx.x & 63) == 0) s_max[ch][threadIdx.x >> 6] = m[ch];
    }
    __syncthreads();
    if (threadIdx.x < 2)                                     // one partial peak per workgroup and channel (no atomics)
        peak[((size_t)capture * 2 + threadIdx.x) * gridDim.x + blockIdx.x] =
            fmaxf(fmaxf(s_max[threadIdx.x][0], s_max[threadIdx.x][1]), fmaxf(s_max[threadIdx.x][2], s_max[threadIdx.x][3]));
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

// raw: [ncaptures][2*npairs] bytes on the device; scratch_sums: 16 B per block; scratch_base: 2 x 16 B per
// 16-block group (entry states, totals) followed by up to 376 partial peaks (float) per capture (sized by ft8gpu_rx_decimate)
hipError_t launch_rx(const uint8_t *raw, int ncaptures, size_t npairs, void *scratch_sums, void *scratch_base,
                     float *iq, int normalise, hipStream_t s) {
    if (ncaptures < 1) return hipSuccess;
    const size_t nb = npairs / kR;
    const int nblocks = (int)(nb > (size_t)kNSamples ? (size_t)kNSamples : nb);   // samples past 48000 are dropped (:196)
    const int ngroups = (nblocks + 15) / 16;
    if (ngroups > 1024 * kScanPer) return hipErrorInvalidValue;            // (48000 outputs at most: 3000 groups)
    int4 *base = (int4 *)scratch_base;
    int4 *gtot = base + (size_t)ncaptures * (ngroups > 0 ? ngroups : 1);
    float *peak = (float *)(gtot + (size_t)ncaptures * (ngroups > 0 ? ngroups : 1));
    constexpr int kFirGrid = (kNSamples + kFirTile - 1) / kFirTile;   // 188 workgroups per capture
    if (nblocks > 0) {
        hipLaunchKernelGGL(ft8_rx_block_kernel, dim3(ngroups, ncaptures), dim3(256), 0, s,
                           raw, npairs * 2, nblocks, (int4 *)scratch_sums, gtot);
        hipLaunchKernelGGL(ft8_rx_scan_kernel, dim3(ncaptures), dim3(1024), 0, s,
                           (const int4 *)gtot, nblocks, ngroups, base);
    }
    hipLaunchKernelGGL(ft8_rx_fir_kernel, dim3(kFirGrid, ncaptures), dim3(256), 0, s,
                       (const int4 *)scratch_sums, (const int4 *)base, nblocks, ngroups, iq, peak);
    if (normalise)
        hipLaunchKernelGGL(ft8_rx_normalise_kernel, dim3((2 * kNSamples / 4 + 255) / 256, ncaptures), dim3(256), 0, s, iq, peak, 2 * kFirGrid);
    return hipGetLastError();
}
