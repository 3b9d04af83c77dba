// waterfall.hip -- stage a2+a3 of the hot path:
//   rtlsdr_ft8d.c:1401-1411  window + 184 complex 1024-point FFTs per frame
//   rtlsdr_ft8d.c:1413-1433  |X|^2 -> 10*log10 -> (int)(2*dB+240) clamp -> uint8, OSR de-interleave
//
// Design (gfx950, wave64):
//   * one wave per FFT row, 16 complex points per lane, four waves (four consecutive rows) per workgroup.  Every wave
//     reads the 1024 samples of its row straight from global memory (32 coalesced 256-byte loads); the 75 % overlap
//     of consecutive rows is served by the CU's vector L1, the overlap between workgroups by the XCD's L2 (xcd_item).
//   * 1024 = 16 x 16 x 4: two radix-4 stages in registers, exchange through a padded (conflict-free) per-wave LDS
//     buffer, two more radix-4 stages, then the last radix-4 stage, of which only the outputs that land in bins
//     0..511 are computed.  The inputs of that last stage differ in the LOWEST digit of the point index only, and the
//     second layout puts that digit into the lane's ROW number (lane >> 4), so the second exchange is a 4 x 4
//     transpose of registers across the four 16-lane rows of the wave: v_permlane16_swap + v_permlane32_swap (gfx950),
//     one instruction per register, no LDS.  (The form with that exchange through LDS is kept behind a per-context
//     debug flag: bit-identical, measured in DESIGN.md.)
//   * stage-0 twiddles live in registers, those of stages 1-3 (multiples of 4) in a 2 KB LDS table, the window is
//     re-read from L1 per row: 37.9 KB of LDS and at most 128 VGPRs, i.e. four workgroups per CU.
//   * the dB quantiser is evaluated against a 256-entry threshold table derived on the host from
//     the reference expression itself (host libm log10f), so the uint8 result is bit-identical to
//     (int)(2*(10.0f*log10f(1e-12f + mag2*4.0f/(NFFT*NFFT)))+240); v_log_f32 only provides a
//     guess known to be the result or one below it.
//   * arithmetic order of the FFT is the "R4DIF-1024" order documented in DESIGN.md; compiled with
//     -ffp-contract=off so every float operation is a single IEEE operation.
#include "ft8gpu_internal.h"
#include <stdlib.h>

namespace {

// complex numbers as packed float2 (x = re, y = im): complex add/sub is one v_pk_add_f32, the
// complex product two v_pk_mul_f32 + one v_pk_add_f32 (gfx950 issues packed f32 at ~1.6x the rate
// of the scalar form).  The individual roundings are those of the R4DIF-1024 specification.
typedef float c32 __attribute__((ext_vector_type(2)));

// The half-swaps and sign flips these need are free operand modifiers of the VOP3P encoding
// (op_sel / op_sel_hi pick the source half per result half, neg_lo / neg_hi negate it); hipcc
// materialises them as v_mov/v_xor, so the four swizzled forms are written in assembly.
// t + (-i)*u = (t.x + u.y, t.y - u.x)
__device__ __forceinline__ c32 add_mul_mi(c32 t, c32 u) {
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(t), "v"(u));
    return r;
}
// t + (+i)*u = (t.x - u.y, t.y + u.x)
__device__ __forceinline__ c32 add_mul_pi(c32 t, c32 u) {
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(t), "v"(u));
    return r;
}
// (yr*wr - yi*wi, yr*wi + yi*wr): four products, one difference, one sum -- no fused operations
__device__ __forceinline__ c32 cmul(c32 y, float2 w) {
    const c32 ww = { w.x, w.y };
    c32 p, q, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(p) : "v"(y), "v"(ww));   // (yr*wr, yi*wr)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(q) : "v"(y), "v"(ww));   // (yi*wi, yr*wi)
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(p), "v"(q));                    // (p.x - q.x, p.y + q.y)
    return r;
}

// radix-4 DIF butterfly, forward transform: y1 = t1 - i*t3, y3 = t1 + i*t3
__device__ __forceinline__ void bfly4(c32 &a0, c32 &a1, c32 &a2, c32 &a3) {
    const c32 t0 = a0 + a2;
    const c32 t1 = a0 - a2;
    const c32 t2 = a1 + a3;
    const c32 t3 = a1 - a3;
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = add_mul_mi(t1, t3);
    a3 = add_mul_pi(t1, t3);
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// rtlsdr_ft8d.c:1415-1427 for two bins (y0 and y1 of a stage-4 butterfly); qthr[k] = smallest float y with
// quantised value >= k (qthr[0] = 0, qthr[256] = NaN: never compares true).
//   * y = 1e-12f + (mag2 * 4.0f) / 1048576.0f: the product and the division by 2^20 are exact scalings, so one
//     multiplication by 2^-18 gives the same y whenever neither intermediate leaves the normal range; a
//     subnormal quotient (< 1.2e-38) vanishes in 1e-12f either way, and where mag2 * 4 overflows to inf the
//     reference quantises inf (255 by the fence of DESIGN.md) while this y stays finite but far above
//     qthr[255] -- 255 as well.
//   * v_log_f32 (1 ulp) puts g = 6.0206 log2(y) + 240 within 1e-4 of the reference's float expression
//     2 * (10 * log10f(y)) + 240, whose truncation is the result q.  With the guess biased DOWN by 0.01,
//     kl = trunc(g - 0.01) satisfies kl <= q <= kl + 1, and q = kl + 1 exactly when y >= qthr[kl + 1]: one table
//     read and one comparison per bin, branch-free.  (y = +inf: the guess saturates, kl = 255, NaN never
//     compares, 255; y = NaN: the guess converts to 0, the comparison fails, 0.)
// Written on pairs so that the scalings and the affine map of the logarithm are packed instructions.
__device__ __forceinline__ void quantise2(c32 a, c32 b, const float *qthr, unsigned &qa, unsigned &qb) {
    const c32 sa = a * a, sb = b * b;
    c32 mag2;                           // horizontal adds, written opaquely: the vectoriser otherwise transposes the two pairs with three moves
    asm("v_add_f32 %0, %1, %2" : "=v"(mag2.x) : "v"(sa.x), "v"(sa.y));
    asm("v_add_f32 %0, %1, %2" : "=v"(mag2.y) : "v"(sb.x), "v"(sb.y));
    const c32 y = mag2 * c32{ 0x1p-18f, 0x1p-18f } + c32{ 1E-12f, 1E-12f };
    const c32 l = { __log2f(y.x), __log2f(y.y) };
    const c32 g = l * c32{ 6.0206f, 6.0206f } + c32{ 239.99f, 239.99f };
    int ka = (int)g.x, kb = (int)g.y;
    ka = ka > 255 ? 255 : ka;
    kb = kb > 255 ? 255 : kb;
    qa = (unsigned)ka + (y.x >= qthr[ka + 1] ? 1u : 0u);
    qb = (unsigned)kb + (y.y >= qthr[kb + 1] ? 1u : 0u);
}

constexpr int kXbuf = 1088;     // 1024 + 4 per 64 padding, complex entries per wave

// XCD-aware work order.  The dispatcher places workgroup w on XCD w % 8 (each XCD has its own L2).
// Consecutive chunks of a frame share 768 of their samples, so every XCD is given whole frames
// (frame f belongs to XCD f % 8) and its workgroups walk that XCD's chunks in order, which keeps the
// overlap re-read inside one L2.  step = k-th item of this workgroup; returns the flat item index
// frame * kWfItemsPerFrame + chunk, or -1 when the XCD has no more work.  (Placement affects cache traffic only.)
__device__ __forceinline__ int xcd_item(int step, int nframes) {
    const int xcd = blockIdx.x & 7, lw = blockIdx.x >> 3, wgs = gridDim.x >> 3;
    const int j = lw + wgs * step;                            // position in this XCD's chunk list
    const int fx = j / kWfItemsPerFrame, chunk = j - fx * kWfItemsPerFrame;
    const int frame = fx * 8 + xcd;
    return frame < nframes ? frame * kWfItemsPerFrame + chunk : -1;
}

// How the four inputs of a last-stage butterfly reach one lane (template parameter of the kernel):
//   kStage4Rows  the product.  The second layout keeps the lowest digit of the point index in the lane's ROW number:
//                lane (j, b) = (lane >> 4, lane & 15) holds point 64 b + j + 4 a in register a, so the inputs 4c .. 4c+3 of
//                butterfly c = 16 b + a are register a of lanes b, 16 + b, 32 + b, 48 + b -- a 4 x 4 transpose of
//                registers across the four rows of the wave, which gfx950 does with v_permlane16_swap (odd rows of one
//                register against even rows of another) and v_permlane32_swap (upper half against lower half): one
//                instruction per register, 32 per row of the waterfall, no select masks, no LDS.
//   kStage4Lds   round 3's product: the exchange through LDS (16 ds_write_b64 + 8 ds_read_b128 per row, XOR-swizzled
//                layout below), digit in the low lane bits.  Instantiated in the A/B build only (FT8GPU_AB_WATERFALL_LDS) as an independent
//                mechanism to hold the transposes against.
// Both run the same butterflies in the same order: bit-identical output (test_waterfall_forms_are_bit_identical).
// Measured, profiles/r04_waterfall_forms.json: 0.881 against 0.904 ms per 4096 frames alone, equal inside the pipeline;
// LDS instruction path 0.92 -> 0.60 busy, VALU 0.62 -> 0.71, 433 against 406 VALU instructions per row.  (Round 3's third
// form -- the same transpose inside the quads with fused select + quad permute, 64 instructions and hand-placed hazard
// padding -- took the same time with 36 more instructions per row and was removed.)
constexpr int kStage4Rows = 0, kStage4Lds = 2;

// first exchange: point p of the row sits at complex slot p + PAD * (p >> 6).  Writes are lane-contiguous whatever PAD is;
// the reads of the second layout want 4 b + j (LDS form: PAD 4) or 2 b + j (row form: PAD 2) to run through
// all 32 eight-byte bank pairs within a 32-lane group.
template <int PAD> __device__ __forceinline__ int pad_idx(int p) { return p + PAD * (p >> 6); }

// 4 x 4 transpose of four VGPRs across the four 16-lane rows of the wave: afterwards register m of a lane in row r holds
// what register r held in the lane of row m (same position in the row).  (The compiler's hazard recogniser knows the
// builtins: it puts the two wait states between a VALU write and the swap that reads it.)
__device__ __forceinline__ void row_transpose4(float &r0, float &r1, float &r2, float &r3) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(r0), __float_as_uint(r1), false, false);   // row bit 0 against register bit 0
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(r2), __float_as_uint(r3), false, false);
    const auto c = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);                                   // row bit 1 against register bit 1
    const auto d = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
    r0 = __uint_as_float(c[0]);
    r1 = __uint_as_float(d[0]);
    r2 = __uint_as_float(c[1]);
    r3 = __uint_as_float(d[1]);
}

// Last radix-4 stage + quantiser + staging of the row's 512 output bytes for the row form.  After the transposes lane
// (j, b) holds all four inputs of the butterflies c = 16 b + 4 i + j, i = 0..3, in registers 4i .. 4i+3.  Outputs are in
// digit-reversed order: butterfly c holds bins rev4(c) and 256 + rev4(c), rev4(c) = (b >> 2) + 4 (b & 3) + 16 i + 64 j.
__device__ __forceinline__ void stage4_in_registers(c32 (&x)[16], int b, int j, const float *s_thr, unsigned char *ob) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float r0 = x[4 * i].x, r1 = x[4 * i + 1].x, r2 = x[4 * i + 2].x, r3 = x[4 * i + 3].x;
        float m0 = x[4 * i].y, m1 = x[4 * i + 1].y, m2 = x[4 * i + 2].y, m3 = x[4 * i + 3].y;
        row_transpose4(r0, r1, r2, r3);
        row_transpose4(m0, m1, m2, m3);
        x[4 * i] = c32{ r0, m0 };
        x[4 * i + 1] = c32{ r1, m1 };
        x[4 * i + 2] = c32{ r2, m2 };
        x[4 * i + 3] = c32{ r3, m3 };
    }
    const int kbase = (b >> 2) + 4 * (b & 3) + 64 * j;
    unsigned char *o0 = ob + 256 * (kbase & 1) + (kbase >> 1);            // [freq_sub = k & 1][pos = k >> 1], bins kbase + 16 i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const c32 a0 = x[4 * i], a1 = x[4 * i + 1], a2 = x[4 * i + 2], a3 = x[4 * i + 3];
        const c32 t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = a1 - a3;
        const c32 y0 = t0 + t2;
        const c32 y1 = add_mul_mi(t1, t3);
        unsigned q0, q1;
        quantise2(y0, y1, s_thr, q0, q1);
        o0[8 * i] = (unsigned char)q0;                                    // bin kbase + 16 i
        o0[128 + 8 * i] = (unsigned char)q1;                              // bin 256 + kbase + 16 i
    }
}

__device__ __forceinline__ void bfly_stage_a(c32 (&x)[16], int a, float2 w1, float2 w2, float2 w3) {
    bfly4(x[a], x[a + 4], x[a + 8], x[a + 12]);
    x[a + 4]  = cmul(x[a + 4],  w1);
    x[a + 8]  = cmul(x[a + 8],  w2);
    x[a + 12] = cmul(x[a + 12], w3);
}
__device__ __forceinline__ void bfly_stage_b(c32 (&x)[16], int q, float2 w1, float2 w2, float2 w3) {
    bfly4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
    x[4 * q + 1] = cmul(x[4 * q + 1], w1);
    x[4 * q + 2] = cmul(x[4 * q + 2], w2);
    x[4 * q + 3] = cmul(x[4 * q + 3], w3);
}

// (Round 3 also had a form that staged a work item's 1792 samples in LDS and kept window and twiddles in 76 registers:
// three workgroups per CU, 0.885 ms against 0.891 ms per 4096 frames for this one in interleaved runs -- equal, and it
// left no room for the heap replay beside it; removed in round 4, profiles/r03_waterfall_forms.json has its counters.)
template <int STAGE4>
__global__ __launch_bounds__(256, 4)
void ft8_waterfall_kernel(const float *__restrict__ iq, uint8_t *__restrict__ mag,
                          const Ft8Tables *__restrict__ tab, int nitems, int nframes, int xcd_order) {
    __shared__ __attribute__((aligned(16))) float2 s_x[4][kXbuf];            // per-wave exchange
    __shared__ __attribute__((aligned(16))) float2 s_tw4[256];               // tw[4 k]
    __shared__ __attribute__((aligned(16))) float s_thr[260];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 260; i += 256) s_thr[i] = tab->qthr[i];
    s_tw4[tid] = tab->tw[4 * tid];

    float2 twA1[4][3];                                                       // stage 0: L = 1024, every index occurs
    // second layout: lane (b16, j2) holds point 64 b16 + j2 + 4 a in register a
    constexpr bool kRows = STAGE4 == kStage4Rows;
    constexpr int kPad = kRows ? 2 : 4;
    const int j2 = kRows ? lane >> 4 : lane & 3, b16 = kRows ? lane & 15 : lane >> 2;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 1; q < 4; ++q) twA1[a][q - 1] = tab->tw[(q * (lane + 64 * a)) & 1023];
    __syncthreads();                                                         // tables are in place (the only barrier)

    float2 *xb = s_x[wave];
    // the 512 output bytes of a row are staged at the front of the wave's exchange buffer (a wave's LDS operations
    // execute in order, and every lane has read its inputs out of the buffer by then)
    unsigned char *ob = reinterpret_cast<unsigned char *>(xb);
    // Layout of the SECOND exchange of the LDS form (between stages 3 and 4).  Stage 4 reads the four inputs 4c .. 4c+3 of
    // butterfly c as two 16-byte units; with the inputs contiguous (32 bytes per lane) every ds_read_b128 hits each
    // bank twice.  So the two halves of a butterfly's inputs live in two regions 260 units apart (unit = two complex
    // values = 16 bytes; 260 = 4 mod 8 staggers the regions by half a bank cycle), and the unit index is XOR-swizzled
    // with bits 4..5 of c:  position of element 4c + e = 2 * ((c ^ ((c >> 4) & 3)) + 260 * (e >> 1)) + (e & 1).
    // Both sides are then conflict-free: the 16 lanes of a ds_read_b128 group cover 16 distinct units mod 16, and the
    // 16 lanes of a ds_write_b64 group (four values of c >> 4, four of e) cover all 32 banks.
    constexpr int kHalfUnits = 260;
    int wbase2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wbase2[r] = 32 * b16 + 2 * kHalfUnits * (j2 >> 1) + (j2 & 1) + 2 * (r ^ (b16 & 3));
    const int rbase2 = 2 * (lane ^ ((lane >> 4) & 3));

    int step = 0;
    int item = xcd_order ? xcd_item(0, nframes) : ((int)blockIdx.x < nitems ? (int)blockIdx.x : -1);
    while (item >= 0) {
        const int frame = item / kWfItemsPerFrame;
        const int chunk = item - frame * kWfItemsPerFrame;
        const int row = chunk * kWfRowsPerItem + wave;                       // = 2*idx_block + time_sub
        const float *pI = iq + (size_t)frame * (2 * kNSamples) + row * 256 + lane;
        const float *pQ = pI + kNSamples;

        // window taps: re-read with every row (4 KB, L1-resident) instead of 16 VGPRs for the life of the wave -- with them
        // four waves would fill the SIMD's register file and no other kernel's wave could run beside this one (the
        // offset is always 0; it only keeps the loads inside the loop)
        const float *hann = tab->hann + lane + (item >> 30);
        c32 x[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) x[a] = c32{ pI[64 * a], pQ[64 * a] };   // rtlsdr_ft8d.c:1407-1410
#pragma unroll
        for (int a = 0; a < 16; ++a) { const float w = hann[64 * a]; x[a] = x[a] * c32{ w, w }; }
        // stages 0, 1
#pragma unroll
        for (int a = 0; a < 4; ++a) bfly_stage_a(x, a, twA1[a][0], twA1[a][1], twA1[a][2]);
        {
            const float2 w1 = s_tw4[lane], w2 = s_tw4[(2 * lane) & 255], w3 = s_tw4[(3 * lane) & 255];   // tw[4 q lane]
#pragma unroll
            for (int q = 0; q < 4; ++q) bfly_stage_b(x, q, w1, w2, w3);
        }
#pragma unroll
        for (int a = 0; a < 16; ++a) xb[pad_idx<kPad>(lane + 64 * a)] = make_float2(x[a].x, x[a].y);
        wave_lds_sync();
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const float2 v = xb[pad_idx<kPad>(64 * b16 + j2 + 4 * a)];
            x[a] = c32{ v.x, v.y };
        }
        // stages 2, 3
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int jj = j2 + 4 * a;                                       // tw[16 q jj] = tw4[4 q jj]
            bfly_stage_a(x, a, s_tw4[(4 * jj) & 255], s_tw4[(8 * jj) & 255], s_tw4[(12 * jj) & 255]);
        }
        {
            const float2 w1 = s_tw4[(16 * j2) & 255], w2 = s_tw4[(32 * j2) & 255], w3 = s_tw4[(48 * j2) & 255];   // tw[64 q j2]
#pragma unroll
            for (int q = 0; q < 4; ++q) bfly_stage_b(x, q, w1, w2, w3);
        }
        if (STAGE4 != kStage4Lds) {
            wave_lds_sync();                            // every lane has read its stage-2 inputs out of xb: its front becomes the row's output bytes
            stage4_in_registers(x, b16, j2, s_thr, ob);
        } else {
#pragma unroll
            for (int a = 0; a < 16; ++a) xb[wbase2[a & 3] + 8 * (a >> 2)] = make_float2(x[a].x, x[a].y);
            wave_lds_sync();

            // stage 4 (L = 4, no twiddles): butterfly c = lane + 64*i holds bins k0+i (y0) and 256+k0+i (y1)
            unsigned q0[4], q1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 *src = reinterpret_cast<const float4 *>(xb + rbase2 + 128 * i);
                const float4 v01 = src[0], v23 = src[kHalfUnits];
                const c32 a0 = { v01.x, v01.y }, a1 = { v01.z, v01.w }, a2 = { v23.x, v23.y }, a3 = { v23.z, v23.w };
                const c32 t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = a1 - a3;
                const c32 y0 = t0 + t2;
                const c32 y1 = add_mul_mi(t1, t3);
                quantise2(y0, y1, s_thr, q0[i], q1[i]);
            }
            wave_lds_sync();                            // every lane's stage-4 loads before the stores into the same bytes
            // bins k0..k0+3 -> [freq_sub = k&1][pos = k>>1]  (rtlsdr_ft8d.c:1420-1428)
            const int k0 = 64 * (lane & 3) + 16 * ((lane >> 2) & 3) + 4 * (lane >> 4);
            const int h = k0 >> 1;
            *reinterpret_cast<unsigned short *>(ob + h)             = (unsigned short)(q0[0] | (q0[2] << 8));
            *reinterpret_cast<unsigned short *>(ob + 256 + h)       = (unsigned short)(q0[1] | (q0[3] << 8));
            *reinterpret_cast<unsigned short *>(ob + 128 + h)       = (unsigned short)(q1[0] | (q1[2] << 8));
            *reinterpret_cast<unsigned short *>(ob + 256 + 128 + h) = (unsigned short)(q1[1] | (q1[3] << 8));
        }
        wave_lds_sync();
        uint2 *dst = reinterpret_cast<uint2 *>(mag + (size_t)frame * kMagArray + (size_t)row * 512);
        dst[lane] = reinterpret_cast<const uint2 *>(ob)[lane];
        wave_lds_sync();                                // ob / xb are rewritten by the next row

        ++step;
        if (xcd_order) item = xcd_item(step, nframes);
        else { item += (int)gridDim.x; if (item >= nitems) item = -1; }
    }
}

}  // namespace

// the product is the row-transpose form; the A/B build (FT8GPU_AB_FORMS) also holds the form with the second exchange through LDS
hipError_t launch_waterfall(const float *iq, uint8_t *mag, const Ft8Tables *tab, int nframes,
                            int num_cus, unsigned debug_flags, hipStream_t s) {
    const int nitems = nframes * kWfItemsPerFrame;
    int grid = num_cus * 16;                     // four resident workgroups per CU (LDS- and VGPR-limited), four rounds of them
    if (grid > nitems) grid = nitems;
    if (grid < 1) return hipSuccess;
    // XCD-aware order needs whole groups of 8 workgroups and enough frames to give every XCD work
    const int xo = (grid % 8 == 0 && nframes >= 64) ? 1 : 0;
#ifdef FT8GPU_AB_FORMS
    if (debug_flags & FT8GPU_AB_WATERFALL_LDS) {
        hipLaunchKernelGGL(ft8_waterfall_kernel<kStage4Lds>, dim3(grid), dim3(256), 0, s, iq, mag, tab, nitems, nframes, xo);
        return hipGetLastError();
    }
#endif
    (void)debug_flags;
    hipLaunchKernelGGL(ft8_waterfall_kernel<kStage4Rows>, dim3(grid), dim3(256), 0, s, iq, mag, tab, nitems, nframes, xo);
    return hipGetLastError();
}
