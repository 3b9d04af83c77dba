// micro-benchmark: does the issue cost of packed f32 VALU instructions on gfx950 depend on WHERE their operands live?
// (tools/ubench/valu_rate.hip found v_pk_mul_f32 at 1.74 issue slots of a v_mul_f32 with compiler-chosen registers.)
// Every variant runs 16 independent instructions per loop trip on explicitly named VGPR pairs:
//   pk_mul  same : v[d], v[d], v[s]      d and s pairs start at the same index mod 4 (same two register banks)
//   pk_mul  diff : d at 0 mod 4, s at 2 mod 4
//   pk_mul  sgpr : second source an SGPR pair
//   pk_mul  lit  : second source an inline constant
//   pk_fma  ...  : the same for the three-operand form
//   mul / fma    : scalar-f32 forms as the yardstick
#include <hip/hip_runtime.h>
#include <stdio.h>

#define N_ITERS 4096
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79"

// 16 destination pairs v[40:41] .. v[70:71]; sources picked by the variant
#define REP16(OP, S0, S1) \
    OP " v[40:41], v[40:41], " S0 "\n" OP " v[42:43], v[42:43], " S1 "\n" OP " v[44:45], v[44:45], " S0 "\n" OP " v[46:47], v[46:47], " S1 "\n" \
    OP " v[48:49], v[48:49], " S0 "\n" OP " v[50:51], v[50:51], " S1 "\n" OP " v[52:53], v[52:53], " S0 "\n" OP " v[54:55], v[54:55], " S1 "\n" \
    OP " v[56:57], v[56:57], " S0 "\n" OP " v[58:59], v[58:59], " S1 "\n" OP " v[60:61], v[60:61], " S0 "\n" OP " v[62:63], v[62:63], " S1 "\n" \
    OP " v[64:65], v[64:65], " S0 "\n" OP " v[66:67], v[66:67], " S1 "\n" OP " v[68:69], v[68:69], " S0 "\n" OP " v[70:71], v[70:71], " S1 "\n"
#define REP16F(OP, S0, S1, C0, C1) \
    OP " v[40:41], v[40:41], " S0 ", " C0 "\n" OP " v[42:43], v[42:43], " S1 ", " C1 "\n" OP " v[44:45], v[44:45], " S0 ", " C0 "\n" OP " v[46:47], v[46:47], " S1 ", " C1 "\n" \
    OP " v[48:49], v[48:49], " S0 ", " C0 "\n" OP " v[50:51], v[50:51], " S1 ", " C1 "\n" OP " v[52:53], v[52:53], " S0 ", " C0 "\n" OP " v[54:55], v[54:55], " S1 ", " C1 "\n" \
    OP " v[56:57], v[56:57], " S0 ", " C0 "\n" OP " v[58:59], v[58:59], " S1 ", " C1 "\n" OP " v[60:61], v[60:61], " S0 ", " C0 "\n" OP " v[62:63], v[62:63], " S1 ", " C1 "\n" \
    OP " v[64:65], v[64:65], " S0 ", " C0 "\n" OP " v[66:67], v[66:67], " S1 ", " C1 "\n" OP " v[68:69], v[68:69], " S0 ", " C0 "\n" OP " v[70:71], v[70:71], " S1 ", " C1 "\n"
// scalar forms on 16 single registers v40..v55
#define REP16S(OP, S) \
    OP " v40, v40, " S "\n" OP " v41, v41, " S "\n" OP " v42, v42, " S "\n" OP " v43, v43, " S "\n" OP " v44, v44, " S "\n" OP " v45, v45, " S "\n" OP " v46, v46, " S "\n" OP " v47, v47, " S "\n" \
    OP " v48, v48, " S "\n" OP " v49, v49, " S "\n" OP " v50, v50, " S "\n" OP " v51, v51, " S "\n" OP " v52, v52, " S "\n" OP " v53, v53, " S "\n" OP " v54, v54, " S "\n" OP " v55, v55, " S "\n"

#define INIT \
    asm volatile("v_mov_b32 v72, %0\n v_mov_b32 v73, %0\n v_mov_b32 v74, %0\n v_mov_b32 v75, %0\n v_mov_b32 v76, %1\n v_mov_b32 v77, %1\n v_mov_b32 v78, %1\n v_mov_b32 v79, %1\n" :: "v"(0.9999f), "v"(1e-6f) : CLOB); \
    for (int r = 0; r < 1; ++r) asm volatile( \
        "v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n" \
        "v_mov_b32 v48, %0\n v_mov_b32 v49, %0\n v_mov_b32 v50, %0\n v_mov_b32 v51, %0\n v_mov_b32 v52, %0\n v_mov_b32 v53, %0\n v_mov_b32 v54, %0\n v_mov_b32 v55, %0\n" \
        "v_mov_b32 v56, %0\n v_mov_b32 v57, %0\n v_mov_b32 v58, %0\n v_mov_b32 v59, %0\n v_mov_b32 v60, %0\n v_mov_b32 v61, %0\n v_mov_b32 v62, %0\n v_mov_b32 v63, %0\n" \
        "v_mov_b32 v64, %0\n v_mov_b32 v65, %0\n v_mov_b32 v66, %0\n v_mov_b32 v67, %0\n v_mov_b32 v68, %0\n v_mov_b32 v69, %0\n v_mov_b32 v70, %0\n v_mov_b32 v71, %0\n" :: "v"(seed + threadIdx.x) : CLOB)
#define FINI \
    float acc; asm volatile("v_add_f32 %0, v40, v41\n v_add_f32 %0, %0, v55\n v_add_f32 %0, %0, v70\n v_add_f32 %0, %0, v71\n" : "=v"(acc) :: CLOB); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float seed, unsigned long long sconst) {
    INIT;
    for (int i = 0; i < N_ITERS; ++i) {
        // sources: v[72:73] (72 = 0 mod 4: same banks as destinations 40, 44, ...), v[74:75] (2 mod 4)
        if (MODE == 0) asm volatile(REP16("v_pk_mul_f32", "v[72:73]", "v[74:75]") ::: CLOB);           // d 0 mod 4 with s 0 mod 4; d 2 mod 4 with s 2 mod 4: SAME banks
        else if (MODE == 1) asm volatile(REP16("v_pk_mul_f32", "v[74:75]", "v[72:73]") ::: CLOB);      // DIFFERENT banks
        else if (MODE == 2) asm volatile(REP16("v_pk_mul_f32", "%0", "%0") :: "s"(sconst) : CLOB);     // SGPR pair
        else if (MODE == 3) asm volatile(REP16("v_pk_mul_f32", "1.0", "1.0") ::: CLOB);                // inline constant
        else if (MODE == 4) asm volatile(REP16F("v_pk_fma_f32", "v[72:73]", "v[74:75]", "v[76:77]", "v[78:79]") ::: CLOB);   // all three on the same banks
        else if (MODE == 5) asm volatile(REP16F("v_pk_fma_f32", "v[74:75]", "v[72:73]", "v[76:77]", "v[78:79]") ::: CLOB);   // multiplier on the other banks
        else if (MODE == 6) asm volatile(REP16F("v_pk_fma_f32", "%0", "%0", "v[78:79]", "v[76:77]") :: "s"(sconst) : CLOB);  // SGPR multiplier
        else if (MODE == 7) asm volatile(REP16S("v_mul_f32", "v72") ::: CLOB);
        else if (MODE == 8) asm volatile(REP16S("v_fma_f32", "v72, v76") ::: CLOB);
        else if (MODE == 9) asm volatile(REP16("v_pk_add_f32", "v[78:79]", "v[76:77]") ::: CLOB);      // different banks
        else if (MODE == 10) asm volatile(REP16("v_pk_mul_f32", "v[74:75] op_sel_hi:[1,0]", "v[72:73] op_sel_hi:[1,0]") ::: CLOB);   // broadcast the low half of the second source
    }
    FINI;
}

template <typename F>
float time_ms(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    float *out;
    const char *names[] = { "pk_mul same banks", "pk_mul other banks", "pk_mul sgpr src", "pk_mul const src", "pk_fma same banks", "pk_fma mixed banks",
                            "pk_fma sgpr mult", "v_mul_f32", "v_fma_f32", "pk_add other banks", "pk_mul op_sel bcast" };
    for (int wps = 8; wps >= 1; wps /= 2) {
        const int blocks = 256 * wps, threads = 256;          // wps waves per SIMD
        hipMalloc(&out, blocks * threads * sizeof(float));
        const double winstr = (double)blocks * threads / 64 * N_ITERS * 16;
        float t[11];
        const unsigned long long sc = 0x3f7fff583f7fff58ull;
        t[0] = time_ms([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, sc); });
        t[1] = time_ms([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, sc); });
        t[2] = time_ms([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, sc); });
        t[3] = time_ms([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, sc); });
        t[4] = time_ms([&] { hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, sc); });
        t[5] = time_ms([&] { hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, sc); });
        t[6] = time_ms([&] { hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, sc); });
        t[7] = time_ms([&] { hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, sc); });
        t[8] = time_ms([&] { hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, sc); });
        t[9] = time_ms([&] { hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, sc); });
        t[10] = time_ms([&] { hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, sc); });
        printf("waves per SIMD %d\n", wps);
        for (int m = 0; m < 11; ++m)
            printf("  %-20s %8.3f ms  %7.2f G wave-instr/s  %5.2f slots of v_mul_f32\n", names[m], t[m], winstr / t[m] / 1e6, t[m] / t[7]);
        hipFree(out);
    }
    return 0;
}
