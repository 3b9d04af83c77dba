// quad_transpose.h -- 4 x 4 transpose of four VGPRs across the four lanes of every quad, without LDS.
// Afterwards register m of lane l holds what register (l & 3) of lane (l & ~3) + m held.  Two butterfly steps
// (lane bit 0 against register bit 0, lane bit 1 against register bit 1), each value moved by ONE instruction:
// v_cndmask_b32_dpp = select between the lane's own register and the quad-permuted register of its partner.
// (The compiler's own lowering of __builtin_amdgcn_update_dpp + select is v_mov + v_mov_dpp + v_cndmask.)
// Hazards are handled here because inline assembly is invisible to the hazard recogniser: a DPP operand must
// not be read within two wait states of the VALU instruction that wrote it.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void quad_transpose4(float &r0, float &r1, float &r2, float &r3) {
    float n0, n1, n2, n3;
    asm volatile(
        "s_nop 1\n\t"                                                                                  // producers outside the block
        "s_mov_b64 vcc, %8\n\t"                                                                        // lanes 1, 3 of a quad
        "v_cndmask_b32_dpp %5, %0, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"         // n1 = bit0 ? r1 : partner's r0
        "v_cndmask_b32_dpp %7, %2, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"         // n3 = bit0 ? r3 : partner's r2
        "s_mov_b64 vcc, %9\n\t"                                                                        // lanes 0, 2
        "v_cndmask_b32_dpp %4, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"         // n0 = !bit0 ? r0 : partner's r1
        "v_cndmask_b32_dpp %6, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"         // n2 = !bit0 ? r2 : partner's r3
        "s_mov_b64 vcc, %10\n\t"                                                                       // lanes 2, 3
        "s_nop 0\n\t"
        "v_cndmask_b32_dpp %3, %5, %7, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"         // r3 = bit1 ? n3 : partner's n1
        "v_cndmask_b32_dpp %2, %4, %6, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"         // r2 = bit1 ? n2 : partner's n0
        "s_mov_b64 vcc, %11\n\t"                                                                       // lanes 0, 1
        "v_cndmask_b32_dpp %1, %7, %5, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"         // r1 = !bit1 ? n1 : partner's n3
        "v_cndmask_b32_dpp %0, %6, %4, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"         // r0 = !bit1 ? n0 : partner's n2
        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)
        : "s"(0xAAAAAAAAAAAAAAAAull), "s"(0x5555555555555555ull), "s"(0xCCCCCCCCCCCCCCCCull), "s"(0x3333333333333333ull)
        : "vcc");
}

// two such transposes (the real and the imaginary parts of four complex registers) interleaved in one block: the four
// mask moves are shared and every instruction is at least three issue slots behind the one that produced its operand
__device__ __forceinline__ void quad_transpose4x2(float &r0, float &r1, float &r2, float &r3, float &q0, float &q1, float &q2, float &q3) {
    float n0, n1, n2, n3, p0, p1, p2, p3;
    asm volatile(
        "s_nop 1\n\t"
        "s_mov_b64 vcc, %16\n\t"
        "v_cndmask_b32_dpp %9, %0, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %11, %2, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %13, %4, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %15, %6, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 vcc, %17\n\t"
        "v_cndmask_b32_dpp %8, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %10, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %12, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %14, %7, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 vcc, %18\n\t"
        "v_cndmask_b32_dpp %3, %9, %11, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %7, %13, %15, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %2, %8, %10, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %6, %12, %14, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 vcc, %19\n\t"
        "v_cndmask_b32_dpp %1, %11, %9, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %5, %15, %13, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %0, %10, %8, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %4, %14, %12, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3),
          "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
        : "s"(0xAAAAAAAAAAAAAAAAull), "s"(0x5555555555555555ull), "s"(0xCCCCCCCCCCCCCCCCull), "s"(0x3333333333333333ull)
        : "vcc");
}
