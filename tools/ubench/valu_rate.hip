// micro-benchmark: issue rate of scalar vs packed f32 VALU ops and IEEE division on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define N_ITERS 4096

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float a8 = a0 + 8, a9 = a0 + 9, a10 = a0 + 10, a11 = a0 + 11, a12 = a0 + 12, a13 = a0 + 13, a14 = a0 + 14, a15 = a0 + 15;
    const float m = 0.999f, c = 0.001f;
    for (int i = 0; i < N_ITERS; ++i) {
        if (MODE == 0) {          // 16 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                         "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
                         "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                         "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                           "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11), "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15)
                         : "v"(m), "v"(c));
        } else if (MODE == 1) {   // 16 independent v_mul_f32 (no fma)
            asm volatile("v_mul_f32 %0, %0, %16\n v_mul_f32 %1, %1, %16\n v_mul_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n"
                         "v_mul_f32 %4, %4, %16\n v_mul_f32 %5, %5, %16\n v_mul_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n"
                         "v_mul_f32 %8, %8, %16\n v_mul_f32 %9, %9, %16\n v_mul_f32 %10, %10, %16\n v_mul_f32 %11, %11, %16\n"
                         "v_mul_f32 %12, %12, %16\n v_mul_f32 %13, %13, %16\n v_mul_f32 %14, %14, %16\n v_mul_f32 %15, %15, %16\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                           "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11), "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15)
                         : "v"(m));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15;
}

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void kp(float *out, float seed) {
    float2_ a[8];
    for (int j = 0; j < 8; ++j) a[j] = float2_{ seed + threadIdx.x + j, seed + j };
    const float2_ m = { 0.999f, 0.998f }, c = { 0.001f, 0.002f };
    for (int i = 0; i < N_ITERS; ++i) {
        if (MODE == 0) {          // 8 independent v_pk_mul_f32 (= 16 mul lanes)
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m));
        } else {                  // 8 independent v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m), "v"(c));
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += a[j].x + a[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void kdiv(float *out, float seed) {   // 8 independent IEEE divisions per iteration
    float a[8];
    for (int j = 0; j < 8; ++j) a[j] = seed + threadIdx.x * 0.01f + j;
    const float b = 1.0001f + seed;
    for (int i = 0; i < N_ITERS / 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = __fdiv_rn(a[j], b) + 1.0f;
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MODE 0: 16 independent v_rcp_f32; MODE 1: the same interleaved with 16 v_fma_f32 (do they overlap?)
template <int MODE>
__global__ __launch_bounds__(256) void krcp(float *out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = a0 + 8, b1 = a0 + 9, b2 = a0 + 10, b3 = a0 + 11, b4 = a0 + 12, b5 = a0 + 13, b6 = a0 + 14, b7 = a0 + 15;
    const float m = 0.999f, c = 0.001f;
    for (int i = 0; i < N_ITERS; ++i) {
        if (MODE == 0) {
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else {
            asm volatile("v_rcp_f32 %0, %0\n v_fma_f32 %8, %8, %16, %17\n v_rcp_f32 %1, %1\n v_fma_f32 %9, %9, %16, %17\n"
                         "v_rcp_f32 %2, %2\n v_fma_f32 %10, %10, %16, %17\n v_rcp_f32 %3, %3\n v_fma_f32 %11, %11, %16, %17\n"
                         "v_rcp_f32 %4, %4\n v_fma_f32 %12, %12, %16, %17\n v_rcp_f32 %5, %5\n v_fma_f32 %13, %13, %16, %17\n"
                         "v_rcp_f32 %6, %6\n v_fma_f32 %14, %14, %16, %17\n v_rcp_f32 %7, %7\n v_fma_f32 %15, %15, %16, %17\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                           "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
                         : "v"(m), "v"(c));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}

// 16 independent integer / select ops of the kinds the BP loop uses besides the float arithmetic
template <int MODE>
__global__ __launch_bounds__(256) void kint(float *out, float seed) {
    unsigned a[16];
    for (int j = 0; j < 16; ++j) a[j] = (unsigned)(seed * 1000) + threadIdx.x * 7 + j;
    const unsigned m = 0x7fffffffu;
    for (int i = 0; i < N_ITERS; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[j]) : "v"(m));
            else if (MODE == 1) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[j]) : "v"(m));
            else asm volatile("v_bfi_b32 %0, %1, %0, %0" : "+v"(a[j]) : "v"(m));
        }
    }
    unsigned s = 0;
    for (int j = 0; j < 16; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
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
    const int blocks = 256 * 8, threads = 256;      // 8 waves/SIMD
    hipMalloc(&out, blocks * threads * sizeof(float));
    const double lanes = (double)blocks * threads;
    float t;
    t = time_ms([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f); });
    printf("v_fma_f32     : %.3f ms  %.1f Gop/s(lane-instr)  %.1f TFLOP/s\n", t, lanes * N_ITERS * 16 / t / 1e6, lanes * N_ITERS * 16 * 2 / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f); });
    printf("v_mul_f32     : %.3f ms  %.1f Gop/s(lane-instr)\n", t, lanes * N_ITERS * 16 / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(kp<0>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f); });
    printf("v_pk_mul_f32  : %.3f ms  %.1f Gop/s(lane-instr)  %.1f Gmul/s\n", t, lanes * N_ITERS * 8 / t / 1e6, lanes * N_ITERS * 16 / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(kp<1>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f); });
    printf("v_pk_fma_f32  : %.3f ms  %.1f Gop/s(lane-instr)  %.1f TFLOP/s\n", t, lanes * N_ITERS * 8 / t / 1e6, lanes * N_ITERS * 16 * 2 / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL(kdiv, dim3(blocks), dim3(threads), 0, 0, out, 1.0f); });
    printf("IEEE fdiv     : %.3f ms  %.1f Gdiv/s\n", t, lanes * (N_ITERS / 8) * 8 / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(krcp<0>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f); });
    printf("v_rcp_f32     : %.3f ms  %.1f Gop/s(lane-instr)\n", t, lanes * N_ITERS * 8 / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(krcp<1>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f); });
    printf("rcp+fma mixed : %.3f ms  (8 rcp + 8 fma per iteration)\n", t);
    t = time_ms([&] { hipLaunchKernelGGL(kint<0>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f); });
    printf("v_min_u32     : %.3f ms  %.1f Gop/s(lane-instr)\n", t, lanes * N_ITERS * 16 / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(kint<1>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f); });
    printf("v_lshl_add_u32: %.3f ms  %.1f Gop/s(lane-instr)\n", t, lanes * N_ITERS * 16 / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(kint<2>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f); });
    printf("v_bfi_b32     : %.3f ms  %.1f Gop/s(lane-instr)\n", t, lanes * N_ITERS * 16 / t / 1e6);
    hipFree(out);
    return 0;
}
