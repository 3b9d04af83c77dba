// test of the fused select + quad-permute 4x4 transpose used by the waterfall kernel's last stage
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../rtlsdr_ft8d_amd/csrc/quad_transpose.h"
__global__ void k(float *o, const float *i) {
    float r0 = i[threadIdx.x], r1 = i[64 + threadIdx.x], r2 = i[128 + threadIdx.x], r3 = i[192 + threadIdx.x];
    r0 = r0 * 1.0f; r1 = r1 + 0.0f;       // produced by VALU right before (hazard path)
    float q0 = r0 + 5000.0f, q1 = r1 + 5000.0f, q2 = r2 + 5000.0f, q3 = r3 + 5000.0f, s0 = r0, s1 = r1, s2 = r2, s3 = r3;
    quad_transpose4(r0, r1, r2, r3);
    quad_transpose4x2(s0, s1, s2, s3, q0, q1, q2, q3);
    if (s0 != r0 || s1 != r1 || s2 != r2 || s3 != r3 || q0 != r0 + 5000.0f || q1 != r1 + 5000.0f || q2 != r2 + 5000.0f || q3 != r3 + 5000.0f) r0 = -1.0f;
    o[threadIdx.x] = r0; o[64 + threadIdx.x] = r1; o[128 + threadIdx.x] = r2; o[192 + threadIdx.x] = r3;
}
int main() {
    float h[256], g[256], *di, *dout;
    for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) h[64 * r + l] = 1000.0f * r + l;     // value encodes (register, lane)
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof h);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di);
    hipMemcpy(g, dout, sizeof g, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 4; ++m) for (int l = 0; l < 64; ++l) {
        const int j = l & 3, src_lane = (l & ~3) + m;          // new register m of lane l = register j of lane (quad base + m)
        const float want = 1000.0f * j + src_lane;
        if (g[64 * m + l] != want) { if (bad < 8) printf("reg %d lane %d: got %g want %g\n", m, l, g[64 * m + l], want); ++bad; }
    }
    printf("quad_transpose4: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    return bad != 0;
}
