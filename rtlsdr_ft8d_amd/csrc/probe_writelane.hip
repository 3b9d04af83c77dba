// Build-time probe (Makefile: build/.writelane_ok).  sync.hip keeps the exact-heap replay in VGPRs and moves
// wave-uniform values into a lane with v_writelane_b32.  This hipcc declares no builtin for that instruction, so the
// LLVM intrinsic is bound by its name through an asm label; if a compiler update renames the intrinsic (or starts
// declaring __builtin_amdgcn_writelane), this translation unit stops producing a v_writelane_b32 and the build fails
// HERE, with a message that says what to change, instead of somewhere inside the kernels.
#include <hip/hip_runtime.h>
extern "C" __device__ int ft8_probe_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
__global__ void ft8_probe_writelane(int *out, int value, int lane) {
    int x = (int)threadIdx.x;
    x = ft8_probe_writelane_i32(__builtin_amdgcn_readfirstlane(value), __builtin_amdgcn_readfirstlane(lane) & 63, x);
    out[threadIdx.x] = x;
}
