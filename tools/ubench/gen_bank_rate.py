#!/usr/bin/env python3
"""Generates tools/ubench/bank_rate.hip: issue cost of f32 VALU instructions on gfx950 as a function of the VGPR banks
(index mod 4) of their operands.  Every variant is a kernel whose loop body is 16 independent instructions on explicitly
named registers: destinations v40..v55 (scalar) or v[40:41]..v[70:71] (packed), constant sources v72..v83.
  python tools/ubench/gen_bank_rate.py && hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bank_rate tools/ubench/bank_rate.hip"""
import os

CLOB = ",".join(f'"v{i}"' for i in range(40, 84))
variants = []       # (name, [16 instruction strings])


def scalar(name, fmt, offs):
    """fmt uses {d} {a} {b} {c}; d = v(40+i) (bank i mod 4); sources from v72..v83 with bank (i + off) mod 4"""
    ins = []
    for i in range(16):
        src = lambda k, off: f"v{72 + 4 * k + ((i + off) % 4)}"
        ins.append(fmt.format(d=f"v{40 + i}", a=src(0, offs[0]), b=src(1, offs[1]), c=src(2, offs[2])))
    variants.append((name, ins))


def packed(name, fmt, offs):
    """d = v[40+2i : 41+2i] (pair bank class i mod 2); sources pairs from v[72..83] with class (i + off) mod 2"""
    ins = []
    for i in range(16):
        def src(k, off):
            base = 72 + 4 * k + 2 * ((i + off) % 2)
            return f"v[{base}:{base + 1}]"
        d = f"v[{40 + 2 * i}:{41 + 2 * i}]"
        ins.append(fmt.format(d=d, a=src(0, offs[0]), b=src(1, offs[1]), c=src(2, offs[2])))
    variants.append((name, ins))


# VOP2, in place: d = d * b
scalar("v_mul_f32 d,d,b   b same bank", "v_mul_f32 {d}, {d}, {a}", (0, 0, 0))
scalar("v_mul_f32 d,d,b   b other bank", "v_mul_f32 {d}, {d}, {a}", (1, 0, 0))
scalar("v_add_f32 d,d,b   b other bank", "v_add_f32 {d}, {d}, {a}", (1, 0, 0))
scalar("v_mul_f32 d,a,b   a,b same bank as d", "v_mul_f32 {d}, {a}, {b}", (0, 0, 0))
scalar("v_mul_f32 d,a,b   a,b one bank, d another", "v_mul_f32 {d}, {a}, {b}", (1, 1, 0))
scalar("v_mul_f32 d,a,b   all three banks differ", "v_mul_f32 {d}, {a}, {b}", (1, 2, 0))
# VOP2 fmac: d += a * b
scalar("v_fmac_f32 d,a,b  all same bank", "v_fmac_f32 {d}, {a}, {b}", (0, 0, 0))
scalar("v_fmac_f32 d,a,b  a,b one bank, d another", "v_fmac_f32 {d}, {a}, {b}", (1, 1, 0))
scalar("v_fmac_f32 d,a,b  all differ", "v_fmac_f32 {d}, {a}, {b}", (1, 2, 0))
# VOP3 fma: d = d * a + b (in place) and d = a * b + c
scalar("v_fma_f32 d,d,a,b  all same bank", "v_fma_f32 {d}, {d}, {a}, {b}", (0, 0, 0))
scalar("v_fma_f32 d,d,a,b  a,b one bank, d another", "v_fma_f32 {d}, {d}, {a}, {b}", (1, 1, 0))
scalar("v_fma_f32 d,d,a,b  all differ", "v_fma_f32 {d}, {d}, {a}, {b}", (1, 2, 0))
scalar("v_fma_f32 d,a,b,c  sources all differ, d = 4th bank", "v_fma_f32 {d}, {a}, {b}, {c}", (1, 2, 3))
scalar("v_fma_f32 d,a,b,c  sources one bank", "v_fma_f32 {d}, {a}, {b}, {c}", (1, 1, 1))
scalar("v_fma_f32 d,d,a,1.0  constant addend", "v_fma_f32 {d}, {d}, {a}, 1.0", (1, 0, 0))
# VOP3 with modifiers, the clamp's bfi, min3
scalar("v_bfi_b32 d,a,1.0,d  a other bank", "v_bfi_b32 {d}, {a}, 1.0, {d}", (1, 0, 0))
scalar("v_min3_f32 d,|d|,|a|,|b| all differ", "v_min3_f32 {d}, |{d}|, |{a}|, |{b}|", (1, 2, 0))
scalar("v_rcp_f32 d,d", "v_rcp_f32 {d}, {d}", (0, 0, 0))
# packed
packed("v_pk_mul_f32 d,d,b  b same class", "v_pk_mul_f32 {d}, {d}, {a}", (0, 0, 0))
packed("v_pk_mul_f32 d,d,b  b other class", "v_pk_mul_f32 {d}, {d}, {a}", (1, 0, 0))
packed("v_pk_mul_f32 d,a,b  a,b same class as d", "v_pk_mul_f32 {d}, {a}, {b}", (0, 0, 0))
packed("v_pk_mul_f32 d,a,b  a other, b same as d", "v_pk_mul_f32 {d}, {a}, {b}", (1, 0, 0))
packed("v_pk_mul_f32 d,a,b  a,b other class than d", "v_pk_mul_f32 {d}, {a}, {b}", (1, 1, 0))
packed("v_pk_add_f32 d,d,b  b other class", "v_pk_add_f32 {d}, {d}, {a}", (1, 0, 0))
packed("v_pk_fma_f32 d,d,a,b all same class", "v_pk_fma_f32 {d}, {d}, {a}, {b}", (0, 0, 0))
packed("v_pk_fma_f32 d,d,a,b a other, b same", "v_pk_fma_f32 {d}, {d}, {a}, {b}", (1, 0, 0))
packed("v_pk_fma_f32 d,d,a,b a,b other", "v_pk_fma_f32 {d}, {d}, {a}, {b}", (1, 1, 0))
packed("v_pk_fma_f32 d,a,b,c sources alternate", "v_pk_fma_f32 {d}, {a}, {b}, {c}", (0, 1, 0))
packed("v_pk_mul_f32 d,d,b op_sel_hi:[1,0] b other", "v_pk_mul_f32 {d}, {d}, {a} op_sel_hi:[1,0]", (1, 0, 0))

# dependency vs in-place: d_i = d_(i+1) * b reads a register another instruction of the block writes (a chain through the
# loop), but never its own destination
def chained(name, op, pk):
    ins = []
    for i in range(16):
        j = (i + 1) % 16
        if pk:
            ins.append(f"{op} v[{40 + 2 * i}:{41 + 2 * i}], v[{40 + 2 * j}:{41 + 2 * j}], v[{72 + 2 * ((i + 1) % 2)}:{73 + 2 * ((i + 1) % 2)}]")
        else:
            ins.append(f"{op} v{40 + i}, v{40 + j}, v{72 + (i + 1) % 4}")
    variants.append((name, ins))


chained("v_mul_f32 d_i = d_(i+1) * b   (chained, not in place)", "v_mul_f32", False)
chained("v_pk_mul_f32 d_i = d_(i+1) * b   (chained, not in place)", "v_pk_mul_f32", True)
chained("v_pk_add_f32 d_i = d_(i+1) + b   (chained, not in place)", "v_pk_add_f32", True)

# which operand position makes the in-place form dearer?
scalar("v_mul_f32 d,b,d   (dst == src1)", "v_mul_f32 {d}, {a}, {d}", (1, 0, 0))
scalar("v_mul_f32_e64 d,d,b  (VOP3 encoding, dst == src0)", "v_mul_f32_e64 {d}, {d}, {a}", (1, 0, 0))
scalar("v_mul_f32_e64 d,a,b  (VOP3 encoding, three-address)", "v_mul_f32_e64 {d}, {a}, {b}", (1, 2, 0))
scalar("v_fma_f32 d,a,d,b  (dst == src1)", "v_fma_f32 {d}, {a}, {d}, {b}", (1, 2, 0))
scalar("v_fma_f32 d,a,b,d  (dst == src2)", "v_fma_f32 {d}, {a}, {b}, {d}", (1, 2, 0))
packed("v_pk_mul_f32 d,b,d  (dst == src1)", "v_pk_mul_f32 {d}, {a}, {d}", (1, 0, 0))
packed("v_pk_fma_f32 d,a,b,d  (dst == src2)", "v_pk_fma_f32 {d}, {a}, {b}, {d}", (1, 0, 0))
packed("v_pk_add_f32 d,a,b  three-address", "v_pk_add_f32 {d}, {a}, {b}", (1, 0, 0))

src = ['// GENERATED by tools/ubench/gen_bank_rate.py -- do not edit', '#include <hip/hip_runtime.h>', '#include <stdio.h>',
       '#define N_ITERS 4096', f'#define CLOB {CLOB}', '']
for k, (name, ins) in enumerate(variants):
    body = "\\n".join(ins)
    src.append(f'__global__ __launch_bounds__(256) void k{k}(float *out, float seed) {{')
    src.append('    asm volatile("' + "\\n".join(f"v_mov_b32 v{r}, %0" for r in range(40, 72)) + '" :: "v"(seed + threadIdx.x * 1e-3f) : CLOB);')
    src.append('    asm volatile("' + "\\n".join(f"v_mov_b32 v{r}, %0" for r in range(72, 84)) + '" :: "v"(0.99999f) : CLOB);')
    src.append('    for (int i = 0; i < N_ITERS; ++i) asm volatile("' + body + '" ::: CLOB);')
    src.append('    float acc; asm volatile("v_add_f32 %0, v40, v41\\nv_add_f32 %0, %0, v55\\nv_add_f32 %0, %0, v70\\nv_add_f32 %0, %0, v71" : "=v"(acc) :: CLOB);')
    src.append('    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;')
    src.append('}')
src.append('''
typedef void (*kern_t)(float *, float);
static float time_ms(kern_t f, int blocks, float *out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}
int main() {''')
src.append('    kern_t ks[] = { ' + ", ".join(f"k{k}" for k in range(len(variants))) + ' };')
src.append('    const char *names[] = { ' + ", ".join('"' + n + '"' for n, _ in variants) + ' };')
src.append(f'''    const int nv = {len(variants)};
    float *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int wps_list[] = {{ 8, 4 }};
    for (int w = 0; w < 2; ++w) {{
        const int wps = wps_list[w], blocks = 256 * wps;
        const double winstr = (double)blocks * 4 * N_ITERS * 16;
        float t[64];
        for (int v = 0; v < nv; ++v) t[v] = time_ms(ks[v], blocks, out);
        printf("waves per SIMD %d   (slot = v_mul_f32 d,d,b with b on another bank)\\n", wps);
        for (int v = 0; v < nv; ++v) printf("  %-52s %7.3f ms %8.1f G wave-instr/s %5.2f slots\\n", names[v], t[v], winstr / t[v] / 1e6, t[v] / t[1]);
    }}
    (void)hipFree(out);
    return 0;
}}''')
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bank_rate.hip"), "w").write("\n".join(src) + "\n")
print(len(variants), "variants")
