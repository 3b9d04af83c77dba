// synth.hip -- bench/test tooling: synthesises 15 s, 3200 sps planar
// I/Q frames directly in HBM so that large batches need no PCIe traffic.
//   signal: plain CPFSK, 512 samples per symbol, tone spacing 6.25 Hz -- the modulation of
//           decoderSelfTest(), rtlsdr_ft8d.c:946-955 (no Gaussian shaping)
//   noise:  complex AWGN from a counter-based generator (splitmix64 hash + Box-Muller)
//   level:  peak-normalised to 0.5 per frame as the decoder thread does, rtlsdr_ft8d.c:248-263
// Not on the decode path; float results need not match any CPU (frames are copied back for the oracle).
#include "ft8gpu_internal.h"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

constexpr int kMaxSig = 64;

__global__ __launch_bounds__(1024)
void ft8_synth_kernel(const ft8gpu_synth_signal *__restrict__ sigs, int nsig, float noise_sigma,
                      uint64_t seed, uint64_t first_frame, float *__restrict__ iq) {
    __shared__ int s_cum[kMaxSig][80];        // prefix sums of tone numbers
    __shared__ float s_max[1024 / 64];
    __shared__ float s_scale;

    const int frame = blockIdx.x, tid = threadIdx.x;
    const ft8gpu_synth_signal *fs = sigs + (size_t)frame * nsig;
    for (int s = tid; s < nsig; s += blockDim.x) {
        int acc = 0;
        for (int k = 0; k < FT8GPU_NN; ++k) { s_cum[s][k] = acc; acc += fs[s].tones[k]; }
        s_cum[s][79] = acc;
    }
    __syncthreads();

    float *fI = iq + (size_t)frame * 2 * kNSamples;
    float *fQ = fI + kNSamples;
    float peak = 0.0f;
    for (int i = tid; i < kNSamples; i += blockDim.x) {
        // keyed on the GLOBAL frame index: frame g of a job gets the same noise whichever rank / shard / batch
        // position synthesises it (first_frame = global index of this launch's frame 0)
        const uint64_t h = splitmix64(seed ^ splitmix64(((first_frame + (uint64_t)frame) << 20) + (uint64_t)i));
        const float u1 = ((float)(uint32_t)(h >> 40) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
        const float u2 = (float)(uint32_t)((h >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);
        const float rad = noise_sigma * sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincospif(2.0f * u2, &sn, &cs);
        float vi = rad * cs, vq = rad * sn;
        for (int s = 0; s < nsig; ++s) {
            const int start = (int)lrintf(fs[s].t0_s * 3200.0f);
            const int n = i - start;
            if (n < 0 || n >= FT8GPU_NN * 512) continue;
            const int sym = n >> 9, within = n & 511;
            const int tone = s_cum[s][sym + 1 > 79 ? 79 : sym + 1] - s_cum[s][sym];
            // phase in cycles: f0*n/3200 + 6.25/3200 * (512*cum[sym] + tone*within)
            const double cyc = (double)fs[s].f0_hz * (double)n * (1.0 / 3200.0) +
                               (double)(512 * s_cum[s][sym] + tone * within) * (6.25 / 3200.0);
            const float frac = (float)(cyc - floor(cyc));
            float ss, cc;
            sincospif(2.0f * frac, &ss, &cc);
            vi += fs[s].amplitude * cc;
            vq += fs[s].amplitude * ss;
        }
        fI[i] = vi;
        fQ[i] = vq;
        peak = fmaxf(peak, fmaxf(fabsf(vi), fabsf(vq)));
    }
    for (int o = 32; o > 0; o >>= 1) peak = fmaxf(peak, __shfl_xor(peak, o, 64));
    if ((tid & 63) == 0) s_max[tid >> 6] = peak;
    __syncthreads();
    if (tid == 0) {
        float m = 1e-24f;                                    // rtlsdr_ft8d.c:249
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, s_max[w]);
        s_scale = 0.5f / m;                                  // :259
    }
    __syncthreads();
    const float scale = s_scale;
    for (int i = tid; i < kNSamples; i += blockDim.x) {      // :260-263
        fI[i] *= scale;
        fQ[i] *= scale;
    }
}

}  // namespace

hipError_t launch_synth(const ft8gpu_synth_signal *sig_dev, int nframes, int nsig, float noise_sigma,
                        uint64_t seed, uint64_t first_frame, float *iq, hipStream_t s) {
    if (nframes < 1) return hipSuccess;
    if (nsig > kMaxSig) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ft8_synth_kernel, dim3(nframes), dim3(1024), 0, s, sig_dev, nsig, noise_sigma, seed, first_frame, iq);
    return hipGetLastError();
}
