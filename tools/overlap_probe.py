"""Do the front-end kernels overlap with the LDPC kernel when both are in flight on different streams?"""
import sys, time
sys.path.insert(0, '/root/repo')
import torch, numpy as np
import rtlsdr_ft8d_amd as ft8
from rtlsdr_ft8d_amd import workload
B = 2048
_, tones = workload.message_pool()
L = ft8.load_library()
A = ft8.Decoder(device=0, max_frames=B)      # decode stream
F = ft8.Decoder(device=0, max_frames=B)      # front-end stream
sig, _ = workload.frame_signals(0, B, 20, tones)
iq = torch.empty((B, 2, 48000), dtype=torch.float32, device='cuda')
A.synth_frames(sig, B, 20, 1.0, workload.SEED_BASE, iq)
mag = torch.empty((B, 94208), dtype=torch.uint8, device='cuda')
mag2 = torch.empty((B, 94208), dtype=torch.uint8, device='cuda')
cands = torch.zeros((B, 120, 8), dtype=torch.uint8, device='cuda')
counts = torch.zeros((B,), dtype=torch.int32, device='cuda')
cands2 = torch.zeros((B, 120, 8), dtype=torch.uint8, device='cuda')
counts2 = torch.zeros((B,), dtype=torch.int32, device='cuda')
status = torch.zeros((B, 120, 48), dtype=torch.uint8, device='cuda')
D = ft8.DEVICE_PTRS
def wf(dec, out): ft8._check(L.ft8gpu_waterfall(dec.h, iq.data_ptr(), B, out.data_ptr(), D))
def sync(dec, m, c, n): ft8._check(L.ft8gpu_find_sync(dec.h, m.data_ptr(), B, c.data_ptr(), n.data_ptr(), D))
def decode(dec): ft8._check(L.ft8gpu_decode_candidates(dec.h, mag.data_ptr(), cands.data_ptr(), counts.data_ptr(), B, status.data_ptr(), D))
wf(A, mag); sync(A, mag, cands, counts); A.synchronize()
def timed(fn, n=10):
    fn(); A.synchronize(); F.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    A.synchronize(); F.synchronize()
    return (time.perf_counter() - t) / n * 1e3
td = timed(lambda: decode(A))
tw = timed(lambda: wf(F, mag2))
ts = timed(lambda: sync(F, mag2, cands2, counts2))
tws = timed(lambda: (wf(F, mag2), sync(F, mag2, cands2, counts2)))
tb_w = timed(lambda: (decode(A), wf(F, mag2)))
tb_s = timed(lambda: (decode(A), sync(F, mag2, cands2, counts2)))
tb_ws = timed(lambda: (decode(A), wf(F, mag2), sync(F, mag2, cands2, counts2)))
print(f"solo: decode {td:.3f}  wf {tw:.3f}  sync(+heap) {ts:.3f}  wf+sync {tws:.3f}")
print(f"together: decode|wf {tb_w:.3f} (sum {td+tw:.3f})  decode|sync {tb_s:.3f} (sum {td+ts:.3f})  decode|wf+sync {tb_ws:.3f} (sum {td+tws:.3f})")
