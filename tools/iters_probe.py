import sys, os
sys.path.insert(0, '/root/repo')
import torch, numpy as np
import rtlsdr_ft8d_amd as ft8
from rtlsdr_ft8d_amd import workload
B=4096
dec = ft8.Decoder(device=0, max_frames=B)
_, tones = workload.message_pool()
sig,_ = workload.frame_signals(0, B, 20, tones)
iq = torch.empty((B,2,48000), dtype=torch.float32, device='cuda')
dec.synth_frames(sig, B, 20, 1.0, workload.SEED_BASE, iq)
spots = torch.zeros((B, 1400), dtype=torch.uint8, device='cuda'); nres = torch.zeros((B,), dtype=torch.int32, device='cuda')
for it in (1, 2, 5, 10, 20):
    dec.set_params(ldpc_iters=it)
    for _ in range(2): dec.decode_batch_dev(iq, B, spots, nres)
    dec.synchronize(); dec.enable_timing(True)
    for _ in range(5): dec.decode_batch_dev(iq, B, spots, nres)
    dec.synchronize(); t = dec.timings(); dec.enable_timing(False)
    print(it, round(t['decode_ms'],3), 'decoded', float(nres.float().mean()))
