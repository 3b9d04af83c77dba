import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import rtlsdr_ft8d_amd as ft8
from rtlsdr_ft8d_amd import workload
B=4096
dec = ft8.Decoder(device=0, max_frames=B)
_, tones = workload.message_pool()
sig,_ = workload.frame_signals(0, B, 20, tones)
iq = torch.empty((B,2,48000), dtype=torch.float32, device='cuda')
dec.synth_frames(sig, B, 20, 1.0, workload.SEED_BASE, iq)
h = iq.cpu().numpy()
hp = torch.from_numpy(h).pin_memory().numpy()
for name, arr in (("pageable", h), ("pinned", hp)):
    dec.decode_batch(arr)
    t=time.perf_counter(); d,n = dec.decode_batch(arr); dt=time.perf_counter()-t
    print(name, round(B/dt), "frames/s", round(dt*1e3,1), "ms", int(n.sum()))
