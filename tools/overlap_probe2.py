"""Round 6: do the front-end kernels (waterfall, sync + heap) co-run profitably with the LDPC kernel IN THE FORM THE BATCH PIPELINE RUNS
(ft8_decode_kernel<false,3>) when they sit on different streams?  Two contexts on one GPU: A decodes, F runs the front end of other frames.
Prints solo times, the sum, and the time with both in flight; then a software-pipelined walk over parts of a 4096-frame batch
(front end of part k+1 beside the LDPC kernel of part k) against the same stage calls issued back to back."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import rtlsdr_ft8d_amd as ft8
from rtlsdr_ft8d_amd import workload
NF = 4096
_, tones = workload.message_pool()
L = ft8.load_library()
A = ft8.Decoder(device=0, max_frames=NF)      # decode stream
F = ft8.Decoder(device=0, max_frames=NF)      # front-end stream
A.set_debug_flags(ft8.DBG_PIPELINE_FORM)
sig, _ = workload.frame_signals(0, NF, 20, tones)
iq = torch.empty((NF, 2, 48000), dtype=torch.float32, device='cuda')
A.synth_frames(sig, NF, 20, 1.0, workload.SEED_BASE, iq)
mag = torch.empty((NF, 94208), dtype=torch.uint8, device='cuda')
cands = torch.zeros((NF, 120, 8), dtype=torch.uint8, device='cuda')
counts = torch.zeros((NF,), dtype=torch.int32, device='cuda')
status = torch.zeros((NF, 120, 48), dtype=torch.uint8, device='cuda')
mag2 = torch.empty_like(mag); cands2 = torch.zeros_like(cands); counts2 = torch.zeros_like(counts)
D = ft8.DEVICE_PTRS
def wf(dec, lo, n, out): ft8._check(L.ft8gpu_waterfall(dec.h, iq[lo:].data_ptr(), n, out[lo:].data_ptr(), D))
def sync(dec, lo, n, m, c, k): ft8._check(L.ft8gpu_find_sync(dec.h, m[lo:].data_ptr(), n, c[lo:].data_ptr(), k[lo:].data_ptr(), D))
def decode(dec, lo, n): ft8._check(L.ft8gpu_decode_candidates(dec.h, mag[lo:].data_ptr(), cands[lo:].data_ptr(), counts[lo:].data_ptr(), n, status[lo:].data_ptr(), D))
torch.cuda.synchronize()
wf(A, 0, NF, mag); sync(A, 0, NF, mag, cands, counts); A.synchronize()
def timed(fn, n=20):
    for _ in range(3): fn()
    A.synchronize(); F.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    A.synchronize(); F.synchronize()
    return (time.perf_counter() - t) / n * 1e3
res = {}
for _ in range(60): decode(A, 0, NF)            # clock settle
A.synchronize()
for B in (1024, 2048):
    td = timed(lambda: decode(A, 0, B))
    tw = timed(lambda: wf(F, 0, B, mag2))
    ts = timed(lambda: sync(F, 0, B, mag2, cands2, counts2))
    tws = timed(lambda: (wf(F, 0, B, mag2), sync(F, 0, B, mag2, cands2, counts2)))
    tb_w = timed(lambda: (decode(A, 0, B), wf(F, 0, B, mag2)))
    tb_ws = timed(lambda: (decode(A, 0, B), wf(F, 0, B, mag2), sync(F, 0, B, mag2, cands2, counts2)))
    res[B] = dict(decode=td, wf=tw, sync_heap=ts, wf_sync=tws, decode_with_wf=tb_w, sum_decode_wf=td + tw, decode_with_wf_sync=tb_ws, sum_decode_wf_sync=td + tws)
    print(B, {k: round(v, 3) for k, v in res[B].items()}, flush=True)
# software pipeline over parts: F runs the front end of part k+1 while A decodes part k (events order F -> A per part)
sA = torch.cuda.ExternalStream(A.stream_handle()); sF = torch.cuda.ExternalStream(F.stream_handle())
for P in (512, 1024, 2048):
    parts = [(lo, min(P, NF - lo)) for lo in range(0, NF, P)]
    evs = [torch.cuda.Event() for _ in parts]
    def serial():
        for lo, n in parts:
            wf(A, lo, n, mag); sync(A, lo, n, mag, cands, counts)
        for lo, n in parts:
            decode(A, lo, n)
    def piped():
        sF.wait_stream(sA)                        # the previous walk's decodes have read mag / cands
        for k, (lo, n) in enumerate(parts):
            wf(F, lo, n, mag); sync(F, lo, n, mag, cands, counts)
            evs[k].record(sF)
            sA.wait_event(evs[k])
            decode(A, lo, n)
    ts_, tp_ = timed(serial, 10), timed(piped, 10)
    t2 = timed(serial, 10); t3 = timed(piped, 10)
    print(f"parts of {P}: back to back {ts_:.3f} / {t2:.3f} ms, software-pipelined {tp_:.3f} / {t3:.3f} ms", flush=True)
    res[f"parts_{P}"] = dict(serial=[ts_, t2], pipelined=[tp_, t3])
print(json.dumps(res))
