#!/usr/bin/env python3
"""What does the RCCL spot exchange cost per step?  One rank, one GPU (the hardware at hand), bench.py's own
step structure, four arms timed back to back in ONE process and GPU session (boxes differ by a few percent):

  plain        no process group at all
  pg_idle      RCCL process group initialised, no collective issued
  gather       + one asynchronous all_gather_into_tensor per step, double-buffered (bench.py --force-dist)
  gather_sync  the same collective waited for inside the step (what it would cost serialised)

Also prints the host time spent inside exch.launch() / exch.buffers() per step.
  python tools/dist_overhead_probe.py [--frames 4096] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B = args.frames
    dec = ft8.Decoder(device=0, max_frames=B)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    dec.set_stream(stream.cuda_stream)
    _, tones = workload.message_pool()
    sig, _ = workload.frame_signals(0, B, 20, tones)
    iq = torch.empty((B, 2, ft8.NSAMPLES), dtype=torch.float32, device=dev)
    dec.synth_frames(sig, B, 20, 1.0, workload.SEED_BASE, iq)

    def run(exch, sync_each):
        host_launch = host_buffers = 0.0
        k = 0

        def step():
            nonlocal k, host_launch, host_buffers
            t = time.perf_counter()
            s_buf, n_buf = exch.buffers(k)
            host_buffers += time.perf_counter() - t
            dec.decode_batch_dev(iq, B, s_buf, n_buf)
            t = time.perf_counter()
            exch.launch(k)
            if sync_each:
                exch.wait_all()
            host_launch += time.perf_counter() - t
            k += 1
        for _ in range(args.warmup):
            step()
        exch.wait_all()
        torch.cuda.synchronize()
        host_launch = host_buffers = 0.0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        exch.wait_all()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"ms_per_step": round(1e3 * dt / args.steps, 4), "host_launch_us": round(1e6 * host_launch / args.steps, 1),
                "host_buffers_us": round(1e6 * host_buffers / args.steps, 1)}

    out = {"frames": B, "steps": args.steps, "arms": {}}
    arms = {"plain": []}
    for _ in range(args.rounds):
        arms["plain"].append(run(workload.SpotExchange(B, 1, dev, collective=False), False))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    for name in ("pg_idle", "gather", "gather_sync", "plain_after"):
        arms[name] = []
    for _ in range(args.rounds):
        arms["pg_idle"].append(run(workload.SpotExchange(B, 1, dev, collective=False), False))
        arms["gather"].append(run(workload.SpotExchange(B, 1, dev, collective=True), False))
        arms["gather_sync"].append(run(workload.SpotExchange(B, 1, dev, collective=True), True))
    out["arms"] = arms
    out["best_ms"] = {k: min(r["ms_per_step"] for r in v) for k, v in arms.items() if v}
    dist.destroy_process_group()
    dec.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
