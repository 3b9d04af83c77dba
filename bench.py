#!/usr/bin/env python3
"""bench.py -- frames decoded per second by the MI355X FT8 hot path (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the whole decode path (waterfall FFT -> Costas sync -> LLR -> LDPC BP -> CRC ->
unpack -> dedup/spots) over the rank's batch of synthetic 15 s frames, resident in HBM before the
timed region starts.  Frames are independent, so N GPUs each take a contiguous shard of the global
batch (weak scaling: frames per GPU fixed); the only collective is one RCCL all-gather of the
fixed-size spot records per step, double-buffered so that it runs under the next step's kernels.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_FRAME = 384000 + 1404          # IQ in + 50 spot records and the count out
HBM_PEAK_GBPS = 8000.0                   # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_PEAK_GBPS = 6290.0              # same guide: measured copy peak (SURVEY.md 8(d) asks for both)
FP32_VALU_PEAK = 157.3e12                # same guide: fp32 vector peak, flop/s (an FMA lane-instruction counts 2)
FRAMES_DEFAULT = 4096                    # the PMC traffic figures in profiles/ were collected at this batch size


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU per step (configs[2])")
    ap.add_argument("--nsig", type=int, default=20, help="FT8 signals per frame")
    ap.add_argument("--snr", type=float, nargs=2, default=(-18.0, 0.0))
    ap.add_argument("--max-candidates", type=int, default=120)
    ap.add_argument("--cpu-frames", type=int, default=2048, help="frames timed on the host CPU (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the spot all-gather even with one rank (exercises the N>1 code path on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B = args.frames
    total = B * world
    lo, hi = workload.shard_range(total, rank, world)
    assert hi - lo == B

    dec = ft8.Decoder(device=local_rank, max_frames=B, min_score=10, max_candidates=args.max_candidates, ldpc_iters=20)
    # one explicit torch stream carries both the decoder kernels and the RCCL gather, so that the
    # collective is ordered after the kernels that produce the spot records
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    dec.set_stream(stream.cuda_stream)

    # ---- synthetic frames, generated in HBM (not timed) ----------------------------------------
    _, pool_tones = workload.message_pool()
    sig, _ = workload.frame_signals(lo, B, args.nsig, pool_tones, snr_range=tuple(args.snr))
    iq = torch.empty((B, 2, ft8.NSAMPLES), dtype=torch.float32, device=dev)
    dec.synth_frames(sig, B, args.nsig, 1.0, workload.SEED_BASE, iq, first_frame=lo)
    # spot records: two buffers per rank; the exchange of step k (one asynchronous RCCL all-gather of
    # records + counts) runs under the kernels of step k + 1 and is drained inside the timed region
    exch = workload.SpotExchange(B, world, dev, collective=use_dist)
    spots, nres = exch.buffers(0)
    state = {"k": 0}

    def step():
        k = state["k"]
        s_buf, n_buf = exch.buffers(k)                  # waits for the exchange that last used this buffer
        dec.decode_batch_dev(iq, B, s_buf, n_buf)
        exch.launch(k)                                  # the whole job's spot list on every rank
        state["k"] = k + 1

    def fence():
        exch.wait_all()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    dec.enable_timing(True)          # stage events are recorded on the stream, read after the fence
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    stage_avg = dec.timings()        # mean over the timed steps (ring of the last 32)
    timed_runs = stage_avg.pop("runs")
    dec.enable_timing(False)
    spots, nres = exch.buffers(state["k"] - 1) if state["k"] > 0 else (spots, nres)    # the last step's local records

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    frames_total = total * args.steps
    value = frames_total / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    n_host = nres.cpu().numpy()
    out = {
        "metric": "15 s FT8 frames decoded/s",
        "value": round(value, 1),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"configs[2]: batch of {B} synthetic 15 s 3200 sps IQ frames per GPU, {args.nsig} CQ signals/frame "
                        f"SNR U[{args.snr[0]:g},{args.snr[1]:g}] dB, full pipeline incl. HIP LDPC(174,91) BP, "
                        f"K_MAX_CANDIDATES={args.max_candidates}",
            "frames_per_gpu": B, "global_frames": total, "parallelism": f"frame-sharded x{world}",
            "decoded_messages_per_frame": round(float(n_host.mean()), 2),
        },
    }
    if rank == 0:
        launches = int(stage_avg.pop("launches_per_stage", 1))
        kernels = {k: v for k, v in stage_avg.items() if k != "total_ms"}
        dom = max(kernels, key=kernels.get)
        dom_ms = kernels[dom]
        achieved = BYTES_PER_FRAME * B / (dom_ms * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm", "kernel": dom.replace("_ms", ""), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": _pmc_traffic(dom.replace("_ms", ""), B, launches),
            "valu_busy_frac_pmc": _pmc_valu_busy(dom.replace("_ms", "")),
            "valu_frac_of_fp32_peak_pmc": _pmc_valu_fraction(dom.replace("_ms", ""), B, launches, dom_ms / launches),
            "frac_of_measured_copy_peak": round(achieved / HBM_COPY_PEAK_GBPS, 5),
            "kernel_ms": round(dom_ms, 4), "kernel_launches_per_step": launches,
            "kernel_ms_per_launch": round(dom_ms / launches, 4), "algorithmic_bytes_per_launch": BYTES_PER_FRAME * B // launches,
            "stage_ms": {k: round(v, 4) for k, v in stage_avg.items()}, "stage_ms_runs": timed_runs,
            "pipeline_achieved_GBps": round(BYTES_PER_FRAME * B / (stage_avg["total_ms"] * 1e-3) / 1e9, 2),
        }
        if world == 1 and not args.no_cpu_baseline and args.cpu_frames > 0:
            out["cpu_baseline"] = cpu_baseline(iq, spots, nres, min(args.cpu_frames, B), args.max_candidates)
    dec.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        sys.stderr.flush()
        try:                                        # RCCL writes its banner through C stdio, which is fully buffered
            import ctypes                           # on a pipe: push it out before the JSON line, not at exit
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)          # the one JSON line, after any library banners


def _pmc_valu_busy(kernel):
    """fraction of SIMD cycles with a VALU instruction in flight for the dominant kernel, from the committed
    PMC summary (the path is VALU-bound, so this is the roof that actually binds)"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(kernel, {}).get("valu_busy_frac")
    except (OSError, ValueError):
        return None


def _pmc_valu_fraction(kernel, frames, launches, ms_per_launch):
    """SURVEY.md 8(d) "valu_fraction": VALU lane-operations per second of the dominant kernel over the fp32
    vector peak.  SQ_INSTS_VALU (wave instructions per launch, committed PMC pass at this batch size) x 64
    lanes / the live launch duration; one operation per lane-instruction (a packed or fused instruction
    carries two, so this is a lower bound), against a peak that counts two per lane and clock."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            if json.load(f).get(kernel, {}).get("frames_per_launch") != frames // launches:
                return None
        with open(os.path.join(ROOT, "profiles", "r01_pmc_counters_final.json")) as f:
            insts = json.load(f)[kernel]["SQ_INSTS_VALU"]
        return round(insts * 64 / (ms_per_launch * 1e-3) / FP32_VALU_PEAK, 4)
    except (OSError, ValueError, KeyError):
        return None


def _pmc_traffic(kernel, frames, launches):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/pmc_traffic.json, FETCH_SIZE / WRITE_SIZE passes over this very command; counters were
    collected with one launch per stage and are scaled to the frames one launch covers now); None
    until it has been collected for this kernel and batch size."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            e = json.load(f).get(kernel, {})
        if not e or e.get("frames_per_launch") != frames // launches:
            return None
        return int(e["hbm_bytes_per_launch"])
    except (OSError, ValueError):
        return None


def usable_cores():
    """host cores this process may actually use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(iq, spots, nres, m, max_candidates):
    """The CPU oracle (a port of the reference path; the reference itself cannot be built here) timed
    on the first m frames of the same batch, all host cores, one frame per OpenMP task.  The same
    sample doubles as a parity check of the GPU results."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import rtlsdr_ft8d_amd as ft8
    oracle_lib.lib()
    cores = usable_cores()
    host_iq = iq[:m].cpu().numpy()
    p = oracle_lib.default_params(10, max_candidates, 20)
    oracle_lib.subsystem_batch(host_iq[:min(m, cores)], p, cores)          # warm-up (page-in, thread pool)
    t0 = time.perf_counter()
    rdec, rn = oracle_lib.subsystem_batch(host_iq, p, cores)
    dt = time.perf_counter() - t0
    m1 = min(m, 64)
    t1 = time.perf_counter()
    oracle_lib.subsystem_batch(host_iq[:m1], p, 1)
    dt1 = time.perf_counter() - t1
    g_dec = spots[:m].cpu().numpy().view(ft8.RESULT_DTYPE).reshape(m, ft8.MAX_MESSAGES)
    g_n = nres[:m].cpu().numpy()
    same = sum(int(g_n[k] == rn[k] and g_dec[k].tobytes() == rdec[k].tobytes()) for k in range(m))
    return {"value": round(m / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"first {m} frames of the bench batch through oracle ft8o_subsystem_batch (gcc -O2, own radix-4 FFT, "
                      f"OpenMP {cores} threads), {dt:.2f} s wall",
            "single_core_frames_per_s": round(m1 / dt1, 2), "os_cpu_count": os.cpu_count(),
            "gpu_vs_oracle_identical_frames": f"{same}/{m}"}


if __name__ == "__main__":
    main()
