#!/usr/bin/env python3
"""The scaling curve of north_star ("frames/s at 1, 2, 4 and 8 GPUs") in one command, for whoever first has a node with
more than one GPU:

    python tools/scaling_curve.py [--gpus 1 2 4 8] [--steps 20] [--warmup 5] [--dry]

For every N it runs `bench.py --gpus N` twice -- with the per-step RCCL all-gather of the spot records, and with
`--no-exchange` (same launcher, shards, barrier and timing reduction, NO collective in the step) -- and prints one JSON
object: per N the two rates, the scaling efficiency against N x the one-GPU rate for both legs (decode scaling, and
decode + gather), what the gather costs (the difference), the sum of the ranks' decode-only rates measured before RCCL
existed, per-rank imbalance, and every failure line a rank or its watchdog left (rank, phase, rccl_error).  Frames are
independent units sharded contiguously: the expectation is a straight line for the control leg and the
same line minus a hidden gather for the other.

--dry: CPU tensors through gloo (launcher / sharding / exchange plumbing only; the rates are not decode rates)."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, extra, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + extra
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        rc, out, err = p.returncode, p.stdout, p.stderr
    except subprocess.TimeoutExpired as e:
        rc, out, err = -1, (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), "timeout"
    lines = []
    for ln in out.splitlines():
        if ln.startswith('{"metric"'):
            try:
                lines.append(json.loads(ln))
            except ValueError:
                pass
    ok = [ln for ln in lines if not ln.get("failed") and ln.get("value")]
    return {"rc": rc, "line": ok[-1] if ok else None, "failures": [ln for ln in lines if ln.get("failed")], "stderr_tail": err[-600:] if rc != 0 else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--dry", action="store_true")
    ap.add_argument("--timeout", type=float, default=1200.0, help="seconds per bench run")
    args = ap.parse_args()
    base = ["--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline", "--no-host-legs"]
    if args.frames:
        base += ["--frames", str(args.frames)]
    if args.dry:
        base += ["--backend", "gloo", "--dry"]
    rows, one = [], {}
    for n in args.gpus:
        row = {"n_gpus": n}
        for leg, extra in (("with_gather", []), ("no_exchange", ["--no-exchange"])):
            if n == 1 and leg == "no_exchange":
                row[leg] = row["with_gather"]          # one rank, no process group: the two legs are the same run
                continue
            r = run(n, base + extra, args.timeout)
            ln = r["line"]
            row[leg] = {"frames_per_s": ln["value"] if ln else None, "ms_per_step": ln["ms_per_step"] if ln else None, "rc": r["rc"],
                        "failures": [{k: f.get(k) for k in ("rank", "phase", "error", "rccl_error")} for f in r["failures"]] or None,
                        "stderr_tail": r["stderr_tail"]}
            if ln:
                d = ln.get("ranks_detail") or {}
                row[leg].update({"imbalance": d.get("imbalance"), "rank_ms_per_step": d.get("ms_per_step"),
                                 "decode_only_frames_per_s_sum": ln.get("decode_only_frames_per_s_sum"),
                                 "gathered_list_ok": ln.get("gathered_list_holds_every_ranks_shard"),
                                 "devices": {r_: v.get("pci") for r_, v in (ln.get("ranks_info") or {}).items()} or None,
                                 "build_id": ln.get("build_id")})
        if n == args.gpus[0]:
            one = {leg: (row[leg]["frames_per_s"] or 0) / n for leg in ("with_gather", "no_exchange")}
        for leg in ("with_gather", "no_exchange"):
            v = row[leg]["frames_per_s"]
            row[leg]["efficiency_vs_first_row"] = round(v / (n * one[leg]), 4) if v and one.get(leg) else None
        a, b = row["with_gather"]["frames_per_s"], row["no_exchange"]["frames_per_s"]
        row["gather_cost_frac"] = round(1.0 - a / b, 4) if a and b else None
        rows.append(row)
        print(f"# N={n}: with gather {a} frames/s, without {b}, gather cost {row['gather_cost_frac']}", file=sys.stderr, flush=True)
    print(json.dumps({"metric": "15 s FT8 frames decoded/s", "scaling": "weak", "dry": args.dry, "curve": rows}))
    return 0 if all(r["with_gather"]["frames_per_s"] for r in rows) else 1


if __name__ == "__main__":
    sys.exit(main())
