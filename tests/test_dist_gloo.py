"""world_size-2 / -3 / -4 CPU tests (gloo) of the multi-GPU plumbing: contiguous frame shards (also ragged ones), seeding
that is independent of the world size, and the single all-gather of the spot records.  The data path
itself has no collective (frames are independent); on the GPU box the same code runs on RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_spots(lo, hi):
    """deterministic stand-in for a rank's decode output: record bytes depend on the global frame index"""
    n = hi - lo
    g = np.arange(lo, hi, dtype=np.int64)
    spots = ((g[:, None] * 131 + np.arange(1400)[None, :] * 7) % 251).astype(np.uint8)
    counts = (g % 50).astype(np.int32)
    return torch.from_numpy(spots), torch.from_numpy(counts)


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtlsdr_ft8d_amd import workload
    lo, hi = workload.shard_range(total, rank, world)
    spots, counts = _fake_spots(lo, hi)
    all_spots, all_counts = workload.gather_spots(spots, counts, world)
    ref_spots, ref_counts = _fake_spots(0, total)
    ok = bool(torch.equal(all_spots, ref_spots) and torch.equal(all_counts, ref_counts))
    # the double-buffered, single-collective, asynchronous form bench.py uses: three steps, each step's
    # records shifted by the step number so that a stale or swapped buffer would show
    ex = workload.SpotExchange(hi - lo, world, "cpu")
    for k in range(3):
        s_buf, c_buf = ex.buffers(k)
        s_buf.copy_((spots.to(torch.int32) + k).to(torch.uint8))
        c_buf.copy_(counts + k)
        ex.launch(k)
    for k in (1, 2):                       # the last two steps are still held (two buffers)
        g_s, g_c = ex.gathered(k)
        ok = ok and bool(torch.equal(g_s, (ref_spots.to(torch.int32) + k).to(torch.uint8)) and torch.equal(g_c, ref_counts + k))
    ex.wait_all()
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t.item()) == float(world)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, lo, hi))


def _worker_ragged(rank, world, port, total, q):
    try:
        _worker_ragged_body(rank, world, port, total, q)
    except Exception as e:                       # a dead rank must fail the test at once, not after the queue's timeout
        q.put((rank, repr(e), -1, -1))
        raise


def _worker_ragged_body(rank, world, port, total, q):
    """the real SpotExchange on a job whose frames do not divide by the ranks: shards of unequal size, one equal-sized
    padded all-gather, padding dropped on the way out; several steps so that both buffers are reused"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtlsdr_ft8d_amd import workload
    lo, hi = workload.shard_range(total, rank, world)
    spots, counts = _fake_spots(lo, hi)
    ref_spots, ref_counts = _fake_spots(0, total)
    ex = workload.SpotExchange(hi - lo, world, "cpu", total_frames=total)
    ok = ex.cap == -(-total // world)
    for k in range(5):
        s_buf, c_buf = ex.buffers(k)
        ok = ok and tuple(s_buf.shape) == (hi - lo, 1400) and tuple(c_buf.shape) == (hi - lo,)
        s_buf.copy_((spots.to(torch.int32) + k).to(torch.uint8))
        c_buf.copy_(counts + k)
        ex.launch(k)
        if k >= 1:                             # the previous step's list while this one is in flight
            g_s, g_c = ex.gathered(k - 1)
            ok = ok and bool(torch.equal(g_s, (ref_spots.to(torch.int32) + k - 1).to(torch.uint8)) and torch.equal(g_c, ref_counts + k - 1))
    g_s, g_c = ex.gathered(4)
    ok = ok and bool(torch.equal(g_s, (ref_spots.to(torch.int32) + 4).to(torch.uint8)) and torch.equal(g_c, ref_counts + 4))
    ex.wait_all()
    # the per-rank diagnostics bench.py gathers for world > 1: one all-gather of a small float vector per rank
    mine = torch.tensor([float(rank), 1.0 + 0.5 * rank, float(hi - lo)], dtype=torch.float64)
    every = torch.empty(world * 3, dtype=torch.float64)           # flat: the form gloo and RCCL both take
    dist.all_gather_into_tensor(every, mine)
    every = every.view(world, 3)
    ok = ok and every[:, 0].tolist() == [float(r) for r in range(world)] and int(every[:, 2].sum().item()) == total
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, lo, hi))


@pytest.mark.parametrize("world,total", [(4, 4 * 37 + 3), (4, 5), (3, 64)])
def test_spot_exchange_world4_ragged_totals(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True] * world, res
    assert res[0][2] == 0 and res[-1][3] == total and all(a[3] == b[2] for a, b in zip(res, res[1:]))
    sizes = [r[3] - r[2] for r in res]
    assert max(sizes) - min(sizes) <= 1 and (total % world == 0 or len(set(sizes)) == 2)


def test_gather_spots_world2():
    world, total = 2, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert (res[0][2], res[0][3], res[1][2], res[1][3]) == (0, 32, 32, 64)


@pytest.mark.parametrize("total,world", [(32768, 8), (4096, 1), (10, 4), (7, 8), (0, 2)])
def test_shard_range_covers_contiguously(total, world):
    from rtlsdr_ft8d_amd import workload
    edges = [workload.shard_range(total, r, world) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    sizes = [b - a for a, b in edges]
    assert max(sizes) - min(sizes) <= 1


def test_frame_description_independent_of_world_size():
    """global frame g gets the same signals whichever rank of whichever world size owns it"""
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    if not os.path.exists(ft8.LIB_PATH):
        pytest.skip("libft8gpu.so not built")
    _, tones = workload.message_pool(64)
    whole, _ = workload.frame_signals(0, 12, 5, tones)
    for world in (2, 3, 4):
        parts = []
        for r in range(world):
            lo, hi = workload.shard_range(12, r, world)
            parts.append(workload.frame_signals(lo, hi - lo, 5, tones)[0])
        assert np.array_equal(np.concatenate(parts), whole)


def _run_bench(*argv, env_extra=None):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    return p.returncode, [json.loads(ln) for ln in lines], p.stderr


def test_bench_self_launches_its_ranks_as_a_plain_command():
    """`python bench.py --gpus 2` with no RANK/WORLD_SIZE in the environment must start its own two ranks (fresh
    child processes through torch.distributed.run, free port on 127.0.0.1), relay ONE JSON line and exit 0;
    --backend gloo --dry runs the real shard_range + SpotExchange on CPU tensors without touching a GPU"""
    rc, lines, err = _run_bench("--gpus", "2", "--backend", "gloo", "--dry", "--frames", "48", "--steps", "3", "--warmup", "1")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["backend"] == "gloo" and line["dry"] is True
    assert line["dry_gather_identical_on_all_ranks"] is True
    assert line["config"]["global_frames"] == 96 and line["config"]["frames_per_gpu"] == 48
    assert line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"


def test_bench_under_an_external_torchrun_still_works():
    """the driver's own form: torch.distributed.run starts bench.py, which must NOT launch again"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--backend", "gloo", "--dry", "--frames", "16", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1 and lines[0]["ranks"] == 2 and lines[0]["dry_gather_identical_on_all_ranks"] is True


def test_bench_launcher_reports_a_failing_rank():
    """a rank that dies makes the plain command exit non-zero (here: world size mismatch inside the children)"""
    rc, lines, err = _run_bench("--gpus", "2", "--backend", "gloo", "--frames", "8")     # gloo without --dry is refused by every rank
    assert rc != 0 and not lines
    assert "2-rank job failed" in err


# ---- first-contact hardening of `bench.py --gpus N`: a hanging or failing rank must leave a diagnosable
# JSON line within the timeout and a non-zero status; exercised on gloo through the same code path the RCCL run takes
def _failed_lines(lines):
    return [ln for ln in lines if ln.get("failed")]


def test_bench_hanging_rank_in_process_group_init_is_named_within_the_timeout():
    """rank 1 never returns from init_process_group (test hook: it sleeps where RCCL's init would hang).  Its watchdog -- a
    child process started before torch was imported, so no Python thread of the rank is needed -- names rank and phase in a
    JSON line and kills the rank; the launcher exits non-zero well within the driver's patience."""
    import time
    t0 = time.time()
    rc, lines, err = _run_bench("--gpus", "2", "--backend", "gloo", "--dry", "--frames", "16", "--steps", "2", "--warmup", "1",
                                "--dist-timeout", "40", "--test-hang", "1:init_process_group:5")   # rank 0 waits for rank 1 inside ITS init with the general
                                                                                                    # limit: only the hung rank's watchdog fires, deterministically
    took = time.time() - t0
    assert rc != 0 and took < 90, (rc, took)
    failed = _failed_lines(lines)
    mine = [ln for ln in failed if ln["rank"] == 1]
    assert mine and mine[0]["phase"] == "init_process_group" and "hung in phase 'init_process_group'" in mine[0]["error"], failed
    assert mine[0]["value"] is None and mine[0]["n_gpus"] == 2
    assert not [ln for ln in lines if not ln.get("failed")]           # no success line from a job that did not finish
    assert "rank 1 of 2 hung in phase 'init_process_group'" in err


def test_bench_hanging_rank_inside_the_exchange_is_named_with_its_history():
    rc, lines, err = _run_bench("--gpus", "2", "--backend", "gloo", "--dry", "--frames", "16", "--steps", "2", "--warmup", "1",
                                "--dist-timeout", "60", "--test-hang", "0:exchange_steps:5")
    assert rc != 0
    mine = [ln for ln in _failed_lines(lines) if ln["rank"] == 0 and ln["phase"] == "exchange_steps"]
    assert mine, lines
    done = [p[0] for p in mine[0]["phases_completed"]]
    assert done[:3] == ["import_torch", "init_process_group", "rank_info"], done


def test_bench_failing_collective_leaves_the_error_and_every_ranks_decode_rate():
    """a collective that raises on one rank after start-up: that rank prints ONE line with `rccl_error`, the phase, and -- read
    from the c10d store, not through the broken group -- each rank's decode-only rate and identity"""
    rc, lines, err = _run_bench("--gpus", "3", "--backend", "gloo", "--dry", "--frames", "16", "--steps", "4", "--warmup", "1",
                                "--dist-timeout", "30", "--test-fail-exchange", "2:2")
    assert rc != 0
    mine = [ln for ln in _failed_lines(lines) if ln["rank"] == 2]
    assert mine, (lines, err[-1500:])
    ln = mine[0]
    assert "simulated collective failure at step 2" in ln["rccl_error"] and ln["phase"] == "exchange_steps" and ln["value"] is None
    assert sorted(ln["per_rank_decode_only"]) == ["0", "1", "2"] and all(v["frames_per_s"] > 0 for v in ln["per_rank_decode_only"].values())
    assert sorted(ln["ranks_info"]) == ["0", "1", "2"] and len({v["pid"] for v in ln["ranks_info"].values()}) == 3
    assert ln["config"]["global_frames"] == 48


def test_bench_no_exchange_control_leg_and_rank_identities_in_the_line():
    """--no-exchange: same launcher, shards and timing reduction, no collective in the step (the control leg that separates
    decode scaling from gather cost); the success line carries what every rank is (pid, host, what it sees)"""
    rc, lines, err = _run_bench("--gpus", "2", "--backend", "gloo", "--dry", "--frames", "24", "--steps", "3", "--warmup", "1", "--no-exchange")
    assert rc == 0 and len(lines) == 1, err[-1500:]
    assert lines[0]["exchange"] == "none" and lines[0]["dry_gather_identical_on_all_ranks"] is True
    rc, lines, err = _run_bench("--gpus", "2", "--backend", "gloo", "--dry", "--frames", "24", "--steps", "3", "--warmup", "1")
    assert rc == 0 and len(lines) == 1, err[-1500:]
    assert lines[0]["exchange"].startswith("all_gather")
    info = lines[0]["ranks_info"]
    assert sorted(info) == ["0", "1"] and info["0"]["pid"] != info["1"]["pid"] and {v["local_rank"] for v in info.values()} == {0, 1}
    assert sorted(lines[0]["per_rank_decode_only"]) == ["0", "1"]


def test_scaling_curve_tool_runs_both_legs_per_world_size():
    """tools/scaling_curve.py: the one command for the first multi-GPU node -- per N a run with the gather and a --no-exchange
    control run, efficiencies against the first row, gather cost, failure lines; here on gloo with pattern records"""
    import json
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scaling_curve.py"), "--gpus", "1", "2", "--dry", "--frames", "24",
                        "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert p.returncode == 0, p.stderr[-1500:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert [r["n_gpus"] for r in d["curve"]] == [1, 2] and d["dry"] is True
    two = d["curve"][1]
    assert two["with_gather"]["frames_per_s"] > 0 and two["no_exchange"]["frames_per_s"] > 0 and two["gather_cost_frac"] is not None
    assert two["with_gather"]["efficiency_vs_first_row"] is not None and not two["with_gather"]["failures"]
