"""Synthetic workload of SURVEY.md section 8(d) and the frame sharding of section 8(e).

Frames are described on the host (which signals, where, how strong) and synthesised on the GPU
(ft8gpu_synth_frames), so that a batch of thousands of 15 s frames needs no PCIe traffic.
"""
import numpy as np

from . import SIGNAL_DTYPE, encode, pack77_std

LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"
SEED_BASE = 0x46543800            # "FT8\0", SURVEY.md section 8(d)


def shard_range(total_frames, rank, world_size):
    """contiguous shard [lo, hi) of `total_frames` for `rank` (frames are independent units)"""
    base, rem = divmod(total_frames, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def message_pool(n=1024, seed=7):
    """n distinct standard messages "CQ <call> <grid>" and their 79 tones"""
    rng = np.random.default_rng(seed)
    msgs, tones = [], np.zeros((n, 79), np.uint8)
    seen = set()
    while len(msgs) < n:
        pfx = rng.choice(["K", "W", "N", "G", "F", "DL", "JA", "VK", "EA", "OH", "SM", "PY"])
        call = pfx + str(rng.integers(0, 10)) + "".join(rng.choice(list(LETTERS), size=rng.integers(1, 4)))
        if call in seen:
            continue
        seen.add(call)
        grid = LETTERS[rng.integers(0, 18)] + LETTERS[rng.integers(0, 18)] + f"{rng.integers(0, 100):02d}"
        msg = f"CQ {call} {grid}"
        tones[len(msgs)] = encode(pack77_std(msg))
        msgs.append(msg)
    return msgs, tones


def amplitude_for_snr(snr_db, noise_sigma=1.0):
    """amplitude of a constant-envelope signal for an SNR quoted in 2500 Hz, with complex noise of
    variance sigma^2 per component over the 3200 Hz sample bandwidth"""
    return np.sqrt(2.0 * noise_sigma ** 2 * 2500.0 / 3200.0 * 10.0 ** (np.asarray(snr_db) / 10.0))


def frame_signals(first_frame, nframes, nsig, pool_tones, snr_range=(-18.0, 0.0), f_range=(100.0, 1500.0),
                  dt_range=(0.0, 1.8)):
    """signal descriptors for global frames [first_frame, first_frame + nframes); frame g is seeded
    with SEED_BASE + g so that any shard of any world size describes the same global batch"""
    sig = np.zeros((nframes, max(nsig, 1)), SIGNAL_DTYPE)
    picks = np.zeros((nframes, max(nsig, 1)), np.int32)
    for k in range(nframes):
        rng = np.random.default_rng(SEED_BASE + first_frame + k)
        idx = rng.integers(0, pool_tones.shape[0], nsig)
        sig[k, :nsig]["tones"] = pool_tones[idx]
        sig[k, :nsig]["f0_hz"] = rng.uniform(*f_range, nsig)
        sig[k, :nsig]["t0_s"] = rng.uniform(*dt_range, nsig)
        sig[k, :nsig]["amplitude"] = amplitude_for_snr(rng.uniform(*snr_range, nsig))
        picks[k, :nsig] = idx
    return sig[:, :nsig].copy(), picks[:, :nsig]


def gather_spots(spots, n_results, world_size):
    """The job's whole spot list on every rank: ONE all-gather of the fixed-size records
    ([frames][50 x 28 B]) and one of the per-frame counts.  Works on the
    RCCL backend (device tensors, xGMI) and on gloo (CPU tensors, used by the CPU tests).
    Returns (all_spots [world*frames, 1400] uint8, all_counts [world*frames] int32) in global frame
    order, because shards are contiguous and ranks are gathered in rank order."""
    import torch
    import torch.distributed as dist
    all_spots = torch.empty((world_size * spots.shape[0],) + tuple(spots.shape[1:]), dtype=spots.dtype, device=spots.device)
    all_counts = torch.empty((world_size * n_results.shape[0],), dtype=n_results.dtype, device=n_results.device)
    dist.all_gather_into_tensor(all_spots, spots)
    dist.all_gather_into_tensor(all_counts, n_results)
    return all_spots, all_counts


class SpotExchange:
    """Double-buffered spot records of one rank and their exchange as ONE asynchronous all-gather.

    A rank's step k writes its [frames][50 x 28 B] records and [frames] int32 counts into buffer k % 2
    (one flat allocation: records, then counts), `launch(k)` starts the all-gather of that buffer on the
    process group's own stream (it waits for the kernels enqueued so far on the current stream), and the
    next step's kernels are enqueued behind it without waiting: the collective (5.7 MB per rank and
    step at 4096 frames) runs under the next batch's decode.  A buffer is waited for just before it is
    reused, and `wait_all()` drains everything (bench.py calls it inside the timed region).
    Works on RCCL (device tensors) and on gloo (CPU tensors, tests/test_dist_gloo.py)."""

    RECORD_BYTES = 1400

    def __init__(self, frames, world_size, device, collective=None):
        import torch
        self.frames, self.world = frames, world_size
        self.collective = (world_size > 1) if collective is None else collective     # True forces it for one rank too
        seg = frames * (self.RECORD_BYTES + 4)
        self._local = [torch.zeros(seg, dtype=torch.uint8, device=device) for _ in range(2)]
        self._all = [torch.zeros(seg * world_size, dtype=torch.uint8, device=device) for _ in range(2)] if self.collective else None
        self._work = [None, None]

    def buffers(self, k):
        """(spots [frames, 1400] uint8, counts [frames] int32) of step k; waits for the exchange that last used them"""
        i = k & 1
        if self._work[i] is not None:
            self._work[i].wait()
            self._work[i] = None
        flat = self._local[i]
        n = self.frames * self.RECORD_BYTES
        return flat[:n].view(self.frames, self.RECORD_BYTES), flat[n:].view(dtype=__import__("torch").int32)

    def launch(self, k):
        import torch.distributed as dist
        if self.collective:
            i = k & 1
            self._work[i] = dist.all_gather_into_tensor(self._all[i], self._local[i], async_op=True)

    def wait_all(self):
        for i in (0, 1):
            if self._work[i] is not None:
                self._work[i].wait()
                self._work[i] = None

    def gathered(self, k):
        """(all_spots [world*frames, 1400], all_counts [world*frames]) of step k in global frame order (copies)"""
        import torch
        i = k & 1
        if self._work[i] is not None:
            self._work[i].wait()
            self._work[i] = None
        if not self.collective:
            s, c = self._local[i][:self.frames * self.RECORD_BYTES], self._local[i][self.frames * self.RECORD_BYTES:]
            return s.view(self.frames, self.RECORD_BYTES).clone(), c.view(dtype=torch.int32).clone()
        seg = self._all[i].view(self.world, -1)
        n = self.frames * self.RECORD_BYTES
        # copies: the receive buffer is overwritten by the exchange of step k + 2
        spots = seg[:, :n].reshape(self.world * self.frames, self.RECORD_BYTES).clone()
        counts = seg[:, n:].clone(memory_format=torch.contiguous_format).view(dtype=torch.int32).reshape(-1)
        return spots, counts
