"""Synthetic workload of SURVEY.md section 8(d) and the frame sharding of section 8(e).

Frames are described on the host (which signals, where, how strong) and synthesised on the GPU
(ft8gpu_synth_frames), so that a batch of thousands of 15 s frames needs no PCIe traffic.
"""
import numpy as np

from . import SIGNAL_DTYPE, encode, pack77_std

MIXED_DUP_FRACTION = 0.25      # frames of the mixed workload in which one message is heard on two frequencies

LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"
SEED_BASE = 0x46543800            # "FT8\0", SURVEY.md section 8(d)


def shard_range(total_frames, rank, world_size):
    """contiguous shard [lo, hi) of `total_frames` for `rank` (frames are independent units)"""
    base, rem = divmod(total_frames, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _call(rng, prefixes=("K", "W", "N", "G", "F", "DL", "JA", "VK", "EA", "OH", "SM", "PY")):
    return rng.choice(list(prefixes)) + str(rng.integers(0, 10)) + "".join(rng.choice(list(LETTERS), size=rng.integers(1, 4)))


def _grid(rng):
    return LETTERS[rng.integers(0, 18)] + LETTERS[rng.integers(0, 18)] + f"{rng.integers(0, 100):02d}"


def message_pool(n=1024, seed=7, traffic="cq"):
    """n distinct messages and their 79 tones.
    traffic="cq"     "CQ <call> <grid>" only: the recipe of SURVEY.md section 8(d) (the headline bench and the soaks of rounds 2-4)
    traffic="mixed"  what a receiver meets on the air, see mixed_message_pool: about a quarter CQ calls, the rest QSO traffic
    Returns (texts, tones [n][79]); a text of None marks a payload that decodes (LDPC + CRC) but that unpack77 refuses."""
    if traffic == "mixed":
        return mixed_message_pool(n, seed)
    if traffic != "cq":
        raise ValueError(f"unknown traffic {traffic!r}")
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


# share of each message shape in the mixed pool (sums to 1).  The reference's candidate loop treats them differently
# (rtlsdr_ft8d.c:1487-1520): every unique message counts, only those whose first token starts with "CQ" fill a slot.
MIXED_SHARES = (
    ("cq_grid", 0.17), ("cq_modifier", 0.04), ("cq_nogrid", 0.015), ("cq_type4", 0.02), ("cq_suffix", 0.005),
    ("grid", 0.19), ("report", 0.17), ("r_report", 0.13), ("rr73", 0.06), ("rrr", 0.02), ("s73", 0.04),
    ("two_calls", 0.03), ("r_grid", 0.02), ("suffix", 0.02), ("hashed", 0.02), ("type4", 0.02),
    ("free_text", 0.012), ("free_text_cq", 0.004), ("telemetry", 0.004), ("not_unpackable", 0.01),
)


def mixed_message_pool(n=1024, seed=7):
    """Messages as a receiver meets them: CQ calls of every shape (plain, with a modifier, without grid, from a
    non-standard call, /P), and the QSO traffic that is most of the band -- grid, report, R-report, RR73 / RRR / 73,
    two bare calls, "R grid", /R /P, hashed <calls> (printed "<...>" by the reference's ft8_lib era), type 4, free text
    (some of it starting with "CQ", which the reference's token test takes for a CQ call), telemetry, and payloads
    of types its unpack77 refuses.  QSO partners come from a pool of 160 calls so that one pair shows up in several
    messages, as on the air.  Deterministic in (n, seed)."""
    from . import pack77
    rng = np.random.default_rng(seed ^ 0x4D495845)
    calls = []
    while len(calls) < 160:
        c = _call(rng, ("K", "W", "N", "G", "F", "DL", "JA", "VK", "EA", "OH", "SM", "PY", "9A", "A6", "ZL", "VE"))
        if c not in calls:
            calls.append(c)
    long_calls = ["PJ4/K1ABC", "KH1/KH7Z", "YW18FIFA", "W9XYZ/QRP", "VP8/G4ABC", "DL1ABC/MM", "EA8/OH2XX", "K1ABC/7"]
    free = ["TNX BOB 73 GL", "HELLO WORLD", "QRP 5W DIPOLE", "GL ES 73", "WX SUNNY 25C", "TEST 123", "PSE QSL BURO", "RR TU 73"]
    free_cq = ["CQ73 GL", "CQ TEST 1/2", "CQDX PSE K"]
    kinds, weights = zip(*MIXED_SHARES)
    texts, payloads, seen = [], [], set()

    def pair():
        a, b = rng.choice(len(calls), 2, replace=False)
        return calls[a], calls[b]

    while len(texts) < n:
        kind = kinds[rng.choice(len(kinds), p=np.asarray(weights) / sum(weights))]
        a, b = pair()
        payload = None
        if kind == "cq_grid":
            text = f"CQ {b} {_grid(rng)}"
        elif kind == "cq_modifier":
            text = f"CQ {rng.choice(['DX', 'NA', 'EU', 'POTA', 'SOTA', 'TEST', f'{rng.integers(0, 1000):03d}'])} {b} {_grid(rng)}"
        elif kind == "cq_nogrid":
            text = f"CQ {b}"
        elif kind == "cq_type4":
            text = f"CQ {rng.choice(long_calls)}"
        elif kind == "cq_suffix":
            text = f"CQ {b}/{rng.choice(['R', 'P'])} {_grid(rng)}"
        elif kind == "grid":
            text = f"{a} {b} {_grid(rng)}"
        elif kind == "report":
            text = f"{a} {b} {int(rng.integers(-24, 11)):+03d}"
        elif kind == "r_report":
            text = f"{a} {b} R{int(rng.integers(-24, 11)):+03d}"
        elif kind in ("rr73", "rrr", "s73"):
            text = f"{a} {b} {dict(rr73='RR73', rrr='RRR', s73='73')[kind]}"
        elif kind == "two_calls":
            text = f"{a} {b}"
        elif kind == "r_grid":
            text = f"{a} {b} R {_grid(rng)}"
        elif kind == "suffix":
            sfx = rng.choice(["R", "P"])
            text = rng.choice([f"{a}/{sfx} {b} {_grid(rng)}", f"{a} {b}/{sfx} {int(rng.integers(-20, 6)):+03d}", f"{a}/{sfx} {b}/{sfx} 73"])
        elif kind == "hashed":
            lc = rng.choice(long_calls)
            text = rng.choice([f"<{lc}> {b} {int(rng.integers(-20, 6)):+03d}", f"{a} <{lc}> RR73", f"<{lc}> {b} {_grid(rng)}"])
        elif kind == "type4":
            lc = rng.choice(long_calls)
            text = rng.choice([f"<{b}> {lc} {rng.choice(['RRR', 'RR73', '73'])}", f"{lc} <{b}> {rng.choice(['RRR', 'RR73', '73'])}", f"<{b}> {lc}", f"{lc} <{b}>"])
        elif kind == "free_text":
            text = str(rng.choice(free)) if rng.random() < 0.5 else "".join(rng.choice(list(" 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ+-./?"), size=rng.integers(3, 14))).strip() or "73"
        elif kind == "free_text_cq":
            text = str(rng.choice(free_cq))
        elif kind == "telemetry":
            text = "".join(rng.choice(list("01234567"), size=1)) + "".join(rng.choice(list("0123456789ABCDEF"), size=17))
        else:                                   # a valid codeword of a type unpack77 has no branch for: i3 = 3, 5, or 0.n3 with n3 in 1..4, 6
            raw = rng.integers(0, 256, 10, dtype=np.uint8)
            i3, n3 = (int(rng.choice([3, 5])), int(rng.integers(0, 8))) if rng.random() < 0.5 else (0, int(rng.choice([1, 2, 3, 4, 6])))
            raw[8] = (raw[8] & 0xFE) | (n3 >> 2)
            raw[9] = ((n3 & 3) << 6) | (i3 << 3)
            payload, text = raw, None
        text = None if text is None else " ".join(str(text).split())
        if payload is None:
            payload = pack77(text)
        key = payload.tobytes()
        if key in seen:
            continue
        seen.add(key)
        texts.append(text)
        payloads.append(payload)
    tones = np.stack([encode(p) for p in payloads])
    return texts, tones


def amplitude_for_snr(snr_db, noise_sigma=1.0):
    """amplitude of a constant-envelope signal for an SNR quoted in 2500 Hz, with complex noise of
    variance sigma^2 per component over the 3200 Hz sample bandwidth"""
    return np.sqrt(2.0 * noise_sigma ** 2 * 2500.0 / 3200.0 * 10.0 ** (np.asarray(snr_db) / 10.0))


def frame_signals(first_frame, nframes, nsig, pool_tones, snr_range=(-18.0, 0.0), f_range=(100.0, 1500.0),
                  dt_range=(0.0, 1.8), dup_fraction=0.0):
    """signal descriptors for global frames [first_frame, first_frame + nframes); frame g is seeded
    with SEED_BASE + g so that any shard of any world size describes the same global batch.
    dup_fraction > 0: in that share of the frames the second signal repeats the first one's message at its own
    frequency, time and strength (one message heard twice: the dedup table of rtlsdr_ft8d.c:1487-1507)."""
    sig = np.zeros((nframes, max(nsig, 1)), SIGNAL_DTYPE)
    picks = np.zeros((nframes, max(nsig, 1)), np.int32)
    for k in range(nframes):
        rng = np.random.default_rng(SEED_BASE + first_frame + k)
        idx = rng.integers(0, pool_tones.shape[0], nsig)
        if dup_fraction > 0 and nsig >= 2 and np.random.default_rng(SEED_BASE + 7919 * (first_frame + k) + 1).random() < dup_fraction:
            idx[1] = idx[0]                                      # (its own generator: the frame's other draws stay what they were)
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
    Ragged jobs: with `total_frames` the shards are those of shard_range(total_frames, r, world) -- sizes that differ by
    one frame when the total does not divide -- every rank's segment is padded to the largest shard so that the
    collective stays ONE equal-sized all-gather, and `gathered()` drops the padding.
    Works on RCCL (device tensors) and on gloo (CPU tensors, tests/test_dist_gloo.py)."""

    RECORD_BYTES = 1400

    def __init__(self, frames, world_size, device, collective=None, total_frames=None):
        import torch
        self.frames, self.world = frames, world_size
        self.collective = (world_size > 1) if collective is None else collective     # True forces it for one rank too
        if total_frames is None:
            self.sizes = [frames] * world_size
        else:
            self.sizes = [hi - lo for lo, hi in (shard_range(total_frames, r, world_size) for r in range(world_size))]
            if frames not in self.sizes:
                raise ValueError(f"{frames} frames is not a shard of {total_frames} frames over {world_size} ranks ({self.sizes})")
        self.cap = max(self.sizes)                       # frames per segment of the exchange
        seg = self.cap * (self.RECORD_BYTES + 4)
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
        n = self.cap * self.RECORD_BYTES
        return flat[:self.frames * self.RECORD_BYTES].view(self.frames, self.RECORD_BYTES), flat[n:n + 4 * self.frames].view(dtype=__import__("torch").int32)

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
        """(all_spots [sum of shards, 1400], all_counts [sum of shards]) of step k in global frame order (copies)"""
        import torch
        i = k & 1
        if self._work[i] is not None:
            self._work[i].wait()
            self._work[i] = None
        n = self.cap * self.RECORD_BYTES
        if not self.collective:
            s, c = self._local[i][:self.frames * self.RECORD_BYTES], self._local[i][n:n + 4 * self.frames]
            return s.view(self.frames, self.RECORD_BYTES).clone(), c.view(dtype=torch.int32).clone()
        seg = self._all[i].view(self.world, -1)
        # copies: the receive buffer is overwritten by the exchange of step k + 2
        if len(set(self.sizes)) == 1:
            spots = seg[:, :n].reshape(self.world * self.cap, self.RECORD_BYTES).clone()
            counts = seg[:, n:].clone(memory_format=torch.contiguous_format).view(dtype=torch.int32).reshape(-1)
            return spots, counts
        spots = torch.cat([seg[r, :m * self.RECORD_BYTES].view(m, self.RECORD_BYTES) for r, m in enumerate(self.sizes)])
        counts = torch.cat([seg[r, n:n + 4 * m].clone().view(dtype=torch.int32) for r, m in enumerate(self.sizes)])
        return spots, counts
