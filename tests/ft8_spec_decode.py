"""A third writing of the ft8_lib stages the reference calls (ft8_find_sync, rtlsdr_ft8d.c:1450; ft8_decode, :1476),
in numpy float32 / Python integers, written from SURVEY.md Appendix A.2-A.4 and the tables of Appendix B -- NOT
from oracle/ft8_oracle.c or the HIP kernels.  Test infrastructure only: it cannot pin anything against upstream
(the submodule is absent), but a transcription slip in the oracle that the kernels merely copied shows up as a
disagreement between two independently written restatements of the same appendix (tests/test_oracle.py).

Everything is evaluated in the appendix's own operation order; float32 operations are single numpy float32
operations (numpy does not fuse), so results are comparable bit for bit.
"""
import pathlib
import re

import numpy as np

F32 = np.float32
ROOT = pathlib.Path(__file__).resolve().parents[1]

COSTAS = (3, 1, 4, 0, 6, 5, 2)
GRAY = (0, 1, 3, 2, 5, 6, 4, 7)
NUM_BLOCKS, NUM_BINS, TIME_OSR, FREQ_OSR = 92, 256, 2, 2
BLOCK_STRIDE = TIME_OSR * FREQ_OSR * NUM_BINS


def tables():
    """Nm[83][7], Mn[174][3] (1-origin, 0 = padding) parsed from SURVEY.md Appendix B.2 / B.3"""
    txt = (ROOT / "SURVEY.md").read_text()
    app = txt[txt.index("## Appendix B"):]
    b2 = app[app.index("### B.2"):app.index("### B.3")]
    b3 = app[app.index("### B.3"):app.index("### B.4")]
    nm = {int(i): [int(x) for x in v.split(",")] for i, v in re.findall(r"(\d+):\{([^}]*)\}", b2)}
    mn = {int(i): [int(x) for x in v.split(",")] for i, v in re.findall(r"(\d+):\{([^}]*)\}", b3)}
    Nm = np.array([nm[m] for m in range(83)], np.int32)
    Mn = np.array([mn[n] for n in range(174)], np.int32)
    return Nm, Mn


# ---- A.2: sync score of every scan position, heap, final order -----------------------------------------
def score_map(mag):
    """int scores [time_sub][freq_sub][36 time offsets -12..23][249 freq offsets], A.2 sync_score"""
    wf = np.asarray(mag, np.uint8).reshape(NUM_BLOCKS, TIME_OSR, FREQ_OSR, NUM_BINS).astype(np.int64)
    out = np.zeros((TIME_OSR, FREQ_OSR, 36, NUM_BINS - 7), np.int64)
    f0 = np.arange(NUM_BINS - 7)
    for ts in range(TIME_OSR):
        for fs in range(FREQ_OSR):
            plane = wf[:, ts, fs, :]                                   # [block][bin]; a time neighbour is one BLOCK away
            for ti, t0 in enumerate(range(-12, 24)):
                score = np.zeros(f0.shape, np.int64)
                n = 0
                for m in range(3):
                    for k in range(7):
                        block_abs = t0 + 36 * m + k
                        if block_abs < 0:
                            continue
                        if block_abs >= NUM_BLOCKS:
                            break                                      # leaves the k loop only
                        sm = COSTAS[k]
                        p = plane[block_abs]
                        here = p[f0 + sm]
                        if sm > 0:
                            score += here - p[f0 + sm - 1]
                            n += 1
                        if sm < 7:
                            score += here - p[f0 + sm + 1]
                            n += 1
                        if k > 0 and block_abs > 0:
                            score += here - plane[block_abs - 1][f0 + sm]
                            n += 1
                        if k + 1 < 7 and block_abs + 1 < NUM_BLOCKS:
                            score += here - plane[block_abs + 1][f0 + sm]
                            n += 1
                if n > 0:
                    score = np.fix(score / n).astype(np.int64)        # C integer division truncates toward zero
                out[ts, fs, ti] = score
    return out


def find_sync(mag, num_candidates=120, min_score=10, scores=None):
    """the appendix's heap, candidate for candidate; returns [(score, time_offset, freq_offset, time_sub, freq_sub)]
    in the final (heap-sorted, descending) order"""
    sc = score_map(mag) if scores is None else scores
    heap = []

    def down(c, size):
        while True:
            smallest, left, right = c, 2 * c + 1, 2 * c + 2
            if left < size and heap[left][0] < heap[smallest][0]:
                smallest = left
            if right < size and heap[right][0] < heap[smallest][0]:
                smallest = right
            if smallest == c:
                return
            heap[c], heap[smallest] = heap[smallest], heap[c]
            c = smallest

    def up(c):
        while c > 0:
            p = (c - 1) // 2
            if heap[c][0] >= heap[p][0]:
                return
            heap[c], heap[p] = heap[p], heap[c]
            c = p

    for ts in range(TIME_OSR):
        for fs in range(FREQ_OSR):
            for ti in range(36):
                row = sc[ts, fs, ti]
                for f in np.nonzero(row >= min_score)[0]:
                    s = int(row[f])
                    if len(heap) == num_candidates and s > heap[0][0]:
                        heap[0] = heap[-1]
                        heap.pop()
                        down(0, len(heap))
                    if len(heap) < num_candidates:
                        heap.append((s, ti - 12, int(f), ts, fs))
                        up(len(heap) - 1)
    size = len(heap)
    length = size
    while length > 1:
        heap[0], heap[length - 1] = heap[length - 1], heap[0]
        length -= 1
        down(0, length)
    return heap[:size]


# ---- A.3: likelihoods ------------------------------------------------------------------------------------
def extract_likelihood(mag, cand):
    _, t0, f0, ts, fs = cand
    flat = np.asarray(mag, np.uint8).reshape(-1)
    index = ((t0 * TIME_OSR + ts) * FREQ_OSR + fs) * NUM_BINS + f0
    log174 = np.zeros(174, F32)
    for k in range(58):
        sym = k + (7 if k < 29 else 14)
        block = t0 + sym
        if block < 0 or block >= NUM_BLOCKS:
            continue
        ps = index + sym * BLOCK_STRIDE
        s2 = [F32(flat[ps + GRAY[j]]) for j in range(8)]
        log174[3 * k + 0] = max(s2[4], s2[5], s2[6], s2[7]) - max(s2[0], s2[1], s2[2], s2[3])
        log174[3 * k + 1] = max(s2[2], s2[3], s2[6], s2[7]) - max(s2[0], s2[1], s2[4], s2[5])
        log174[3 * k + 2] = max(s2[1], s2[3], s2[5], s2[7]) - max(s2[0], s2[2], s2[4], s2[6])
    return log174


def normalize_logl(log174):
    s = F32(0)
    s2 = F32(0)
    for v in log174:                                   # index order, float accumulators
        s = F32(s + v)
        s2 = F32(s2 + F32(v * v))
    inv_n = F32(1.0) / F32(174)
    variance = F32(F32(s2 - F32(F32(s * s) * inv_n)) * inv_n)
    with np.errstate(divide="ignore", invalid="ignore"):
        norm = np.sqrt(F32(24.0) / variance, dtype=F32)
    return (log174 * norm).astype(F32)


# ---- A.4: sum-product decoder ---------------------------------------------------------------------------
def fast_tanh(x):
    x = x.astype(F32)
    x2 = x * x
    a = x * (F32(945) + x2 * (F32(105) + x2))
    b = F32(945) + x2 * (F32(420) + x2 * F32(15))
    with np.errstate(all="ignore"):
        r = a / b
    r = np.where(x < F32(-4.97), F32(-1), r)
    return np.where(x > F32(4.97), F32(1), r).astype(F32)


def fast_atanh(x):
    x = x.astype(F32)
    x2 = x * x
    a = x * (F32(945) + x2 * (F32(-735) + x2 * F32(64)))
    b = F32(945) + x2 * (F32(-1050) + x2 * F32(225))
    with np.errstate(all="ignore"):
        return (a / b).astype(F32)


class BP:
    def __init__(self):
        self.Nm, self.Mn = tables()
        self.num_rows = (self.Nm > 0).sum(axis=1)
        # for variable n, edge m_idx: the row and this variable's position in it
        self.edge_row = self.Mn - 1                                                   # [174][3]
        self.edge_pos = np.zeros((174, 3), np.int64)
        for n in range(174):
            for e in range(3):
                self.edge_pos[n, e] = list(self.Nm[self.edge_row[n, e]]).index(n + 1)

    def check(self, plain):
        errors = 0
        for m in range(83):
            x = 0
            for j in range(self.num_rows[m]):
                x ^= int(plain[self.Nm[m, j] - 1])
            errors += x
        return errors

    def decode(self, codeword, max_iters):
        """returns (min_errors, iterations entered, plain174 of the LAST hard decision)"""
        cw = np.asarray(codeword, F32)
        tov = np.zeros((174, 3), F32)
        toc = np.zeros((83, 7), F32)
        min_errors = 83
        plain = np.zeros(174, np.uint8)
        iters = 0
        for it in range(max_iters):
            iters = it
            total = ((cw + tov[:, 0]) + tov[:, 1]) + tov[:, 2]
            plain = (total > 0).astype(np.uint8)
            if not plain.any():
                break
            errors = self.check(plain)
            if errors < min_errors:
                min_errors = errors
                if errors == 0:
                    break
            # messages to the checks: Tnm = codeword[n] + the OTHER two tov[n][.] in ascending m_idx order
            for e in range(3):
                others = [o for o in range(3) if o != e]
                tnm = (cw + tov[:, others[0]]) + tov[:, others[1]]
                x = -tnm / F32(2)
                toc[self.edge_row[:, e], self.edge_pos[:, e]] = fast_tanh(x)
            # messages to the variables: product of the OTHER toc of the row, ascending n_idx, starting from 1.0f
            new = np.zeros((174, 3), F32)
            for e in range(3):
                rows, pos = self.edge_row[:, e], self.edge_pos[:, e]
                acc = np.ones(174, F32)
                for j in range(7):
                    use = (j < self.num_rows[rows]) & (j != pos)
                    acc = np.where(use, acc * toc[rows, j], acc).astype(F32)
                new[:, e] = F32(-2) * fast_atanh(acc)
            tov = new
            iters = it + 1
        return min_errors, iters, plain


def pack_bits(plain, nbits=91):
    out = bytearray(12)
    for i in range(nbits):
        if plain[i]:
            out[i >> 3] |= 0x80 >> (i & 7)
    return bytes(out)


def decode_candidate(bp, mag, cand, max_iters=20):
    """A.3 up to the packed bits: (ldpc_errors, iterations entered, a91 bytes)"""
    logl = normalize_logl(extract_likelihood(mag, cand))
    errors, iters, plain = bp.decode(logl, max_iters)
    return errors, iters, pack_bits(plain)
