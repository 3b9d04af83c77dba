"""Seeded synthetic FT8 frames on the host (numpy), SURVEY.md section 8(d) recipe:
complex AWGN + S plain-CPFSK signals (512 samples/symbol, as rtlsdr_ft8d.c:946-955), then
peak-normalised to 0.5 (rtlsdr_ft8d.c:248-263).  Messages are standard "CQ <call> <grid>"."""
import numpy as np

NSAMPLES = 48000
LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"


def random_message(rng, cq=True):
    pfx = rng.choice(["K", "W", "N", "G", "F", "DL", "JA", "VK", "EA", "OH"])
    call = pfx + str(rng.integers(0, 10)) + "".join(rng.choice(list(LETTERS), size=rng.integers(1, 4)))
    grid = LETTERS[rng.integers(0, 18)] + LETTERS[rng.integers(0, 18)] + str(rng.integers(0, 10)) + str(rng.integers(0, 10))
    if cq:
        return f"CQ {call} {grid}"
    pfx2 = rng.choice(["K", "W", "G", "DL"])
    call2 = pfx2 + str(rng.integers(0, 10)) + "".join(rng.choice(list(LETTERS), size=rng.integers(1, 4)))
    return f"{call2} {call} {grid}"


def amplitude_for_snr(snr_db, noise_sigma):
    """SNR in a 2500 Hz reference bandwidth; complex noise of variance sigma^2 per component
    spread over 3200 Hz."""
    noise_power_2500 = 2.0 * noise_sigma ** 2 * 2500.0 / 3200.0
    return float(np.sqrt(noise_power_2500 * 10.0 ** (snr_db / 10.0)))


def cpfsk(tones, f0_hz, start_sample, amplitude, n=NSAMPLES):
    """plain FSK with continuous phase: tone k at f0 + 6.25*tone Hz for 512 samples"""
    out_i = np.zeros(n, np.float64)
    out_q = np.zeros(n, np.float64)
    freqs = f0_hz + 6.25 * np.repeat(np.asarray(tones, np.float64), 512)
    phase = 2.0 * np.pi * np.concatenate([[0.0], np.cumsum(freqs)[:-1]]) / 3200.0
    idx = start_sample + np.arange(freqs.size)
    ok = (idx >= 0) & (idx < n)
    out_i[idx[ok]] = amplitude * np.cos(phase[ok])
    out_q[idx[ok]] = amplitude * np.sin(phase[ok])
    return out_i, out_q


def make_frame(seed, nsig, encode_fn, snr_range=(-18.0, 0.0), noise_sigma=1.0, cq_fraction=1.0,
               f_range=(100.0, 1500.0), dt_range=(0.0, 1.8)):
    """returns (iq float32 [2][48000], list of messages)"""
    rng = np.random.default_rng(seed)
    fi = rng.normal(0.0, noise_sigma, NSAMPLES)
    fq = rng.normal(0.0, noise_sigma, NSAMPLES)
    msgs = []
    for _ in range(nsig):
        msg = random_message(rng, cq=rng.random() < cq_fraction)
        tones = encode_fn(msg)
        f0 = rng.uniform(*f_range)
        t0 = rng.uniform(*dt_range)
        amp = amplitude_for_snr(rng.uniform(*snr_range), noise_sigma)
        si, sq = cpfsk(tones, f0, int(round(t0 * 3200)), amp)
        fi += si
        fq += sq
        msgs.append(msg)
    i32 = fi.astype(np.float32)
    q32 = fq.astype(np.float32)
    peak = max(np.abs(i32).max(), np.abs(q32).max(), np.float32(1e-24))
    scale = np.float32(0.5) / peak
    iq = np.stack([i32 * scale, q32 * scale]).astype(np.float32)
    return iq, msgs


def make_mixed_frame(seed, nsig, snr_range, texts, tones, noise_sigma=1.0):
    """a frame of on-air style traffic: nsig messages drawn from a pool (rtlsdr_ft8d_amd.workload.mixed_message_pool:
    texts / tones) plus the first one again at another frequency (one message heard twice), in AWGN, peak-normalised
    to 0.5.  Returns (iq float32 [2][48000], list of planted texts; None = a payload unpack77 refuses)."""
    rng = np.random.default_rng(seed)
    fi, fq = rng.normal(0.0, noise_sigma, NSAMPLES), rng.normal(0.0, noise_sigma, NSAMPLES)
    picks = list(rng.integers(0, len(texts), nsig))
    if picks:
        picks.append(picks[0])
    for k in picks:
        si, sq = cpfsk(tones[k], rng.uniform(100.0, 1500.0), int(round(rng.uniform(0.0, 1.8) * 3200)),
                       amplitude_for_snr(rng.uniform(*snr_range), noise_sigma))
        fi += si
        fq += sq
    i32, q32 = fi.astype(np.float32), fq.astype(np.float32)
    scale = np.float32(0.5) / max(np.abs(i32).max(), np.abs(q32).max(), np.float32(1e-24))
    return np.stack([i32 * scale, q32 * scale]).astype(np.float32), [texts[k] for k in picks]


def oracle_encode_fn(oracle):
    def enc(msg):
        rc, p = oracle.pack77(msg)
        assert rc == 0, msg
        return oracle.encode(p)
    return enc
