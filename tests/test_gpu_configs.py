"""GPU tests at the sizes of BASELINE.json's configs, through size-independent properties
(determinism, batch-composition independence, permutation equivariance, planted-signal recall,
checksums) plus the CPU oracle on samples."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _synth(ft8, workload, dec, first, n, nsig, snr, pool_tones, seed_off=0):
    import torch
    sig, picks = workload.frame_signals(first, n, nsig, pool_tones, snr_range=snr)
    iq = torch.empty((n, 2, ft8.NSAMPLES), dtype=torch.float32, device="cuda")
    dec.synth_frames(sig, n, nsig, 1.0, workload.SEED_BASE + first + seed_off, iq)
    return iq, sig, picks


def _decode_dev(ft8, dec, iq, n):
    import torch
    spots = torch.zeros((n, ft8.MAX_MESSAGES * 28), dtype=torch.uint8, device="cuda")
    nres = torch.zeros((n,), dtype=torch.int32, device="cuda")
    dec.decode_batch_dev(iq, n, spots, nres)
    dec.synchronize()
    return spots.cpu().numpy().view(ft8.RESULT_DTYPE).reshape(n, ft8.MAX_MESSAGES), nres.cpu().numpy()


def test_config3_full_batch_properties(oracle):
    """configs[2]: 4096 synthetic frames, full pipeline on one GPU"""
    import torch
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    B, S = 4096, 20
    msgs, tones = workload.message_pool()
    pool_calls = {m.split()[1] for m in msgs}
    with ft8.Decoder(device=0, max_frames=B) as dec:
        iq, sig, picks = _synth(ft8, workload, dec, 0, B, S, (-18.0, 0.0), tones)
        d1, n1 = _decode_dev(ft8, dec, iq, B)
        d2, n2 = _decode_dev(ft8, dec, iq, B)
        # determinism (checksum of checksums)
        assert hashlib.sha256(d1.tobytes()).hexdigest() == hashlib.sha256(d2.tobytes()).hexdigest()
        assert np.array_equal(n1, n2)
        # batch-composition independence: a sub-batch alone gives the same records
        ds, ns = _decode_dev(ft8, dec, iq[1000:1064].contiguous(), 64)
        assert np.array_equal(ns, n1[1000:1064]) and ds.tobytes() == d1[1000:1064].tobytes()
        # permutation equivariance
        perm = torch.randperm(256, generator=torch.Generator().manual_seed(5))
        dp, npm = _decode_dev(ft8, dec, iq[:256][perm.cuda()].contiguous(), 256)
        assert np.array_equal(npm, n1[:256][perm.numpy()]) and dp.tobytes() == d1[:256][perm.numpy()].tobytes()
        # chunked context (max_frames smaller than the batch) gives the same result
        with ft8.Decoder(device=0, max_frames=1000) as dec_small:
            dc, nc = _decode_dev(ft8, dec_small, iq, B)
        assert np.array_equal(nc, n1) and dc.tobytes() == d1.tobytes()
        # host-buffer entry: uploads are pipelined in 512-frame chunks (ragged last chunk) under the kernels
        hn = 512 * 2 + 77
        host_all = iq[:hn].cpu().numpy()
        dh, nh = dec.decode_batch(host_all)
        assert np.array_equal(nh, n1[:hn]) and dh.tobytes() == d1[:hn].tobytes()
        host_iq = host_all[:24]
    # recall / false decodes against what was planted
    found = planted = false_calls = total_calls = 0
    for f in range(B):
        calls = {x["call"].decode() for x in d1[f][:n1[f]] if x["call"]}
        total_calls += len(calls)
        false_calls += sum(1 for c in calls if c not in pool_calls)
        strong = [msgs[picks[f, s]].split()[1] for s in range(S)
                  if sig[f, s]["amplitude"] >= workload.amplitude_for_snr(-8.0)]
        planted += len(strong)
        found += sum(1 for c in strong if c in calls)
    assert n1.sum() > 8 * B                         # the batch really decodes (about 12 messages per frame)
    assert found >= 0.75 * planted                  # strong planted signals are recovered (collisions cost some)
    assert false_calls <= 1e-3 * total_calls + 2    # CRC-14 false decodes are rare
    # oracle on a sample of the very same frames
    for f in range(host_iq.shape[0]):
        rdec, rn = oracle.subsystem(host_iq[f, 0], host_iq[f, 1])
        assert n1[f] == rn and d1[f].tobytes() == rdec.tobytes()


def test_config2_gpu_waterfall_sync_cpu_ldpc(oracle):
    """configs[1]: 256 frames, waterfall + sync on the GPU, LDPC on the CPU (the oracle's ft8_decode
    and spot loop fed with the GPU's waterfall) must equal the all-GPU result"""
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    B, S = 256, 20
    _, tones = workload.message_pool()
    with ft8.Decoder(device=0, max_frames=B) as dec:
        iq, _, _ = _synth(ft8, workload, dec, 5000, B, S, (-18.0, 0.0), tones)
        host_iq = iq.cpu().numpy()
        mag = dec.waterfall(host_iq)
        cands, counts = dec.find_sync(mag)
        gdec, gn = dec.decode_batch(host_iq)
    for f in range(0, B, 4):                                 # every 4th frame through the CPU LDPC
        rc = oracle.find_sync(mag[f])
        assert counts[f] == len(rc) and np.array_equal(cands[f, :counts[f]], rc)
        rdec, rn = oracle.subsystem_from_waterfall(mag[f])
        assert gn[f] == rn and gdec[f].tobytes() == rdec.tobytes()


def test_config5_oversubscribed_candidates(oracle):
    """configs[4]: K_MAX_CANDIDATES x 4, 60 weak signals per frame: stresses heap eviction and BP occupancy"""
    import rtlsdr_ft8d_amd as ft8
    from rtlsdr_ft8d_amd import workload
    B, S = 48, 60
    _, tones = workload.message_pool()
    with ft8.Decoder(device=0, max_frames=B, max_candidates=480) as dec:
        iq, _, _ = _synth(ft8, workload, dec, 9000, B, S, (-24.0, -14.0), tones)
        host_iq = iq.cpu().numpy()
        mag = dec.waterfall(host_iq)
        cands, counts = dec.find_sync(mag)
        gdec, gn = dec.decode_batch(host_iq)
    assert counts.max() > 120                                # the cap of 120 would have been exceeded
    p = oracle.default_params(10, 480, 20)
    for f in range(0, B, 3):
        rc = oracle.find_sync(mag[f], 480, 10)
        assert counts[f] == len(rc) and np.array_equal(cands[f, :counts[f]], rc)
        rdec, rn = oracle.subsystem(host_iq[f, 0], host_iq[f, 1], p)
        assert gn[f] == rn and gdec[f].tobytes() == rdec.tobytes()


def test_non_overlapped_pipeline_gives_the_same_records(oracle):
    """FT8GPU_OVERLAP=0: one launch per stage for the whole batch, no side stream, no chunked upload.
    Same spot records as the default two-half pipeline on 1200 frames (device and host buffers)."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, '.')
import torch, rtlsdr_ft8d_amd as ft8
from rtlsdr_ft8d_amd import workload
n = 1200
_, tones = workload.message_pool()
with ft8.Decoder(device=0, max_frames=n) as dec:
    sig, _ = workload.frame_signals(0, n, 20, tones)
    iq = torch.empty((n, 2, ft8.NSAMPLES), dtype=torch.float32, device='cuda')
    dec.synth_frames(sig, n, 20, 1.0, workload.SEED_BASE, iq)
    spots = torch.zeros((n, 1400), dtype=torch.uint8, device='cuda'); nres = torch.zeros((n,), dtype=torch.int32, device='cuda')
    dec.decode_batch_dev(iq, n, spots, nres); dec.synchronize()
    d, c = dec.decode_batch(iq.cpu().numpy())
assert np.array_equal(c, nres.cpu().numpy()) and d.tobytes() == spots.cpu().numpy().tobytes()
print('DIGEST', hashlib.sha256(spots.cpu().numpy().tobytes() + nres.cpu().numpy().tobytes()).hexdigest(), int(nres.sum()))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for overlap in ("1", "0"):
        env = dict(os.environ, FT8GPU_OVERLAP=overlap)
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append([l for l in out.stdout.splitlines() if l.startswith("DIGEST")][-1])
    assert outs[0] == outs[1] and int(outs[0].split()[-1]) > 8 * 1200
