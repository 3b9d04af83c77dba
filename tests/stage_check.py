"""Every stage boundary of a device-resident batch against the CPU oracle -- shared by the GPU tests and tools/soak_parity.py.

What the spot records cannot show (Weak 1): a deviation inside a candidate that does not decode reaches nothing,
the text of a message that is not a CQ call never reaches a record.  Here, for every frame of a batch:
  * all 94 208 waterfall bytes                                   (rtlsdr_ft8d.c:1395-1435)
  * the ordered candidate list                                    (ft8_find_sync, call :1450)
  * the 48-byte status record of EVERY candidate -- parity errors, iterations entered, packed bits, both CRCs, unpack status,
    ok, text -- from BOTH forms of the LDPC kernel                (ft8_decode, call :1476):
      - ft8_decode_kernel<true,1>, the stage entry's own form: every byte;
      - ft8_decode_kernel<false,3>, the form the batch pipeline runs (FT8GPU_DBG_PIPELINE_FORM): it reports ldpc_errors
        as 0 or 83 only (it never counts the failing checks of a candidate that does not converge), so that field must be in
        {0, 83} and agree with the oracle on being zero; every other byte is compared.
The iteration cap (K_LDPC_ITERS, rtlsdr_ft8d.h:45) is a parameter on both sides."""
import numpy as np

FT8_LDPC_M = 83


def new_counters():
    return dict(frames=0, waterfall_cells=0, waterfall_cells_differing=0, waterfall_frames_differing=0, candidate_lists_differing=0,
                candidate_records=0, candidate_records_decoded_ok=0, records_differing_stage_form=0, records_differing_pipeline_form=0,
                pipeline_form_ldpc_errors_not_0_or_83=0)


def compare_with_oracle(O, iq_host, h_mag, h_cands, h_counts, g_stage, g_pipe, cap, min_score, iters, cores, c, first_bad=None, f0=0):
    """the comparison itself, numpy only (a CPU test feeds it doctored arrays: a checker must be seen to fail).
    iq_host [m][2][48000] f32; h_mag [m][94208] u8, h_cands [m][cap] CAND_DTYPE, h_counts [m] i32: the device's waterfall and
    candidate lists; g_stage / g_pipe [m][cap][48] u8: the status records of the two kernel forms."""
    m = len(h_counts)
    ref_mag = O.waterfall_batch(iq_host, False, cores)
    wdiff = h_mag != ref_mag
    c["waterfall_cells"] += h_mag.size
    c["waterfall_cells_differing"] += int(wdiff.sum())
    c["waterfall_frames_differing"] += int(wdiff.any(axis=1).sum())
    ref_cands, ref_counts = O.find_sync_batch(ref_mag, cap, min_score, cores)
    lbad = (h_counts != ref_counts) | (h_cands.view(np.uint64) != ref_cands.view(np.uint64)).any(axis=1).reshape(m)
    c["candidate_lists_differing"] += int(lbad.sum())
    want = O.decode_candidates_batch(h_mag, h_cands, h_counts, iters, cores)
    live = np.arange(cap)[None, :] < h_counts[:, None]
    sbad = (g_stage != want).any(axis=2)             # slots at and beyond the count included: nothing may be written there
    # pipeline form: the error count is 0 or 83 and is zero exactly where the oracle's is; all other bytes equal
    pe = g_pipe[:, :, 0:2].copy().view(np.int16)[..., 0]
    we = want[:, :, 0:2].copy().view(np.int16)[..., 0]
    c["pipeline_form_ldpc_errors_not_0_or_83"] += int((live & (pe != 0) & (pe != FT8_LDPC_M)).sum())
    pbad = ((g_pipe[:, :, 2:] != want[:, :, 2:]).any(axis=2) | ((pe == 0) != (we == 0)))
    c["frames"] += m
    c["candidate_records"] += int(h_counts.sum())
    c["candidate_records_decoded_ok"] += int((want[:, :, 9] == 1)[live].sum())
    c["records_differing_stage_form"] += int(sbad.sum())
    c["records_differing_pipeline_form"] += int(pbad.sum())
    if first_bad is not None and len(first_bad) < 8:
        for name, arr in (("waterfall", wdiff.any(axis=1)), ("candidate_list", lbad)):
            for f in np.flatnonzero(arr)[:2]:
                first_bad.append((name, f0 + int(f), -1))
        for name, arr in (("record_stage_form", sbad), ("record_pipeline_form", pbad)):
            for f, k in np.argwhere(arr)[:2]:
                first_bad.append((name, f0 + int(f), int(k)))
    return c


def stage_boundaries_vs_oracle(ft8, O, dec, iq, nframes, cap, min_score, iters, cores, counters=None, chunk=1024, first_bad=None, base_flags=0):
    """iq: [nframes][2][48000] float32 on the device; dec's parameters must already be (min_score, cap, iters) and cap must not
    exceed what dec was created / set for.  base_flags: the FT8GPU_DBG_* bits the context runs with (both passes keep them; they are
    what the context is left with).  Adds to `counters` (new_counters()) and
    returns it; `first_bad` (a list) receives up to 8 (stage, frame, candidate) tuples."""
    import torch
    c = counters if counters is not None else new_counters()
    for f0 in range(0, nframes, chunk):
        m = min(chunk, nframes - f0)
        part = iq[f0:f0 + m]
        mag = torch.empty((m, ft8.MAG_ARRAY), dtype=torch.uint8, device="cuda")
        cands = torch.zeros((m, cap, 8), dtype=torch.uint8, device="cuda")
        counts = torch.zeros((m,), dtype=torch.int32, device="cuda")
        st_stage = torch.zeros((m, cap, 48), dtype=torch.uint8, device="cuda")
        st_pipe = torch.zeros((m, cap, 48), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()                        # the fills run on torch's stream, the decoder on its own
        dec.waterfall_dev(part, m, mag)
        dec.find_sync_dev(mag, m, cands, counts)
        dec.decode_candidates_dev(mag, cands, counts, m, st_stage)
        dec.synchronize()
        dec.set_debug_flags(base_flags | ft8.DBG_PIPELINE_FORM)
        try:
            dec.decode_candidates_dev(mag, cands, counts, m, st_pipe)
            dec.synchronize()
        finally:
            dec.set_debug_flags(base_flags)
        h_mag, h_counts = mag.cpu().numpy(), counts.cpu().numpy()
        h_cands = cands.cpu().numpy().view(O.CAND_DTYPE).reshape(m, cap)
        g_stage, g_pipe = st_stage.cpu().numpy(), st_pipe.cpu().numpy()
        del mag, cands, counts, st_stage, st_pipe
        compare_with_oracle(O, part.cpu().numpy(), h_mag, h_cands, h_counts, g_stage, g_pipe, cap, min_score, iters, cores, c, first_bad, f0)
    return c


def differing(c):
    """total number of differences of any kind in a counter set"""
    return (c["waterfall_cells_differing"] + c["candidate_lists_differing"] + c["records_differing_stage_form"] +
            c["records_differing_pipeline_form"] + c["pipeline_form_ldpc_errors_not_0_or_83"])


def assert_clean(c, what, first_bad=None):
    assert differing(c) == 0, f"{what}: stage boundaries differ from the oracle: { {k: v for k, v in c.items() if 'differing' in k or 'not_0' in k} }, first {first_bad}"
    assert c["candidate_records"] > 0 and c["frames"] > 0
