"""No kernel of libft8gpu.so may use scratch memory, and nothing may follow bp_math.h's EXEC-narrowing assembly too
closely (CPU box: hipcc cross-compiles gfx950 without a GPU).  tools/kernel_resources.py does the work: it compiles
every csrc/*.hip to assembly with the product's flags and reads the kernels' metadata.

Round 3's LDPC kernel had a 128-byte scratch segment (the status record and unpack77's byte buffers were indexed
dynamically in private memory) and wrote 5.8 x the bytes of its own records to HBM because of it; round 4 composes
the record in LDS.  This test keeps it that way."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_kernel_uses_scratch_and_no_dpp_follows_a_cmpx(tmp_path):
    out = tmp_path / "res.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "--json", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.load(open(out))
    assert d["problems"] == []
    ks = {k["kernel"]: k for k in d["kernels"]}
    # the kernels of the hot path are all there, and the LDPC kernel keeps eight waves per SIMD
    for name in ("ft8_waterfall_kernel<0>", "ft8_sync_kernel<false>", "ft8_heap_kernel", "ft8_heap_simt_kernel",
                 "ft8_decode_kernel<false, 3>", "ft8_decode_kernel<true, 1>", "ft8_spots_kernel"):
        assert name in ks, (name, sorted(ks))
        assert ks[name]["scratch_bytes_per_lane"] == 0
    assert ks["ft8_decode_kernel<false, 3>"]["waves_per_simd_by_vgprs"] == 8
    assert ks["ft8_decode_kernel<false, 3>"]["sgprs"] <= 80          # 256-thread workgroups: 8 per CU only up to 80 SGPRs
    # four waves of at most 120 registers leave 32 per lane free: the heap replay's waves run beside the waterfall kernel
    assert ks["ft8_waterfall_kernel<0>"]["vgprs"] <= 120 and ks["ft8_waterfall_kernel<0>"]["waves_per_simd_by_vgprs"] == 4


def test_committed_counter_evidence_describes_these_kernel_sources():
    """profiles/pmc_traffic.json carries the hash of csrc/ it was collected on; bench.py reports the counter-derived roofline
    figures (traffic, valu_*) only while that hash matches the tree.  A mismatch is not an error of the code -- it means
    tools/gpu_round.sh has to run again on a GPU box -- so it is reported as a SKIP with the reason, not as a failure."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    with open(os.path.join(root, "profiles", "pmc_traffic.json")) as f:
        t = json.load(f)
    assert isinstance(t.get("csrc_sha"), str) and len(t["csrc_sha"]) == 16
    for kernel in ("waterfall", "sync", "heap", "decode", "spots"):
        # the busy fraction is a ratio of two counters read in different profiler passes: a saturated pipe can read 1.00x, so
        # tools/pmc_summary.py stores it clamped to 1.0 and keeps the raw quotient beside it
        assert t[kernel]["hbm_bytes_per_frame"] > 0 and 0.0 < t[kernel]["valu_busy_frac"] <= 1.0
        assert t[kernel]["valu_busy_frac"] == min(1.0, t[kernel]["valu_busy_frac_raw"]) and t[kernel]["valu_busy_frac_raw"] <= 1.02
    if t["csrc_sha"] != bench.csrc_hash():
        import pytest
        pytest.skip(f"profiles/pmc_traffic.json was collected on csrc {t['csrc_sha']}, the tree is {bench.csrc_hash()}: "
                    "re-run TAG=rNN bash tools/gpu_round.sh on a GPU box and copy gpurun_out/profiles_rNN/* to profiles/")


def test_hazard_scan_follows_the_control_flow():
    """the v_cmpx -> DPP scan of tools/kernel_resources.py walks branches (round 4's was a window of the next lines of
    text): a DPP instruction behind a taken branch is found, enough wait states on EVERY path clear it, and a DPP
    instruction that only follows in the text -- behind an unconditional branch elsewhere -- is not reported"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as K
    asm = """
kernel:
  v_cmpx_lt_f32 v1, v2
  s_cbranch_scc1 .LBB0_2
  s_nop 4
  v_add_f32 v1, v2, v3
.LBB0_1:
  s_endpgm
.LBB0_2:
  s_nop 1
  v_mov_b32_dpp v1, v2 row_shr:1
  s_branch .LBB0_1
"""
    assert K.cmpx_dpp_hazards(asm) == [(11, "v_mov_b32_dpp v1, v2 row_shr:1")]
    assert K.cmpx_dpp_hazards(asm.replace("s_nop 1", "s_nop 4")) == []
    loop = """
.LBB1_0:
  v_add_f32_dpp v5, v5, v5 row_shr:1
  s_nop 0
  v_cmpx_gt_u32 v0, v1
  s_cbranch_execnz .LBB1_0
  s_nop 4
  v_add_f32_dpp v6, v6, v6 row_shr:2
"""
    assert [ln for ln, _ in K.cmpx_dpp_hazards(loop)] == [3]           # the loop's back edge lands on DPP code; the exit path waits
    skipped = """
  v_cmpx_lt_f32 v1, v2
  s_branch .LBB2_9
  v_mov_b32_dpp v1, v2 row_shr:1
.LBB2_9:
  s_endpgm
"""
    assert K.cmpx_dpp_hazards(skipped) == []
