export TMPDIR=/tmp
CMD0="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs --no-clock-sampler"
# FORMS: selector of the last-stage form of the waterfall kernel: 0 rows (product), 8 lds (A/B build only: csrc/ft8gpu_internal.h)
for f in ${FORMS:-0 8}; do
  CMD="$CMD0 --ab-lib --debug-flags $f"
  rm -rf gpurun_out/wfpmc$f; mkdir -p gpurun_out/wfpmc$f
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/wfpmc$f/sq -o sq -- $CMD > gpurun_out/wfpmc$f/sq.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/wfpmc$f/sq2 -o sq2 -- $CMD > gpurun_out/wfpmc$f/sq2.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d gpurun_out/wfpmc$f/sq3 -o sq3 -- $CMD > gpurun_out/wfpmc$f/sq3.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/wfpmc$f/fetch -o fetch -- $CMD > gpurun_out/wfpmc$f/fetch.log 2>&1
  python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for p in glob.glob("gpurun_out/wfpmc$f/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "waterfall" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("form $f", {k: round(sum(v) / 1e6, 2) for k, v in sorted(acc.items())}, "dispatches", max(len(v) for v in acc.values()))
PY
done
