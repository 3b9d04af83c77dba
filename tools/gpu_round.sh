#!/bin/bash
# one GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace + PMC passes of the bench command
TAG=${TAG:-r01}
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
python bench.py --steps ${STEPS:-10} --warmup 3 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --force-dist > gpurun_out/bench_dist1.log 2>&1; tail -1 gpurun_out/bench_dist1.log | cut -c1-200
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun1.log 2>&1; tail -1 gpurun_out/bench_torchrun1.log | cut -c1-200
export TMPDIR=/tmp
rm -rf gpurun_out/prof gpurun_out/pmc
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- $CMD > gpurun_out/prof.log 2>&1
mkdir -p gpurun_out/pmc
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc/sq -o sq -- $CMD > gpurun_out/pmc/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc/sq2 -o sq2 -- $CMD > gpurun_out/pmc/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/fetch -o fetch -- $CMD > gpurun_out/pmc/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/write -o write -- $CMD > gpurun_out/pmc/write.log 2>&1
python tools/pmc_summary.py --traffic gpurun_out/pmc_traffic.json gpurun_out/pmc/*/*_counter_collection.csv > gpurun_out/pmc_counters.json
cat gpurun_out/prof/trace_kernel_stats.csv | cut -c1-160 | head -9
