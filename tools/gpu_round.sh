#!/bin/bash
# one GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace of the bench command
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
python bench.py --steps ${STEPS:-5} --warmup 2 > gpurun_out/bench.log 2>&1; tail -3 gpurun_out/bench.log
export TMPDIR=/tmp
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof.log 2>&1
tail -2 gpurun_out/prof.log
find gpurun_out/prof -name "*stats*" | head; cat gpurun_out/prof/*kernel_stats.csv 2>/dev/null | head -12; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
