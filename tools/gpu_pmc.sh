#!/bin/bash
# rocprofv3 PMC passes over the bench command (separate passes; HBM counters per MI355X_MICROARCH.md)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc/sq -o sq -- $CMD > gpurun_out/pmc/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc/sq2 -o sq2 -- $CMD > gpurun_out/pmc/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/fetch -o fetch -- $CMD > gpurun_out/pmc/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/write -o write -- $CMD > gpurun_out/pmc/write.log 2>&1
ls -R gpurun_out/pmc | head -30
tail -2 gpurun_out/pmc/sq.log gpurun_out/pmc/sq2.log
