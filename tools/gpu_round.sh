#!/bin/bash
# one GPU-box session: [tests] [bench] [prof] [pmc] -- stages picked by $STAGES (default: all)
#   tests  parity tests + smoke            bench  bench.py (configs 2, 4, 1, 3) + the RCCL one-rank legs (+ SUSTAIN=60: a sustained run)
#   prof   rocprofv3 kernel trace + stats  pmc    the four PMC passes (SQ / SQ+GRBM / FETCH_SIZE / WRITE_SIZE) for configs 2, 4 and 1
# Summaries land in gpurun_out/profiles_$TAG/; copy them to profiles/ to have them judged.  A profiler pass that
# fails (non-zero exit, or no CSV where one is expected) aborts its stage and leaves profiles/ untouched.
TAG=${TAG:-r06}
STAGES=${STAGES:-"tests bench prof pmc"}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }
die_stage() { echo "*** stage '$1' aborted: $2 (profiles/ left untouched)"; }
if has tests; then
  python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  grep -E "passed|failed|rc=" gpurun_out/pytest_gpu.log | tail -3
  python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
fi
CMD="python3 bench.py --steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-host-legs"
# one rocprofv3 counter pass; fails unless the profiler exits 0 AND wrote its counter CSV
pmc_pass() {   # $1 = dir, $2 = name, $3 = counters (quoted), $4.. = the program
  local d=$1 n=$2 c=$3; shift 3
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d/$n -o $n -- "$@" > $d/$n.log 2>&1 || { echo "rocprofv3 pass '$n' exited with $?"; return 1; }
  [ -s $d/$n/${n}_counter_collection.csv ] || { echo "rocprofv3 pass '$n' wrote no counter CSV"; return 1; }
}
pmc_passes() {   # $1 = output dir, $2.. = the program
  local d=$1; shift
  rm -rf $d; mkdir -p $d
  pmc_pass $d sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "$@" &&
  pmc_pass $d sq2 "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "$@" &&
  pmc_pass $d fetch "FETCH_SIZE" "$@" &&
  pmc_pass $d write "WRITE_SIZE" "$@"
}
if has prof; then
  ok=1
  rm -rf gpurun_out/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- $CMD > gpurun_out/prof.log 2>&1 || ok=0
  [ -s gpurun_out/prof/trace_kernel_stats.csv ] || ok=0
  rm -rf gpurun_out/prof4
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -o trace -- $CMD --config 4 > gpurun_out/prof4.log 2>&1 || ok=0
  [ -s gpurun_out/prof4/trace_kernel_stats.csv ] || ok=0
  if [ $ok = 1 ]; then
    cp gpurun_out/prof/trace_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
    cp gpurun_out/prof4/trace_kernel_stats.csv $OUT/${TAG}_bench_config4_kernel_stats.csv
    cut -c1-160 $OUT/${TAG}_bench_kernel_stats.csv | head -9
  else
    die_stage prof "a rocprofv3 --stats run failed or wrote no trace_kernel_stats.csv (see gpurun_out/prof*.log)"
  fi
fi
if has pmc; then
  if pmc_passes gpurun_out/pmc $CMD && pmc_passes gpurun_out/pmc4 $CMD --config 4 && pmc_passes gpurun_out/pmc1 $CMD --config 1; then
    python tools/pmc_summary.py --traffic $OUT/pmc_traffic.json gpurun_out/pmc/*/*_counter_collection.csv > $OUT/pmc_counters.json &&
    python tools/pmc_summary.py --traffic $OUT/pmc_traffic.json --frames 1024 --config-key config4 --command "$CMD --config 4" gpurun_out/pmc4/*/*_counter_collection.csv > $OUT/pmc_counters_config4.json &&
    python tools/pmc_summary.py --traffic $OUT/pmc_traffic.json --frames 256 --no-overlap --config-key config1 --command "$CMD --config 1" gpurun_out/pmc1/*/*_counter_collection.csv > $OUT/pmc_counters_config1.json &&
    python -c "import json; d=json.load(open('$OUT/pmc_traffic.json')); print({k: (v.get('hbm_bytes_per_launch'), v.get('valu_busy_frac')) for k, v in d.items() if isinstance(v, dict) and 'hbm_bytes_per_launch' in v})" &&
    cp $OUT/pmc_traffic.json $OUT/pmc_counters.json profiles/ ||   # so that the bench runs below report them
    die_stage pmc "tools/pmc_summary.py failed"
  else
    die_stage pmc "a counter pass failed"
  fi
fi
if has bench; then
  python bench.py --steps ${STEPS:-20} --warmup 5 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log > $OUT/${TAG}_bench.json; cut -c1-400 $OUT/${TAG}_bench.json
  python bench.py --config 4 --steps 40 --warmup 10 --cpu-frames 128 --no-host-legs > gpurun_out/bench_c4.log 2>&1; tail -1 gpurun_out/bench_c4.log > $OUT/${TAG}_bench_config4.json; cut -c1-300 $OUT/${TAG}_bench_config4.json
  python bench.py --config 1 --steps 5 --warmup 1 > gpurun_out/bench_c1.log 2>&1; tail -1 gpurun_out/bench_c1.log > $OUT/${TAG}_bench_config1.json; cut -c1-300 $OUT/${TAG}_bench_config1.json
  python bench.py --config 3 --steps 5 --warmup 2 > gpurun_out/bench_c3.log 2>&1; tail -1 gpurun_out/bench_c3.log > $OUT/${TAG}_bench_config3_one_gpu.json; cut -c1-300 $OUT/${TAG}_bench_config3_one_gpu.json
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-legs --force-dist > gpurun_out/bench_dist1.log 2>&1; tail -1 gpurun_out/bench_dist1.log > $OUT/${TAG}_bench_rccl_one_rank.json; cut -c1-200 $OUT/${TAG}_bench_rccl_one_rank.json
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-legs --force-dist --ctx-last > gpurun_out/bench_dist1_last.log 2>&1; tail -1 gpurun_out/bench_dist1_last.log > $OUT/${TAG}_bench_rccl_one_rank_ctx_last.json; cut -c1-200 $OUT/${TAG}_bench_rccl_one_rank_ctx_last.json
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-legs --force-dist --no-exchange > gpurun_out/bench_dist1_noex.log 2>&1; tail -1 gpurun_out/bench_dist1_noex.log > $OUT/${TAG}_bench_rccl_one_rank_no_exchange.json; cut -c1-200 $OUT/${TAG}_bench_rccl_one_rank_no_exchange.json
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-legs > gpurun_out/bench_plain2.log 2>&1; tail -1 gpurun_out/bench_plain2.log > $OUT/${TAG}_bench_no_dist_same_session.json; cut -c1-200 $OUT/${TAG}_bench_no_dist_same_session.json
  python bench.py --steps 20 --warmup 5 --traffic mixed --no-host-legs > gpurun_out/bench_mixed.log 2>&1; tail -1 gpurun_out/bench_mixed.log > $OUT/${TAG}_bench_mixed.json; cut -c1-200 $OUT/${TAG}_bench_mixed.json
  [ -n "${SUSTAIN:-}" ] && { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-legs --sustain-seconds $SUSTAIN > gpurun_out/bench_sustain.log 2>&1; tail -1 gpurun_out/bench_sustain.log > $OUT/${TAG}_bench_sustained_${SUSTAIN}s.json; cut -c1-200 $OUT/${TAG}_bench_sustained_${SUSTAIN}s.json; }
  [ -f tools/ab/libft8gpu_r05.so ] && { python tools/ab_libs.py --libs tools/ab/libft8gpu_r05.so rtlsdr_ft8d_amd/libft8gpu.so --rounds 3 > gpurun_out/ab_r05.log 2>&1; tail -1 gpurun_out/ab_r05.log > $OUT/${TAG}_ab_r05_vs_${TAG}.json; cut -c1-300 $OUT/${TAG}_ab_r05_vs_${TAG}.json; }
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-host-legs > gpurun_out/bench_torchrun1.log 2>&1; tail -1 gpurun_out/bench_torchrun1.log | cut -c1-200
fi
