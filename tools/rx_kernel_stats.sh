#!/bin/bash
# Per-kernel times of the RX front end (rx.hip) on the GPU box:  gpurun -- 'bash tools/rx_kernel_stats.sh [lib.so]'
# rocprofv3 --kernel-trace --stats around tools/ab_rx.py (69 launches of 16 captures); prints calls, average and minimum ns.
LIB=${1:-$GRAFT_REPO_ROOT/rtlsdr_ft8d_amd/libft8gpu.so}
OUT=$GRAFT_REPO_ROOT/gpurun_out/rx_kernel_stats
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/ab_rx.py --libs $LIB --rounds 3 > /dev/null 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -f "$f" ] || { echo "no kernel_stats.csv under $OUT"; exit 1; }
cp $f $GRAFT_REPO_ROOT/gpurun_out/rx_kernel_stats.csv
grep -E 'ft8_rx' $f | awk -F, '{print substr($1,1,52), $(NF-6), $(NF-4), $(NF-2)}'
