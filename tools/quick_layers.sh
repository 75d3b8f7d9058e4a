#!/bin/bash
# Per-layer table of one bench run under rocprofv3 (kernel trace only): gpurun -- bash tools/quick_layers.sh [fp32|split_f16] [tag]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
CONV=${1:-split_f16}
TAG=${2:-quick}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_q && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_q -o run --output-format csv -- python3 $R/bench.py --conv $CONV --no-fp32-mode --steps 5 --warmup 1 --parity-frames 0 --cpu-frames 0 > $R/gpurun_out/${TAG}_under_rocprof.log 2>&1 || exit 1
python3 $R/tools/layer_profile.py /tmp/prof_q/run_kernel_trace.csv 4096 4096 > $R/gpurun_out/${TAG}_layers.txt 2>&1 || exit 1
cp /tmp/prof_q/run_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
cat $R/gpurun_out/${TAG}_layers.txt
