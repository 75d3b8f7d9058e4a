#!/bin/bash
# (the rocprofv3 passes run bench.py without its checker legs - --parity-frames 0 --cpu-frames 0 - so that the kernel
# statistics hold the timed steps and the roofline pass only, not the per-frame launches of the recording_00 check)
# CONV=fp32 profiles the exact-fp32 arithmetic instead of the default split_f16 (ut_set_conv_arithmetic).
# Regenerate the judged summaries under gpurun_out/profiles_new/ on the GPU box (copy them into profiles/ after
# review): bench line, rocprofv3 kernel stats + per-layer table of the same command, PMC traffic (two passes).
#   gpurun -- bash tools/refresh_profiles.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
CONV=${CONV:-split_f16}
O=$R/gpurun_out/profiles_r04_$CONV
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py --conv $CONV > $O/bench_stdout.log 2>$O/bench_stderr.log || exit 1
tail -1 $O/bench_stdout.log > $O/bench_line.json
rm -rf /tmp/prof_s && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o run --output-format csv -- python3 $R/bench.py --conv $CONV --no-fp32-mode --steps 5 --warmup 1 --parity-frames 0 --cpu-frames 0 > $O/bench_under_rocprof.log 2>&1 || exit 1
grep -o "{\"metric.*" $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
cp /tmp/prof_s/run_kernel_stats.csv $O/bench_kernel_stats.csv
python3 $R/tools/layer_profile.py /tmp/prof_s/run_kernel_trace.csv 4096 4096 > $O/bench_conv_layers.txt 2>&1 || exit 1
rm -rf /tmp/prof_f && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o run --output-format csv -- python3 $R/bench.py --conv $CONV --no-fp32-mode --steps 2 --warmup 1 --no-roofline --parity-frames 0 --cpu-frames 0 > /dev/null 2>&1 || exit 1
rm -rf /tmp/prof_w && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_w -o run --output-format csv -- python3 $R/bench.py --conv $CONV --no-fp32-mode --steps 2 --warmup 1 --no-roofline --parity-frames 0 --cpu-frames 0 > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_traffic.py /tmp/prof_f/run_counter_collection.csv /tmp/prof_w/run_counter_collection.csv $O/conv_traffic.json "bench.py --conv $CONV --steps 2 --warmup 1, 4096 crops / 2048 hand-frames per step" 3 $CONV || exit 1
tail -3 $O/bench_conv_layers.txt
cat $O/bench_line.json
