#!/bin/bash
# The two PMC passes of tools/refresh_profiles.sh alone (HBM traffic of the conv kernels per launch and of one whole
# hot-path step over all kernels): gpurun -- bash tools/refresh_traffic.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profiles_new
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_f && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-roofline --parity-frames 0 --cpu-frames 0 > /dev/null 2>&1 || exit 1
rm -rf /tmp/prof_w && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_w -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-roofline --parity-frames 0 --cpu-frames 0 > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_traffic.py /tmp/prof_f/run_counter_collection.csv /tmp/prof_w/run_counter_collection.csv $O/conv_traffic.json "bench.py --steps 2 --warmup 1, 4096 crops / 2048 hand-frames per step" 3 || exit 1
