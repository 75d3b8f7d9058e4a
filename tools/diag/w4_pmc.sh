#!/bin/bash
# Diagnostic (not part of the product): SQ counters of conv_w4.hip and of the chunked kernel on one layer shape, three passes.
#   gpurun -- bash tools/diag/w4_pmc.sh <cin> <cout> <hw> <n_img>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/w4pmc_$1_$2_$3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/w4pmc_$i
  AB_ROUNDS=2 timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/w4pmc_$i -o run --output-format csv -- python3 $R/tools/diag/w4_ab.py $1 $2 $3 $4 > $O/pass_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pass_$i.log; continue; }
  python3 - /tmp/w4pmc_$i/run_counter_collection.csv /tmp/w4pmc_$i/run_kernel_trace.csv <<'PY'
import csv, sys, collections
for pat in ("conv_w4", "conv_split", "conv_c64k"):
    tr = [r for r in csv.DictReader(open(sys.argv[2])) if pat in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
    if d:
        print(f"{pat}: duration under this pass: n={len(d)} avg={sum(d)/len(d):.1f} us min={min(d):.1f} us")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, v in agg.items():
        print(f"   {n:28s} n={len(v):4d} avg={sum(v)/len(v):.5g}")
PY
done 2>&1 | tee $O/summary.txt
