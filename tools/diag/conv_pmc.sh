#!/bin/bash
# Diagnostic (not part of the product): SQ counters of one conv_igemm layer shape.
#   gpurun -- bash tools/diag/conv_pmc.sh <cin> <cout> <hw> <n_img> [name=src.hip[:flags]]
# Runs tools/diag/conv_ab.py under rocprofv3 --pmc in separate passes (8 SQ slots per pass) and prints the
# per-dispatch averages of the conv kernel's counters.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$1_$2_$3
mkdir -p $O
SPEC=${5:-base=absolutetrack_amd/csrc/conv_igemm.hip}
cd /tmp && export TMPDIR=/tmp
[ -f $R/gpurun_out/counters.txt ] || rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$i -o run --output-format csv -- python3 $R/tools/diag/conv_ab.py $1 $2 $3 $4 $SPEC > $O/pass_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pass_$i.log; continue; }
  python3 - /tmp/pmc_$i/run_counter_collection.csv /tmp/pmc_$i/run_kernel_trace.csv <<'PY'
import csv, sys, collections
tr = [r for r in csv.DictReader(open(sys.argv[2])) if "conv" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
if d:
    print(f"kernel duration under this pass: n={len(d)} avg={sum(d)/len(d):.1f} us min={min(d):.1f} us")
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "conv" not in k:
        continue
    agg[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for n, v in c.items():
        print(f"   {n:28s} n={len(v):4d} avg={sum(v)/len(v):.4g}")
PY
done 2>&1 | tee $O/summary.txt
