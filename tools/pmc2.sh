#!/bin/bash
# usage: tools/pmc2.sh <prof_one name> <tag>  — SQ activity counters (separate passes), printed per kernel
name=$1; tag=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS" "SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass -d gpurun_out/pmc2_${tag}_$i -o pmc --output-format csv -- python3 tools/prof_one.py $name > gpurun_out/pmc2_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(list); dur = []
for f in glob.glob(f"gpurun_out/pmc2_{tag}_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "svtdev" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items()):
    print(tag, k[0], k[1], sum(v) / len(v))
print(tag, "avg_ns", sum(dur) / max(len(dur), 1))
PY
