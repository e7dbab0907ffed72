#!/bin/bash
# usage: tools/pmc.sh <prof_one name> <tag>   -> gpurun_out/pmc_<tag>/*.csv summary lines
name=$1; tag=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  t=$(echo $pass | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass -d gpurun_out/pmc_${tag}_$t -o pmc --output-format csv -- python3 tools/prof_one.py $name > gpurun_out/pmc_${tag}_$t.log 2>&1 || echo "pass $t failed"
done
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(list); dur = []
for f in glob.glob(f"gpurun_out/pmc_{tag}_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "svtdev" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items()):
    print(k[0], k[1], sum(v) / len(v))
print("avg_ns", sum(dur) / max(len(dur), 1))
PY
