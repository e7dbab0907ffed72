#!/bin/bash
# usage: tools/pmc3.sh <prof_one name> <tag> "<counters pass 1>" ["<pass 2>" ...]
name=$1; tag=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass -d gpurun_out/pmc3_${tag}_$i -o pmc --output-format csv -- python3 tools/prof_one.py $name > gpurun_out/pmc3_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(list); dur = []
for f in glob.glob(f"gpurun_out/pmc3_{tag}_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "svtdev" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items()):
    print(tag, k[0], k[1], sum(v) / len(v))
print(tag, "avg_ns", sum(dur) / max(len(dur), 1))
PY
