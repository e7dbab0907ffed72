#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_r02.sh <out tag> <prof_one name> [<prof_one name> ...]
# Per kernel: one rocprofv3 --kernel-trace --stats pass and separate --pmc passes (SQ activity, LDS, HBM bytes); the summary
# goes to gpurun_out/<tag>/<name>.json (copy into profiles/).
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for name in "$@"; do
  i=0
  for pass in "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass -d $out/${name}_$i -o pmc --output-format csv -- python3 tools/prof_one.py $name > $out/${name}_$i.log 2>&1 || echo "$name pass $i failed"
  done
  python3 - "$out" "$name" <<'PY'
import csv, glob, json, sys, collections
out, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/{name}_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "svtdev" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0][:90]
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {}
for (k, c), v in sorted(agg.items()):
    res.setdefault(k, {})[c] = sum(v) / len(v)
for k in res:
    res[k]["avg_ns_under_pmc"] = sum(dur[k]) / len(dur[k])
    res[k]["launches"] = len(dur[k])
json.dump({"prof_one": name, "command": "rocprofv3 --kernel-trace --pmc <group> -- python3 tools/prof_one.py " + name + " (one pass per counter group)", "kernels": res},
          open(f"{out}/{name}.json", "w"), indent=1)
print(json.dumps(res))
PY
done
