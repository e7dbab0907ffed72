#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# Produces under gpurun_out/<tag>/: bench.json (plain run), kernel_stats.csv (rocprofv3 --kernel-trace --stats of
# the same bench.py command), pmc_*.csv + pmc.json (separate --pmc passes: FETCH_SIZE, WRITE_SIZE, SQ activity).
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py > $out/bench.log 2>&1 && tail -1 $out/bench.log > $out/bench.json || { echo "bench failed"; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-probes > $out/stats.log 2>&1 || echo "stats pass failed"
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass -d $out/pmc_$i -o pmc --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-probes > $out/pmc_$i.log 2>&1 || echo "pmc pass $i failed"
  f=$(find $out/pmc_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc_$(echo $pass | cut -d" " -f1).csv
done
python3 - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]; out = f"gpurun_out/{tag}"
agg = collections.defaultdict(list)
for f in glob.glob(f"{out}/pmc_*.csv"):
    for r in csv.DictReader(open(f)):
        if "fwd32_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in agg.items()}
blocks = 1 << 20
res = {"round": int(tag[1:3]) if tag[:1] == "r" and tag[1:3].isdigit() else None, "tag": tag, "kernel": "fwd32_kernel<true,true,true,1,false,2> (headline fused chain)", "blocks": blocks,
       "command": "rocprofv3 --kernel-trace --pmc <C> -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-probes (one pass per counter group)"}
if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    rd = 2 * mean["FETCH_SIZE"] * 1024; wr = mean["WRITE_SIZE"] * 1024
    res.update({"FETCH_SIZE_KB_per_launch": mean["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": mean["WRITE_SIZE"],
                "correction": "gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md HBM): read bytes = 2*FETCH_SIZE*1024; WRITE_SIZE exact",
                "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                "algorithmic_bytes_per_launch": 14342 * blocks, "traffic_over_algorithmic": (rd + wr) / (14342 * blocks)})
    commit = open("GIT_COMMIT").read().strip() if __import__("os").path.exists("GIT_COMMIT") else None
    json.dump({"blocks": blocks, "hbm_bytes_per_launch": rd + wr, "source": f"profiles/{tag}_pmc.json", "commit": commit}, open(f"{out}/traffic_latest.json", "w"))
for k, v in mean.items():
    if k not in ("FETCH_SIZE", "WRITE_SIZE"): res[k] = v
json.dump(res, open(f"{out}/pmc.json", "w"), indent=1)
print(json.dumps(res))
PY
head -5 $out/kernel_stats.csv
