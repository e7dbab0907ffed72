#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_frame.sh <tag>  -> gpurun_out/<tag>/frame_kernel_stats.csv (rocprofv3 --kernel-trace --stats of
# tools/bench_frame.py), gpurun_out/frame_c4.json (timings)
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 tools/bench_frame.py > $out/frame.log 2>&1 || { echo "bench_frame failed"; tail -5 $out/frame.log; exit 1; }
tail -1 $out/frame.log
tail -1 $out/frame.log > $out/frame_c4.json          # the plain run's numbers (the profiled run below rewrites gpurun_out/frame_c4.json)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/fstats -o fstats --output-format csv -- python3 tools/bench_frame.py > $out/fstats.log 2>&1 || echo "stats pass failed"
f=$(find $out/fstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/frame_kernel_stats.csv && cut -c1-160 $out/frame_kernel_stats.csv | head -30
