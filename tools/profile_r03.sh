#!/bin/bash
# usage (GPU box, repo root): bash tools/profile_r03.sh <tag>     e.g. r03_a
# One call collects what the round's DESIGN numbers cite: the headline bench line + rocprofv3 kernel stats + PMC traffic
# (tools/profile_round.sh), PMC per kernel for the encode / search kernels touched this round (tools/pmc_r02.sh), the per-kernel
# table, configs[3] frame modes and configs[4] at 240 frames.  Everything lands under gpurun_out/<tag>/.
tag=$1
out=gpurun_out/$tag; mkdir -p $out
bash tools/profile_round.sh $tag > $out/profile_round.log 2>&1
bash tools/pmc_r02.sh $tag enc8 enc16 me_sb bip ois8 > $out/pmc_kernels.log 2>&1
timeout -k 10 300 python3 tools/bench_frame.py > $out/bench_frame.log 2>&1 && cp gpurun_out/frame_c4.json $out/frame_c4.json
timeout -k 10 300 python3 tools/bench_c5.py --frames 240 --json-out $out/c5_240frames_1gpu.json > $out/c5_240.log 2>&1
timeout -k 10 300 python3 tools/bench_c5.py --frames 240 --stack 30 --json-out $out/c5_240frames_1gpu_stack30.json > $out/c5_240_stack30.log 2>&1
timeout -k 10 500 python3 tools/bench_kernels.py > $out/bench_kernels.log 2>&1 && cp gpurun_out/kernels.json $out/kernels.json
ls $out | head -50
