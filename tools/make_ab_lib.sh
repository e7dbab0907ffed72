#!/bin/bash
# usage: tools/make_ab_lib.sh <git ref | WORK> <a|b>
# Builds libsvt_hip_dsp.so of a commit (in a scratch worktree under /tmp) or of the working tree (WORK) into tools/ab/lib_<tag>.so
# for tools/ab_kernels.py: two builds of the library timed interleaved on ONE box - the only comparison that means anything
# here, boxes differ by up to 25 %.  tools/ab/ is git-ignored and travels with the gpurun snapshot.
set -e
ref=$1; tag=$2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/tools/ab"
if [ "$ref" = "WORK" ]; then
  python3 "$root/cidana-svt-av1_amd/build.py" > /dev/null
  cp "$root/cidana-svt-av1_amd/libsvt_hip_dsp.so" "$root/tools/ab/lib_$tag.so"
else
  wt=/tmp/svt_ab_wt_$tag
  rm -rf "$wt"; git -C "$root" worktree prune
  git -C "$root" worktree add --detach "$wt" "$ref" > /dev/null
  python3 "$wt/cidana-svt-av1_amd/build.py" > /dev/null
  cp "$wt/cidana-svt-av1_amd/libsvt_hip_dsp.so" "$root/tools/ab/lib_$tag.so"
  git -C "$root" worktree remove --force "$wt"
fi
echo "tools/ab/lib_$tag.so <- $ref"
