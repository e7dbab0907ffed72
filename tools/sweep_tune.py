#!/usr/bin/env python3
"""Sweep one svt_hip_tune key over values on one workload of tools/ab_kernels.py (library: tools/ab/lib_b.so), interleaved rounds.
usage: sweep_tune.py <workload> <key> <v1> <v2> ..."""
import json
import sys

import ab_kernels as ab


def main():
    name, key, vals = sys.argv[1], sys.argv[2].encode(), [int(v) for v in sys.argv[3:]]
    d = ab.load("b")
    fn = ab.workload(name, d)[0]
    times = {v: [] for v in vals}
    for _ in range(5):
        for v in vals:
            assert d.lib.svt_hip_tune(key, v) == 0
            times[v].append(ab.timeit(fn, iters=10))
    for v in vals:
        t = sorted(times[v])
        print(json.dumps({"workload": name, "key": key.decode(), "value": v, "ms_min": round(t[0], 4), "ms_med": round(t[2], 4)}), flush=True)


if __name__ == "__main__":
    main()
