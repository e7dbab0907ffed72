#!/usr/bin/env python3
"""Probe variants of the inverse 32x32 kernel on the GPU box (svt_hip_tune inv32_waves / inv32_var).
var bit0 = no destination prefetch, bit1 = no transforms (memory-only), bit2 = no global loads (compute-only)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
dsp = pkg.SvtHipDsp(0)
dev = torch.device("cuda:0")
n = 1 << 20
c = torch.randint(-500, 501, (n, 1024), dtype=torch.int32, device=dev)
d = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev)
def run(iters=8):
    for _ in range(2): dsp.inv_txfm2d_add(c, d, 3, 0, 8)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): dsp.inv_txfm2d_add(c, d, 3, 0, 8)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
variants = [tuple(int(x) for x in v.split(",")) for v in (sys.argv[1:] or ["4,0", "4,1", "4,2", "4,4", "4,5", "1,0", "1,1", "2,0", "2,1"])]
times = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        dsp.lib.svt_hip_tune(b"inv32_waves", v[0]); dsp.lib.svt_hip_tune(b"inv32_var", v[1])
        times[v].append(run())
for v in variants:
    t = sorted(times[v])
    print(json.dumps({"waves": v[0], "var": v[1], "ms_min": t[0], "ms_med": t[1], "frac_at_min": 6144 * n / t[0] / 1e6 / 8000}), flush=True)
