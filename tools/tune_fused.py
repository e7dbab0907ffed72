#!/usr/bin/env python3
"""Sweep the tuning knobs of the fused 32x32 kernel on the GPU box (svt_hip_tune)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
dsp = pkg.SvtHipDsp(0)
dev = torch.device("cuda:0")
n = 1 << 20
g = torch.Generator(device=dev); g.manual_seed(13596)
src = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev, generator=g)
pred = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev, generator=g)
qt = pkg.tables.quant_tables(8); qrow = {k: v[100].copy() for k, v in qt.items()}
_, isc = pkg.tables.scan_tables(3, 0); iscan = torch.from_numpy(isc).to(dev)
outs = (torch.empty((n, 1024), dtype=torch.int32, device=dev), torch.empty((n, 1024), dtype=torch.int32, device=dev),
        torch.empty((n, 1024), dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int16, device=dev),
        torch.zeros(n, dtype=torch.int32, device=dev))
def run(iters=10):
    for _ in range(2): dsp.fwd_quant_sad(src, pred, 3, 0, qrow, iscan, outs=outs)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): dsp.fwd_quant_sad(src, pred, 3, 0, qrow, iscan, outs=outs)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
variants = [(nt, mw, wg) for nt in (0, 1) for mw in (1,) for wg in (0, 2, 3, 4, 6)]
times = {v: [] for v in variants}
for rnd in range(8):
    for v in variants:
        nt, mw, wg = v
        dsp.lib.svt_hip_tune(b"f32_nt", nt); dsp.lib.svt_hip_tune(b"f32_wg_per_cu", wg)      # (the 128-register variant, min_waves 4, spilled to scratch and was removed)
        times[v].append(run(8))
res = []
for v in variants:
    t = sorted(times[v])
    r = {"nt": v[0], "min_waves": v[1], "wg_per_cu": v[2], "ms_min": t[0], "ms_med": (t[3] + t[4]) / 2, "ms_max": t[-1],
         "frac_at_min": 14342 * n / t[0] / 1e6 / 8000}
    res.append(r); print(r, flush=True)
print("BEST", json.dumps(min(res, key=lambda r: r["ms_med"])))
