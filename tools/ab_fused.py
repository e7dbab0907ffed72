#!/usr/bin/env python3
"""A/B of two builds of the library on the same box, interleaved: tools/ab/lib_a.so vs tools/ab/lib_b.so
(headline fused kernel and the fused encode-recon chain).  usage: ab_fused.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
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
dsps = {}
for tag in ("a", "b"):
    d = pkg.SvtHipDsp.__new__(pkg.SvtHipDsp)
    d.torch = torch; d.lib = pkg.load_library(os.path.join(ROOT, "tools", "ab", f"lib_{tag}.so")); d.lib.svt_hip_init(0); d.device = dev
    dsps[tag] = d
def run(d, iters=8):
    for _ in range(2): d.fwd_quant_sad(src, pred, 3, 0, qrow, iscan, outs=outs)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): d.fwd_quant_sad(src, pred, 3, 0, qrow, iscan, outs=outs)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
t = {"a": [], "b": []}
for rnd in range(6):
    for tag in ("a", "b"):
        t[tag].append(run(dsps[tag]))
for tag in ("a", "b"):
    v = sorted(t[tag]); print(json.dumps({"lib": tag, "ms_min": v[0], "ms_med": (v[2] + v[3]) / 2, "frac_at_med": 14342 * n / ((v[2] + v[3]) / 2) / 1e6 / 8000}))
