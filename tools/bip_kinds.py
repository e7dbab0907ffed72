#!/usr/bin/env python3
"""svt_hip_build_intra_predictors_batch, 2^20 16x16 8-bit blocks of ONE mode each: where the mixed batch's time goes.
usage: bip_kinds.py [lib tag under tools/ab, default: the built library]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

import ab_kernels as ab

pkg, dev = ab.pkg, ab.dev
d = ab.load(sys.argv[1]) if len(sys.argv) > 1 else pkg.SvtHipDsp(0)
n = 1 << 20
g = torch.Generator(device=dev); g.manual_seed(5)
top = torch.randint(0, 256, (n, 48), dtype=torch.uint8, device=dev, generator=g); left = torch.randint(0, 256, (n, 48), dtype=torch.uint8, device=dev, generator=g)
out = torch.empty((n, 16, 16), dtype=torch.uint8, device=dev)
cases = [("DC", 0, 0), ("V", 1, 0), ("H", 2, 0), ("D45 z1", 3, 0), ("D45-9 z1", 3, -3), ("D135 z2", 4, 0), ("D113+3 z2", 5, 1), ("D203 z3", 7, 0), ("D67 z1", 8, 0), ("V+3 z2", 1, 1), ("V-3 z1", 1, -1),
         ("SMOOTH", 9, 0), ("SMOOTH_V", 10, 0), ("SMOOTH_H", 11, 0), ("PAETH", 12, 0)]
for name, mode, delta in cases:
    blk = torch.zeros((n, 8), dtype=torch.uint8, device=dev)
    blk[:, 0] = mode
    blk[:, 1] = torch.tensor([delta], dtype=torch.int8).view(torch.uint8).item()
    blk[:, 4] = 16; blk[:, 5] = 16; blk[:, 6] = 16; blk[:, 7] = 16
    ms = ab.timeit(lambda: d.build_intra_predictors(top, left, blk, 2, dst=out, dst_stride=16), iters=6)
    print(json.dumps({"mode": name, "ms": round(ms, 4), "frac_hbm": round(330 * n / ms / 1e6 / 8000, 3)}), flush=True)
