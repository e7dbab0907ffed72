#!/usr/bin/env python3
"""Placement and the kernels with ONE large output: inverse 32x32 (4 GiB of coefficients read, 1 GiB of samples updated) and the
residual kernel (two 1 GiB inputs, one 2 GiB output): arrays back to back in one pool against one of them 32 GiB away."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); dsp = pkg.SvtHipDsp(0); dev = torch.device("cuda:0")
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
G = 1 << 30
pool = torch.empty(44 * G, dtype=torch.uint8, device=dev)
def view(off_gib, shape, dt):
    es = torch.tensor([], dtype=dt).element_size(); numel = 1
    for d in shape: numel *= d
    return pool[off_gib * G: off_gib * G + numel * es].view(dt).view(shape)
n = 1 << 20
g = torch.Generator(device=dev); g.manual_seed(3)
for tag, (oc, od) in (("back to back", (0, 4)), ("32 GiB apart", (0, 36))):
    c = view(oc, (n, 1024), torch.int32); c.copy_(torch.randint(-500, 501, (n, 1024), dtype=torch.int32, device=dev, generator=g))
    dst = view(od, (n, 32, 32), torch.uint8); dst.fill_(128)
    ms = timeit(lambda: dsp.inv_txfm2d_add(c, dst, 3, 0, 8))
    print(json.dumps({"kernel": "inv_txfm2d_add 32x32 u8", "arrays": tag, "ms": round(ms, 4), "frac_hbm": round(6144 * n / ms / 1e6 / 8000, 3)}), flush=True)
for tag, offs in (("back to back", (0, 1, 2)), ("output 32 GiB away", (0, 1, 36)), ("all three apart", (0, 16, 36))):
    a = view(offs[0], (n, 32, 32), torch.uint8); b = view(offs[1], (n, 32, 32), torch.uint8); r = view(offs[2], (n, 32, 32), torch.int16)
    a.fill_(7); b.fill_(3)
    try:
        ms = timeit(lambda: dsp.residual(a, b, out=r))
    except TypeError:
        print(json.dumps({"kernel": "residual", "note": "mirror takes no preallocated output"})); break
    print(json.dumps({"kernel": "residual 32x32", "arrays": tag, "ms": round(ms, 4), "frac_hbm": round(4096 * n / ms / 1e6 / 8000, 3)}), flush=True)
