#!/usr/bin/env python3
"""Does the placement rule of the headline kernel (output arrays 32 GiB apart) matter for the other multi-output kernels?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); dsp = pkg.SvtHipDsp(0); dev = torch.device("cuda:0")
qrow = {k: v[100].copy() for k, v in pkg.tables.quant_tables(8).items()}
g = torch.Generator(device=dev); g.manual_seed(1)
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
    nbytes = int(torch.tensor([], dtype=dt).element_size())
    numel = 1
    for d in shape: numel *= d
    return pool[off_gib * G: off_gib * G + numel * nbytes].view(dt).view(shape)
for S, ts in ((32, 3), (16, 2), (8, 1)):
    n = (1 << 20) * 1024 // (S * S)
    nc = S * S
    src = torch.randint(0, 256, (n, S, S), dtype=torch.uint8, device=dev, generator=g)
    pred = (src.to(torch.int16) + torch.randint(-20, 21, (n, S, S), dtype=torch.int16, device=dev, generator=g)).clamp(0, 255).to(torch.uint8)
    iscan = torch.from_numpy(pkg.tables.scan_tables(ts, 0)[1]).to(dev)
    for tag, offs in (("back to back", (0, 4, 8, 12)), ("32 GiB apart", (0, 36, 4, 40))):
        # forward + quantiser + SAD: coeff, qcoeff, dqcoeff (4 GiB each)
        outs = (view(offs[0], (n, nc), torch.int32), view(offs[1], (n, nc), torch.int32), view(offs[2], (n, nc), torch.int32),
                torch.zeros(n, dtype=torch.int16, device=dev), torch.zeros(n, dtype=torch.int32, device=dev))
        ms = timeit(lambda: dsp.fwd_quant_sad(src, pred, ts, 0, qrow, iscan, outs=outs))
        print(json.dumps({"kernel": f"fwd+quant+sad {S}x{S}", "outputs": tag, "ms": round(ms, 4), "frac_hbm": round((2 * S * S + 12 * nc + 6) * n / ms / 1e6 / 8000, 3)}), flush=True)
    del src, pred
# encode pass 32x32: qcoeff (4 GiB) + recon (1 GiB)
n = 1 << 20
src = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev, generator=g)
pred = (src.to(torch.int16) + torch.randint(-20, 21, (n, 32, 32), dtype=torch.int16, device=dev, generator=g)).clamp(0, 255).to(torch.uint8)
iscan = torch.from_numpy(pkg.tables.scan_tables(3, 0)[1]).to(dev)
L = dsp.lib
import ctypes
tabs = [pkg._np16(qrow[k]) for k in ("zbin", "round", "quant", "quant_shift", "dequant")] if hasattr(pkg, "_np16") else None
for tag, offs in (("back to back", (0, 4)), ("32 GiB apart", (0, 36))):
    q = view(offs[0], (n, 1024), torch.int32); rec = view(offs[1], (n, 32, 32), torch.uint8)
    eob = torch.zeros(n, dtype=torch.int16, device=dev)
    try:
        ms = timeit(lambda: dsp.encode_recon(src, pred, 3, 0, qrow, iscan, keep_coeff=False, want_sad=False, outs=(q, eob, rec)))
    except TypeError:
        print(json.dumps({"kernel": "encode_recon 32x32", "note": "the mirror's encode_recon takes no preallocated outputs"})); break
    print(json.dumps({"kernel": "encode_recon 32x32 (qcoeff + recon)", "outputs": tag, "ms": round(ms, 4), "frac_hbm": round(7174 * n / ms / 1e6 / 8000, 3)}), flush=True)
