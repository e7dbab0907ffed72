#!/usr/bin/env python3
"""One 44 GiB allocation; the headline kernel's three output arrays placed at different BASE offsets inside it (regions) and with
different SKEWS between them.  Which of the two decides between the 350 and the 450 M blocks/s seen for one binary?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); dsp = pkg.SvtHipDsp(0); dev = torch.device("cuda:0")
n = 1 << 20
qrow = {k: v[100].copy() for k, v in pkg.tables.quant_tables(8).items()}
iscan = torch.from_numpy(pkg.tables.scan_tables(pkg.TX_32X32, pkg.DCT_DCT)[1]).to(dev)
g = torch.Generator(device=dev); g.manual_seed(13596)
src = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev, generator=g)
pred = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev, generator=g)
eob = torch.zeros(n, dtype=torch.int16, device=dev); sad = torch.zeros(n, dtype=torch.int32, device=dev)
G = 1 << 28                                  # int32 elements per GiB
pool = torch.empty(44 * G, dtype=torch.int32, device=dev)
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
def run(offs):
    outs = tuple(pool[o:o + n * 1024].view(n, 1024) for o in offs) + (eob, sad)
    ms = timeit(lambda: dsp.fwd_quant_sad(src, pred, pkg.TX_32X32, pkg.DCT_DCT, qrow, iscan, outs=outs))
    return round(n / ms / 1e3, 1)
print(json.dumps({"pool_ptr_GiB": round(pool.data_ptr() / 2 ** 30, 3)}), flush=True)
for base in range(0, 32, 4):                 # three arrays back to back, 4 GiB each, window moved through the pool
    print(json.dumps({"base_GiB": base, "Mblocks_per_s": run([base * G, (base + 4) * G, (base + 8) * G])}), flush=True)
for skew_kib in (0, 4, 12, 68, 260, 1028, 4100, 65540, 1048580):       # same base, growing distance between the arrays
    sk = skew_kib * 256
    print(json.dumps({"base_GiB": 0, "skew_KiB": skew_kib, "Mblocks_per_s": run([0, 4 * G + sk, 8 * G + 2 * sk])}), flush=True)
# each array alone in a different region: coeff at a, qcoeff at b, dqcoeff at c
for offs in ((0, 16, 32), (0, 20, 40), (2, 17, 33), (1, 14, 27), (0, 8, 16), (0, 12, 24), (4, 16, 28), (8, 20, 36), (30, 34, 38), (32, 36, 40), (0, 4, 32), (0, 4, 36), (0, 32, 36), (28, 4, 8)):
    print(json.dumps({"arrays_at_GiB": offs, "Mblocks_per_s": run([o * G for o in offs])}), flush=True)
