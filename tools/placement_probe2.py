#!/usr/bin/env python3
"""Buffer sets of the headline kernel allocated one after the other and ALL KEPT: is it the first allocation of the process that is slow?"""
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
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
sets = []
for i in range(5):
    src = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev, generator=g)
    pred = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev, generator=g)
    big = torch.empty(3 * n * 1024, dtype=torch.int32, device=dev)
    outs = (big[:n * 1024].view(n, 1024), big[n * 1024:2 * n * 1024].view(n, 1024), big[2 * n * 1024:].view(n, 1024),
            torch.zeros(n, dtype=torch.int16, device=dev), torch.zeros(n, dtype=torch.int32, device=dev))
    sets.append((src, pred, outs, big))
def run(i):
    src, pred, outs, _ = sets[i]
    return timeit(lambda: dsp.fwd_quant_sad(src, pred, pkg.TX_32X32, pkg.DCT_DCT, qrow, iscan, outs=outs))
for rnd in range(2):
    for i in range(5):
        ms = run(i)
        print(json.dumps({"round": rnd, "set": i, "ms": round(ms, 4), "Mblocks_per_s": round(n / ms / 1e3, 1), "big_ptr_GiB": round(sets[i][3].data_ptr() / 2 ** 30, 3)}), flush=True)
# mixed: inputs of one set, outputs of another
for a, b in ((0, 1), (1, 0), (0, 0)):
    src, pred = sets[a][0], sets[a][1]; outs = sets[b][2]
    ms = timeit(lambda: dsp.fwd_quant_sad(src, pred, pkg.TX_32X32, pkg.DCT_DCT, qrow, iscan, outs=outs))
    print(json.dumps({"inputs_of_set": a, "outputs_of_set": b, "ms": round(ms, 4), "Mblocks_per_s": round(n / ms / 1e3, 1)}), flush=True)
