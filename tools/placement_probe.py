#!/usr/bin/env python3
"""Does the headline kernel's time depend on WHERE its buffers were allocated?  One process, the same 2^20 blocks, the buffers
allocated several times in different ways / orders; the kernel timed on each set.  (bench.py's value varies 370 - 450 M blocks/s
between processes on one box while the memory probes do not.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import __graft_entry__ as ge

pkg = ge.load_package()
dsp = pkg.SvtHipDsp(0)
dev = torch.device("cuda:0")
n = 1 << 20
qrow = {k: v[100].copy() for k, v in pkg.tables.quant_tables(8).items()}
iscan = torch.from_numpy(pkg.tables.scan_tables(pkg.TX_32X32, pkg.DCT_DCT)[1]).to(dev)
g = torch.Generator(device=dev); g.manual_seed(13596)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def make(tag, pad_elems=0, separate=False, hold=None):
    src = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev, generator=g)
    pred = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev, generator=g)
    if separate:
        o = [torch.empty((n, 1024), dtype=torch.int32, device=dev) for _ in range(3)]
    else:
        big = torch.empty(3 * n * 1024 + 2 * pad_elems, dtype=torch.int32, device=dev)
        o1, o2 = n * 1024 + pad_elems, 2 * (n * 1024 + pad_elems)
        o = [big[:n * 1024].view(n, 1024), big[o1:o1 + n * 1024].view(n, 1024), big[o2:o2 + n * 1024].view(n, 1024)]
    outs = (o[0], o[1], o[2], torch.zeros(n, dtype=torch.int16, device=dev), torch.zeros(n, dtype=torch.int32, device=dev))
    ms = timeit(lambda: dsp.fwd_quant_sad(src, pred, pkg.TX_32X32, pkg.DCT_DCT, qrow, iscan, outs=outs))
    row = {"case": tag, "ms": round(ms, 4), "Mblocks_per_s": round(n / ms / 1e3, 1), "ptrs_mod_1GiB_MiB": [round((t.data_ptr() % (1 << 30)) / 2 ** 20, 2) for t in (src, pred, o[0], o[1], o[2])]}
    print(json.dumps(row), flush=True)
    return src, pred, outs


keep = []
make("first")
make("second (same sizes again: the allocator reuses the freed blocks)")
keep.append(torch.empty(3 << 30, dtype=torch.uint8, device=dev))            # shift everything that follows by 3 GiB
make("after a 3 GiB spacer")
keep.append(torch.empty((1 << 30) + (37 << 20), dtype=torch.uint8, device=dev))
make("after another 1 GiB + 37 MiB spacer")
make("separate output tensors", separate=True)
torch.cuda.empty_cache()
make("after empty_cache()")
keep.clear(); torch.cuda.empty_cache()
make("after freeing the spacers + empty_cache()")
