#!/usr/bin/env python3
"""MotionEstimateLcu's per-SB glue on one 1080p picture, two reference pictures: set-up (best HME region,
CheckZeroZeroCenter, search area), the full-pel search with per-SB areas (85 / 209 PUs) and the bi-prediction + me_results rows.
HME outputs are synthetic (random vectors inside the reference's range).  One JSON line; also gpurun_out/me_glue.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import __graft_entry__ as ge

pkg = ge.load_package()
dsp = pkg.SvtHipDsp(0)
dev = torch.device("cuda:0")
W, H, PAD = 1920, 1080, 128
g = torch.Generator(device=dev); g.manual_seed(7)
planes = [torch.randint(0, 256, (H + 2 * PAD, W + 2 * PAD), dtype=torch.uint8, device=dev, generator=g) for _ in range(3)]      # source, ref 0, ref 1
stride = W + 2 * PAD
at = lambda p: p[PAD:, PAD:]
sbx, sby = (W + 63) // 64, (H + 63) // 64
orig = np.array([(x * 64, y * 64) for y in range(sby) for x in range(sbx)], np.int16)
size = np.array([(min(64, W - x), min(64, H - y)) for x, y in orig], np.int16)
n = orig.shape[0]
d_orig, d_size = torch.from_numpy(orig).to(dev), torch.from_numpy(size).to(dev)
hme_sad = torch.randint(1000, 200000, (4, n), dtype=torch.int64, device=dev, generator=g)
hme_mv = torch.randint(-60, 61, (4, n, 2), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
prm = pkg.SvtHipDsp.MeSetupParams(W, H, W, H, 64, 64, 2, 2, 0, 1)
offs = torch.from_numpy(((orig[:, 1].astype(np.int64) + PAD) * stride + orig[:, 0] + PAD).astype(np.uint32).view(np.int32)).to(dev)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


rows = {}
areas = []
for r in (1, 2):
    rows[f"setup_ref{r - 1}_ms"] = round(timeit(lambda: dsp.me_setup(at(planes[0]), stride, at(planes[r]), stride, d_orig, d_size, hme_sad, hme_mv, prm)), 4)
    areas.append(dsp.me_setup(at(planes[0]), stride, at(planes[r]), stride, d_orig, d_size, hme_sad, hme_mv, prm)[1])
a = areas[0].cpu().numpy()
rows["areas_w_h_histogram"] = {f"{w}x{h}": int(c) for (w, h), c in zip(*np.unique(a[:, 2:], axis=0, return_counts=True))}
res = {}
for nsq in (False, True):
    outs = []
    for r in (1, 2):
        bs = torch.full((n, 209 if nsq else 85), dsp.MAX_SAD_VALUE, dtype=torch.int32, device=dev); bm = torch.zeros_like(bs)

        def run(r=r, bs=bs, bm=bm):
            bs.fill_(dsp.MAX_SAD_VALUE)
            return dsp.me_fullpel_search_areas(planes[0], stride, offs, planes[r], stride, offs, areas[r - 1], 64, 64, nsq=nsq, best_sad=bs, best_mv=bm)
        rows[f"search_{'209' if nsq else '85'}pus_ref{r - 1}_ms"] = round(timeit(run), 4)
        outs.append(run())
    res[nsq] = outs
for nsq in (False, True):
    (s0, m0), (s1, m1) = res[nsq]
    npus = 209 if nsq else 85
    rows[f"bipred_results_{npus}pus_ms"] = round(timeit(lambda: dsp.me_bipred(at(planes[0]), stride, at(planes[1]), stride, at(planes[2]), stride, d_orig, s0, m0, s1, m1, npus=npus)), 4)
rows.update({"picture": "1920x1080, 510 SBs, 2 reference pictures, 64x64 nominal search area, 2 x 2 HME regions", "device": dsp.device_name()})
print(json.dumps(rows), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "me_glue.json"), "w"), indent=1)
