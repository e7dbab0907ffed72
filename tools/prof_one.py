#!/usr/bin/env python3
"""Run ONE kernel a few times (for rocprofv3 --pmc / --kernel-trace).  usage: prof_one.py <name>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); dsp = pkg.SvtHipDsp(0)
dev = torch.device("cuda:0")
name = sys.argv[1]
qt = pkg.tables.quant_tables(8); qrow = {k: v[100].copy() for k, v in qt.items()}
if name == "sad_search":
    n = 1 << 20
    src = torch.randint(0, 256, (n, 16, 16), dtype=torch.uint8, device=dev); ref = torch.randint(0, 256, (n, 23, 23), dtype=torch.uint8, device=dev)
    fn = lambda: dsp.sad_search(src, ref, 8, 8)
elif name == "inv32":
    n = 1 << 20
    c = torch.randint(-500, 501, (n, 1024), dtype=torch.int32, device=dev); d = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev)
    if os.environ.get("INV32_VAR"):
        wv, vr = (int(t) for t in os.environ["INV32_VAR"].split(","))
        dsp.lib.svt_hip_tune(b"inv32_waves", wv); dsp.lib.svt_hip_tune(b"inv32_var", vr)
    fn = lambda: dsp.inv_txfm2d_add(c, d, 3, 0, 8)
elif name == "intra_dc":
    n = 1 << 21
    ab = torch.randint(0, 256, (n, 160), dtype=torch.uint8, device=dev); lf = torch.randint(0, 256, (n, 160), dtype=torch.uint8, device=dev)
    out = torch.empty((n, 32, 32), dtype=torch.uint8, device=dev)
    fn = lambda: dsp.intra_pred(ab, lf, 0, 32, 32, 8, out=out)
elif name.startswith("intra:"):
    mode = int(name.split(":")[1]); n = 1 << 21
    ab = torch.randint(0, 256, (n, 160), dtype=torch.uint8, device=dev); lf = torch.randint(0, 256, (n, 160), dtype=torch.uint8, device=dev)
    out = torch.empty((n, 32, 32), dtype=torch.uint8, device=dev)
    fn = lambda: dsp.intra_pred(ab, lf, mode, 32, 32, 8, 0, 0, 64, 64, out=out)
elif name == "me_sb":
    n = 510
    src = torch.randint(0, 256, (n, 64, 64), dtype=torch.uint8, device=dev); ref = torch.randint(0, 256, (n, 127, 128), dtype=torch.uint8, device=dev)
    fn = lambda: dsp.me_sb_search(src, ref, 64, 64)
elif name == "oisall":                # the four block sizes of a 1080p picture in one svt_hip_ois_search_frame call
    import numpy as np
    W, H, pad = 1920, 1080, 64
    plane = torch.randint(0, 256, (H + 2 * pad, W + 2 * pad), dtype=torch.uint8, device=dev); pic = plane[pad:, pad:]
    groups = []
    for bsize in (8, 16, 32, 64):
        blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
        xy = torch.from_numpy(np.array([(y << 16) | x for x, y in blocks], np.uint32).view(np.int32)).to(dev)
        modes, deltas = dsp.ois_candidates(bsize)
        groups.append((xy, bsize, modes, deltas))
    fn = lambda: dsp.ois_search_frame(pic, W + 2 * pad, W, H, groups)
elif name.startswith("ois"):          # ois8 / ois16: the open-loop intra search of one 1080p picture at that block size
    import numpy as np
    bsize = int(name[3:]); W, H, pad = 1920, 1080, 64
    plane = torch.randint(0, 256, (H + 2 * pad, W + 2 * pad), dtype=torch.uint8, device=dev); pic = plane[pad:, pad:]
    blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
    xy = torch.from_numpy(np.array([(y << 16) | x for x, y in blocks], np.uint32).view(np.int32)).to(dev)
    modes, deltas = dsp.ois_candidates(bsize)
    fn = lambda: dsp.ois_search(pic, W + 2 * pad, W, H, xy, bsize, modes, deltas)
elif name.startswith("enc"):          # enc32 / enc64 / enc16 / enc8: fused encode_recon (qcoeff + recon), as tools/bench_kernels.py
    S = int(name[3:]); s_ = {8: 1, 16: 2, 32: 3, 64: 4}[S]
    n = (1 << 20) * 1024 // (S * S) if S <= 32 else 1 << 18
    src = torch.randint(0, 256, (n, S, S), dtype=torch.uint8, device=dev)
    pred = (src.to(torch.int16) + torch.randint(-20, 21, (n, S, S), dtype=torch.int16, device=dev)).clamp(0, 255).to(torch.uint8)
    iscan = torch.from_numpy(pkg.tables.scan_tables(s_, 0)[1]).to(dev)
    fn = lambda: dsp.encode_recon(src, pred, s_, 0, qrow, iscan, keep_coeff=False)
elif name == "bip":                  # build_intra_predictors glue, 16x16, mixed modes (as tools/bench_kernels.py)
    n = 1 << 20
    top = torch.randint(0, 256, (n, 48), dtype=torch.uint8, device=dev); left = torch.randint(0, 256, (n, 48), dtype=torch.uint8, device=dev)
    blk = torch.zeros((n, 8), dtype=torch.uint8, device=dev)
    blk[:, 0] = torch.arange(n, device=dev) % 13
    blk[:, 1] = ((torch.arange(n, device=dev) // 13) % 7 - 3).to(torch.int8).view(torch.uint8) * ((blk[:, 0] >= 1) & (blk[:, 0] <= 8)).to(torch.uint8)
    blk[:, 4] = 16; blk[:, 5] = 16; blk[:, 6] = 16; blk[:, 7] = 16
    out = torch.empty((n, 16, 16), dtype=torch.uint8, device=dev)
    def fn():                         # this round: the batch is ordered by predictor kind on the device first (both launches are in the trace)
        order = dsp.intra_order_blocks(blk, 2)
        return dsp.build_intra_predictors(top, left, blk, 2, dst=out, dst_stride=16, order=order)
else:
    raise SystemExit("unknown " + name)
for _ in range(5): fn()
torch.cuda.synchronize()
