#!/usr/bin/env python3
"""A/B of two builds of the library on the same box, interleaved: tools/ab/lib_a.so against tools/ab/lib_b.so (tools/make_ab_lib.sh).
usage: ab_kernels.py <workload> [<workload> ...]      workloads: enc8 enc16 enc32 enc64 inv8 inv16 inv32 fq8 fq16 (dense 8-bit batches),
       c4 (one 1080p frame, five sizes, svt_hip_encode_recon_frame), ois8 ois16 (open-loop intra search of a 1080p picture), bip, me85 me209
Prints per workload the minimum and median of 6 interleaved rounds per library and whether the two libraries' outputs are equal."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import __graft_entry__ as ge

pkg = ge.load_package()
dev = torch.device("cuda:0")
TW = pkg.TX_W; TH = pkg.TX_H


def load(tag):
    d = pkg.SvtHipDsp.__new__(pkg.SvtHipDsp)
    d.torch = torch; d.lib = pkg.load_library(os.path.join(ROOT, "tools", "ab", f"lib_{tag}.so")); d.device = dev
    assert d.lib.svt_hip_init(0) == 0
    return d


def timeit(fn, iters=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def digest(o):
    if isinstance(o, (tuple, list)):
        return [digest(x) for x in o]
    if isinstance(o, torch.Tensor):
        return int((o.to(torch.int64) & 0xffffffff).sum())
    return None


def workload(name, d):
    g = torch.Generator(device=dev); g.manual_seed(13596)
    qt = pkg.tables.quant_tables(8); qrow = {k: v[100].copy() for k, v in qt.items()}
    if name[:3] in ("enc", "inv") or name[:2] == "fq":
        S = int(name[3:] if name[:2] != "fq" else name[2:]); s_ = {4: 0, 8: 1, 16: 2, 32: 3, 64: 4}[S]
        n = (1 << 20) * 1024 // (S * S) if S <= 32 else 1 << 18
        nc = min(S, 32) ** 2
        src = torch.randint(0, 256, (n, S, S), dtype=torch.uint8, device=dev, generator=g)
        pred = (src.to(torch.int16) + torch.randint(-20, 21, (n, S, S), dtype=torch.int16, device=dev, generator=g)).clamp(0, 255).to(torch.uint8)
        iscan = torch.from_numpy(pkg.tables.scan_tables(s_, 0)[1]).to(dev)
        if name.startswith("enc"):
            return (lambda: d.encode_recon(src, pred, s_, 0, qrow, iscan, keep_coeff=False)), n, 2 * S * S + 4 * nc + S * S + 6
        if name.startswith("fq"):
            outs = (torch.empty((n, nc), dtype=torch.int32, device=dev), torch.empty((n, nc), dtype=torch.int32, device=dev),
                    torch.empty((n, nc), dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int16, device=dev), torch.zeros(n, dtype=torch.int32, device=dev))
            return (lambda: d.fwd_quant_sad(src, pred, s_, 0, qrow, iscan, outs=outs)), n, 2 * S * S + 12 * nc + 6
        c = torch.randint(-500, 501, (n, nc), dtype=torch.int32, device=dev, generator=g)
        dst0 = src.clone()

        def inv():
            dst = dst0.clone() if False else dst0
            return d.inv_txfm2d_add(c, dst, s_, 0, 8)
        return inv, n, 4 * nc + 2 * S * S
    if name == "c4":
        from cidana_svt_av1_amd import frames
        W, H = 1920, 1080
        src, pred = {}, {}
        for nm, (ph, pw) in (("Y", (H, W)), ("U", (H // 2, W // 2)), ("V", (H // 2, W // 2))):
            src[nm] = torch.randint(0, 256, (ph, pw), dtype=torch.uint8, device=dev, generator=g)
            pred[nm] = (src[nm].to(torch.int16) + torch.randint(-20, 21, (ph, pw), dtype=torch.int16, device=dev, generator=g)).clamp(0, 255).to(torch.uint8)
        fp = frames.FramePass(d, pkg, src, pred)
        return (lambda: fp.run(qrow)), fp.pixels, 7, (lambda: fp.digest())
    if name == "oisall":        # the four block sizes of a 1080p picture in one svt_hip_ois_search_frame call
        W, H, pad = 1920, 1080, 64
        plane = torch.randint(0, 256, (H + 2 * pad, W + 2 * pad), dtype=torch.uint8, device=dev, generator=g); pic = plane[pad:, pad:]
        groups, nb = [], 0
        for bsize in (8, 16, 32, 64):
            blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
            xy = torch.from_numpy(np.array([(y << 16) | x for x, y in blocks], np.uint32).view(np.int32)).to(dev)
            modes, deltas = d.ois_candidates(bsize)
            groups.append((xy, bsize, modes, deltas)); nb += len(blocks)
        return (lambda: d.ois_search_frame(pic, W + 2 * pad, W, H, groups)), nb, 0
    if name.startswith("ois"):
        bsize = int(name[3:]); W, H, pad = 1920, 1080, 64
        plane = torch.randint(0, 256, (H + 2 * pad, W + 2 * pad), dtype=torch.uint8, device=dev, generator=g); pic = plane[pad:, pad:]
        blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
        xy = torch.from_numpy(np.array([(y << 16) | x for x, y in blocks], np.uint32).view(np.int32)).to(dev)
        modes, deltas = d.ois_candidates(bsize)
        return (lambda: d.ois_search(pic, W + 2 * pad, W, H, xy, bsize, modes, deltas)), len(blocks), bsize * bsize + 4 * len(modes)
    if name.startswith("intra:"):        # intra:<mode> - dense 32x32 8-bit prediction batch, SVT_INTRA_* mode number (z1 / z2 / z3: 10 / 11 / 12)
        parts = name.split(":")          # intra:<mode>[:<neighbour row pitch>]
        mode = int(parts[1]); n = 1 << 21
        pitch = int(parts[2]) if len(parts) > 2 else 16 + 2 * 64 + 16
        ab_ = torch.randint(0, 256, (n, pitch), dtype=torch.uint8, device=dev, generator=g); lf = torch.randint(0, 256, (n, pitch), dtype=torch.uint8, device=dev, generator=g)
        out = torch.empty((n, 32, 32), dtype=torch.uint8, device=dev)
        return (lambda: d.intra_pred(ab_, lf, mode, 32, 32, 8, 0, 0, 64, 64, out=out)), n, 1024 + 130
    if name == "bip":
        n = 1 << 20
        top = torch.randint(0, 256, (n, 48), dtype=torch.uint8, device=dev, generator=g); left = torch.randint(0, 256, (n, 48), dtype=torch.uint8, device=dev, generator=g)
        blk = torch.zeros((n, 8), dtype=torch.uint8, device=dev)
        blk[:, 0] = torch.arange(n, device=dev) % 13
        blk[:, 1] = ((torch.arange(n, device=dev) // 13) % 7 - 3).to(torch.int8).view(torch.uint8) * ((blk[:, 0] >= 1) & (blk[:, 0] <= 8)).to(torch.uint8)
        blk[:, 4] = 16; blk[:, 5] = 16; blk[:, 6] = 16; blk[:, 7] = 16
        out = torch.empty((n, 16, 16), dtype=torch.uint8, device=dev)
        if hasattr(d.lib, "svt_hip_intra_order_blocks_batch"):          # this round: order the batch by predictor kind first (both timed)
            def run():
                order = d.intra_order_blocks(blk, 2)
                return d.build_intra_predictors(top, left, blk, 2, dst=out, dst_stride=16, order=order)
            return run, n, 330
        return (lambda: d.build_intra_predictors(top, left, blk, 2, dst=out, dst_stride=16)), n, 330
    if name == "bip_plain":
        n = 1 << 20
        top = torch.randint(0, 256, (n, 48), dtype=torch.uint8, device=dev, generator=g); left = torch.randint(0, 256, (n, 48), dtype=torch.uint8, device=dev, generator=g)
        blk = torch.zeros((n, 8), dtype=torch.uint8, device=dev)
        blk[:, 0] = torch.arange(n, device=dev) % 13
        blk[:, 1] = ((torch.arange(n, device=dev) // 13) % 7 - 3).to(torch.int8).view(torch.uint8) * ((blk[:, 0] >= 1) & (blk[:, 0] <= 8)).to(torch.uint8)
        blk[:, 4] = 16; blk[:, 5] = 16; blk[:, 6] = 16; blk[:, 7] = 16
        out = torch.empty((n, 16, 16), dtype=torch.uint8, device=dev)
        return (lambda: d.build_intra_predictors(top, left, blk, 2, dst=out, dst_stride=16)), n, 330
    if name in ("me85", "me209"):
        n = 2040
        src = torch.randint(0, 256, (n, 64, 64), dtype=torch.uint8, device=dev, generator=g); ref = torch.randint(0, 256, (n, 127, 128), dtype=torch.uint8, device=dev, generator=g)
        return (lambda: d.me_fullpel_search(src, ref, 64, 64, nsq=(name == "me209"))), n, 64 * 64 + 127 * 127
    raise SystemExit(f"unknown workload {name}")


def main():
    libs = {t: load(t) for t in ("a", "b")}
    for name in sys.argv[1:]:
        fns, outs = {}, {}
        for t in ("a", "b"):
            w = workload(name, libs[t])
            fns[t], units, bpu = w[:3]
            r = fns[t]()
            outs[t] = digest(w[3]() if len(w) > 3 else r)
        times = {"a": [], "b": []}
        for rnd in range(6):
            for t in ("a", "b"):
                times[t].append(timeit(fns[t]))
        row = {"workload": name, "units": units, "equal_outputs": outs["a"] == outs["b"]}
        for t in ("a", "b"):
            v = sorted(times[t])
            med = (v[2] + v[3]) / 2
            row[t] = {"ms_min": round(v[0], 4), "ms_med": round(med, 4), "frac_hbm_at_med": round(bpu * units / med / 1e6 / 8000, 4)}
        row["b_over_a_speed"] = round(row["a"]["ms_med"] / row["b"]["ms_med"], 4)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
