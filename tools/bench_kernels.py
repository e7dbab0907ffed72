#!/usr/bin/env python3
"""Per-kernel throughput on the GPU box: algorithmic bytes / measured time vs the 8 TB/s
HBM peak (SURVEY §8d figures).  Not the headline bench (bench.py) — a profiling aid whose
table goes to profiles/rNN_kernels.json."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); dsp = pkg.SvtHipDsp(0)
dev = torch.device("cuda:0")
TW, TH = pkg.TX_W, pkg.TX_H
def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
def timeit_graph(fn, iters=40):
    """the same call captured once into a HIP graph and replayed: GPU time without the host's per-launch cost (a call of a few
    launches of ~10 us each is host-bound from Python); None when the call cannot be captured"""
    try:
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                fn()
        torch.cuda.synchronize()
        return timeit(g.replay, iters=iters)
    except Exception as e:                                   # noqa: BLE001
        print(json.dumps({"graph_capture_failed": str(e)[:200]}), flush=True)
        torch.cuda.synchronize()
        return None
rows = []
def rec(name, n, bytes_per, ms, extra=None):
    r = {"kernel": name, "units": n, "bytes_per_unit": bytes_per, "ms": round(ms, 4), "Munits_per_s": round(n / ms / 1e3, 2),
         "GBps": round(bytes_per * n / ms / 1e6, 1), "frac_hbm_peak": round(bytes_per * n / ms / 1e6 / 8000, 4)}
    if extra: r.update(extra)
    rows.append(r); print(json.dumps(r), flush=True)
qt = pkg.tables.quant_tables(8); qrow = {k: v[100].copy() for k, v in qt.items()}
only = sys.argv[1:] 
def want(k): return not only or any(o in k for o in only)
# forward transforms
for s, n in ((0, 1 << 23), (1, 1 << 22), (2, 1 << 21), (3, 1 << 20), (4, 1 << 18), (9, 1 << 20), (16, 1 << 21)):
    if not want("fwd_txfm"): break
    w, h = TW[s], TH[s]
    x = torch.randint(-255, 256, (n, h, w), dtype=torch.int16, device=dev)
    out = torch.empty((n, w * h), dtype=torch.int32, device=dev)
    ms = timeit(lambda: dsp.fwd_txfm2d(x, s, 0, 8, out=out))
    rec(f"fwd_txfm2d_{w}x{h}", n, 6 * w * h, ms)
    del x, out
# quantize
if want("quantize"):
    for s, ls, n in ((3, 1, 1 << 20), (2, 0, 1 << 22)):
        w, h = TW[s], TH[s]
        c = torch.randint(-2000, 2001, (n, w * h), dtype=torch.int32, device=dev)
        _, isc = pkg.tables.scan_tables(s, 0); iscan = torch.from_numpy(isc).to(dev)
        ms = timeit(lambda: dsp.quantize_b(c, qrow, iscan, ls))
        rec(f"quantize_b_{w}x{h}", n, 12 * w * h + 2, ms)
        del c
# generic fused chain
for s, n in ((1, 1 << 22), (2, 1 << 21), (4, 1 << 18), (0, 1 << 23)):
    if not want("fused_generic"): break
    w, h = TW[s], TH[s]
    src = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev); pred = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
    _, isc = pkg.tables.scan_tables(s, 0); iscan = torch.from_numpy(isc).to(dev)
    nc = min(w, 32) * min(h, 32)
    outs = (torch.empty((n, nc), dtype=torch.int32, device=dev), torch.empty((n, nc), dtype=torch.int32, device=dev),
            torch.empty((n, nc), dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int16, device=dev), torch.zeros(n, dtype=torch.int32, device=dev))
    ms = timeit(lambda: dsp.fwd_quant_sad(src, pred, s, 0, qrow, iscan, outs=outs))
    rec(f"fused_generic_{w}x{h}", n, 2 * w * h + 12 * nc + 6, ms)
    del src, pred, outs
# inverse
for s, n in ((3, 1 << 20), (1, 1 << 22), (2, 1 << 21), (4, 1 << 18)):
    if not want("inv_txfm"): break
    w, h = TW[s], TH[s]
    nc = min(w, 32) * min(h, 32)
    c = torch.randint(-500, 501, (n, nc), dtype=torch.int32, device=dev)
    d = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
    ms = timeit(lambda: dsp.inv_txfm2d_add(c, d, s, 0, 8))
    rec(f"inv_txfm2d_add_u8_{w}x{h}", n, 4 * nc + 2 * w * h, ms)
    del c, d
# SAD search C3
if want("sad_search"):
    n = 1 << 20
    src = torch.randint(0, 256, (n, 16, 16), dtype=torch.uint8, device=dev); ref = torch.randint(0, 256, (n, 23, 23), dtype=torch.uint8, device=dev)
    ms = timeit(lambda: dsp.sad_search(src, ref, 8, 8))
    rec("sad_search_16x16_64cand(C3)", n, 797, ms, {"abs_diff_per_s_T": round(n * 64 * 256 / ms / 1e9, 2)})
    del src, ref
    # HME level 1 / 2 shapes: 32x32 and 64x64 blocks, 16x16 refinement area
    for bs, n in ((32, 1 << 17), (64, 1 << 15)):
        src = torch.randint(0, 256, (n, bs, bs), dtype=torch.uint8, device=dev); ref = torch.randint(0, 256, (n, bs + 15, bs + 15), dtype=torch.uint8, device=dev)
        ms = timeit(lambda: dsp.sad_search(src, ref, 16, 16))
        rec(f"sad_search_{bs}x{bs}_256cand", n, bs * bs + (bs + 15) ** 2 + 12, ms, {"abs_diff_per_s_T": round(n * 256 * bs * bs / ms / 1e9, 2)})
        del src, ref
# plain sad / sse / residual 32x32
if want("pixel"):
    n = 1 << 21
    a = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev); b = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev)
    rec("sad_32x32", n, 2052, timeit(lambda: dsp.sad(a, b)))
    rec("sse_32x32", n, 2056, timeit(lambda: dsp.sse(a, b)))
    rec("residual_32x32", n, 4096, timeit(lambda: dsp.residual(a, b)))
    del a, b
# 10-bit (BASELINE configs[4]): fused chain and inverse on dense 16-bit blocks
if want("bd10"):
    qt10 = pkg.tables.quant_tables(10); qrow10 = {k: v[100].copy() for k, v in qt10.items()}
    for s, n in ((3, 1 << 20), (2, 1 << 21)):
        w, h = TW[s], TH[s]
        src = torch.randint(0, 1024, (n, h, w), dtype=torch.int16, device=dev); pred = torch.randint(0, 1024, (n, h, w), dtype=torch.int16, device=dev)
        _, isc = pkg.tables.scan_tables(s, 0); iscan = torch.from_numpy(isc).to(dev)
        # the batch as ONE plane 16 384 samples wide (origins are 16-bit x | y << 16: a single column of blocks would wrap y and
        # measure a cache-resident handful of blocks - the first version of this row did, and read 1.1 of the HBM peak)
        PWD = 16384
        bi = np.arange(n, dtype=np.uint32)
        assert int(bi[-1] // (PWD // w)) * h + h <= 65536
        xy = torch.from_numpy((((bi // (PWD // w)) * h << 16) | ((bi % (PWD // w)) * w)).astype(np.uint32).view(np.int32)).to(dev)
        ms = timeit(lambda: dsp.fwd_quant_planes(src.view(-1, PWD), PWD, pred.view(-1, PWD), PWD, xy, s, 0, qrow10, iscan, bd=10), iters=4)
        rec(f"fused_generic_bd10_{w}x{h}", n, 4 * w * h + 12 * w * h + 2, ms)
        co = torch.randint(-2000, 2001, (n, w * h), dtype=torch.int32, device=dev)
        d = torch.randint(0, 1024, (n, h, w), dtype=torch.int16, device=dev)
        ms = timeit(lambda: dsp.inv_txfm2d_add(co, d, s, 0, 10))
        rec(f"inv_txfm2d_add_u16_bd10_{w}x{h}", n, 4 * w * h + 4 * w * h, ms)
        del src, pred, co, d
# BASELINE configs[4] shape: a shard of a 10-bit GOP, 16 luma frames of 1080p stacked in one plane, blocks addressed by
# origin tables; forward+quant on planes, then inverse + reconstruction in place (15 B/px at 8-bit, 18 B/px at 10-bit)
if want("gop"):
    qt10 = pkg.tables.quant_tables(10); qrow10 = {k: v[100].copy() for k, v in qt10.items()}
    FR, PH, PW = 16, 1080, 1920
    srcp = torch.randint(0, 1024, (FR * PH, PW), dtype=torch.int16, device=dev)
    predp = (srcp + torch.randint(-12, 13, (FR * PH, PW), dtype=torch.int16, device=dev)).clamp(0, 1023)
    for s in (3, 2, 1):
        S = TW[s]
        xs = np.arange(0, PW - S + 1, S); ys = np.concatenate([f * PH + np.arange(0, PH - S + 1, S) for f in range(FR)])
        xy = torch.from_numpy(np.array([(y << 16) | x for y in ys for x in xs], np.uint32).view(np.int32)).to(dev)
        offs = torch.from_numpy(np.array([y * PW + x for y in ys for x in xs], np.uint32).view(np.int32)).to(dev)
        _, isc = pkg.tables.scan_tables(s, 0); iscan = torch.from_numpy(isc).to(dev)
        recon = predp.clone()
        def gop():
            co, q, dq, eob, _, _ = dsp.fwd_quant_planes(srcp, PW, predp, PW, xy, s, 0, qrow10, iscan, bd=10)
            dsp.inv_txfm2d_add(dq, recon, s, 0, 10, dst_stride=PW, dst_block_pitch=0, offsets=offs)
        ms = timeit(gop, iters=4)
        n = xy.numel()
        rec(f"gop16_1080p_bd10_planes_fwd+quant+inv_{S}x{S}", n, 18 * S * S, ms, {"ms_per_frame": round(ms / FR, 4), "Mpx_per_s": round(n * S * S / ms / 1e3, 1)})
        recon2 = predp.clone()
        ms = timeit(lambda: dsp.encode_recon_planes(srcp, PW, predp, PW, recon2, PW, xy, s, 0, qrow10, iscan, bd=10), iters=4)
        rec(f"gop16_1080p_bd10_planes_encode_recon_fused_{S}x{S}", n, 10 * S * S, ms, {"ms_per_frame": round(ms / FR, 4), "Mpx_per_s": round(n * S * S / ms / 1e3, 1)})
        del recon2
        del recon, xy, offs
    del srcp, predp
# fused encode-pass chain (residual -> fwd -> quant/dequant -> inverse -> recon), 32x32
if want("encode_recon"):
    n = 1 << 20
    src = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device=dev)
    pred = (src.to(torch.int16) + torch.randint(-20, 21, (n, 32, 32), dtype=torch.int16, device=dev)).clamp(0, 255).to(torch.uint8)
    _, isc = pkg.tables.scan_tables(3, 0); iscan = torch.from_numpy(isc).to(dev)
    rec(f"encode_recon_32x32_fused(qcoeff+recon)", n, 2048 + 4096 + 1024 + 6, timeit(lambda: dsp.encode_recon(src, pred, 3, 0, qrow, iscan, keep_coeff=False)))
    rec(f"encode_recon_32x32_fused(+coeff,dqcoeff)", n, 2048 + 3 * 4096 + 1024 + 6, timeit(lambda: dsp.encode_recon(src, pred, 3, 0, qrow, iscan, keep_coeff=True)))
    del src, pred
    for s_, n in ((1, 1 << 22), (2, 1 << 21), (4, 1 << 18)):
        w, h = TW[s_], TH[s_]
        src = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
        pred = (src.to(torch.int16) + torch.randint(-20, 21, (n, h, w), dtype=torch.int16, device=dev)).clamp(0, 255).to(torch.uint8)
        _, isc = pkg.tables.scan_tables(s_, 0); iscan = torch.from_numpy(isc).to(dev)
        kc = min(w, 32) * min(h, 32)
        rec(f"encode_recon_{w}x{h}_fused(qcoeff+recon)", n, 2 * w * h + 4 * kc + w * h + 6, timeit(lambda: dsp.encode_recon(src, pred, s_, 0, qrow, iscan, keep_coeff=False)))
        del src, pred
# intra
if want("intra"):
    n = 1 << 21
    ab = torch.randint(0, 256, (n, 160), dtype=torch.uint8, device=dev); lf = torch.randint(0, 256, (n, 160), dtype=torch.uint8, device=dev)
    out = torch.empty((n, 32, 32), dtype=torch.uint8, device=dev)
    for mode, nm in ((0, "dc"), (9, "dc128"), (1, "v"), (2, "h"), (3, "smooth"), (6, "paeth"), (10, "z1"), (11, "z2"), (12, "z3")):
        ms = timeit(lambda: dsp.intra_pred(ab, lf, mode, 32, 32, 8, 0, 0, 64, 64, out=out))
        rec(f"intra_{nm}_32x32_u8", n, 1024 + 2 * 65, ms)
    del ab, lf, out
# ME 85-PU search, 1080p worth of SBs
if want("cfl"):
    # K11 + level map: algorithmic bytes per chroma block, compact Q3 layout (line = W) for the chain
    for (w, h, n) in ((16, 16, 1 << 21), (8, 8, 1 << 22), (32, 32, 1 << 19)):
        luma = torch.randint(0, 256, (n, 2 * h, 2 * w), dtype=torch.uint8, device=dev)
        q3 = torch.empty((n, h, w), dtype=torch.int16, device=dev)
        pr = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
        al = torch.randint(-16, 17, (n,), dtype=torch.int32, device=dev)
        def f_ac():
            dsp._check(dsp.lib.svt_hip_cfl_luma_subsampling_420_batch(dsp._p(luma), 2 * w, 4 * w * h, None, 0, dsp._p(q3), w, w * h,
                                                                      2 * w, 2 * h, 1, n, dsp._stream()), "cfl_ac")
        def f_pred():
            dsp._check(dsp.lib.svt_hip_cfl_predict_batch(dsp._p(q3), w, w * h, dsp._p(pr), w, dsp._p(pr), w, None, dsp._p(al), 8, w, h, 0,
                                                         n, dsp._stream()), "cfl_predict")
        rec(f"cfl_luma_ac_{w}x{h}_u8(subsample+subtract_average)", n, 6 * w * h, timeit(f_ac))
        rec(f"cfl_predict_{w}x{h}_u8(in place)", n, 4 * w * h + 4, timeit(f_pred))
        del luma, q3, pr, al
    for (w, h, n) in ((32, 32, 1 << 19), (16, 16, 1 << 21), (4, 4, 1 << 23)):
        co = torch.randint(-300, 301, (n, w * h), dtype=torch.int32, device=dev)
        size = (w + 4) * (h + 6) + 16
        lv = torch.empty((n, (size + 15) // 16 * 16), dtype=torch.uint8, device=dev)       # 16-byte-aligned block buffers
        rec(f"txb_init_levels_{w}x{h}", n, 4 * w * h + size, timeit(lambda: dsp.txb_init_levels(co, w, h, lv)))
        del co, lv
if want("ois"):
    # open-loop intra search of a whole 1080p luma picture: every block of each size, the reference's candidate lists
    W, H, pad = 1920, 1080, 64
    plane = torch.randint(0, 256, (H + 2 * pad, W + 2 * pad), dtype=torch.uint8, device=dev)
    pic = plane[pad:, pad:]
    tot = 0.0
    for bsize in (8, 16, 32, 64):
        blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
        xy = torch.from_numpy(np.array([(y << 16) | x for x, y in blocks], np.uint32).view(np.int32)).to(dev)
        modes, deltas = dsp.ois_candidates(bsize)
        call = lambda: dsp.ois_search(pic, W + 2 * pad, W, H, xy, bsize, modes, deltas)
        ms_call = timeit(call, iters=20)
        ms_g = timeit_graph(call)
        ms = min(ms_call, ms_g) if ms_g is not None else ms_call
        tot += ms
        rec(f"ois_search_1080p_{bsize}x{bsize}_{len(modes)}cand", len(blocks), bsize * bsize * (1 + 2 * len(modes)), ms,
            {"candidate_predictions_per_s_M": round(len(blocks) * len(modes) / ms / 1e3, 1), "ms_per_call_from_python": round(ms_call, 4),
             "ms_graph_replay": None if ms_g is None else round(ms_g, 4)})
    groups = []
    for bsize in (8, 16, 32, 64):
        blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
        xy = torch.from_numpy(np.array([(y << 16) | x for x, y in blocks], np.uint32).view(np.int32)).to(dev)
        modes, deltas = dsp.ois_candidates(bsize)
        groups.append((xy, bsize, modes, deltas))
    ms_frame = timeit(lambda: dsp.ois_search_frame(pic, W + 2 * pad, W, H, groups), iters=20)
    rows.append({"kernel": "ois_search_1080p_all_sizes_one_call", "ms": round(ms_frame, 4), "units": 1, "bytes_per_unit": 0, "Munits_per_s": 0, "GBps": 0, "frac_hbm_peak": 0})
    print(json.dumps({"ois_search_1080p_all_sizes_ms": round(tot, 3), "ois_search_1080p_all_sizes_one_call_ms": round(ms_frame, 4)}), flush=True)
if want("me_sb"):
    for n, label in ((510, "1 ref"), (2040, "4 refs")):      # 510 SBs of a 1080p frame x reference pictures
        src = torch.randint(0, 256, (n, 64, 64), dtype=torch.uint8, device=dev); ref = torch.randint(0, 256, (n, 127, 128), dtype=torch.uint8, device=dev)
        ms = timeit(lambda: dsp.me_sb_search(src, ref, 64, 64), iters=4)
        rec(f"me_sb_search_64x64area_{n}SBs({label})", n, 4096 + 127 * 127 + 680, ms, {"search_points_per_s_G": round(n * 4096 / ms / 1e6, 3)})
        del src, ref
if want("me_fullpel"):
    # K6 in the reference's own layout (p_sb_best_sad / p_sb_best_mv): the 85 square PUs (fast kernel) and all 209 PUs with the NSQ
    # shapes (exact kernel: the reference's search-point order and update rules, both flavours)
    n = 2040
    src = torch.randint(0, 256, (n, 64, 64), dtype=torch.uint8, device=dev); ref = torch.randint(0, 256, (n, 127, 128), dtype=torch.uint8, device=dev)
    for nsq in (False, True):
        ms = timeit(lambda: dsp.me_fullpel_search(src, ref, 64, 64, nsq=nsq), iters=3)
        rec(f"me_fullpel_search_64x64area_2040SBs({'209 PUs, NSQ' if nsq else '85 PUs'})", n, 4096 + 127 * 127 + (209 if nsq else 85) * 8, ms,
            {"search_points_per_s_G": round(n * 4096 / ms / 1e6, 3)})
    del src, ref
if want("bip"):
    # build_intra_predictors glue (a14): 16x16 blocks, per-block modes (all 13 x angle deltas) and availability, one launch
    n = 1 << 20
    top = torch.randint(0, 256, (n, 48), dtype=torch.uint8, device=dev); left = torch.randint(0, 256, (n, 48), dtype=torch.uint8, device=dev)
    blk = torch.zeros((n, 8), dtype=torch.uint8, device=dev)
    blk[:, 0] = torch.arange(n, device=dev) % 13
    blk[:, 1] = ((torch.arange(n, device=dev) // 13) % 7 - 3).to(torch.int8).view(torch.uint8) * ((blk[:, 0] >= 1) & (blk[:, 0] <= 8)).to(torch.uint8)
    blk[:, 4] = 16; blk[:, 5] = 16; blk[:, 6] = 16; blk[:, 7] = 16
    out = torch.empty((n, 16, 16), dtype=torch.uint8, device=dev)
    ms = timeit(lambda: dsp.build_intra_predictors(top, left, blk, 2, dst=out, dst_stride=16))
    rec("build_intra_predictors_16x16_u8(mixed modes)", n, 256 + 2 * 33 + 8, ms)

    def ordered():
        order = dsp.intra_order_blocks(blk, 2)
        return dsp.build_intra_predictors(top, left, blk, 2, dst=out, dst_stride=16, order=order)
    rec("build_intra_predictors_16x16_u8(mixed modes, ordered by kind on the device: ordering included)", n, 256 + 2 * 33 + 8, timeit(ordered))
    del top, left, blk, out
if want("hme"):
    # HME level 0 on a 1/16-resolution 4K picture pair (960x540, 16x16 SBs... the level's own SB size), every SB, one launch
    W, H, sb, pad = 960, 540, 16, 24
    stride = W + 2 * pad
    refb = torch.randint(0, 256, (H + 2 * pad, stride), dtype=torch.uint8, device=dev); srcp = torch.randint(0, 256, (H, W), dtype=torch.uint8, device=dev)
    org = np.array([(x, y) for y in range(0, H, sb) for x in range(0, W, sb)], np.int16)
    size = np.array([(min(sb, W - x), min(sb, H - y)) for x, y in org.tolist()], np.int16)
    hw = np.array([64, 64], np.uint16); hh = np.array([32, 32], np.uint16)
    prm = dsp.hme_level_params(0, hw, hh, 0, 0, 128, 64, 100, 100, pad, pad, W, H)
    d_org = torch.from_numpy(org).to(dev); d_size = torch.from_numpy(size).to(dev)
    ref00 = refb.view(-1)[pad * stride + pad:]
    ms = timeit(lambda: dsp.hme_level(srcp, W, ref00, stride, d_org, d_size, None, 0, prm), iters=4)
    rec("hme_level0_960x540_64x32area", org.shape[0], 16 * 8 + 79 * 47, ms, {"search_points_per_s_G": round(org.shape[0] * 64 * 32 / ms / 1e6, 3)})
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "kernels.json"), "w"), indent=1)
