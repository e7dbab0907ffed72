#!/usr/bin/env python3
"""BASELINE.json configs[3]: one 1080p yuv420p frame, every CU size: residual -> FwdTxfm2d ->
quant -> dequant -> InvTxfm2d + add on PLANES (xy-addressed blocks).  Reports ms/frame per size
and blocks/s; bytes: 15 B/px (src,pred u8 in; coeff,qcoeff,dqcoeff i32 + recon u8 out) — SURVEY §8d."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); dsp = pkg.SvtHipDsp(0)
dev = torch.device("cuda:0")
qt = pkg.tables.quant_tables(8); qrow = {k: v[100].copy() for k, v in qt.items()}
g = torch.Generator(device=dev); g.manual_seed(13596)
planes = {"Y": (1080, 1920), "U": (540, 960), "V": (540, 960)}
src = {k: torch.randint(0, 256, s, dtype=torch.uint8, device=dev, generator=g) for k, s in planes.items()}
pred = {k: torch.randint(0, 256, s, dtype=torch.uint8, device=dev, generator=g) for k, s in planes.items()}
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
rows = []
for tx_size in (4, 3, 2, 1, 0):
    S = pkg.TX_W[tx_size]
    work = []
    for name, (ph, pw) in planes.items():
        s_c = S if name == "Y" else S // 2
        if s_c < 4: continue
        ts = {64: 4, 32: 3, 16: 2, 8: 1, 4: 0}[s_c]
        xs = np.arange(0, pw - s_c + 1, s_c); ys = np.arange(0, ph - s_c + 1, s_c)
        xy = torch.from_numpy(np.array([(y << 16) | x for y in ys for x in xs], np.uint32).view(np.int32)).to(dev)
        offs = torch.from_numpy(np.array([y * pw + x for y in ys for x in xs], np.uint32).view(np.int32)).to(dev)
        _, isc = pkg.tables.scan_tables(ts, 0)
        work.append((name, ts, pw, xy, offs, torch.from_numpy(isc).to(dev), pred[name].clone()))
    nblk = sum(w[3].numel() for w in work)
    npx = sum(w[3].numel() * pkg.TX_W[w[1]] ** 2 for w in work)
    def frame():
        for (name, ts, pw, xy, offs, iscan, recon) in work:
            co, q, dq, eob, _, _ = dsp.fwd_quant_planes(src[name], pw, pred[name], pw, xy, ts, 0, qrow, iscan)
            dsp.inv_txfm2d_add(dq, recon, ts, 0, 8, dst_stride=pw, dst_block_pitch=0, offsets=offs)
    ms = timeit(frame)
    def frame_fused():                      # one fused launch per plane (svt_hip_encode_recon_planes_batch), recon in place
        for (name, ts, pw, xy, offs, iscan, recon) in work:
            if ts == 0:                     # 4x4 has no fused kernel
                co, q, dq, eob, _, _ = dsp.fwd_quant_planes(src[name], pw, pred[name], pw, xy, ts, 0, qrow, iscan)
                dsp.inv_txfm2d_add(dq, recon, ts, 0, 8, dst_stride=pw, dst_block_pitch=0, offsets=offs)
            else:
                dsp.encode_recon_planes(src[name], pw, pred[name], pw, recon, pw, xy, ts, 0, qrow, iscan)
    ms_fused = timeit(frame_fused)
    # the same launches captured once in a HIP graph and replayed (every entry point only enqueues work on
    # the caller's stream, so a frame pass is capturable as is)
    ms_graph = None
    try:
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            frame(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                frame()
        torch.cuda.synchronize()
        ms_graph = timeit(gr.replay)
    except Exception as e:       # report, do not hide
        print("graph capture failed:", repr(e), flush=True)
    r = {"luma_size": S, "blocks": nblk, "pixels": npx, "ms_per_frame": round(ms, 4), "ms_per_frame_fused": round(ms_fused, 4), "ms_per_frame_hipgraph": None if ms_graph is None else round(ms_graph, 4), "Mblocks_per_s": round(nblk / ms / 1e3, 1),
         "GBps_at_15B_per_px": round(15 * npx / ms / 1e6, 1), "frac_hbm_peak": round(15 * npx / ms / 1e6 / 8000, 4)}
    rows.append(r); print(json.dumps(r), flush=True)
print(json.dumps({"total_ms_all_sizes": round(sum(r["ms_per_frame"] for r in rows), 3), "total_ms_all_sizes_fused": round(sum(r["ms_per_frame_fused"] for r in rows), 3), "total_ms_all_sizes_hipgraph": round(sum((r["ms_per_frame_hipgraph"] or 0) for r in rows), 3)}))
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "frame_c4.json"), "w"), indent=1)
