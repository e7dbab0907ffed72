#!/usr/bin/env python3
"""The hot path end to end on a y4m file (every row of SURVEY 8(f) working on the others' outputs), per frame:
   picture input    file -> pinned -> HBM -> padded planes + 1/4 and 1/16 luma pictures   (frames.PictureInput, n4)
   HME 0 / 1 / 2    every 64x64 SB against the previous picture's pyramid, vectors stay on the device   (svt_hip_hme_level_batch, n1)
   ME set-up        best HME region, CheckZeroZeroCenter, per-SB search area clipped as MotionEstimateLcu does   (svt_hip_me_setup_batch, n1)
   full-pel ME      209 PUs per SB over its own (up to 64x64) area, one launch   (svt_hip_me_fullpel_search_areas_batch, a11)
   open-loop intra  every 8x8 ... 64x64 block, the reference's candidate lists   (svt_hip_ois_search_frame, n2)
   encode pass      residual -> FwdTxfm2d -> quant / dequant -> InvTxfm2d -> recon, five CU sizes, luma + chroma   (svt_hip_encode_recon_frame, n3)
The prediction of the encode pass is the previous source picture at zero motion (inter prediction itself is outside SURVEY 8).
Prints frames/s of the whole chain (file reading and PCIe included) and a digest of every stage's outputs; with no file argument a
synthetic moving-texture 1080p clip is written first.  usage: tools/pipeline_y4m.py [in.y4m] [--frames N]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import __graft_entry__ as ge

pkg = ge.load_package()
from cidana_svt_av1_amd import frames


def synthetic_clip(path, w, h, nf, seed=3, pan=(3, 1), bd=8):
    """a textured picture panning by `pan` samples per frame: the motion search has something to find (bd 8, or 10 as 16-bit samples)"""
    rng = np.random.default_rng(seed)
    dx, dy = pan
    big = rng.integers(0, 256, (h + dy * nf + 64, w + dx * nf + 64)).astype(np.float32)
    k = np.ones(5, np.float32) / 5
    big = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, big)
    big = np.apply_along_axis(lambda c: np.convolve(c, k, "same"), 0, big)
    top = (1 << bd) - 1
    big = ((big - big.min()) / (big.max() - big.min()) * top).astype(np.uint8 if bd == 8 else np.dtype("<u2"))
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F30:1 Ip {'C420jpeg' if bd == 8 else 'C420p10'}\n".encode())
        for i in range(nf):
            y = big[dy * i:dy * i + h, dx * i:dx * i + w]
            f.write(b"FRAME\n")
            f.write(np.ascontiguousarray(y).tobytes())
            f.write(np.ascontiguousarray(y[::2, ::2]).tobytes())
            f.write(np.ascontiguousarray(top - y[::2, ::2]).tobytes())


class Pipeline:
    def __init__(self, dsp, path, qindex=100, use_graph=False):
        self.dsp = dsp
        self.use_graph, self.graph, self.graph_out, self.count = use_graph, None, None, 0
        self.pi = frames.PictureInput(dsp, pkg, path, origin=(68, 68))
        pi = self.pi
        # a 10-bit clip: the analysis stages read the 8-bit plane (the samples' top 8 bits, as in the reference), the encode pass the
        # 16-bit samples at bd 10
        dev = pi.planes[0].device
        self.W, self.H, self.pad = pi.W, pi.H, pi.ox
        W, H = self.W, self.H
        sbs = [(x, y) for y in range(0, H, 64) for x in range(0, W, 64)]
        self.nsb = len(sbs)
        self.sb_xy = torch.tensor(sbs, dtype=torch.int32, device=dev)
        # HME: per level the SB origins / sizes in that level's picture, and the level parameters (EbMotionEstimation.c:5689-6016)
        # the search area of level 0 is split into 2 x 2 regions (EbMotionEstimation.c: number_hme_search_region_in_width / _height);
        # each region's vector is refined by levels 1 and 2 and the best level-2 SAD wins, as MotionEstimateLcu does
        self.hme = {}
        hw = {0: (np.array([32, 32], np.uint16), np.array([16, 16], np.uint16)), 1: (np.array([16, 16], np.uint16), np.array([8, 8], np.uint16)),
              2: (np.array([16, 16], np.uint16), np.array([8, 8], np.uint16))}
        self.hme_geo = []
        for level in (0, 1, 2):
            sh = 2 - level
            lw, lh, lpad = (W + (1 << sh) - 1) >> sh, (H + (1 << sh) - 1) >> sh, self.pad >> sh
            org = torch.tensor([(x >> sh, y >> sh) for x, y in sbs], dtype=torch.int16, device=dev)
            size = torch.tensor([(min(64, W - x) >> sh, min(64, H - y) >> sh) for x, y in sbs], dtype=torch.int16, device=dev)
            self.hme_geo.append((org, size, lpad))
            w_, h_ = hw[level]
            for rw in (0, 1):
                for rh in (0, 1):
                    self.hme[(level, rw, rh)] = dsp.hme_level_params(level, w_, h_, rw, rh, int(w_.sum()), int(h_.sum()), 100, 100, lpad, lpad, lw, lh)
        # full-pel ME: plane form, byte offsets of each SB's source block and of its search window's origin
        self.stride = pi.planes[0].stride(0)
        self.src_off = ((self.sb_xy[:, 1] + self.pad) * self.stride + self.sb_xy[:, 0] + self.pad).to(torch.int32)
        self.SW = self.SH = 64
        self.me_setup_params = dsp.MeSetupParams(W, H, W, H, self.SW, self.SH, 2, 2, 0, 1)      # 2 x 2 HME regions, CheckZeroZeroCenter on
        # open-loop intra search groups
        self.ois_groups = []
        for bsize in (8, 16, 32, 64):
            blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
            xy = torch.from_numpy(np.array([(y << 16) | x for x, y in blocks], np.uint32).view(np.int32)).to(dev)
            modes, deltas = dsp.ois_candidates(bsize)
            self.ois_groups.append((xy, bsize, modes, deltas))
        qt = pkg.tables.quant_tables(10 if pi.is16 else 8)
        self.qrow = {k: v[qindex].copy() for k, v in qt.items()}
        self.prev = None
        self.fp = None

    def interior(self, planes):
        p = self.pad
        return {"Y": planes[0][p:p + self.H, p:p + self.W], "U": planes[1][p // 2:p // 2 + self.H // 2, p // 2:p // 2 + self.W // 2],
                "V": planes[2][p // 2:p // 2 + self.H // 2, p // 2:p // 2 + self.W // 2]}

    def analyse(self, planes):
        """every stage after the picture import, on the current stream; reads self.prev, returns the stages' output tensors"""
        dsp, pi, t = self.dsp, self.pi, torch
        out = {}
        y = pi.luma8 if pi.is16 else planes[0]             # the plane HME / ME / the intra search read
        pic = y[self.pad:, self.pad:]
        # open-loop intra search on the source picture
        out["ois"] = dsp.ois_search_frame(pic, y.stride(0), self.W, self.H, self.ois_groups)
        if self.prev is not None:
            pyr_cur = {0: pi.sixteenth, 1: pi.quarter, 2: y}
            # the four regions of a level in one launch, three launches per picture
            centres = None
            for level in (0, 1, 2):
                org, size, lpad = self.hme_geo[level]
                cur, ref = pyr_cur[level], self.prev["pyr"][level]
                b4, centres = dsp.hme_level_regions(cur[lpad:, lpad:], cur.stride(0), ref[lpad:, lpad:], ref.stride(0), org, size, centres,
                                                    1 if level == 1 else 0, [self.hme[(level, rw, rh)] for rh in (0, 1) for rw in (0, 1)])   # region r = rh * 2 + rw
            # MotionEstimateLcu's glue on the device (svt_hip_me_setup_batch): best region, CheckZeroZeroCenter against the previous
            # picture, the search area clipped against the picture per SB - then ONE search launch whose SBs each read their own area
            # (interior 64x64 areas and the clipped ones of the edge SBs alike), nothing returns to the host in between
            org, size, _ = self.hme_geo[2]
            cur00, ref00 = y[self.pad:, self.pad:], self.prev["pyr"][2][self.pad:, self.pad:]
            centre, area = dsp.me_setup(cur00, self.stride, ref00, self.stride, org, size, b4, centres, self.me_setup_params)
            out["hme_sad"], out["hme_regions_mv"], out["hme_mv"], out["me_area"] = b4, centres, centre, area
            out["_cur_luma"] = y                                                  # (a view, for tests; not part of the digest)
            out["me_sad"], out["me_mv"] = dsp.me_fullpel_search_areas(y, self.stride, self.src_off, self.prev["pyr"][2], self.stride, self.src_off, area,
                                                                      self.SW, self.SH, nsq=True)
            # encode pass: source against the previous picture at zero motion (views into the padded buffers, no copies)
            if self.fp is None:
                self.fp = frames.FramePass(dsp, pkg, self.interior(planes), self.interior(self.prev["planes"]), is_16bit=pi.is16)
            self.fp.run(self.qrow)
            out["enc"] = self.fp                               # outputs stay in the pass's buffers; digest_of() sums them on request
            # the buffers of this picture become the reference of the next (PictureInput overwrites its planes on the next call)
            for d, s in zip(self.prev["planes"], planes):
                d.copy_(s)
            self.prev["pyr"][0].copy_(pi.sixteenth); self.prev["pyr"][1].copy_(pi.quarter)
            if pi.is16:
                self.prev["pyr"][2].copy_(pi.luma8)
        return out

    def step(self):
        """the next picture through the chain; None at the end of the file.  With use_graph the analysis of the third picture is
        captured into a HIP graph (every buffer it touches is static by then) and replayed for the rest of the clip: one graph
        launch per picture instead of ~30 kernel launches and a dozen torch operations issued from Python."""
        planes = self.pi.next()
        if planes is None:
            return None
        self.count += 1
        if self.prev is None:                              # first picture: nothing to search against yet
            out = self.analyse(planes)
            self.prev = {"planes": tuple(p.clone() for p in planes), "pyr": {0: self.pi.sixteenth.clone(), 1: self.pi.quarter.clone(), 2: None}}
            self.prev["pyr"][2] = self.pi.luma8.clone() if self.pi.is16 else self.prev["planes"][0]
            return out
        if not self.use_graph or self.count == 2:
            return self.analyse(planes)
        if self.graph is None:
            cur = torch.cuda.current_stream()
            st = torch.cuda.Stream()
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    self.graph_out = self.analyse(planes)
            cur.wait_stream(st)
            self.graph = g
        self.graph.replay()
        return self.graph_out


def digest_of(out):
    d = {}
    for k, v in out.items():
        if k.startswith("_"):
            continue
        if k == "ois":
            d["ois_best_sum"] = int(sum(int(b.to(torch.int64).sum()) for _, b in v))
            d["ois_dist_sum"] = int(sum(int(dd.to(torch.int64).sum()) for dd, _ in v))
        elif k == "enc":
            d["enc_digest"] = [int(x) for x in v.digest().cpu().tolist()]
        else:
            d[k + "_sum"] = int((v.to(torch.int64) & 0xffffffff).sum())
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file", nargs="?")
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--bd", type=int, default=8, choices=(8, 10), help="bit depth of the synthetic clip")
    ap.add_argument("--graph", action="store_true", help="capture the per-picture analysis into a HIP graph and replay it")
    a = ap.parse_args()
    dsp = pkg.SvtHipDsp(0)
    tmp = None
    path = a.file
    if path is None:
        w, h = (int(v) for v in a.size.split("x"))
        tmp = tempfile.mkdtemp(prefix="svt_pipe_")
        path = os.path.join(tmp, "clip.y4m")
        synthetic_clip(path, w, h, a.frames, bd=a.bd)
    results = []
    for rep in range(2):                                  # the second pass is timed (page cache, allocator, first-launch costs settled)
        p = Pipeline(dsp, path, use_graph=a.graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n, last = 0, None
        while True:
            o = p.step()
            if o is None:
                break
            last, n = o, n + 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        results.append((n, dt, digest_of(last)))
        del p.graph
        p.pi.close()
    n, dt, dg = results[-1]
    assert results[0][2] == dg, "the two passes disagree"
    print(json.dumps({"file": os.path.basename(path), "picture": f"{p.W}x{p.H}", "frames": n, "seconds": round(dt, 4), "frames_per_s": round(n / dt, 1),
                      "ms_per_frame": round(1e3 * dt / n, 3), "hip_graph": bool(a.graph), "stages": "input + decimation, OIS (4 sizes), HME 0/1/2, ME set-up + 209 PUs per-SB areas, encode pass (5 sizes, YUV)",
                      "last_frame_digest": dg, "device": dsp.device_name()}), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"picture": f"{p.W}x{p.H}", "bd": a.bd, "frames": n, "seconds": dt, "frames_per_s": n / dt, "ms_per_frame": 1e3 * dt / n, "hip_graph": bool(a.graph),
               "stages": "input + decimation, OIS (4 sizes), HME 0/1/2 (2 x 2 regions), ME set-up (svt_hip_me_setup_batch) + 209 PUs per-SB areas, encode pass (5 sizes, YUV)",
               "digest": dg, "device": dsp.device_name()}, open(os.path.join(ROOT, "gpurun_out", "pipeline.json"), "w"), indent=1)
    if tmp:
        os.remove(path); os.rmdir(tmp)


if __name__ == "__main__":
    main()
