"""GPU: svt_hip_hme_level_batch - hierarchical ME levels 0 / 1 / 2 with the per-SB search-area clipping on the device -
against the reference's own HmeLevel0 / HmeLevel1 / HmeLevel2 outputs (tests/golden/hme.npz) and, chained on the device
(level 0 -> 1 -> 2 without the centres leaving HBM), against the oracle on every SB of a synthetic picture pyramid."""
import ctypes
import os

import numpy as np
import pytest
import torch

import svtlibs
from svtlibs import ptr

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_hme_levels_golden_gpu(dsp):
    g = np.load(os.path.join(G, "hme.npz"))
    hme_w, hme_h = g["hme_w"], g["hme_h"]
    for level in range(3):
        W, H, sb, pad, stride = (int(v) for v in g[f"l{level}_dims"])
        d_src = dev(g[f"l{level}_src"]); d_ref = dev(g[f"l{level}_ref"])
        ref00 = d_ref.view(-1)[pad * stride + pad:]
        cases = g[f"l{level}_cases"]; exp = g[f"l{level}_out"]
        # group the cases by parameter set (region, multipliers): one launch per set, as a caller would
        keys = sorted({tuple(c[7:11]) for c in cases.tolist()})
        for (rw, rh, mx, my) in keys:
            idx = [i for i, c in enumerate(cases.tolist()) if tuple(c[7:11]) == (rw, rh, mx, my)]
            sel = cases[idx]
            p = dsp.hme_level_params(level, hme_w, hme_h, rw, rh, int(hme_w.sum()), int(hme_h.sum()), mx, my, pad, pad, W, H)
            po = svtlibs.hme_params(level, hme_w, hme_h, rw, rh, mx, my, pad, W, H)
            assert bytes(p) == bytes(po)                       # the product's parameter derivation == the oracle's
            best, mv = dsp.hme_level(d_src, W, ref00, stride, dev(sel[:, 3:5].astype(np.int16)), dev(sel[:, 1:3].astype(np.int16)),
                                     dev(sel[:, 5:7].astype(np.int16)), 0, p)
            got = np.concatenate([best.cpu().numpy()[:, None], mv.cpu().numpy().astype(np.int64)], axis=1)
            assert np.array_equal(got, exp[idx]), (level, rw, rh, mx, my, np.nonzero((got != exp[idx]).any(axis=1))[0][:5])


def test_hme_three_levels_chained_on_device_vs_oracle(dsp):
    """a 3-level pyramid of a 480x272 picture (partial SBs right and bottom), every SB, two search regions at level 0: the
    level-0 vectors feed level 1 (>> 1) and those feed level 2 entirely on the device; the oracle walks the same chain"""
    O = svtlibs.oracle()
    rng = np.random.default_rng(77)
    W2, H2 = 480, 272
    pics = {}
    for level, (W, H, pad) in {2: (W2, H2, 72), 1: (W2 // 2, H2 // 2, 40), 0: (W2 // 4, H2 // 4, 24)}.items():
        stride = W + 2 * pad + 3
        ref = rng.integers(0, 256, (H + 2 * pad, stride), dtype=np.uint8)
        src = rng.integers(0, 256, (H, W), dtype=np.uint8)
        if level == 0:
            src[:, : W // 2] = (src[:, : W // 2] >> 6) << 6; ref[:, : stride // 2] = (ref[:, : stride // 2] >> 6) << 6
        pics[level] = (src, ref, W, H, pad, stride)
    sbs = [(x, y, min(64, W2 - x), min(64, H2 - y)) for y in range(0, H2, 64) for x in range(0, W2, 64)]
    n = len(sbs)
    hme = {0: (np.array([48, 32], np.uint16), np.array([24, 16], np.uint16)), 1: (np.array([16, 16], np.uint16), np.array([8, 8], np.uint16)),
           2: (np.array([16, 16], np.uint16), np.array([8, 8], np.uint16))}
    for (rw, rh) in ((0, 0), (1, 1)):
        centers = None
        o_centers = np.zeros((n, 2), np.int16)
        for level in (0, 1, 2):
            src, ref, W, H, pad, stride = pics[level]
            sh = 2 - level
            org = np.array([(x >> sh, y >> sh) for (x, y, w, h) in sbs], np.int16)
            size = np.array([(w >> sh, h >> sh) for (x, y, w, h) in sbs], np.int16)
            hw, hh = hme[level]
            p = dsp.hme_level_params(level, hw, hh, rw, rh, int(hw.sum()), int(hh.sum()), 100, 100, pad, pad, W, H)
            d_src = dev(src); d_ref = dev(ref)
            cshift = 1 if level == 1 else 0                   # HmeLevel1 takes the level-0 vector >> 1 (:7791-7792)
            best, mv = dsp.hme_level(d_src, W, d_ref.view(-1)[pad * stride + pad:], stride, dev(org), dev(size), centers, cshift, p)
            gb = best.cpu().numpy(); gm = mv.cpu().numpy()
            po = svtlibs.hme_params(level, hw, hh, rw, rh, 100, 100, pad, W, H)
            ref00 = ctypes.c_void_p(ref.ctypes.data + pad * stride + pad)
            for i in range(n):
                b = np.zeros(1, np.uint64); x = np.zeros(1, np.int16); y = np.zeros(1, np.int16)
                xc, yc = (int(o_centers[i, 0]) >> cshift, int(o_centers[i, 1]) >> cshift) if level else (0, 0)
                O.svt_oracle_hme_level(ptr(src), W, ref00, stride, int(org[i, 0]), int(org[i, 1]), int(size[i, 0]), int(size[i, 1]), xc, yc,
                                       ctypes.byref(po), ptr(b), ptr(x), ptr(y))
                assert (int(gb[i]), int(gm[i, 0]), int(gm[i, 1])) == (int(b[0]), int(x[0]), int(y[0])), (level, rw, i, sbs[i])
                o_centers[i] = (x[0], y[0])
            centers = mv


def test_hme_regions_in_one_launch_equal_per_region_calls(dsp):
    """svt_hip_hme_level_regions_batch: the 2 x 2 search regions of a level in ONE launch, chained through the three levels with
    [region][task] centre planes, against one svt_hip_hme_level_batch call per region and level"""
    rng = np.random.default_rng(5)
    W2, H2 = 448, 256
    pics = {}
    for level, (W, H, pad) in {2: (W2, H2, 68), 1: (W2 // 2, H2 // 2, 34), 0: (W2 // 4, H2 // 4, 17)}.items():
        stride = W + 2 * pad + 5
        pics[level] = (dev(rng.integers(0, 256, (H, W), dtype=np.uint8)), dev(rng.integers(0, 256, (H + 2 * pad, stride), dtype=np.uint8)), W, H, pad, stride)
    sbs = [(x, y, min(64, W2 - x), min(64, H2 - y)) for y in range(0, H2, 64) for x in range(0, W2, 64)]
    hme = {0: (np.array([32, 32], np.uint16), np.array([16, 16], np.uint16)), 1: (np.array([16, 16], np.uint16), np.array([8, 8], np.uint16)),
           2: (np.array([16, 16], np.uint16), np.array([8, 8], np.uint16))}
    regions = [(rw, rh) for rw in (0, 1) for rh in (0, 1)]
    c_all, c_each = None, [None] * 4
    for level in (0, 1, 2):
        src, ref, W, H, pad, stride = pics[level]
        sh = 2 - level
        org = dev(np.array([(x >> sh, y >> sh) for (x, y, w, h) in sbs], np.int16)); size = dev(np.array([(w >> sh, h >> sh) for (x, y, w, h) in sbs], np.int16))
        hw, hh = hme[level]
        prm = [dsp.hme_level_params(level, hw, hh, rw, rh, int(hw.sum()), int(hh.sum()), 100, 100, pad, pad, W, H) for rw, rh in regions]
        ref00 = ref.view(-1)[pad * stride + pad:]
        cs = 1 if level == 1 else 0
        b_all, c_all = dsp.hme_level_regions(src, W, ref00, stride, org, size, c_all, cs, prm)
        for r in range(4):
            b, c_each[r] = dsp.hme_level(src, W, ref00, stride, org, size, c_each[r], cs, prm[r])
            assert torch.equal(b, b_all[r]) and torch.equal(c_each[r], c_all[r]), (level, r)


def test_hme_argument_errors(dsp, pkg):
    p = dsp.HmeParams()
    z = torch.zeros((1, 2), dtype=torch.int16, device="cuda")
    pic = torch.zeros((64, 64), dtype=torch.uint8, device="cuda")
    with pytest.raises(pkg.SvtHipError):
        dsp.hme_level(pic, 64, pic, 64, z, z, None, 0, p)          # zeroed parameter block
