"""Frame-level (plane-addressed) search entry points: SURVEY §8(f) n1.  A padded picture pair, one 64x64
superblock per grid cell, windows clipped to the padded reference as HmeLevel0 / FullPelSearch_LCU do
(EbMotionEstimation.c:5720-5798, 3210-3225); results must equal the per-block oracle on the same windows."""
import ctypes

import numpy as np
import pytest
import torch

import svtlibs
from svtlibs import ptr
from test_gpu_parity import dev

pytestmark = pytest.mark.gpu


def make_frame(rng, w, h, pad):
    src = rng.integers(0, 256, size=(h + 2 * pad, w + 2 * pad), dtype=np.uint8)
    # reference = shifted source + noise, so the search has real structure
    ref = np.roll(src, (3, -5), axis=(0, 1))
    ref = np.clip(ref.astype(int) + rng.integers(-4, 5, size=ref.shape), 0, 255).astype(np.uint8)
    return src, ref


def test_me_sb_search_on_planes_matches_oracle(dsp):
    O = svtlibs.oracle()
    rng = np.random.default_rng(2024)
    W, H, PAD, SW, SH = 320, 192, 80, 64, 32
    src, ref = make_frame(rng, W, H, PAD)
    stride = src.shape[1]
    sbs = [(x, y) for y in range(0, H, 64) for x in range(0, W, 64)]
    org = np.array([[-(SW // 2), -(SH // 2)]] * len(sbs), np.int16)      # search area centred on the SB
    soff = np.array([(PAD + y) * stride + PAD + x for x, y in sbs], np.uint32)
    roff = np.array([(PAD + y + int(o[1])) * stride + PAD + x + int(o[0]) for (x, y), o in zip(sbs, org)], np.uint32)
    bs, bm = dsp.me_sb_search_planes(dev(src), stride, dev(soff.view(np.int32)), dev(ref), stride, dev(roff.view(np.int32)),
                                     SW, SH, origins=dev(org))
    torch.cuda.synchronize()
    bs = bs.cpu().numpy().view(np.uint32); bm = bm.cpu().numpy().view(np.uint32)
    for i, (x, y) in enumerate(sbs):
        rs = np.full(85, 128 * 128 * 255, np.uint32); rm = np.zeros(85, np.uint32)
        sblk = np.ascontiguousarray(src[PAD + y:PAD + y + 64, PAD + x:PAD + x + 64])
        r0 = PAD + y + int(org[i, 1]); c0 = PAD + x + int(org[i, 0])
        win = np.ascontiguousarray(ref[r0:r0 + 64 + SH - 1, c0:c0 + 64 + SW - 1 + 1])
        O.svt_oracle_me_sb_search(ptr(sblk), 64, ptr(win), win.shape[1], SW, SH, int(org[i, 0]), int(org[i, 1]), ptr(rs), ptr(rm))
        assert np.array_equal(bs[i], rs), f"sad SB {i}"
        assert np.array_equal(bm[i], rm), f"mv SB {i}"


@pytest.mark.parametrize("bw,bh,sw,sh", [(16, 16, 8, 8), (8, 8, 16, 5), (32, 32, 24, 9), (64, 64, 16, 16)])
def test_sad_search_on_planes_matches_oracle(dsp, bw, bh, sw, sh):
    O = svtlibs.oracle()
    rng = np.random.default_rng(bw * 7 + sw)
    W, H, PAD = 256, 128, 48
    src, ref = make_frame(rng, W, H, PAD)
    stride = src.shape[1]
    blocks = [(x, y) for y in range(0, H, bh) for x in range(0, W, bw)]
    soff = np.array([(PAD + y) * stride + PAD + x for x, y in blocks], np.uint32)
    roff = np.array([(PAD + y - sh // 2) * stride + PAD + x - sw // 2 for x, y in blocks], np.uint32)
    best, bx, by = dsp.sad_search_planes(dev(src), stride, dev(soff.view(np.int32)), dev(ref), stride, dev(roff.view(np.int32)),
                                         bw, bh, sw, sh)
    torch.cuda.synchronize()
    best = best.cpu().numpy(); bx = bx.cpu().numpy(); by = by.cpu().numpy()
    for i, (x, y) in enumerate(blocks):
        sblk = np.ascontiguousarray(src[PAD + y:PAD + y + bh, PAD + x:PAD + x + bw])
        r0 = PAD + y - sh // 2; c0 = PAD + x - sw // 2
        win = np.ascontiguousarray(ref[r0:r0 + bh + sh - 1, c0:c0 + bw + sw - 1])
        rb = np.zeros(1, np.uint64); rx = np.zeros(1, np.int16); ry = np.zeros(1, np.int16)
        O.svt_oracle_sad_loop(ptr(sblk), bw, ptr(win), win.shape[1], bh, bw, ptr(rb), ptr(rx), ptr(ry), win.shape[1],
                              ctypes.c_int16(sw), ctypes.c_int16(sh))
        assert (int(best[i]), int(bx[i]), int(by[i])) == (int(rb[0]), int(rx[0]), int(ry[0])), (i, x, y)
