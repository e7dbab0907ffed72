"""Open-loop intra search through the C ABI: bit-exact against the outputs of the reference's own
open_loop_intra_search_sb (tests/golden/ois.npz, made by oracle/ref_ois.c) and against oracle/ois.c on a whole
picture."""
import ctypes
import os

import numpy as np
import pytest
import torch

import svtlibs
from svtlibs import ptr
from test_gpu_parity import dev
from test_oracle_golden import ois_md_scan, ois_raster_idx

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
c_int = ctypes.c_int


def _xy(blocks):
    return dev(np.array([(y << 16) | x for (x, y) in blocks], np.uint32).view(np.int32))


def test_ois_golden(dsp):
    g = np.load(os.path.join(G, "ois.npz"))
    buf = np.ascontiguousarray(g["pic"]); W, H, pad = (int(v) for v in g["dims"])
    stride = buf.shape[1]
    plane = dev(buf)
    pic = plane[pad:, pad:]
    md = ois_md_scan()
    checked = 0
    for k, (sx, sy, tl, ipm, isref) in enumerate(g["cases"].tolist()):
        for bsize in (8, 16, 32, 64):
            idx = [i for i, (x, y, s) in enumerate(md) if s == bsize and g[f"c{k}_valid"][ois_raster_idx(x, y, s)]]
            if not idx:
                continue
            modes, deltas = dsp.ois_candidates(bsize, tl, ipm, bool(isref))
            n = len(modes)
            for i in idx:                                    # the host-side candidate list is the reference's
                assert int(g[f"c{k}_count"][i]) == n
                assert np.array_equal(g[f"c{k}_mode"][i, :n], modes) and np.array_equal(g[f"c{k}_delta"][i, :n], deltas)
            dist, best = dsp.ois_search(pic, stride, W, H, _xy([(sx + md[i][0], sy + md[i][1]) for i in idx]), bsize, modes, deltas)
            want = np.stack([g[f"c{k}_dist"][i, :n] for i in idx]).astype(np.int64)
            assert np.array_equal(dist.cpu().numpy().astype(np.int64), want), (k, bsize)
            assert np.array_equal(best.cpu().numpy(), np.array([g[f"c{k}_best"][i] for i in idx], np.int8)), (k, bsize)
            checked += len(idx)
    assert checked > 400


@pytest.mark.parametrize("bsize,tl", [(8, 0), (16, 0), (32, 0), (64, 0), (16, 1)])
def test_ois_whole_picture_vs_oracle(dsp, bsize, tl):
    """every block of one size of a 416x240 picture (not a multiple of 64: partial SBs; blocks on all four borders)"""
    O = svtlibs.oracle()
    rng = np.random.default_rng(bsize + tl)
    W, H, pad = 416, 240, 80
    buf = rng.integers(0, 256, size=(H + 2 * pad, W + 2 * pad), dtype=np.uint8)
    buf[pad + 100:pad + 180, pad + 200:pad + 330] = 90       # flat area: ties, first strict minimum
    buf[pad:pad + 64, pad:pad + 64] = 255; buf[pad:pad + 64, pad + 64:pad + 128] = 0
    stride = buf.shape[1]
    blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
    rng.shuffle(blocks)
    modes, deltas = dsp.ois_candidates(bsize, tl)
    n = len(modes)
    plane = dev(buf)
    dist, best = dsp.ois_search(plane[pad:, pad:], stride, W, H, _xy(blocks), bsize, modes, deltas)
    dist = dist.cpu().numpy(); best = best.cpu().numpy()
    pic = ctypes.c_void_p(buf.ctypes.data + pad * stride + pad)
    om = np.zeros(61, np.uint8); od = np.zeros(61, np.int8)
    assert O.svt_oracle_ois_candidates(c_int(bsize), c_int(tl), c_int(0), c_int(1), c_int(0), ptr(om), ptr(od)) == n
    assert np.array_equal(om[:n], modes) and np.array_equal(od[:n], deltas)
    ds = np.zeros(61, np.uint32)
    for i, (x, y) in enumerate(blocks):
        bi = O.svt_oracle_ois_block(pic, c_int(stride), c_int(W), c_int(H), c_int(x), c_int(y), c_int(bsize), c_int(n), ptr(om), ptr(od), ptr(ds))
        assert np.array_equal(dist[i].astype(np.int64), ds[:n].astype(np.int64)), (i, x, y)
        assert int(best[i]) == bi, (i, x, y)


@pytest.mark.parametrize("bsize", [8, 16])
def test_ois_fold_variant(dsp, bsize):
    """SAD folded into the directional kernels (default for 8x8 / 16x16) == predictions through scratch + SAD kernel"""
    rng = np.random.default_rng(bsize)
    W, H, pad = 352, 288, 32
    buf = rng.integers(0, 256, size=(H + 2 * pad, W + 2 * pad), dtype=np.uint8)
    blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
    modes, deltas = dsp.ois_candidates(bsize)
    plane = dev(buf)
    out = []
    try:
        for v in (0, 1):
            assert dsp.lib.svt_hip_tune(b"ois_no_fold", v) == 0
            out.append(dsp.ois_search(plane[pad:, pad:], W + 2 * pad, W, H, _xy(blocks), bsize, modes, deltas))
    finally:
        dsp.lib.svt_hip_tune(b"ois_no_fold", 0)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("bsize", [8, 16])
def test_ois_three_zones_in_one_launch_variant(dsp, bsize):
    """ois_dir3_kernel (the three directional zones of the list in one launch, default) == one launch per zone; also a list whose
    directional candidates all lie in one zone (stays on the per-zone kernel) against the oracle-pinned default"""
    rng = np.random.default_rng(900 + bsize)
    W, H, pad = 416, 240, 16
    buf = rng.integers(0, 256, size=(H + 2 * pad, W + 2 * pad), dtype=np.uint8)
    blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
    modes, deltas = dsp.ois_candidates(bsize)
    plane = dev(buf)
    lists = [(modes, deltas), (np.array([0, 3, 3, 8, 8, 1], np.uint8), np.array([0, -2, 1, 0, 3, 0], np.int8)),      # zone 1 only (+ DC, V)
             (np.array([4, 7, 2, 12], np.uint8), np.array([1, -3, 0, 0], np.int8))]                                 # zones 2 and 3
    for m, d in lists:
        out = []
        try:
            for v in (0, 1):
                assert dsp.lib.svt_hip_tune(b"ois_no_dir3", v) == 0
                out.append(dsp.ois_search(plane[pad:, pad:], W + 2 * pad, W, H, _xy(blocks), bsize, m, d))
        finally:
            dsp.lib.svt_hip_tune(b"ois_no_dir3", 0)
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("bsize", [8, 16])
def test_ois_directional_angles_split_over_grid_variant(dsp, bsize):
    """small batches spread the directional angles of a zone over grid.y (dir_no_split = 0) == every workgroup walks all angles"""
    rng = np.random.default_rng(300 + bsize)
    W, H = 320, 192
    buf = rng.integers(0, 256, size=(H, W + 24), dtype=np.uint8)
    blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
    modes, deltas = dsp.ois_candidates(bsize)
    plane = dev(buf)
    out = []
    try:
        for v in (0, 1):
            assert dsp.lib.svt_hip_tune(b"dir_no_split", v) == 0
            out.append(dsp.ois_search(plane, W + 24, W, H, _xy(blocks), bsize, modes, deltas))
    finally:
        dsp.lib.svt_hip_tune(b"dir_no_split", 0)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("bsize,tl", [(8, 0), (16, 0), (8, 2), (32, 0), (64, 0), (32, 2)])
def test_ois_fused_non_directional_variant(dsp, bsize, tl):
    """ois_nd_kernel (non-directional candidates predicted and summed straight from the picture, one launch when the list has no
    directional candidate) == the gather -> per-candidate prediction batches -> SAD kernel path; blocks at every picture border"""
    rng = np.random.default_rng(100 + bsize + tl)
    W, H, pad = 320, 192, 0
    buf = rng.integers(0, 256, size=(H, W + 24), dtype=np.uint8)
    buf[: H // 2] = (buf[: H // 2] >> 5) << 5                 # coarse half: ties between candidates
    blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
    modes, deltas = dsp.ois_candidates(bsize, tl)
    plane = dev(buf)
    out = []
    try:
        for v in (0, 1):
            assert dsp.lib.svt_hip_tune(b"ois_no_nd", v) == 0
            out.append(dsp.ois_search(plane, W + 24, W, H, _xy(blocks), bsize, modes, deltas))
    finally:
        dsp.lib.svt_hip_tune(b"ois_no_nd", 0)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_ois_candidate_lists_match_oracle(dsp):
    O = svtlibs.oracle()
    for bsize in (8, 16, 32, 64):
        for tl in (0, 2):
            for ipm in (0, 4, 5):
                for isref in (0, 1):
                    for is16 in (0, 1):
                        om = np.zeros(61, np.uint8); od = np.zeros(61, np.int8)
                        n = O.svt_oracle_ois_candidates(c_int(bsize), c_int(tl), c_int(ipm), c_int(isref), c_int(is16), ptr(om), ptr(od))
                        m, d = dsp.ois_candidates(bsize, tl, ipm, bool(isref), bool(is16))
                        assert n == len(m) and np.array_equal(om[:n], m) and np.array_equal(od[:n], d)


def test_ois_argument_errors(dsp):
    L = dsp.lib
    z = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    p = z.data_ptr()
    m = np.array([0, 1], np.uint8); d = np.array([0, 9], np.int8)
    wb = L.svt_hip_ois_work_bytes(8, 2, 4)
    assert wb > 0 and L.svt_hip_ois_work_bytes(12, 2, 4) == 0 and L.svt_hip_ois_work_bytes(8, 62, 4) == 0
    assert L.svt_hip_ois_search_batch(p, 64, 64, 64, p, 12, m.ctypes.data, d.ctypes.data, 1, p, p, p, wb, 4, None) != 0       # size
    assert L.svt_hip_ois_search_batch(p, 64, 64, 64, p, 8, m.ctypes.data, d.ctypes.data, 2, p, p, p, wb, 4, None) != 0        # 90 + 27 has no derivative
    assert L.svt_hip_ois_search_batch(p, 64, 64, 64, p, 8, m.ctypes.data, d.ctypes.data, 1, p, p, p, L.svt_hip_ois_work_bytes(8, 1, 4) - 1, 4, None) != 0    # work buffer
    assert L.svt_hip_ois_search_batch(p, 64, 64, 64, p, 8, m.ctypes.data, d.ctypes.data, 1, p, p, p, wb, 0, None) == 0        # empty


def test_ois_search_frame_equals_per_size_calls(dsp):
    """svt_hip_ois_search_frame (the block sizes of a picture in one call: every group's chain, then ONE non-directional launch for
    all of them) == one svt_hip_ois_search_batch per size"""
    rng = np.random.default_rng(2024)
    W, H = 384, 256
    buf = rng.integers(0, 256, size=(H, W + 16), dtype=np.uint8)
    plane = dev(buf)
    groups, single = [], []
    for bsize in (8, 16, 32, 64):
        blocks = [(x, y) for y in range(0, H - bsize + 1, bsize) for x in range(0, W - bsize + 1, bsize)]
        modes, deltas = dsp.ois_candidates(bsize)
        xy = _xy(blocks)
        groups.append((xy, bsize, modes, deltas))
        single.append(dsp.ois_search(plane, W + 16, W, H, xy, bsize, modes, deltas))
    # six groups (more than one merged non-directional launch holds) and, second, one non-directional launch per group
    groups6 = groups + [groups[1], groups[3]]
    single6 = single + [single[1], single[3]]
    orders = [list(range(6)), [0, 2, 3, 5, 4, 1], [1, 4, 0, 2, 3, 5], [2, 3, 1, 0, 5, 4]]      # the merged launches must not depend on the order
    try:
        for knob in (0, 1):
            assert dsp.lib.svt_hip_tune(b"ois_no_nd_multi", knob) == 0
            for od in orders:
                outs = dsp.ois_search_frame(plane, W + 16, W, H, [groups6[i] for i in od])
                torch.cuda.synchronize()
                for i, (d2, b2) in zip(od, outs):
                    assert torch.equal(single6[i][0], d2) and torch.equal(single6[i][1], b2), (knob, od, i)
    finally:
        dsp.lib.svt_hip_tune(b"ois_no_nd_multi", 0)
