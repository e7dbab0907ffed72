"""K11 chroma-from-luma helpers and av1_txb_init_levels through the C ABI: bit-exact against
the reference's outputs (tests/golden/cfl_levels.npz) and against oracle/cfl.c on planes at larger scale."""
import ctypes
import os

import numpy as np
import pytest
import torch

import svtlibs
from svtlibs import ptr
from test_gpu_parity import dev

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
c_int = ctypes.c_int
SHAPES = [(4, 4), (8, 8), (16, 16), (32, 32), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (4, 16), (16, 4), (8, 32), (32, 8)]


def _t(a):
    return dev(a.view(np.int16) if a.dtype == np.uint16 else a)


def _np(t, dt):
    a = t.cpu().numpy()
    return a.view(np.uint16) if dt == np.uint16 else a


@pytest.mark.parametrize("w,h", SHAPES)
@pytest.mark.parametrize("bd", [8, 10])
def test_cfl_golden(dsp, w, h, bd):
    g = np.load(os.path.join(G, "cfl_levels.npz"))
    key = f"cfl_{w}x{h}_{bd}"
    luma = g[key + "_luma"]
    dt = luma.dtype
    n, _, ls = luma.shape
    q3 = dsp.cfl_luma_subsampling_420(_t(luma), ls, 2 * w, 2 * h, luma_block_pitch=2 * h * ls, n=n)
    assert np.array_equal(q3.cpu().numpy()[:, :h, :w], g[key + "_q3"][:, :h, :w])
    assert not q3.cpu().numpy()[:, h:, :].any() and not q3.cpu().numpy()[:, :, w:].any()      # nothing outside the block
    ac1 = dsp.cfl_luma_subsampling_420(_t(luma), ls, 2 * w, 2 * h, luma_block_pitch=2 * h * ls, n=n, subtract_average=True)
    assert np.array_equal(ac1.cpu().numpy()[:, :h, :w], g[key + "_ac"][:, :h, :w])
    ac2 = dsp.subtract_average(q3.clone(), w, h, w * h // 2, int(np.log2(w * h)))
    assert torch.equal(ac1, ac2)
    pred = g[key + "_pred"]
    ps = pred.shape[2]
    dst = torch.zeros_like(_t(pred))
    dsp.cfl_predict(dev(np.ascontiguousarray(g[key + "_ac"])), _t(pred), ps, dst, ps, dev(g[key + "_alpha"]), bd, w, h)
    assert np.array_equal(_np(dst, dt), g[key + "_dst"])


@pytest.mark.parametrize("w,h", [(4, 4), (8, 8), (16, 16), (32, 32), (4, 8), (8, 4), (16, 32), (32, 16), (4, 16), (16, 4), (8, 32), (32, 8), (8, 16), (16, 8)])
def test_txb_init_levels_golden(dsp, w, h):
    g = np.load(os.path.join(G, "cfl_levels.npz"))
    coeff = g[f"lv_{w}x{h}_coeff"]
    n = coeff.shape[0]
    size = (w + 4) * (h + 6) + 16
    for pitch in ((size + 15) // 16 * 16, (size + 15) // 16 * 16 + 4):       # 16-byte stores / dword stores
        buf = torch.full((n, pitch), 0x55, dtype=torch.uint8, device="cuda")
        dsp.txb_init_levels(dev(coeff.reshape(n, -1)), w, h, buf)
        got = buf.cpu().numpy()
        assert np.array_equal(got[:, :size], g[f"lv_{w}x{h}_levels"])
        assert (got[:, size:] == 0x55).all()                                  # slack between buffers untouched


@pytest.mark.parametrize("w,h,bd", [(4, 4, 8), (8, 8, 8), (16, 16, 10), (32, 32, 8), (32, 32, 12), (16, 8, 8), (8, 32, 10), (4, 16, 12)])
def test_cfl_chain_on_planes(dsp, w, h, bd):
    """The encode pass's order on picture planes: subsample + subtract average from the luma recon plane, then
    predict both chroma planes IN PLACE with per-block alphas; oracle per block."""
    O = svtlibs.oracle()
    rng = np.random.default_rng(w * 100 + h + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    PW, PH = 256 + 6, 128                       # chroma plane; luma is twice as large
    luma = rng.integers(0, 1 << bd, size=(2 * PH, 2 * PW)).astype(dt)
    cb = rng.integers(0, 1 << bd, size=(PH, PW)).astype(dt)
    xs = np.arange(0, 256 - w + 1, w); ys = np.arange(0, PH - h + 1, h)
    pos = [(int(x) + 2, int(y)) for y in ys for x in xs]          # x offset 2: chunks not 16-byte aligned
    rng.shuffle(pos)
    n = len(pos)
    xy_c = np.array([(y << 16) | x for x, y in pos], np.uint32).view(np.int32)
    xy_l = np.array([((2 * y) << 16) | (2 * x) for x, y in pos], np.uint32).view(np.int32)
    alpha = rng.integers(-16, 17, size=n).astype(np.int32)
    ac = dsp.cfl_luma_subsampling_420(_t(luma), 2 * PW, 2 * w, 2 * h, xy=dev(xy_l), subtract_average=True)
    plane = _t(cb.copy())
    dsp.cfl_predict(ac, plane, PW, plane, PW, dev(alpha), bd, w, h, xy=dev(xy_c))
    got_ac = ac.cpu().numpy(); got = _np(plane, dt)
    want = cb.copy()
    lg = int(np.log2(w * h))
    for i, (x, y) in enumerate(pos):
        q3 = np.zeros((32, 32), np.int16)
        lv = luma[2 * y:, 2 * x:]
        O.svt_oracle_cfl_luma_subsampling_420(ctypes.c_void_p(luma.ctypes.data + (2 * y * 2 * PW + 2 * x) * luma.itemsize), c_int(bd > 8),
                                              c_int(2 * PW), ptr(q3), c_int(2 * w), c_int(2 * h))
        O.svt_oracle_subtract_average(ptr(q3), c_int(w), c_int(h), c_int(w * h // 2), c_int(lg))
        assert np.array_equal(got_ac[i], q3), (i, x, y)
        at = ctypes.c_void_p(want.ctypes.data + (y * PW + x) * want.itemsize)
        O.svt_oracle_cfl_predict(ptr(q3), at, c_int(PW), at, c_int(PW), c_int(int(alpha[i])), c_int(bd), c_int(w), c_int(h), c_int(bd > 8))
    assert np.array_equal(got, want)


def test_txb_init_levels_at_scale(dsp):
    O = svtlibs.oracle()
    rng = np.random.default_rng(77)
    for (w, h) in ((32, 32), (64, 64), (64, 16), (4, 4)):
        n = 301
        coeff = rng.integers(-200, 201, size=(n, h * w)).astype(np.int32)
        coeff[::7] *= 1 << 16
        pitch = ((w + 4) * (h + 6) + 16 + 8 + 15) // 16 * 16      # 16-byte-aligned buffers; the slack is not touched
        buf = torch.full((n, pitch), 0x77, dtype=torch.uint8, device="cuda")
        dsp.txb_init_levels(dev(coeff), w, h, buf)
        got = buf.cpu().numpy()
        for i in range(0, n, 13):
            lv = np.full(pitch, 0x77, np.uint8)
            O.svt_oracle_txb_init_levels(ptr(coeff[i]), c_int(w), c_int(h), ctypes.c_void_p(lv.ctypes.data + 2 * (w + 4)))
            assert np.array_equal(got[i], lv), (w, h, i)
        # size-independent property on every block: the interior equals min(|c|, 127), everything else is zero
        body = got[:, :(w + 4) * (h + 6)].reshape(n, h + 6, w + 4)
        assert np.array_equal(body[:, 2:2 + h, :w], np.minimum(np.abs(coeff.astype(np.int64)), 127).astype(np.uint8).reshape(n, h, w))
        assert not body[:, :2].any() and not body[:, 2 + h:].any() and not body[:, :, w:].any()


def test_cfl_argument_errors(dsp):
    L = dsp.lib
    z = torch.zeros(4096, dtype=torch.int16, device="cuda")
    p = ctypes.c_void_p(z.data_ptr())
    assert L.svt_hip_cfl_luma_subsampling_420_batch(p, 64, 0, None, 0, p, 32, 1024, 6, 8, 0, 1, None) != 0      # chroma 3x4
    assert L.svt_hip_cfl_luma_subsampling_420_batch(p, 64, 0, None, 0, p, 4, 1024, 16, 16, 0, 1, None) != 0     # q3 line < width
    assert L.svt_hip_subtract_average_batch(None, 32, 1024, 8, 8, 32, 6, 1, None) != 0
    assert L.svt_hip_cfl_predict_batch(p, 32, 1024, p, 8, p, 8, None, p, 9, 8, 8, 0, 1, None) != 0                # 8-bit samples, bd 9
    assert L.svt_hip_txb_init_levels_batch(p, 16, p, 90, 4, 4, 1, None) != 0                                      # buffer < 96 B
    assert L.svt_hip_txb_init_levels_batch(p, 16, p, 96, 4, 4, 0, None) == 0                                      # empty batch
