"""GPU: the HIP path against the round-2 fixtures, which are outputs of the reference's OWN caller-level functions
(tests/golden/me.npz: FullPelSearch_LCU / open_loop_me_fullpel_search_sblock for asm_type 0 and 1; tests/golden/pins.npz:
av1_estimate_transform, av1_inv_txfm_add_c == _ssse3, full_distortion_kernel32_bits, av1_inv_txfm2d_add_*_c at the clamp
limits), and against the oracle on seeded inputs.  All calls go through the C ABI."""
import ctypes
import os

import numpy as np
import pytest
import torch

import svtlibs
from svtlibs import TX_H, TX_SIZES, TX_TYPES, TX_W, ptr, txfm_allowed

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ME_MAX = 128 * 128 * 255


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def me_g():
    return np.load(os.path.join(G, "me.npz"))


@pytest.fixture(scope="module")
def pins():
    return np.load(os.path.join(G, "pins.npz"))


@pytest.mark.parametrize("exact", [0, 1])
@pytest.mark.parametrize("nsq", [0, 1])
@pytest.mark.parametrize("flavour", [0, 1])
def test_me_fullpel_golden(dsp, me_g, flavour, nsq, exact):
    """K6 == the reference's own search drivers, asm_type 0 / 1, square and all 209 PUs, incl. the chained (IN/OUT) cases;
    exact = 1 forces the general search-point-by-search-point kernel, 0 lets the dispatcher take the fast kernels."""
    dsp.lib.svt_hip_tune(b"me_exact", exact)
    try:
        cases = me_g["cases"]
        for k, (sw, sh, xo, yo, _) in enumerate(cases):
            src = dev(me_g[f"c{k}_src"][None]); win = me_g[f"c{k}_win"]
            stride = win.shape[1]
            flat = dev(win.reshape(-1))
            zero = torch.zeros(1, dtype=torch.int32, device="cuda")
            bs, bm = dsp.me_fullpel_search(src, flat, int(sw), int(sh), int(xo), int(yo), flavour=flavour, nsq=bool(nsq),
                                           src_offsets=zero, ref_stride=stride, ref_offsets=zero, n=1)
            if k >= len(cases) - 2:      # chained second search on the running bests, window shifted by (1, 1)
                off = torch.tensor([stride + 1], dtype=torch.int32, device="cuda")
                dsp.me_fullpel_search(src, flat, max(1, int(sw) - 1), int(sh), int(xo) + 1, int(yo) + 1, flavour=flavour,
                                      nsq=bool(nsq), best_sad=bs, best_mv=bm, src_offsets=zero, ref_stride=stride,
                                      ref_offsets=off, n=1)
            gs = bs.cpu().numpy().view(np.uint32)[0]; gm = bm.cpu().numpy().view(np.uint32)[0]
            assert np.array_equal(gs, me_g[f"c{k}_sad_a{flavour}_n{nsq}"]), (k, "sad", np.nonzero(gs != me_g[f"c{k}_sad_a{flavour}_n{nsq}"])[0][:8])
            assert np.array_equal(gm, me_g[f"c{k}_mv_a{flavour}_n{nsq}"]), (k, "mv", np.nonzero(gm != me_g[f"c{k}_mv_a{flavour}_n{nsq}"])[0][:8])
    finally:
        dsp.lib.svt_hip_tune(b"me_exact", 0)


@pytest.mark.parametrize("sw,sh", [(8, 3), (16, 5), (64, 64), (24, 7), (7, 9), (20, 4), (48, 16), (1, 1), (40, 33)])
@pytest.mark.parametrize("nsq", [0, 1])
@pytest.mark.parametrize("flavour", [0, 1])
def test_me_fullpel_vs_oracle_batches(dsp, sw, sh, nsq, flavour):
    """dense batches, per-SB origins, ties and maximal SADs; fast kernels and the general kernel must both equal the oracle"""
    O = svtlibs.oracle()
    rng = np.random.default_rng(sw * 131 + sh * 7 + nsq * 3 + flavour)
    n = 6
    rw, rh = 64 + sw - 1 + 21, 64 + sh - 1 + 9       # slack rows / columns: the flavour-1 NSQ single-point row fetch
    src = rng.integers(0, 256, size=(n, 64, 64), dtype=np.uint8)
    ref = rng.integers(0, 256, size=(n, rh, rw), dtype=np.uint8)
    ref[0] = 100; src[0] = 101
    ref[3] = 255; src[3] = 0
    ref[4] = (ref[4] >> 6) << 6; src[4] = (src[4] >> 6) << 6
    ref[5] = (ref[5] >> 7) << 7; src[5] = (src[5] >> 7) << 7
    if sw > 4 and sh > 3:
        ref[1, 2:66, 3:67] = src[1]
        ref[2, 0:64, 1:65] = src[2]; ref[2, 1:65, 0:64] = src[2]
    org = np.array([[-32, -16], [0, 0], [-7, 5], [100, -100], [-64, -64], [3, -3]], np.int16)
    npu = 209 if nsq else 85
    exp_s = np.full((n, npu), ME_MAX, np.uint32); exp_m = np.zeros((n, npu), np.uint32)
    for i in range(n):
        bs = np.full(209, ME_MAX, np.uint32); bm = np.zeros(209, np.uint32)
        O.svt_oracle_me_sb_search_full(ptr(src[i]), 64, ptr(ref[i]), rw, sw, sh, int(org[i, 0]), int(org[i, 1]), flavour, nsq, ptr(bs), ptr(bm))
        exp_s[i] = bs[:npu]; exp_m[i] = bm[:npu]
    for exact in (0, 1):
        dsp.lib.svt_hip_tune(b"me_exact", exact)
        try:
            bs, bm = dsp.me_fullpel_search(dev(src), dev(ref), sw, sh, origins=dev(org), flavour=flavour, nsq=bool(nsq))
            gs = bs.cpu().numpy().view(np.uint32); gm = bm.cpu().numpy().view(np.uint32)
        finally:
            dsp.lib.svt_hip_tune(b"me_exact", 0)
        for i in range(n):
            assert np.array_equal(gs[i], exp_s[i]), (exact, i, "sad", np.nonzero(gs[i] != exp_s[i])[0][:8])
            assert np.array_equal(gm[i], exp_m[i]), (exact, i, "mv", np.nonzero(gm[i] != exp_m[i])[0][:8])


def test_estimate_transform_pack64_energy_golden_gpu(dsp, pins):
    """a5: svt_hip_fwd_txfm2d_batch + svt_hip_pack64_batch == the reference's av1_estimate_transform (coefficients re-packed
    to stride 32 and three_quad_energy), all five 64-point sizes and control sizes, bd 8 / 10"""
    n = 0
    for key in [k[:-3] for k in pins.files if k.startswith("est_") and k.endswith("_in")]:
        _, s, t, bd = key.split("_"); s, t, bd = int(s), int(t), int(bd)
        w, h = TX_W[s], TX_H[s]
        m = min(w, 32) * min(h, 32)
        co = dsp.fwd_txfm2d(dev(pins[key + "_in"]), s, t, bd)
        e = dsp.pack64(co, s)
        assert np.array_equal(co.cpu().numpy()[:, :m], pins[key + "_coeff"]), key
        assert np.array_equal(e.cpu().numpy().view(np.uint64), pins[key + "_energy"]), key
        n += 1
    assert n >= 20


def test_inv_txfm_add_u8_entry_golden_gpu(dsp, pins):
    """a6: the 8-bit reconstruction entry == av1_inv_txfm_add_c == av1_inv_txfm_add_ssse3 (production) for every size / type"""
    for s in range(19):
        for t in range(16):
            if not txfm_allowed(s, t):
                continue
            co, d0, d1 = pins[f"inv8_{s}_{t}_coeff"], pins[f"inv8_{s}_{t}_dst_in"], pins[f"inv8_{s}_{t}_dst_out"]
            d = dev(d0)
            dsp.inv_txfm2d_add(dev(co), d, s, t, bd=8)
            assert np.array_equal(d.cpu().numpy(), d1), (TX_SIZES[s], TX_TYPES[t])


def test_full_distortion32_golden_gpu(dsp, pins):
    """a12 coefficient domain: full_distortion_kernel32_bits / _cbf_zero32_bits (C) incl. 2^26-magnitude coefficients"""
    for key in [k[:-2] for k in pins.files if k.startswith("dist_") and k.endswith("_a")]:
        w, h = (int(v) for v in key.split("_")[1].split("x"))
        a = np.ascontiguousarray(pins[key + "_a"][:, :, :w]); b = np.ascontiguousarray(pins[key + "_b"][:, :, :w])
        out = pins[key + "_out"]
        got = dsp.full_distortion32(dev(a), dev(b), w, h).cpu().numpy().view(np.uint64)
        got0 = dsp.full_distortion32(dev(a), None, w, h, cbf_zero=True).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, out[:, 0]), key
        assert np.array_equal(got0, out[:, 1]), key


def test_inv_txfm2d_add_clamp_limits_golden_gpu(dsp, pins):
    """coefficients at +-2^(bd+7) and beyond, bd 8 / 10 / 12 (the add on 16-bit lanes must saturate, not wrap)"""
    n = 0
    for key in [k[:-6] for k in pins.files if k.startswith("sat_") and k.endswith("_coeff")]:
        _, s, t, bd = key.split("_"); s, t, bd = int(s), int(t), int(bd)
        d = dev(pins[key + "_dst_in"].view(np.int16))
        dsp.inv_txfm2d_add(dev(pins[key + "_coeff"]), d, s, t, bd=bd)
        got = d.cpu().numpy().view(np.uint16)
        assert np.array_equal(got, pins[key + "_dst_out"]), (key, np.argwhere(got != pins[key + "_dst_out"])[:4])
        n += 1
    assert n == 39


def test_calls_from_a_fresh_thread_use_the_library_device(dsp):
    """the HIP current device is per thread; a batched call from a new thread must run on the library's device"""
    import threading
    res = {}

    def work():
        try:
            a = torch.randint(0, 256, (64, 16, 16), dtype=torch.uint8, device="cuda")
            b = torch.randint(0, 256, (64, 16, 16), dtype=torch.uint8, device="cuda")
            s = dsp.sad(a, b)
            torch.cuda.synchronize()
            exp = (a.int() - b.int()).abs().sum(dim=(1, 2))
            res["ok"] = bool((s.int() == exp.int()).all())
        except Exception as e:       # noqa: BLE001
            res["err"] = repr(e)
    th = threading.Thread(target=work)
    th.start(); th.join()
    assert res.get("ok"), res
