"""GPU: the boundary closed in round 2 - every per-size RTCD slot drop-in reached BY NAME through the slot registry (as a
host that overrides its dispatch pointers would), the new batched SAD / residual / distortion entry points, and a plain-C
host (tests/c/rtcd_caller.c) that links -lsvt_hip_dsp and calls through its own pointer block.  Checker: the oracle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

import svtlibs
from svtlibs import TX_H, TX_W, ptr

pytestmark = pytest.mark.gpu
c_int = ctypes.c_int
U = ctypes.c_uint32
S = ctypes.c_ssize_t
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INTRA = ["dc", "v", "h", "smooth", "smooth_v", "smooth_h", "paeth", "dc_top", "dc_left", "dc_128"]     # oracle mode numbering


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def slot(dsp):
    L = dsp.lib
    L.svt_hip_rtcd_slot_function.restype = ctypes.c_void_p
    L.svt_hip_rtcd_slot_function.argtypes = [ctypes.c_char_p]

    def get(name, restype, *argtypes):
        addr = L.svt_hip_rtcd_slot_function(name.encode())
        assert addr, name
        return ctypes.CFUNCTYPE(restype, *argtypes)(addr)
    return get


VP = ctypes.c_void_p


def test_every_intra_slot_by_name(slot):
    """380 slots: aom_<mode>_predictor_WxH and aom_highbd_<mode>_predictor_WxH with the exact intra_pred_fn signature"""
    O = svtlibs.oracle()
    rng = np.random.default_rng(31)
    n = 0
    for s in range(19):
        bw, bh = TX_W[s], TX_H[s]
        for mode, name in enumerate(INTRA):
            a = rng.integers(0, 256, size=16 + 2 * 64 + 16, dtype=np.uint8); l = rng.integers(0, 256, size=16 + 2 * 64 + 16, dtype=np.uint8)
            d1 = np.full((bh, 80), 3, np.uint8); d2 = d1.copy()
            f = slot(f"aom_{name}_predictor_{bw}x{bh}", None, VP, S, VP, VP)
            f(d1.ctypes.data, 80, a.ctypes.data + 16, l.ctypes.data + 16)
            O.svt_oracle_intra_pred(mode, ptr(d2), S(80), bw, bh, VP(a.ctypes.data + 16), VP(l.ctypes.data + 16))
            assert np.array_equal(d1, d2), (name, bw, bh)
            a16 = rng.integers(0, 1024, size=16 + 2 * 64 + 16).astype(np.uint16); l16 = rng.integers(0, 1024, size=16 + 2 * 64 + 16).astype(np.uint16)
            h1 = np.full((bh, 72), 5, np.uint16); h2 = h1.copy()
            fh = slot(f"aom_highbd_{name}_predictor_{bw}x{bh}", None, VP, S, VP, VP, c_int)
            fh(h1.ctypes.data, 72, a16.ctypes.data + 32, l16.ctypes.data + 32, 10)
            O.svt_oracle_intra_pred_hbd(mode, ptr(h2), S(72), bw, bh, VP(a16.ctypes.data + 32), VP(l16.ctypes.data + 32), 10)
            assert np.array_equal(h1, h2), ("highbd", name, bw, bh)
            n += 2
    assert n == 380


def test_sad_slots_by_name(slot):
    O = svtlibs.oracle()
    O.svt_oracle_sad.restype = ctypes.c_uint32
    rng = np.random.default_rng(32)
    sizes = [(128, 128), (128, 64), (64, 128), (64, 64), (64, 32), (32, 64), (32, 32), (32, 16), (16, 32), (16, 16), (16, 8), (8, 16),
             (8, 8), (8, 4), (4, 8), (4, 4), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]
    for (w, h) in sizes:
        s = rng.integers(0, 256, (h, w + 3), dtype=np.uint8)
        refs = [rng.integers(0, 256, (h + 2, w + 7), dtype=np.uint8) for _ in range(4)]
        exp = [O.svt_oracle_sad(ptr(s), U(w + 3), ptr(r), U(w + 7), U(h), U(w)) for r in refs]
        f = slot(f"aom_sad{w}x{h}", ctypes.c_uint32, VP, c_int, VP, c_int)
        assert [f(s.ctypes.data, w + 3, r.ctypes.data, w + 7) for r in refs] == exp, (w, h)
        arr = (VP * 4)(*[r.ctypes.data for r in refs])
        out = np.zeros(4, np.uint32)
        slot(f"aom_sad{w}x{h}x4d", None, VP, c_int, VP, c_int, VP)(s.ctypes.data, w + 3, ctypes.addressof(arr), w + 7, out.ctypes.data)
        assert list(out) == exp, (w, h, "x4d")


def test_helper_slots_by_name(slot, dsp):
    """edge filter / upsample, subtract_average, cfl_predict, txb_init_levels, eb_smooth_*, combined_averaging_sad, residual16"""
    O = svtlibs.oracle()
    rng = np.random.default_rng(33)
    for sz in (5, 9, 17, 33, 65):
        for strength in (0, 1, 2, 3):
            e = rng.integers(0, 256, size=sz + 8).astype(np.uint8); e2 = e.copy()
            slot("av1_filter_intra_edge", None, VP, c_int, c_int)(e.ctypes.data, sz, strength)
            O.svt_oracle_filter_intra_edge(ptr(e2), sz, strength)
            assert np.array_equal(e, e2), (sz, strength)
            h = rng.integers(0, 1024, size=sz + 8).astype(np.uint16); h2 = h.copy()
            slot("av1_filter_intra_edge_high", None, VP, c_int, c_int)(h.ctypes.data, sz, strength)
            O.svt_oracle_filter_intra_edge_hbd(ptr(h2), sz, strength)
            assert np.array_equal(h, h2), (sz, strength, "high")
    for sz in (4, 8, 16):
        b = rng.integers(0, 256, size=64).astype(np.uint8); b2 = b.copy()
        slot("av1_upsample_intra_edge", None, VP, c_int)(b.ctypes.data + 16, sz)
        O.svt_oracle_upsample_intra_edge(VP(b2.ctypes.data + 16), sz)
        assert np.array_equal(b, b2), sz
        h = rng.integers(0, 1024, size=64).astype(np.uint16); h2 = h.copy()
        slot("av1_upsample_intra_edge_high", None, VP, c_int, c_int)(h.ctypes.data + 32, sz, 10)
        O.svt_oracle_upsample_intra_edge_hbd(VP(h2.ctypes.data + 32), sz, 10)
        assert np.array_equal(h, h2), (sz, "high")
    g = np.load(os.path.join(G, "cfl_levels.npz"))
    for (w, h) in ((4, 4), (8, 8), (16, 16), (32, 32), (8, 16), (32, 8)):
        for bd in (8, 10):
            key = f"cfl_{w}x{h}_{bd}"
            q3 = g[key + "_q3"][0].copy()
            slot("subtract_average", None, VP, c_int, c_int, c_int, c_int)(q3.ctypes.data, w, h, w * h // 2, int(np.log2(w * h)))
            assert np.array_equal(q3[:h, :w], g[key + "_ac"][0][:h, :w]), key
            pred = np.ascontiguousarray(g[key + "_pred"][0]); out = np.zeros_like(pred)
            ac = np.ascontiguousarray(g[key + "_ac"][0])
            name = "cfl_predict_lbd" if bd == 8 else "cfl_predict_hbd"
            slot(name, None, VP, VP, c_int, VP, c_int, c_int, c_int, c_int, c_int)(ac.ctypes.data, pred.ctypes.data, pred.shape[1], out.ctypes.data,
                                                                                     pred.shape[1], int(g[key + "_alpha"][0]), bd, w, h)
            assert np.array_equal(out[:, :w], g[key + "_dst"][0][:, :w]), (key, "predict")
    for (w, h) in ((4, 4), (16, 16), (32, 32), (8, 32), (16, 4)):
        co = np.ascontiguousarray(g[f"lv_{w}x{h}_coeff"][1]); exp = g[f"lv_{w}x{h}_levels"][1]
        lv = np.full(exp.shape, 0xAA, np.uint8)
        slot("av1_txb_init_levels", None, VP, c_int, c_int, VP)(co.ctypes.data, w, h, lv.ctypes.data + 2 * (w + 4))
        assert np.array_equal(lv, exp), (w, h)
    for (bw, bh) in ((4, 4), (16, 8), (32, 32), (64, 16)):
        a = rng.integers(0, 256, size=200, dtype=np.uint8); l = rng.integers(0, 256, size=200, dtype=np.uint8)
        for nm, mode in (("eb_smooth_v_predictor", 4), ("eb_smooth_h_predictor", 5)):
            d1 = np.zeros((bh, 80), np.uint8); d2 = d1.copy()
            slot(nm, None, VP, S, c_int, c_int, VP, VP)(d1.ctypes.data, 80, bw, bh, a.ctypes.data + 16, l.ctypes.data + 16)
            O.svt_oracle_intra_pred(mode, ptr(d2), S(80), bw, bh, VP(a.ctypes.data + 16), VP(l.ctypes.data + 16))
            assert np.array_equal(d1, d2), (nm, bw, bh)
    L = dsp.lib
    L.svt_hip_combined_averaging_sad.restype = ctypes.c_uint32
    for (w, h) in ((8, 8), (16, 16), (32, 32), (64, 64), (24, 16), (4, 4)):
        s = rng.integers(0, 256, (h, w + 5), dtype=np.uint8); r1 = rng.integers(0, 256, (h, w + 9), dtype=np.uint8)
        r2 = rng.integers(0, 256, (h, w + 1), dtype=np.uint8)
        got = L.svt_hip_combined_averaging_sad(ptr(s), U(w + 5), ptr(r1), U(w + 9), ptr(r2), U(w + 1), U(h), U(w))
        assert got == O.svt_oracle_sad_avg(ptr(s), U(w + 5), ptr(r1), U(w + 9), ptr(r2), U(w + 1), U(h), U(w)), (w, h)
    for (w, h) in ((4, 4), (8, 16), (32, 32), (64, 64), (12, 6)):
        a = rng.integers(0, 1024, (h, w + 3)).astype(np.uint16); b = rng.integers(0, 1024, (h, w + 6)).astype(np.uint16)
        r = np.zeros((h, w + 2), np.int16)
        L.svt_hip_residual_kernel16bit(ptr(a), U(w + 3), ptr(b), U(w + 6), ptr(r), U(w + 2), U(w), U(h))
        assert np.array_equal(r[:, :w], (a[:, :w].astype(np.int32) - b[:, :w].astype(np.int32)).astype(np.int16)), (w, h)


def test_batched_sad_variants_and_residual16(dsp):
    """(every device tensor is held in a variable for the duration of the call: a temporary's memory goes back to torch's
    caching allocator as soon as .data_ptr() has been taken)"""
    O = svtlibs.oracle()
    L = dsp.lib
    rng = np.random.default_rng(34)
    n = 301
    for (w, h) in ((16, 16), (8, 8), (64, 64), (32, 8), (4, 16), (128, 64)):
        pw, ph = 400, 300
        src = rng.integers(0, 256, (ph, pw), dtype=np.uint8); ref = rng.integers(0, 256, (ph, pw), dtype=np.uint8)
        so = (rng.integers(0, ph - h, n) * pw + rng.integers(0, pw - w, n)).astype(np.int32)
        ro = (rng.integers(0, ph - h, (n, 4)) * pw + rng.integers(0, pw - w, (n, 4))).astype(np.int32)
        assert int(so.max()) + (h - 1) * pw + w <= pw * ph and int(ro.max()) + (h - 1) * pw + w <= pw * ph
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_src, d_ref, d_so, d_ro0, d_ro = dev(src), dev(ref), dev(so), dev(ro[:, 0]), dev(ro)
        rc = L.svt_hip_sad_planes_batch(d_src.data_ptr(), pw, d_so.data_ptr(), d_ref.data_ptr(), pw, d_ro0.data_ptr(), w, h, out.data_ptr(), n, None)
        assert rc == 0, L.svt_hip_last_error()
        exp = np.array([[O.svt_oracle_sad(VP(src.ctypes.data + int(so[i])), U(pw), VP(ref.ctypes.data + int(ro[i, k])), U(pw), U(h), U(w))
                         for k in range(4)] for i in range(n)], np.uint32)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), exp[:, 0]), (w, h)
        out4 = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
        rc = L.svt_hip_sad_x4d_batch(d_src.data_ptr(), pw, 0, d_so.data_ptr(), d_ref.data_ptr(), pw, d_ro.data_ptr(), w, h, out4.data_ptr(), n, None)
        assert rc == 0, L.svt_hip_last_error()
        assert np.array_equal(out4.cpu().numpy().view(np.uint32), exp), (w, h, "x4d")
    for (w, h) in ((8, 8), (16, 16), (32, 32), (64, 64), (12, 4)):
        s = rng.integers(0, 256, (n, h, w), dtype=np.uint8); r1 = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        r2 = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        r1[0] = 255; r2[0] = 254; s[0] = 0
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_s, d_r1, d_r2 = dev(s), dev(r1), dev(r2)
        assert L.svt_hip_sad_avg_batch(d_s.data_ptr(), w, w * h, d_r1.data_ptr(), w, w * h, d_r2.data_ptr(), w, w * h, w, h, out.data_ptr(), n, None) == 0
        exp = np.array([O.svt_oracle_sad_avg(ptr(s[i]), U(w), ptr(r1[i]), U(w), ptr(r2[i]), U(w), U(h), U(w)) for i in range(n)], np.uint32)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), exp), (w, h, "avg")
        a = rng.integers(0, 1024, (n, h, w)).astype(np.uint16); b = rng.integers(0, 1024, (n, h, w)).astype(np.uint16)
        res = torch.zeros((n, h, w), dtype=torch.int16, device="cuda")
        d_a, d_b = dev(a.view(np.int16)), dev(b.view(np.int16))
        assert L.svt_hip_residual16_batch(d_a.data_ptr(), w, w * h, d_b.data_ptr(), w, w * h, res.data_ptr(), w, w * h, w, h, n, None) == 0
        assert np.array_equal(res.cpu().numpy(), (a.astype(np.int32) - b.astype(np.int32)).astype(np.int16)), (w, h, "res16")


def test_picture_full_distortion32_c_and_avx2_flavours(dsp):
    """picture_full_distortion32_bits (luma): size clamp, per-block cbf by count_non_zero_coeffs, and the AVX2 kernel's
    carry-less residual sum; fixture = the reference's own outputs (tests/golden/pins.npz), random batch = the oracle"""
    O = svtlibs.oracle()
    L = dsp.lib
    pins = np.load(os.path.join(G, "pins.npz"))
    for (w, h) in ((4, 4), (8, 8), (16, 16), (32, 32), (64, 64)):
        k = min(w, 32)
        a = np.ascontiguousarray(pins[f"dist_{w}x{h}_a"][:, :k, :k]); b = np.ascontiguousarray(pins[f"dist_{w}x{h}_b"][:, :k, :k])
        d_a, d_b = dev(a), dev(b)
        for nz in (1, 0):
            out = torch.zeros((4, 2), dtype=torch.int64, device="cuda")
            cnt = dev(np.full(4, nz, np.int32))
            assert L.svt_hip_picture_full_distortion32_batch(d_a.data_ptr(), k * k, d_b.data_ptr(), k * k, w, h, cnt.data_ptr(), 0,
                                                             out.data_ptr(), 4, None) == 0, L.svt_hip_last_error()
            got = out.cpu().numpy().view(np.uint64)
            for i in range(4):
                assert np.array_equal(got[i], pins[f"pdist_{w}_{i}_{nz}"]), (w, i, nz)
    rng = np.random.default_rng(35)
    n = 97
    for (w, h) in ((4, 4), (8, 16), (32, 32), (64, 64), (16, 64)):
        kw, kh = min(w, 32), min(h, 32)
        for mag in (12, 17, 22):
            a = rng.integers(-(1 << mag), 1 << mag, (n, kh, kw)).astype(np.int32); b = rng.integers(-(1 << mag), 1 << mag, (n, kh, kw)).astype(np.int32)
            cnt = rng.integers(0, 3, n).astype(np.int32)
            d_a, d_b, d_cnt = dev(a), dev(b), dev(cnt)
            for flavour in (0, 1):
                out = torch.zeros((n, 2), dtype=torch.int64, device="cuda")
                assert L.svt_hip_picture_full_distortion32_batch(d_a.data_ptr(), kw * kh, d_b.data_ptr(), kw * kh, w, h, d_cnt.data_ptr(), flavour,
                                                                 out.data_ptr(), n, None) == 0
                got = out.cpu().numpy().view(np.uint64)
                for i in range(n):
                    e = np.zeros(2, np.uint64)
                    f = O.svt_oracle_full_distortion32_avx2 if flavour else O.svt_oracle_full_distortion32
                    f(ptr(a[i]), U(kw), ptr(b[i]), U(kw), ptr(e), U(kw), U(kh))
                    if cnt[i] == 0:
                        e[0] = e[1]
                    assert np.array_equal(got[i], e), (w, h, mag, flavour, i)


def test_c_host_calls_through_its_own_dispatch_block(tmp_path):
    from test_abi_and_host import build_c_caller
    exe = build_c_caller(tmp_path)
    pr = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert pr.returncode == 0, (pr.returncode, pr.stdout, pr.stderr)
    assert "all results equal" in pr.stdout
