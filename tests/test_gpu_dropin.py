"""GPU: the DROP-IN entry points (reference signatures, host pointers, one block per call)
called exactly as the reference's call sites / unit tests call the RTCD slots, checked against
the oracle; plus re-entrancy from several host threads (the encoder calls these kernels
concurrently from its ME / EncDec pthreads)."""
import ctypes
import threading

import numpy as np
import pytest

import svtlibs
from svtlibs import TX_H, TX_SIZES, TX_W, ptr, txfm_allowed

pytestmark = pytest.mark.gpu
c_int = ctypes.c_int
SQ = {0, 1, 2, 3, 4}
NO_EOB = {5, 6, 13, 14}


@pytest.fixture(scope="module")
def lib(pkg, dsp):
    L = dsp.lib
    L.svt_hip_nxm_sad_kernel.restype = ctypes.c_uint32
    L.svt_hip_spatial_full_distortion_kernel.restype = ctypes.c_uint64
    return L


@pytest.mark.parametrize("tx_size", [0, 1, 2, 3, 4, 5, 10, 12, 14, 17])
def test_dropin_fwd_and_inv(lib, tx_size):
    O = svtlibs.oracle()
    w, h = TX_W[tx_size], TX_H[tx_size]
    kw, kh = min(w, 32), min(h, 32)
    rng = np.random.default_rng(tx_size)
    fwd = getattr(lib, f"svt_hip_av1_fwd_txfm2d_{w}x{h}")
    inv = getattr(lib, f"svt_hip_av1_inv_txfm2d_add_{w}x{h}")
    for tx_type in (0, 3, 9, 15):
        if not txfm_allowed(tx_size, tx_type):
            continue
        for bd in (8, 10):
            x = np.zeros((h, 72), np.int16)                     # stride 72 != width
            x[:, :w] = rng.integers(-(1 << bd) + 1, 1 << bd, size=(h, w))
            out = np.zeros(w * h, np.int32); ref = np.zeros(w * h, np.int32)
            fwd(ptr(x), ptr(out), ctypes.c_uint32(72), ctypes.c_uint8(tx_type), ctypes.c_uint8(bd))
            O.svt_oracle_fwd_txfm2d(ptr(x), ptr(ref), ctypes.c_uint32(72), tx_type, tx_size, bd)
            assert np.array_equal(out, ref), (TX_SIZES[tx_size], tx_type, bd)
            O.svt_oracle_fwd_txfm2d_pack64(ptr(ref), tx_size)
            co = np.ascontiguousarray(ref[:kw * kh])
            d1 = rng.integers(0, 1 << bd, size=(h, 80)).astype(np.uint16); d2 = d1.copy()
            if tx_size in SQ:
                inv(ptr(co), ptr(d1), c_int(80), ctypes.c_uint8(tx_type), c_int(bd))
            elif tx_size in NO_EOB:
                inv(ptr(co), ptr(d1), c_int(80), ctypes.c_uint8(tx_type), ctypes.c_uint8(tx_size), c_int(bd))
            else:
                inv(ptr(co), ptr(d1), c_int(80), ctypes.c_uint8(tx_type), ctypes.c_uint8(tx_size), c_int(kw * kh), c_int(bd))
            O.svt_oracle_inv_txfm2d_add(ptr(co), ptr(d2), c_int(80), tx_type, tx_size, bd)
            assert np.array_equal(d1, d2), (TX_SIZES[tx_size], tx_type, bd)


def test_dropin_lowbd_inv_txfm_add(lib):
    O = svtlibs.oracle()
    rng = np.random.default_rng(1)

    class TxfmParam(ctypes.Structure):     # EbDefinitions.h:764-776
        _fields_ = [("tx_type", ctypes.c_uint8), ("tx_size", ctypes.c_uint8), ("lossless", ctypes.c_int32), ("bd", ctypes.c_int32),
                    ("is_hbd", ctypes.c_int32), ("tx_set_type", ctypes.c_uint8), ("eob", ctypes.c_int32)]
    for tx_size in (1, 3, 9):
        w, h = TX_W[tx_size], TX_H[tx_size]
        co = rng.integers(-4000, 4001, size=w * h).astype(np.int32)
        d1 = rng.integers(0, 256, size=(h, 96)).astype(np.uint8); d2 = d1.copy()
        p = TxfmParam(0, tx_size, 0, 8, 0, 0, w * h)
        lib.svt_hip_av1_inv_txfm_add(ptr(co), ptr(d1), c_int(96), ctypes.byref(p))
        O.svt_oracle_inv_txfm2d_add_u8(ptr(co), ptr(d2), c_int(96), 0, tx_size)
        assert np.array_equal(d1, d2)


@pytest.mark.parametrize("name,ls,tx_size", [("svt_hip_aom_highbd_quantize_b", 0, 2), ("svt_hip_aom_highbd_quantize_b_32x32", 1, 3),
                                             ("svt_hip_aom_highbd_quantize_b_64x64", 2, 4), ("svt_hip_aom_quantize_b_32x32", 1, 3)])
def test_dropin_quantize(lib, name, ls, tx_size):
    O = svtlibs.oracle()
    rng = np.random.default_rng(ls)
    qt = svtlibs.quant_tables(8)
    scan, iscan = svtlibs.scan_tables(tx_size, 0)
    n = len(scan)
    for q in (0, 77, 255):
        tabs = [np.ascontiguousarray(qt[k][q]) for k in ("zbin", "round", "quant", "quant_shift", "dequant")]
        co = rng.integers(-(1 << 15), (1 << 15) + 1, size=n).astype(np.int32)
        a = [np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(1, np.uint16)]
        b = [np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(1, np.uint16)]
        getattr(lib, name)(ptr(co), ctypes.c_ssize_t(n), c_int(0), ptr(tabs[0]), ptr(tabs[1]), ptr(tabs[2]), ptr(tabs[3]),
                           ptr(a[0]), ptr(a[1]), ptr(tabs[4]), ptr(a[2]), ptr(scan), ptr(iscan))
        O.svt_oracle_quantize_b(ptr(co), ctypes.c_ssize_t(n), 0, ptr(tabs[0]), ptr(tabs[1]), ptr(tabs[2]), ptr(tabs[3]),
                                ptr(b[0]), ptr(b[1]), ptr(tabs[4]), ptr(b[2]), ptr(scan), ptr(iscan), ls, 0)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), (name, q)


def test_dropin_pixel_kernels(lib):
    O = svtlibs.oracle()
    rng = np.random.default_rng(9)
    for (w, h) in ((8, 8), (16, 16), (32, 32), (64, 64), (24, 32)):
        a = rng.integers(0, 256, size=(h, 100), dtype=np.uint8)
        b = rng.integers(0, 256, size=(h, 90), dtype=np.uint8)
        assert lib.svt_hip_nxm_sad_kernel(ptr(a), 100, ptr(b), 90, h, w) == O.svt_oracle_sad(ptr(a), 100, ptr(b), 90, h, w)
        assert lib.svt_hip_spatial_full_distortion_kernel(ptr(a), 100, ptr(b), 90, w, h) == O.svt_oracle_sse(ptr(a), 100, ptr(b), 90, w, h)
        r1 = np.zeros((h, 70), np.int16); r2 = r1.copy()
        lib.svt_hip_residual_kernel(ptr(a), 100, ptr(b), 90, ptr(r1), 70, w, h)
        O.svt_oracle_residual(ptr(a), 100, ptr(b), 90, ptr(r2), 70, w, h)
        assert np.array_equal(r1, r2)
    # coefficient-domain distortion
    c = rng.integers(-(1 << 17), 1 << 17, size=(32, 40)).astype(np.int32)
    r = rng.integers(-(1 << 17), 1 << 17, size=(32, 36)).astype(np.int32)
    o1 = np.zeros(2, np.uint64); o2 = np.zeros(2, np.uint64)
    lib.svt_hip_full_distortion_kernel32_bits(ptr(c), 40, ptr(r), 36, ptr(o1), 32, 32)
    O.svt_oracle_full_distortion32(ptr(c), 40, ptr(r), 36, ptr(o2), 32, 32)
    assert np.array_equal(o1, o2)
    lib.svt_hip_full_distortion_kernel_cbf_zero32_bits(ptr(c), 40, ptr(r), 36, ptr(o1), 32, 32)
    assert int(o1[0]) == int(o2[1]) and int(o1[1]) == int(o2[1])


def test_dropin_sad_loop_incl_line_skipping(lib):
    """sad_loop_kernel as HME calls it: ref_stride doubled, src_stride_raw = true stride (EbMotionEstimation.c:5799-5875)"""
    O = svtlibs.oracle()
    rng = np.random.default_rng(4)
    for (w, h, sw, sh, skip) in ((16, 16, 16, 9, 1), (16, 8, 12, 6, 2), (32, 32, 8, 8, 1), (64, 32, 5, 4, 2)):
        stride = 200
        rows = sh + h * skip + 2
        src = rng.integers(0, 256, size=(h * skip, 64 + 8), dtype=np.uint8)
        ref = rng.integers(0, 256, size=(rows, stride), dtype=np.uint8)
        b1 = np.zeros(1, np.uint64); x1 = np.full(1, -7, np.int16); y1 = np.full(1, -7, np.int16)
        b2 = np.zeros(1, np.uint64); x2 = np.full(1, -7, np.int16); y2 = np.full(1, -7, np.int16)
        args = (ptr(src), ctypes.c_uint32(72 * skip), ptr(ref), ctypes.c_uint32(stride * skip), ctypes.c_uint32(h), ctypes.c_uint32(w))
        lib.svt_hip_sad_loop_kernel(*args, ptr(b1), ptr(x1), ptr(y1), ctypes.c_uint32(stride), ctypes.c_int16(sw), ctypes.c_int16(sh))
        O.svt_oracle_sad_loop(*args, ptr(b2), ptr(x2), ptr(y2), ctypes.c_uint32(stride), ctypes.c_int16(sw), ctypes.c_int16(sh))
        assert (int(b1[0]), int(x1[0]), int(y1[0])) == (int(b2[0]), int(x2[0]), int(y2[0])), (w, h, sw, sh, skip)


def test_dropin_intra(lib):
    O = svtlibs.oracle()
    rng = np.random.default_rng(6)
    S = ctypes.c_ssize_t
    for (bw, bh) in ((4, 4), (8, 16), (32, 32), (64, 16)):
        a = rng.integers(0, 256, size=400, dtype=np.uint8); l = rng.integers(0, 256, size=400, dtype=np.uint8)
        pa = ctypes.c_void_p(a.ctypes.data + 16); pl = ctypes.c_void_p(l.ctypes.data + 16)
        for mode in range(10):
            d1 = np.zeros((bh, 80), np.uint8); d2 = d1.copy()
            lib.svt_hip_intra_predictor(mode, bw, bh, ptr(d1), S(80), pa, pl)
            O.svt_oracle_intra_pred(mode, ptr(d2), S(80), bw, bh, pa, pl)
            assert np.array_equal(d1, d2), (bw, bh, mode)
        d1 = np.zeros((bh, 80), np.uint8); d2 = d1.copy()
        lib.svt_hip_av1_dr_prediction_z1(ptr(d1), S(80), bw, bh, pa, pl, 0, 151, 1)
        O.svt_oracle_dr_prediction(1, ptr(d2), S(80), bw, bh, pa, pl, 0, 0, 151, 1)
        assert np.array_equal(d1, d2)
        lib.svt_hip_av1_dr_prediction_z2(ptr(d1), S(80), bw, bh, pa, pl, 0, 0, 64, 64)
        O.svt_oracle_dr_prediction(2, ptr(d2), S(80), bw, bh, pa, pl, 0, 0, 64, 64)
        assert np.array_equal(d1, d2)
        lib.svt_hip_av1_dr_prediction_z3(ptr(d1), S(80), bw, bh, pa, pl, 0, 1, 27)
        O.svt_oracle_dr_prediction(3, ptr(d2), S(80), bw, bh, pa, pl, 0, 0, 1, 27)
        assert np.array_equal(d1, d2)
        ah = rng.integers(0, 1024, size=400).astype(np.uint16); lh = rng.integers(0, 1024, size=400).astype(np.uint16)
        pah = ctypes.c_void_p(ah.ctypes.data + 32); plh = ctypes.c_void_p(lh.ctypes.data + 32)
        h1 = np.zeros((bh, 80), np.uint16); h2 = h1.copy()
        lib.svt_hip_highbd_intra_predictor(6, bw, bh, ptr(h1), S(80), pah, plh, 10)
        O.svt_oracle_intra_pred_hbd(6, ptr(h2), S(80), bw, bh, pah, plh, 10)
        assert np.array_equal(h1, h2)


def test_rtcd_override_fills_the_table(lib):
    class Table(ctypes.Structure):
        _fields_ = [("fwd", ctypes.POINTER(ctypes.c_void_p) * 19), ("inv", ctypes.POINTER(ctypes.c_void_p) * 19),
                    ("inv_add", ctypes.POINTER(ctypes.c_void_p))] + [(n, ctypes.POINTER(ctypes.c_void_p)) for n in
                    ("q", "q32", "q64", "hq", "hq32", "hq64", "residual")]
    slots = (ctypes.c_void_p * 8)()
    t = Table()
    t.fwd[3] = ctypes.cast(ctypes.byref(slots, 0), ctypes.POINTER(ctypes.c_void_p))
    t.inv[3] = ctypes.cast(ctypes.byref(slots, 8), ctypes.POINTER(ctypes.c_void_p))
    t.hq32 = ctypes.cast(ctypes.byref(slots, 16), ctypes.POINTER(ctypes.c_void_p))
    t.residual = ctypes.cast(ctypes.byref(slots, 24), ctypes.POINTER(ctypes.c_void_p))
    assert lib.svt_hip_rtcd_override(ctypes.byref(t)) == 0
    want = [ctypes.cast(getattr(lib, n), ctypes.c_void_p).value for n in
            ("svt_hip_av1_fwd_txfm2d_32x32", "svt_hip_av1_inv_txfm2d_add_32x32", "svt_hip_aom_highbd_quantize_b_32x32", "svt_hip_residual_kernel")]
    assert [slots[i] for i in range(4)] == want and slots[4] is None


def test_dropins_are_reentrant_from_threads(lib):
    """8 host threads hammer different drop-ins concurrently (per-thread stream + staging)"""
    O = svtlibs.oracle()
    errors = []

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            for it in range(25):
                x = rng.integers(-255, 256, size=(32, 32)).astype(np.int16)
                out = np.zeros(1024, np.int32); ref = np.zeros(1024, np.int32)
                lib.svt_hip_av1_fwd_txfm2d_32x32(ptr(x), ptr(out), ctypes.c_uint32(32), ctypes.c_uint8(0), ctypes.c_uint8(8))
                O.svt_oracle_fwd_txfm2d(ptr(x), ptr(ref), ctypes.c_uint32(32), 0, 3, 8)
                if not np.array_equal(out, ref):
                    errors.append(("fwd", seed, it))
                a = rng.integers(0, 256, size=(16, 16), dtype=np.uint8); b = rng.integers(0, 256, size=(16, 16), dtype=np.uint8)
                if lib.svt_hip_nxm_sad_kernel(ptr(a), 16, ptr(b), 16, 16, 16) != O.svt_oracle_sad(ptr(a), 16, ptr(b), 16, 16, 16):
                    errors.append(("sad", seed, it))
        except Exception as e:      # noqa: BLE001
            errors.append(("exc", seed, repr(e)))
    threads = [threading.Thread(target=worker, args=(s,)) for s in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
