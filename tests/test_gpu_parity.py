"""GPU parity tests: the HIP path, called through the C ABI
(include/svt_hip_dsp.h), against the CPU oracle on the same seeded inputs.
Bit-exact everywhere (all outputs on this path are integers).  Mirrors the
reference's own unit tests: FwdTxfm2dAsmTest.cc:74-166 (all sizes x allowed types
x bd{8,10}, uniform +-(2^bd-1) inputs), InvTxfm2dAsmTest.cc:166-510 (coefficients
from the forward transform), QuantAsmTest.cc:84-308 (q sweep, +-2^(7+bd))."""
import ctypes

import numpy as np
import pytest
import torch

import svtlibs
from svtlibs import TX_H, TX_SIZES, TX_TYPES, TX_W, ptr, txfm_allowed

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def oracle_chain(src, pred, tx_size, tx_type, qrow):
    O = svtlibs.oracle()
    n = src.shape[0]
    h, w = src.shape[1:]
    nc = min(w, 32) * min(h, 32)
    co = np.zeros((n, w * h), np.int32); q = np.zeros((n, w * h), np.int32); dq = np.zeros((n, w * h), np.int32)
    eob = np.zeros(n, np.uint16); sad = np.zeros(n, np.uint32)
    for i in range(n):
        O.svt_oracle_fwd_quant_sad(ptr(src[i]), w, ptr(pred[i]), w, tx_size, tx_type, ptr(qrow["zbin"]),
                                   ptr(qrow["round"]), ptr(qrow["quant"]), ptr(qrow["quant_shift"]),
                                   ptr(qrow["dequant"]), ptr(co[i]), ptr(q[i]), ptr(dq[i]),
                                   ptr(eob[i:i + 1]), ptr(sad[i:i + 1]))
    return co[:, :nc], q[:, :nc], dq[:, :nc], eob, sad


def make_pixels(rng, n, h, w, kind):
    src = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
    pred = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
    if kind == "extreme":      # residual +-255 everywhere / checker / flat
        src[0] = 255; pred[0] = 0
        src[1] = 0; pred[1] = 255
        src[2] = pred[2]                      # zero residual -> eob 0
        yy, xx = np.mgrid[0:h, 0:w]
        src[3] = np.where((yy + xx) & 1, 255, 0); pred[3] = 255 - src[3]
        src[4] = 128; pred[4] = 127          # DC only
    if kind == "smooth":
        pred = np.clip(src.astype(int) + rng.integers(-3, 4, size=src.shape), 0, 255).astype(np.uint8)
    return src, pred


@pytest.mark.parametrize("qindex", [0, 1, 25, 100, 180, 255])
@pytest.mark.parametrize("kind", ["random", "extreme", "smooth"])
def test_fused_fwd_quant_sad_32x32(dsp, pkg, qindex, kind):
    rng = np.random.default_rng(13596 + qindex)
    n = 37                                   # odd: exercises the half-empty last wave
    src, pred = make_pixels(rng, n, 32, 32, kind)
    qt = svtlibs.quant_tables(8)
    qrow = {k: v[qindex].copy() for k, v in qt.items()}
    _, iscan = svtlibs.scan_tables(3, 0)
    co, q, dq, eob, sad = dsp.fwd_quant_sad(dev(src), dev(pred), 3, 0, qrow, dev(iscan))
    torch.cuda.synchronize()
    rco, rq, rdq, reob, rsad = oracle_chain(src, pred, 3, 0, qrow)
    assert np.array_equal(co.cpu().numpy(), rco)
    assert np.array_equal(q.cpu().numpy(), rq)
    assert np.array_equal(dq.cpu().numpy(), rdq)
    assert np.array_equal(eob.cpu().numpy().view(np.uint16), reob)
    assert np.array_equal(sad.cpu().numpy().view(np.uint32), rsad)


def test_fused_empty_and_single(dsp, pkg):
    qt = svtlibs.quant_tables(8)
    qrow = {k: v[100].copy() for k, v in qt.items()}
    _, iscan = svtlibs.scan_tables(3, 0)
    e = torch.empty((0, 32, 32), dtype=torch.uint8, device=DEV)
    dsp.fwd_quant_sad(e, e, 3, 0, qrow, dev(iscan))          # n = 0 is a no-op
    rng = np.random.default_rng(1)
    src, pred = make_pixels(rng, 1, 32, 32, "random")
    co, q, dq, eob, sad = dsp.fwd_quant_sad(dev(src), dev(pred), 3, 0, qrow, dev(iscan))
    rco, rq, rdq, reob, rsad = oracle_chain(src, pred, 3, 0, qrow)
    assert np.array_equal(co.cpu().numpy(), rco) and np.array_equal(q.cpu().numpy(), rq)
    assert int(eob.cpu().numpy().view(np.uint16)[0]) == int(reob[0])


def test_fused_invalid_combination_is_loud(dsp, pkg):
    """TX_64X64 only exists with DCT_DCT (is_txfm_allowed): anything else is an error, never a guess"""
    qt = svtlibs.quant_tables(8)
    qrow = {k: v[100].copy() for k, v in qt.items()}
    _, iscan = svtlibs.scan_tables(4, 0)
    x = torch.zeros((4, 64, 64), dtype=torch.uint8, device=DEV)
    with pytest.raises(pkg.SvtHipError):
        dsp.fwd_quant_sad(x, x, 4, 1, qrow, dev(iscan))


@pytest.mark.parametrize("tx_size", range(19))
def test_fused_chain_every_size_and_type(dsp, tx_size):
    """the generic fused kernel behind svt_hip_fwd_quant_sad_batch (dense 8-bit blocks)"""
    w, h = TX_W[tx_size], TX_H[tx_size]
    rng = np.random.default_rng(4000 + tx_size)
    qt = svtlibs.quant_tables(8)
    n = 11
    for tx_type in range(16):
        if not txfm_allowed(tx_size, tx_type):
            continue
        for qindex in (0, 60, 255):
            qrow = {k: v[qindex].copy() for k, v in qt.items()}
            _, iscan = svtlibs.scan_tables(tx_size, tx_type)
            src, pred = make_pixels(rng, n, h, w, "extreme" if qindex == 60 else "random")
            co, q, dq, eob, sad = dsp.fwd_quant_sad(dev(src), dev(pred), tx_size, tx_type, qrow, dev(iscan))
            rco, rq, rdq, reob, rsad = oracle_chain(src, pred, tx_size, tx_type, qrow)
            tag = (TX_SIZES[tx_size], TX_TYPES[tx_type], qindex)
            assert np.array_equal(co.cpu().numpy(), rco), tag
            assert np.array_equal(q.cpu().numpy(), rq), tag
            assert np.array_equal(dq.cpu().numpy(), rdq), tag
            assert np.array_equal(eob.cpu().numpy().view(np.uint16), reob), tag
            assert np.array_equal(sad.cpu().numpy().view(np.uint32), rsad), tag


def _plane_case(dsp, rng, bd, tx_size, tx_type, pw, ph, qindex):
    O = svtlibs.oracle()
    w, h = TX_W[tx_size], TX_H[tx_size]
    dt = np.uint8 if bd == 8 else np.uint16
    src = rng.integers(0, 1 << bd, size=(ph, pw)).astype(dt)
    pred = rng.integers(0, 1 << bd, size=(ph, pw + 16)).astype(dt)        # different stride on purpose
    xs = np.arange(0, pw - w + 1, w); ys = np.arange(0, ph - h + 1, h)
    xy = np.array([(y << 16) | x for y in ys for x in xs], np.uint32)
    qt = svtlibs.quant_tables(bd)
    qrow = {k: v[qindex].copy() for k, v in qt.items()}
    _, iscan = svtlibs.scan_tables(tx_size, tx_type)
    as_dev = (lambda a: dev(a)) if bd == 8 else (lambda a: dev(a.view(np.int16)))
    co, q, dq, eob, sad, en = dsp.fwd_quant_planes(as_dev(src), pw, as_dev(pred), pw + 16, dev(xy.view(np.int32)), tx_size,
                                                   tx_type, qrow, dev(iscan), bd=bd, want_sad=(bd == 8), want_energy=True)
    co = co.cpu().numpy(); q = q.cpu().numpy(); dq = dq.cpu().numpy()
    eob = eob.cpu().numpy().view(np.uint16); en = en.cpu().numpy().view(np.uint64)
    nc = co.shape[1]
    for i, v in enumerate(xy):
        y, x = int(v >> 16), int(v & 0xffff)
        rc = np.zeros(1024, np.int32); rq = np.zeros(1024, np.int32); rdq = np.zeros(1024, np.int32)
        reob = np.zeros(1, np.uint16); rsad = np.zeros(1, np.uint32); ren = np.zeros(1, np.uint64)
        sp = ctypes.c_void_p(src.ctypes.data + (y * pw + x) * src.itemsize)
        pp = ctypes.c_void_p(pred.ctypes.data + (y * (pw + 16) + x) * pred.itemsize)
        O.svt_oracle_fwd_quant_planes(sp, pw, pp, pw + 16, int(bd != 8), bd, tx_size, tx_type, ptr(qrow["zbin"]), ptr(qrow["round"]),
                                      ptr(qrow["quant"]), ptr(qrow["quant_shift"]), ptr(qrow["dequant"]), ptr(rc), ptr(rq), ptr(rdq),
                                      ptr(reob), ptr(rsad), ptr(ren))
        assert np.array_equal(co[i], rc[:nc]) and np.array_equal(q[i], rq[:nc]) and np.array_equal(dq[i], rdq[:nc]), (i, bd)
        assert eob[i] == reob[0] and en[i] == ren[0]
        if bd == 8:
            assert int(sad[i].item()) & 0xffffffff == int(rsad[0])
    return src, pred, xy, qrow, dq


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("tx_size,tx_type", [(0, 0), (1, 3), (2, 9), (3, 0), (4, 0), (5, 6), (10, 0), (11, 0), (17, 0), (14, 12)])
def test_fwd_quant_planes_xy_addressing(dsp, bd, tx_size, tx_type):
    rng = np.random.default_rng(tx_size * 3 + bd)
    _plane_case(dsp, rng, bd, tx_size, tx_type, 192, 128, 80)


def test_planes_sad_is_refused_for_16bit(dsp, pkg):
    qt = svtlibs.quant_tables(10)
    qrow = {k: v[100].copy() for k, v in qt.items()}
    _, iscan = svtlibs.scan_tables(1, 0)
    p = torch.zeros((64, 64), dtype=torch.int16, device=DEV)
    xy = torch.zeros(4, dtype=torch.int32, device=DEV)
    with pytest.raises(pkg.SvtHipError):
        dsp.fwd_quant_planes(p, 64, p, 64, xy, 1, 0, qrow, dev(iscan), bd=10, want_sad=True)


@pytest.mark.parametrize("tx_size", range(19))
@pytest.mark.parametrize("bd", [8, 10])
def test_fwd_txfm2d_all_types(dsp, tx_size, bd):
    O = svtlibs.oracle()
    w, h = TX_W[tx_size], TX_H[tx_size]
    rng = np.random.default_rng(100 * tx_size + bd)
    n = 21
    for tx_type in range(16):
        if not txfm_allowed(tx_size, tx_type):
            continue
        x = rng.integers(-(1 << bd) + 1, 1 << bd, size=(n, h, w)).astype(np.int16)
        x[0] = (1 << bd) - 1; x[1] = -((1 << bd) - 1); x[2] = 0
        x[3] = 0; x[3, 0, 0] = (1 << bd) - 1
        out = dsp.fwd_txfm2d(dev(x), tx_size, tx_type, bd).cpu().numpy()
        ref = np.zeros((n, w * h), np.int32)
        for i in range(n):
            O.svt_oracle_fwd_txfm2d(ptr(x[i]), ptr(ref[i]), ctypes.c_uint32(w), tx_type, tx_size, bd)
        assert np.array_equal(out, ref), f"{TX_SIZES[tx_size]} {TX_TYPES[tx_type]} bd{bd}"


@pytest.mark.parametrize("tx_size", [4, 11, 12, 17, 18])
def test_pack64(dsp, tx_size):
    O = svtlibs.oracle()
    w, h = TX_W[tx_size], TX_H[tx_size]
    rng = np.random.default_rng(tx_size)
    n = 5
    c = rng.integers(-(1 << 17), 1 << 17, size=(n, w * h)).astype(np.int32)
    d = dev(c)
    energy = dsp.pack64(d, tx_size).cpu().numpy()
    ref = c.copy()
    for i in range(n):
        e = O.svt_oracle_fwd_txfm2d_pack64(ptr(ref[i]), tx_size)
        assert int(energy[i]) == int(e)
    assert np.array_equal(d.cpu().numpy(), ref)


@pytest.mark.parametrize("tx_size", range(19))
@pytest.mark.parametrize("bd", [8, 10])
def test_inv_txfm2d_add_all_types(dsp, tx_size, bd):
    O = svtlibs.oracle()
    w, h = TX_W[tx_size], TX_H[tx_size]
    kw, kh = min(w, 32), min(h, 32)
    rng = np.random.default_rng(7 * tx_size + bd)
    n = 13
    for tx_type in range(16):
        if not txfm_allowed(tx_size, tx_type):
            continue
        co = np.zeros((n, kw * kh), np.int32)
        for i in range(n):
            if i < 9:    # coefficients of a forward transform (InvTxfm2dAsmTest.cc:398-462)
                x = rng.integers(-(1 << bd) + 1, 1 << bd, size=(h, w)).astype(np.int16)
                full = np.zeros(w * h, np.int32)
                O.svt_oracle_fwd_txfm2d(ptr(x), ptr(full), ctypes.c_uint32(w), tx_type, tx_size, bd)
                O.svt_oracle_fwd_txfm2d_pack64(ptr(full), tx_size)
                co[i] = full[:kw * kh]
                if i >= 5:   # eob truncation: clear the tail in raster order
                    co[i, rng.integers(1, kw * kh):] = 0
            else:        # out-of-range garbage exercises every clamp
                co[i] = rng.integers(-(1 << 20), 1 << 20, size=kw * kh)
        dst = rng.integers(0, 1 << bd, size=(n, h, w)).astype(np.uint16)
        ref = dst.copy()
        for i in range(n):
            O.svt_oracle_inv_txfm2d_add(ptr(co[i]), ptr(ref[i]), w, tx_type, tx_size, bd)
        d = dev(dst.view(np.int16))
        dsp.inv_txfm2d_add(dev(co), d, tx_size, tx_type, bd)
        assert np.array_equal(d.cpu().numpy().view(np.uint16), ref), f"{TX_SIZES[tx_size]} {TX_TYPES[tx_type]} bd{bd}"
        if bd == 8:      # the 8-bit recon entry (av1_inv_txfm_add)
            dst8 = dst.astype(np.uint8)
            ref8 = dst8.copy()
            for i in range(n):
                O.svt_oracle_inv_txfm2d_add_u8(ptr(co[i]), ptr(ref8[i]), w, tx_type, tx_size)
            d8 = dev(dst8)
            dsp.inv_txfm2d_add(dev(co), d8, tx_size, tx_type, 8)
            assert np.array_equal(d8.cpu().numpy(), ref8)


@pytest.mark.parametrize("tx_size,log_scale", [(0, 0), (1, 0), (2, 0), (3, 1), (4, 2), (9, 1), (7, 0), (5, 0)])
@pytest.mark.parametrize("bd", [8, 10])
def test_quantize_b(dsp, tx_size, log_scale, bd):
    O = svtlibs.oracle()
    qt = svtlibs.quant_tables(bd)
    scan, iscan = svtlibs.scan_tables(tx_size, 0)
    nc = len(scan)
    rng = np.random.default_rng(tx_size * 31 + bd)
    n = 11
    for qindex in (0, 1, 25, 100, 200, 255):
        qrow = {k: v[qindex].copy() for k, v in qt.items()}
        co = rng.integers(-(1 << (7 + bd)), (1 << (7 + bd)) + 1, size=(n, nc)).astype(np.int32)
        co[0] = 0
        co[1] = 0; co[1, 0] = 30000
        co[2, rng.random(nc) < 0.95] = 0
        co[3] = rng.integers(-40, 41, size=nc)
        q, dq, eob = dsp.quantize_b(dev(co), qrow, dev(iscan), log_scale)
        rq = np.zeros_like(co); rdq = np.zeros_like(co); reob = np.zeros(n, np.uint16)
        for i in range(n):
            O.svt_oracle_quantize_b(ptr(co[i]), ctypes.c_ssize_t(nc), 0, ptr(qrow["zbin"]), ptr(qrow["round"]),
                                    ptr(qrow["quant"]), ptr(qrow["quant_shift"]), ptr(rq[i]), ptr(rdq[i]),
                                    ptr(qrow["dequant"]), ptr(reob[i:i + 1]), ptr(scan), ptr(iscan), log_scale, 0)
        assert np.array_equal(q.cpu().numpy(), rq)
        assert np.array_equal(dq.cpu().numpy(), rdq)
        assert np.array_equal(eob.cpu().numpy().view(np.uint16), reob)
    # skip_block: everything zero, eob 0
    q, dq, eob = dsp.quantize_b(dev(co), qrow, dev(iscan), log_scale, skip_block=1)
    assert not q.any() and not dq.any() and not eob.any()


@pytest.mark.parametrize("w,h", [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (24, 16), (48, 32), (8, 32), (64, 16)])
def test_sad_sse_residual(dsp, w, h):
    O = svtlibs.oracle()
    rng = np.random.default_rng(w * 100 + h)
    n = 19
    a = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
    b = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
    a[0] = 255; b[0] = 0
    b[1] = a[1]
    sad = dsp.sad(dev(a), dev(b)).cpu().numpy().view(np.uint32)
    sse = dsp.sse(dev(a), dev(b)).cpu().numpy().view(np.uint64)
    res = dsp.residual(dev(a), dev(b)).cpu().numpy()
    for i in range(n):
        assert int(sad[i]) == O.svt_oracle_sad(ptr(a[i]), w, ptr(b[i]), w, h, w)
        assert int(sse[i]) == O.svt_oracle_sse(ptr(a[i]), w, ptr(b[i]), w, w, h)
    assert np.array_equal(res, a.astype(np.int16) - b.astype(np.int16))


@pytest.mark.parametrize("w,h,sw,sh", [(16, 16, 8, 8), (16, 16, 1, 1), (8, 8, 16, 7), (32, 32, 13, 5), (64, 64, 9, 9),
                                       (4, 4, 8, 8), (24, 24, 8, 4), (48, 48, 5, 3), (16, 8, 32, 20)])
def test_sad_search(dsp, w, h, sw, sh):
    O = svtlibs.oracle()
    rng = np.random.default_rng(w + 7 * sw)
    n = 23
    rw, rh = w + sw - 1, h + sh - 1
    src = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
    ref = rng.integers(0, 256, size=(n, rh, rw), dtype=np.uint8)
    # ties: constant planes (every candidate equal -> (0,0)), exact duplicates at two places
    ref[0] = 7; src[0] = 9
    if sw > 3 and sh > 2:
        ref[1, 2:2 + h, 3:3 + w] = src[1]
        ref[2, 0:h, 1:1 + w] = src[2]; ref[2, 1:1 + h, 0:w] = src[2]
    best, x, y = dsp.sad_search(dev(src), dev(ref), sw, sh)
    best = best.cpu().numpy(); x = x.cpu().numpy(); y = y.cpu().numpy()
    for i in range(n):
        rb = np.zeros(1, np.uint64); rx = np.zeros(1, np.int16); ry = np.zeros(1, np.int16)
        O.svt_oracle_sad_loop(ptr(src[i]), w, ptr(ref[i]), rw, h, w, ptr(rb), ptr(rx), ptr(ry), rw,
                              ctypes.c_int16(sw), ctypes.c_int16(sh))
        assert (int(best[i]), int(x[i]), int(y[i])) == (int(rb[0]), int(rx[0]), int(ry[0])), f"block {i}"


# ---------------------------------------------------------------------------------------
# intra prediction (no reference unit test exists: SURVEY F5 — oracle is pinned by the
# reference's scalar C functions, tests/test_oracle_vs_ref.py + tests/golden/intra.npz)
# ---------------------------------------------------------------------------------------
NB = 16
DR_DERIV = {3: 1023, 6: 547, 9: 372, 14: 273, 17: 215, 20: 178, 23: 151, 26: 132, 29: 116, 32: 102, 36: 90, 39: 80,
            42: 71, 45: 64, 48: 57, 51: 51, 54: 45, 58: 40, 61: 35, 64: 31, 67: 27, 70: 23, 73: 19, 76: 15, 81: 11,
            84: 7, 87: 3}


def _neighbours(rng, n, bd, pitch=304):
    dt = np.uint8 if bd == 8 else np.uint16
    a = rng.integers(0, 1 << bd, size=(n, pitch)).astype(dt)
    l = rng.integers(0, 1 << bd, size=(n, pitch)).astype(dt)
    a[0] = (1 << bd) - 1; l[0] = 0
    a[1] = 0; l[1] = (1 << bd) - 1
    return a, l


def _as_dev(a):
    return dev(a if a.dtype == np.uint8 else a.view(np.int16))


@pytest.mark.parametrize("tx_size", range(19))
@pytest.mark.parametrize("bd", [8, 10])
def test_intra_nondirectional(dsp, tx_size, bd):
    O = svtlibs.oracle()
    bw, bh = TX_W[tx_size], TX_H[tx_size]
    rng = np.random.default_rng(tx_size * 13 + bd)
    n = 9
    a, l = _neighbours(rng, n, bd)
    es = a.itemsize
    for mode in range(10):
        out = dsp.intra_pred(_as_dev(a), _as_dev(l), mode, bw, bh, bd).cpu().numpy()
        out = out if bd == 8 else out.view(np.uint16)
        ref = np.zeros((n, bh, bw), a.dtype)
        for i in range(n):
            pa = ctypes.c_void_p(a[i].ctypes.data + NB * es); pl = ctypes.c_void_p(l[i].ctypes.data + NB * es)
            if bd == 8:
                O.svt_oracle_intra_pred(mode, ptr(ref[i]), ctypes.c_ssize_t(bw), bw, bh, pa, pl)
            else:
                O.svt_oracle_intra_pred_hbd(mode, ptr(ref[i]), ctypes.c_ssize_t(bw), bw, bh, pa, pl, bd)
        assert np.array_equal(out, ref), (TX_SIZES[tx_size], mode, bd)


@pytest.mark.parametrize("tx_size", [0, 1, 2, 3, 4, 5, 6, 8, 9, 13, 14, 16, 17, 18])
@pytest.mark.parametrize("bd", [8, 10])
def test_intra_directional(dsp, tx_size, bd):
    O = svtlibs.oracle()
    bw, bh = TX_W[tx_size], TX_H[tx_size]
    rng = np.random.default_rng(tx_size * 17 + bd)
    n = 7
    a, l = _neighbours(rng, n, bd)
    es = a.itemsize
    angles = sorted(DR_DERIV)
    for zone in (1, 2, 3):
        for ang in angles[::2] + [87]:
            ups = ((0, 0), (1, 1), (1, 0), (0, 1)) if bw + bh <= 16 else ((0, 0),)
            for ua, ul in ups:
                dx = DR_DERIV[ang] if zone in (1, 2) else 1
                dy = DR_DERIV[90 - ang] if zone == 2 else (DR_DERIV[ang] if zone == 3 else 1)
                out = dsp.intra_pred(_as_dev(a), _as_dev(l), 9 + zone, bw, bh, bd, ua, ul, dx, dy).cpu().numpy()
                out = out if bd == 8 else out.view(np.uint16)
                ref = np.zeros((n, bh, bw), a.dtype)
                for i in range(n):
                    pa = ctypes.c_void_p(a[i].ctypes.data + NB * es); pl = ctypes.c_void_p(l[i].ctypes.data + NB * es)
                    if bd == 8:
                        O.svt_oracle_dr_prediction(zone, ptr(ref[i]), ctypes.c_ssize_t(bw), bw, bh, pa, pl, ua, ul, dx, dy)
                    else:
                        O.svt_oracle_dr_prediction_hbd(zone, ptr(ref[i]), ctypes.c_ssize_t(bw), bw, bh, pa, pl, ua, ul, dx, dy, bd)
                assert np.array_equal(out, ref), (TX_SIZES[tx_size], zone, ang, ua, ul, bd)


@pytest.mark.parametrize("bd", [8, 10])
def test_intra_edge_filter_and_upsample(dsp, bd):
    O = svtlibs.oracle()
    rng = np.random.default_rng(bd)
    n = 11
    dt = np.uint8 if bd == 8 else np.uint16
    for sz in (5, 9, 17, 33, 65, 129):
        for strength in (0, 1, 2, 3):
            e = rng.integers(0, 1 << bd, size=(n, 160)).astype(dt)
            ref = e.copy()
            for i in range(n):
                p = ctypes.c_void_p(ref[i].ctypes.data + NB * e.itemsize)
                (O.svt_oracle_filter_intra_edge if bd == 8 else O.svt_oracle_filter_intra_edge_hbd)(p, sz, strength)
            d = _as_dev(e)
            dsp.filter_intra_edge(d, sz, strength)
            got = d.cpu().numpy(); got = got if bd == 8 else got.view(np.uint16)
            assert np.array_equal(got, ref), (sz, strength)
    for sz in (4, 8, 16):
        e = rng.integers(0, 1 << bd, size=(n, 64)).astype(dt)
        ref = e.copy()
        for i in range(n):
            p = ctypes.c_void_p(ref[i].ctypes.data + NB * e.itemsize)
            if bd == 8: O.svt_oracle_upsample_intra_edge(p, sz)
            else: O.svt_oracle_upsample_intra_edge_hbd(p, sz, bd)
        d = _as_dev(e)
        dsp.upsample_intra_edge(d, sz, bd)
        got = d.cpu().numpy(); got = got if bd == 8 else got.view(np.uint16)
        assert np.array_equal(got, ref), sz


@pytest.mark.parametrize("w,h", [(4, 4), (8, 8), (16, 16), (32, 32), (32, 16), (8, 32)])
def test_full_distortion32(dsp, w, h):
    O = svtlibs.oracle()
    rng = np.random.default_rng(w + h)
    n = 21
    c = rng.integers(-(1 << 18), 1 << 18, size=(n, h, w)).astype(np.int32)
    r = rng.integers(-(1 << 18), 1 << 18, size=(n, h, w)).astype(np.int32)
    r[0] = c[0]
    got = dsp.full_distortion32(dev(c), dev(r), w, h).cpu().numpy().view(np.uint64)
    got0 = dsp.full_distortion32(dev(c), None, w, h, cbf_zero=True).cpu().numpy().view(np.uint64)
    for i in range(n):
        out = np.zeros(2, np.uint64)
        O.svt_oracle_full_distortion32(ptr(c[i]), w, ptr(r[i]), w, ptr(out), w, h)
        assert (int(got[i, 0]), int(got[i, 1])) == (int(out[0]), int(out[1]))
        assert (int(got0[i, 0]), int(got0[i, 1])) == (int(out[1]), int(out[1]))


@pytest.mark.parametrize("sw,sh", [(1, 1), (8, 8), (13, 7), (64, 64), (48, 16), (5, 33), (16, 1), (32, 40), (64, 3), (20, 20)])
def test_me_sb_search_85_pus(dsp, sw, sh):
    """K6: all 85 PU bests + packed MVs of FullPelSearch_LCU, incl. ties (first in raster order)."""
    O = svtlibs.oracle()
    rng = np.random.default_rng(sw * 100 + sh)
    n = 5
    rw, rh = 64 + sw - 1 + 3, 64 + sh - 1
    src = rng.integers(0, 256, size=(n, 64, 64), dtype=np.uint8)
    ref = rng.integers(0, 256, size=(n, rh, rw), dtype=np.uint8)
    ref[0] = 100; src[0] = 101                      # every point ties -> first point wins everywhere
    ref[3] = 255; src[3] = 0                        # largest possible SADs (packed 16-bit partial sums at their limit)
    if sw > 4 and sh > 3:
        ref[1, 2:66, 3:67] = src[1]                 # exact match at (3, 2)
        ref[2, 0:64, 1:65] = src[2]; ref[2, 1:65, 0:64] = src[2]   # two exact matches: (1,0) before (0,1)
    org = np.array([[-32, -16], [0, 0], [-7, 5], [100, -100], [-64, -64]], np.int16)
    bs, bm = dsp.me_sb_search(dev(src), dev(ref), sw, sh, origins=dev(org))
    bs = bs.cpu().numpy().view(np.uint32); bm = bm.cpu().numpy().view(np.uint32)
    for i in range(n):
        rs = np.full(85, 128 * 128 * 255, np.uint32); rm = np.zeros(85, np.uint32)
        O.svt_oracle_me_sb_search(ptr(src[i]), 64, ptr(ref[i]), rw, sw, sh, int(org[i, 0]), int(org[i, 1]), ptr(rs), ptr(rm))
        assert np.array_equal(bs[i], rs), f"sad block {i}"
        assert np.array_equal(bm[i], rm), f"mv block {i}"
    # running-best semantics: a second call with a worse window must not change anything
    bs2, bm2 = dsp.me_sb_search(dev(src), dev((255 - ref)), sw, sh, origins=dev(org), best_sad=dev(bs.view(np.int32)), best_mv=dev(bm.view(np.int32)))
    b2 = bs2.cpu().numpy().view(np.uint32)
    assert (b2 <= bs).all()


@pytest.mark.parametrize("tx_size,tx_type", [(3, 0), (3, 9), (2, 0), (1, 5), (9, 0)])
@pytest.mark.parametrize("qindex", [0, 100, 255])
def test_config2_fwd_quant_on_residual_batch(dsp, tx_size, tx_type, qindex):
    """BASELINE.json configs[1]: FwdTxfm2d + quantize, int16 residual in (tuned 32x32 kernel + composed path)"""
    O = svtlibs.oracle()
    w, h = TX_W[tx_size], TX_H[tx_size]
    rng = np.random.default_rng(tx_size + qindex)
    n = 29
    res = rng.integers(-255, 256, size=(n, h, w)).astype(np.int16)
    res[0] = 255; res[1] = -255; res[2] = 0
    qt = svtlibs.quant_tables(8)
    qrow = {k: v[qindex].copy() for k, v in qt.items()}
    scan, iscan = svtlibs.scan_tables(tx_size, tx_type)
    co, q, dq, eob = dsp.fwd_quant(dev(res), tx_size, tx_type, qrow, dev(iscan))
    ls = 1 if w * h > 256 else 0
    for i in range(n):
        rc = np.zeros(w * h, np.int32); rq = np.zeros(w * h, np.int32); rdq = np.zeros(w * h, np.int32); reob = np.zeros(1, np.uint16)
        O.svt_oracle_fwd_txfm2d(ptr(res[i]), ptr(rc), ctypes.c_uint32(w), tx_type, tx_size, 8)
        O.svt_oracle_quantize_b(ptr(rc), ctypes.c_ssize_t(w * h), 0, ptr(qrow["zbin"]), ptr(qrow["round"]), ptr(qrow["quant"]),
                                ptr(qrow["quant_shift"]), ptr(rq), ptr(rdq), ptr(qrow["dequant"]), ptr(reob), ptr(scan), ptr(iscan), ls, 0)
        assert np.array_equal(co[i].cpu().numpy(), rc) and np.array_equal(q[i].cpu().numpy(), rq)
        assert np.array_equal(dq[i].cpu().numpy(), rdq) and int(eob[i].item()) & 0xffff == int(reob[0])


def test_fused32_nonstandard_quant_tables_take_the_general_path(dsp):
    """quant_shift that is not a power of two must still be exact (QMODE 1 fallback)"""
    rng = np.random.default_rng(5)
    src, pred = make_pixels(rng, 9, 32, 32, "random")
    qrow = {"zbin": np.array([37, 41] + [41] * 6, np.int16), "round": np.array([19, 23] + [23] * 6, np.int16),
            "quant": np.array([-12345, 7001] + [7001] * 6, np.int16), "quant_shift": np.array([12000, 777] + [777] * 6, np.int16),
            "dequant": np.array([53, 61] + [61] * 6, np.int16)}
    _, iscan = svtlibs.scan_tables(3, 0)
    co, q, dq, eob, sad = dsp.fwd_quant_sad(dev(src), dev(pred), 3, 0, qrow, dev(iscan))
    rco, rq, rdq, reob, rsad = oracle_chain(src, pred, 3, 0, qrow)
    assert np.array_equal(q.cpu().numpy(), rq) and np.array_equal(dq.cpu().numpy(), rdq)
    assert np.array_equal(eob.cpu().numpy().view(np.uint16), reob)


@pytest.mark.parametrize("tx_size", [2, 3, 4, 9, 10, 17, 18])
@pytest.mark.parametrize("bd", [8, 10])
def test_inverse_clamp_free_threshold(dsp, tx_size, bd):
    """the inverse kernels drop the stage clamps while gain * L1(row or column) + slack <= the stage bound (csrc/kernel_txfm.h
    inv1d, kernel_fused32.h idct32_pass).  Blocks whose rows carry all their L1 mass in one or two coefficients, with magnitudes
    that straddle that limit (the worst case of the bound), and blocks that pass the row test but fail the column one."""
    O = svtlibs.oracle()
    rng = np.random.default_rng(900 + tx_size + bd)
    w, h = TX_W[tx_size], TX_H[tx_size]
    kw, kh = min(w, 32), min(h, 32)
    hi = (1 << (bd + 8 - 1)) - 1
    mags = sorted({hi, hi * 3 // 4, hi * 9 // 16, hi // 2, hi * 7 // 16, hi * 3 // 8, hi // 4, 18258, 18259, 13410, 13411, 25192, 25193})
    blocks = []
    for m in mags:
        for rep in range(3):
            co = np.zeros((kh, kw), np.int32)
            if rep == 0:                      # one coefficient per row, random column and sign
                co[np.arange(kh), rng.integers(0, kw, kh)] = m * rng.choice([-1, 1], kh)
            elif rep == 1:                    # the mass split over two coefficients of every row
                a = rng.integers(0, m + 1, kh)
                c0 = rng.integers(0, kw // 2, kh); c1 = rng.integers(kw // 2, kw, kh)
                co[np.arange(kh), c0] = a * rng.choice([-1, 1], kh); co[np.arange(kh), c1] = (m - a) * rng.choice([-1, 1], kh)
            else:                             # one column: quiet rows, a loud column for the second pass
                co[:, int(rng.integers(0, kw))] = (m // 2) * rng.choice([-1, 1], kh)
            blocks.append(co.ravel())
    co = np.stack(blocks)
    n = co.shape[0]
    dst = rng.integers(0, 1 << bd, size=(n, h, w)).astype(np.uint16)
    ref = dst.copy()
    for i in range(n):
        O.svt_oracle_inv_txfm2d_add(ptr(co[i]), ptr(ref[i]), w, 0, tx_size, bd)
    d = dev(dst.view(np.int16))
    dsp.inv_txfm2d_add(dev(co), d, tx_size, 0, bd)
    got = d.cpu().numpy().view(np.uint16)
    bad = [i for i in range(n) if not np.array_equal(got[i], ref[i])]
    assert not bad, (TX_SIZES[tx_size], bd, bad[:5])
    if bd == 8:
        d8 = dev(dst.astype(np.uint8))
        ref8 = dst.astype(np.uint8)
        for i in range(n):
            O.svt_oracle_inv_txfm2d_add_u8(ptr(co[i]), ptr(ref8[i]), w, 0, tx_size)
        dsp.inv_txfm2d_add(dev(co), d8, tx_size, 0, 8)
        assert np.array_equal(d8.cpu().numpy(), ref8)
