"""Every svt_hip_tune knob selects another kernel for the same job: results must not change.  This keeps the
first-generation / general kernels (which the default routing now rarely reaches) under test."""
import numpy as np
import pytest
import torch

import svtlibs
from test_gpu_parity import dev, make_pixels

pytestmark = pytest.mark.gpu


def _qrow(bd, q):
    qt = svtlibs.quant_tables(bd)
    return {k: v[q].copy() for k, v in qt.items()}


def _tune(dsp, key, value):
    assert dsp.lib.svt_hip_tune(key.encode(), int(value)) == 0


def _eq(a, b):
    if isinstance(a, dict):
        return all(_eq(a[k], b[k]) for k in a)
    if isinstance(a, (tuple, list)):
        return all(_eq(x, y) for x, y in zip(a, b))
    if a is None or b is None:
        return a is b
    return torch.equal(a, b)


@pytest.mark.parametrize("knob", ["no_staged", "no_f32p"])
@pytest.mark.parametrize("tx_size", [3, 2, 1, 4, 9])
@pytest.mark.parametrize("bd", [8, 10])
def test_fwd_quant_planes_variants(dsp, knob, tx_size, bd):
    w, h = svtlibs.TX_W[tx_size], svtlibs.TX_H[tx_size]
    rng = np.random.default_rng(tx_size * 10 + bd)
    PH, PW = 3 * h + 5, 4 * w + 9
    hi = 1 << bd
    src = rng.integers(0, hi, size=(PH, PW)); pred = np.clip(src + rng.integers(-20, 21, size=src.shape), 0, hi - 1)
    conv = (lambda a: dev(a.astype(np.uint8))) if bd == 8 else (lambda a: dev(a.astype(np.uint16).view(np.int16)))
    xy = dev(np.array([(y << 16) | x for y in range(2, PH - h + 1, h) for x in range(5, PW - w + 1, w)], np.uint32).view(np.int32))
    _, iscan = svtlibs.scan_tables(tx_size, 0)
    call = lambda: dsp.fwd_quant_planes(conv(src), PW, conv(pred), PW, xy, tx_size, 0, _qrow(bd, 90), dev(iscan), bd=bd)
    try:
        _tune(dsp, knob, 0); a = call()
        _tune(dsp, knob, 1); b = call()
    finally:
        _tune(dsp, knob, 0)
    torch.cuda.synchronize()
    assert _eq(a[:4], b[:4])


@pytest.mark.parametrize("knob", ["no_enc_staged"])
@pytest.mark.parametrize("tx_size", [1, 2, 4, 7])
def test_encode_recon_variants(dsp, knob, tx_size):
    rng = np.random.default_rng(5 + tx_size)
    src, pred = make_pixels(rng, 19, svtlibs.TX_H[tx_size], svtlibs.TX_W[tx_size], "smooth")
    _, iscan = svtlibs.scan_tables(tx_size, 0)
    call = lambda: dsp.encode_recon(dev(src), dev(pred), tx_size, 0, _qrow(8, 70), dev(iscan), keep_coeff=True)
    try:
        _tune(dsp, knob, 0); a = call()
        _tune(dsp, knob, 1); b = call()          # composed two-kernel path
    finally:
        _tune(dsp, knob, 0)
    torch.cuda.synchronize()
    assert _eq(a, b)


@pytest.mark.parametrize("n,keep,kind", [(1, False, "smooth"), (2, True, "smooth"), (5, False, "random"), (64, True, "extreme"), (33, False, "edge")])
def test_encode_recon_64x64_two_blocks_per_wave_vs_one(dsp, n, keep, kind):
    """enc64_kernel (two blocks per wave, pruned 64-point networks, clamp-free inverse under the L1 bound) against the generic
    staged kernel: odd batches (the last wave's second block is a repeat), quiet / noisy / saturated residuals (the last takes
    the clamped inverse), with and without coeff / dqcoeff outputs"""
    rng = np.random.default_rng(640 + n)
    if kind == "edge":       # full-swing residuals: rows whose L1 exceeds the clamp-free bound
        src = np.where(rng.integers(0, 2, size=(n, 64, 64)) > 0, 255, 0).astype(np.uint8); pred = 255 - src
        src[: n // 2] = rng.integers(0, 256, size=(n // 2, 64, 64)); pred[: n // 2] = 128
    else:
        src, pred = make_pixels(rng, n, 64, 64, kind)
    _, iscan = svtlibs.scan_tables(4, 0)
    for q in (30, 200):
        call = lambda: dsp.encode_recon(dev(src), dev(pred), 4, 0, _qrow(8, q), dev(iscan), keep_coeff=keep)
        try:
            _tune(dsp, "no_enc64", 0); a = call()
            _tune(dsp, "no_enc64", 1); b = call()
        finally:
            _tune(dsp, "no_enc64", 0)
        torch.cuda.synchronize()
        assert _eq(a, b), (n, keep, kind, q)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("n", [1, 6, 33])
def test_fwd_quant_and_inverse_64x64_two_blocks_per_wave_vs_one(dsp, n, bd):
    """fq64_kernel (the forward half of the two-blocks-per-wave 64x64 kernel) against the one-block staged kernel, on planes:
    coeff / qcoeff / dqcoeff / eob, then the reconstruction from those dequantised coefficients"""
    rng = np.random.default_rng(64000 + n + bd)
    PH, PW = 2 * 64 + 8, ((n + 1) // 2) * 64 + 24
    dt = np.uint8 if bd == 8 else np.uint16
    srcp = rng.integers(0, 1 << bd, size=(PH, PW)).astype(dt)
    predp = np.clip(srcp.astype(np.int32) + rng.integers(-60, 61, size=(PH, PW)), 0, (1 << bd) - 1).astype(dt)
    if n > 2:
        srcp[:64, :64] = (1 << bd) - 1; predp[:64, :64] = 0          # a saturated block: the clamped inverse
    xs = np.arange(0, PW - 63, 64); ys = np.arange(0, PH - 63, 64)
    xy = ((ys[:, None] << 16) | xs[None, :]).reshape(-1)[:n].astype(np.uint32)
    offs = ((xy >> 16) * PW + (xy & 0xffff)).astype(np.uint32)
    _, iscan = svtlibs.scan_tables(4, 0)
    view = (lambda a: a) if bd == 8 else (lambda a: a.view(np.int16))
    outs = []
    try:
        for knob in (0, 1):
            _tune(dsp, "no_enc64", knob)
            co, q, dq, eob, _, _ = dsp.fwd_quant_planes(dev(view(srcp)), PW, dev(view(predp)), PW, dev(xy.view(np.int32)), 4, 0, _qrow(bd, 60), dev(iscan), bd=bd)
            recon = dev(view(predp))
            dsp.inv_txfm2d_add(dq, recon, 4, 0, bd, dst_stride=PW, dst_block_pitch=0, offsets=dev(offs.view(np.int32)))
            outs.append({"coeff": co, "qcoeff": q, "dqcoeff": dq, "eob": eob, "recon": recon})
    finally:
        _tune(dsp, "no_enc64", 0)
    torch.cuda.synchronize()
    assert _eq(outs[0], outs[1])
    assert not torch.equal(outs[0]["recon"], dev(view(predp)))


@pytest.mark.parametrize("bd", [8, 10])
def test_encode_recon_64x64_planes_two_blocks_per_wave_vs_one(dsp, bd):
    """the same A/B on picture planes (origin table, in-place reconstruction into a copy of the prediction), 8 and 10 bit"""
    rng = np.random.default_rng(6400 + bd)
    PH, PW = 3 * 64 + 8, 5 * 64 + 24
    dt = np.uint8 if bd == 8 else np.uint16
    srcp = rng.integers(0, 1 << bd, size=(PH, PW)).astype(dt)
    predp = np.clip(srcp.astype(np.int32) + rng.integers(-30, 31, size=(PH, PW)), 0, (1 << bd) - 1).astype(dt)
    xs = np.arange(0, PW - 63, 64); ys = np.arange(0, PH - 63, 64)
    xy = ((ys[:, None] << 16) | xs[None, :]).reshape(-1).astype(np.uint32)
    _, iscan = svtlibs.scan_tables(4, 0)
    view = (lambda a: a) if bd == 8 else (lambda a: a.view(np.int16))
    outs = []
    try:
        for knob in (0, 1):
            _tune(dsp, "no_enc64", knob)
            recon = dev(view(predp))
            r = dsp.encode_recon_planes(dev(view(srcp)), PW, dev(view(predp)), PW, recon, PW, dev(xy.view(np.int32)), 4, 0, _qrow(bd, 90), dev(iscan), bd=bd)
            r["recon"] = recon
            outs.append(r)
    finally:
        _tune(dsp, "no_enc64", 0)
    torch.cuda.synchronize()
    assert _eq(outs[0], outs[1])
    assert not torch.equal(outs[0]["recon"], dev(view(predp)))            # something was reconstructed


@pytest.mark.parametrize("knob", ["no_inv_planes", "no_staged"])
@pytest.mark.parametrize("tx_size,bd", [(2, 8), (1, 8), (4, 8), (11, 8), (2, 10), (4, 10)])
def test_inverse_on_planes_variants(dsp, knob, tx_size, bd):
    w, h = svtlibs.TX_W[tx_size], svtlibs.TX_H[tx_size]
    kw, kh = min(w, 32), min(h, 32)
    rng = np.random.default_rng(tx_size + bd)
    PH, PW = 2 * h + 3, 3 * w + 4
    offs = np.array([y * PW + x for y in range(1, PH - h + 1, h) for x in range(2, PW - w + 1, w)], np.uint32)
    co = dev(rng.integers(-900, 901, size=(len(offs), kw * kh)).astype(np.int32))
    plane = rng.integers(0, 1 << bd, size=(PH, PW))
    mk = (lambda: dev(plane.astype(np.uint8))) if bd == 8 else (lambda: dev(plane.astype(np.uint16).view(np.int16)))
    outs = []
    try:
        for v in (0, 1):
            _tune(dsp, knob, v)
            d = mk()
            dsp.inv_txfm2d_add(co, d, tx_size, 0, bd, dst_stride=PW, dst_block_pitch=0, offsets=dev(offs.view(np.int32)))
            outs.append(d)
    finally:
        _tune(dsp, knob, 0)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("n,bw,sw,sh", [(1, 16, 8, 8), (7, 16, 8, 8), (8, 16, 8, 8), (9, 16, 8, 8), (5000, 16, 8, 8), (4099, 8, 8, 8), (333, 16, 5, 3), (333, 8, 13, 5),
                                        (333, 16, 11, 8), (77, 8, 4, 2), (77, 16, 1, 1)])
def test_sad_search_pipelined_vs_one_shot(dsp, n, bw, sw, sh):
    """sad_search_q2p_kernel (persistent, next set's chunks prefetched during the search) against sad_search_q2_kernel: batch sizes
    around one wave's set of blocks, more sets than waves, ragged search areas, ties, footprint-tail chunks; both against the
    SADs recomputed in numpy for a few blocks"""
    rng = np.random.default_rng(n + bw + sw)
    src = rng.integers(0, 256, size=(n, bw, bw), dtype=np.uint8)
    ref = rng.integers(0, 256, size=(n, bw + sh - 1, bw + sw - 1), dtype=np.uint8)
    ref[0] = 3; src[0] = 4                                   # ties everywhere: the first candidate wins
    if n > 2:
        ref[n - 1, sh - 1:, sw - 1:] = src[n - 1]            # exact match at the LAST candidate of the LAST block
    try:
        _tune(dsp, "no_q2p", 0); a = dsp.sad_search(dev(src), dev(ref), sw, sh)
        _tune(dsp, "no_q2p", 1); b = dsp.sad_search(dev(src), dev(ref), sw, sh)
    finally:
        _tune(dsp, "no_q2p", 0)
    torch.cuda.synchronize()
    assert _eq(a, b)
    sad, bx, by = (t.cpu().numpy() for t in a)
    for i in sorted({0, n - 1, n // 2}):
        best = None
        for y in range(sh):
            for x in range(sw):
                v = int(np.abs(src[i].astype(int) - ref[i, y:y + bw, x:x + bw].astype(int)).sum())
                if best is None or v < best[0]:
                    best = (v, x, y)
        assert (int(sad[i]), int(bx[i]), int(by[i])) == best, (i, best)


@pytest.mark.parametrize("knob,bw,sw,sh", [("q2_su4", 16, 8, 8), ("q2_su4", 8, 16, 16), ("no_q2", 16, 8, 8), ("no_q2", 8, 13, 5), ("no_qsad", 16, 8, 8), ("no_qsad", 32, 9, 6),
                                             ("no_q16", 32, 9, 6), ("no_q16", 64, 16, 16), ("no_q16", 32, 37, 3)])
def test_sad_search_variants(dsp, knob, bw, sw, sh):
    rng = np.random.default_rng(bw + sw)
    n = 37
    src = rng.integers(0, 256, size=(n, bw, bw), dtype=np.uint8)
    ref = rng.integers(0, 256, size=(n, bw + sh - 1, bw + sw - 1), dtype=np.uint8)
    ref[0] = 3; src[0] = 4                                   # ties everywhere
    try:
        _tune(dsp, knob, 0); a = dsp.sad_search(dev(src), dev(ref), sw, sh)
        _tune(dsp, knob, 1); b = dsp.sad_search(dev(src), dev(ref), sw, sh)
    finally:
        _tune(dsp, knob, 0)
    torch.cuda.synchronize()
    assert _eq(a, b)


@pytest.mark.parametrize("bw,bh,sw,sh", [(32, 8, 16, 16), (32, 16, 5, 9), (32, 64, 33, 2), (64, 16, 16, 7), (64, 32, 70, 3),
                                         (64, 64, 1, 1), (32, 32, 1, 40), (64, 64, 64, 17), (16, 16, 16, 16), (16, 16, 32, 3),
                                         (16, 32, 7, 5), (16, 64, 24, 9), (16, 32, 80, 2)])
def test_sad_search_wide_variants(dsp, bw, bh, sw, sh):
    """sad_search_q16_kernel (16 candidates per lane) vs the 4-candidate kernel: rectangular blocks, search
    widths that are not a multiple of 16, more tasks than lanes, row splits, ties and the maximum SAD."""
    rng = np.random.default_rng(bw * bh + sw)
    n = 19
    src = rng.integers(0, 256, size=(n, bh, bw), dtype=np.uint8)
    ref = rng.integers(0, 256, size=(n, bh + sh - 1, bw + sw - 1), dtype=np.uint8)
    ref[0] = 3; src[0] = 4
    ref[1] = 255; src[1] = 0
    ref[2] = src[2, :1, :1]; ref[2, sh - 1:, sw - 1:] = src[2]             # exact match at the LAST candidate
    try:
        _tune(dsp, "no_q16", 0); a = dsp.sad_search(dev(src), dev(ref), sw, sh)
        _tune(dsp, "no_q16", 1); b = dsp.sad_search(dev(src), dev(ref), sw, sh)
    finally:
        _tune(dsp, "no_q16", 0)
    torch.cuda.synchronize()
    assert _eq(a, b)
    assert int(a[0][2]) == 0 and int(a[1][2]) == sw - 1 and int(a[2][2]) == sh - 1


@pytest.mark.parametrize("sw,sh", [(64, 64), (16, 5), (48, 16), (13, 7), (1, 1), (37, 21)])
def test_me_sb_search_variants(dsp, sw, sh):
    rng = np.random.default_rng(sw + sh)
    n = 6
    src = rng.integers(0, 256, size=(n, 64, 64), dtype=np.uint8)
    ref = rng.integers(0, 256, size=(n, 64 + sh - 1, 64 + sw - 1 + 2), dtype=np.uint8)
    ref[0] = 9; src[0] = 7
    ref[1] = 255; src[1] = 0
    # the 16-points-per-lane kernel (legacy result layout 8x8 | 16x16 | 32x32 | 64x64) against the search-point-by-search-point
    # kernel in the reference's layout (svt_hip_tune("me_exact", 1)): two independent implementations
    a = dsp.me_sb_search(dev(src), dev(ref), sw, sh)
    try:
        _tune(dsp, "me_exact", 1); e = dsp.me_fullpel_search(dev(src), dev(ref), sw, sh, nsq=False)
    finally:
        _tune(dsp, "me_exact", 0)
    torch.cuda.synchronize()
    order = [21 + i for i in range(64)] + [5 + i for i in range(16)] + [1 + i for i in range(4)] + [0]     # legacy slot -> EbMeTierZeroPu index
    b = tuple(t[:, order].contiguous() for t in e)
    assert _eq(a, b)
