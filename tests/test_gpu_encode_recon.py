"""GPU parity of the encode-pass chain svt_hip_encode_recon_batch (residual -> forward -> quantize /
dequantize -> inverse + prediction) against the CPU oracle's separate stages, i.e. the reference call
sequence of Av1EncodeLoop (EbCodingLoop.c:545-950).  Bit-exact; there is no reference unit test for the
chain as a whole, so the pins are the per-stage oracle functions (themselves pinned in
test_oracle_vs_ref.py / test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

import svtlibs
from svtlibs import TX_H, TX_W, ptr
from test_gpu_parity import dev, make_pixels, oracle_chain

pytestmark = pytest.mark.gpu


def oracle_recon(pred, dq, tx_size, tx_type):
    O = svtlibs.oracle()
    n, h, w = pred.shape
    rec = pred.copy()
    full = np.zeros(w * h, np.int32)
    for i in range(n):
        full[:] = 0
        full[:dq.shape[1]] = dq[i]
        O.svt_oracle_inv_txfm2d_add_u8(ptr(full), ptr(rec[i]), w, tx_type, tx_size)
    return rec


def check(dsp, src, pred, tx_size, tx_type, qindex, keep):
    qt = svtlibs.quant_tables(8)
    qrow = {k: v[qindex].copy() for k, v in qt.items()}
    _, iscan = svtlibs.scan_tables(tx_size, tx_type)
    out = dsp.encode_recon(dev(src), dev(pred), tx_size, tx_type, qrow, dev(iscan), keep_coeff=keep)
    torch.cuda.synchronize()
    rco, rq, rdq, reob, rsad = oracle_chain(src, pred, tx_size, tx_type, qrow)
    assert np.array_equal(out["qcoeff"].cpu().numpy(), rq)
    assert np.array_equal(out["eob"].cpu().numpy().view(np.uint16), reob)
    assert np.array_equal(out["sad"].cpu().numpy().view(np.uint32), rsad)
    if keep:
        assert np.array_equal(out["coeff"].cpu().numpy(), rco)
        assert np.array_equal(out["dqcoeff"].cpu().numpy(), rdq)
    assert np.array_equal(out["recon"].cpu().numpy(), oracle_recon(pred, rdq, tx_size, tx_type))


@pytest.mark.parametrize("qindex", [0, 25, 100, 255])
@pytest.mark.parametrize("kind", ["random", "extreme", "smooth"])
@pytest.mark.parametrize("keep", [False, True])
def test_fused_encode_recon_32x32(dsp, qindex, kind, keep):
    rng = np.random.default_rng(4242 + qindex)
    src, pred = make_pixels(rng, 37, 32, 32, kind)             # odd count: half-empty last wave
    check(dsp, src, pred, 3, 0, qindex, keep)


def test_fused_encode_recon_32x32_idtx(dsp):
    rng = np.random.default_rng(99)
    src, pred = make_pixels(rng, 21, 32, 32, "smooth")
    check(dsp, src, pred, 3, 9, 100, False)                     # IDTX


@pytest.mark.parametrize("keep", [True, False])
@pytest.mark.parametrize("tx_size,tx_type", [(0, 0), (1, 5), (1, 0), (2, 1), (2, 6), (4, 0), (5, 3), (6, 0), (7, 8), (9, 0), (10, 9),
                                             (11, 0), (12, 0), (13, 1), (14, 0), (15, 0), (16, 10), (17, 0), (18, 0)])
def test_encode_recon_other_sizes(dsp, tx_size, tx_type, keep):
    """enc_staged_kernel (every size but 4x4 and 32x32) and, for 4x4, the one-lane-per-block enc4_kernel."""
    if not svtlibs.txfm_allowed(tx_size, tx_type):
        pytest.skip("type not defined for this size")
    rng = np.random.default_rng(7 + tx_size)
    for kind, n in (("smooth", 11), ("extreme", 7), ("random", 70)):       # 70: more than one wave of small blocks
        src, pred = make_pixels(rng, n, TX_H[tx_size], TX_W[tx_size], kind)
        check(dsp, src, pred, tx_size, tx_type, 60, keep)


def test_encode_recon_argument_errors(dsp, pkg):
    qt = svtlibs.quant_tables(8)
    qrow = {k: v[100].copy() for k, v in qt.items()}
    _, iscan = svtlibs.scan_tables(0, 0)
    src = torch.zeros((3, 4, 4), dtype=torch.uint8, device="cuda:0")
    dsp.lib.svt_hip_tune(b"no_enc_staged", 1)
    try:
        with pytest.raises(RuntimeError):                       # the composed two-kernel path needs coeff / dqcoeff buffers
            dsp.encode_recon(src, src.clone(), 0, 0, qrow, dev(iscan), keep_coeff=False)
    finally:
        dsp.lib.svt_hip_tune(b"no_enc_staged", 0)
    out = dsp.encode_recon(src[:0], src[:0].clone(), 0, 0, qrow, dev(iscan))     # empty batch is a no-op
    assert out["recon"].shape[0] == 0


def test_encode_recon_full_size_round_trip_property(dsp):
    """BASELINE-size property: at qindex 0 (lossless-ish step 4) the reconstruction stays within the
    quantiser's error bound of the source on 2^17 smooth blocks, and equals the two-kernel path."""
    rng = np.random.default_rng(5)
    n = 1 << 17
    g = torch.Generator(device="cuda:0"); g.manual_seed(77)
    src = torch.randint(0, 256, (n, 32, 32), dtype=torch.uint8, device="cuda:0", generator=g)
    pred = (src.to(torch.int16) + torch.randint(-6, 7, (n, 32, 32), dtype=torch.int16, device="cuda:0", generator=g)).clamp(0, 255).to(torch.uint8)
    qt = svtlibs.quant_tables(8)
    qrow = {k: v[40].copy() for k, v in qt.items()}
    _, iscan = svtlibs.scan_tables(3, 0)
    isc = dev(iscan)
    a = dsp.encode_recon(src, pred, 3, 0, qrow, isc, keep_coeff=False)
    co, q, dq, eob, sad = dsp.fwd_quant_sad(src, pred, 3, 0, qrow, isc)
    rec2 = pred.clone()
    dsp.inv_txfm2d_add(dq, rec2, 3, 0, 8)
    torch.cuda.synchronize()
    assert torch.equal(a["recon"], rec2) and torch.equal(a["qcoeff"], q) and torch.equal(a["eob"], eob) and torch.equal(a["sad"], sad)
    err = (a["recon"].to(torch.int16) - src.to(torch.int16)).abs().max().item()
    assert err <= 24, err


@pytest.mark.parametrize("tx_size,tx_type", [(3, 0), (3, 9), (2, 0), (2, 5), (1, 0), (4, 0), (9, 0), (7, 3), (5, 0), (13, 0), (17, 0)])
@pytest.mark.parametrize("inplace", [False, True])
def test_encode_recon_on_planes(dsp, tx_size, tx_type, inplace):
    """Plane-addressed chain (svt_hip_encode_recon_planes_batch) on a frame whose size is not a multiple of the block:
    every block against the oracle stages, untouched margin == prediction, in-place reconstruction allowed."""
    if not svtlibs.txfm_allowed(tx_size, tx_type):
        pytest.skip("type not defined for this size")
    w, h = TX_W[tx_size], TX_H[tx_size]
    rng = np.random.default_rng(31 * tx_size + tx_type)
    PH, PW = 200, 328
    src = rng.integers(0, 256, size=(PH, PW), dtype=np.uint8)
    pred = np.clip(src.astype(int) + rng.integers(-9, 10, size=src.shape), 0, 255).astype(np.uint8)
    xs = np.arange(3, PW - w + 1, w); ys = np.arange(1, PH - h + 1, h)          # unaligned origins on purpose
    xy = np.array([(y << 16) | x for y in ys for x in xs], np.uint32)
    qt = svtlibs.quant_tables(8)
    qrow = {k: v[80].copy() for k, v in qt.items()}
    _, iscan = svtlibs.scan_tables(tx_size, tx_type)
    d_pred = dev(pred)
    d_recon = d_pred if inplace else torch.full_like(d_pred, 7)
    out = dsp.encode_recon_planes(dev(src), PW, d_pred, PW, d_recon, PW, dev(xy.view(np.int32)), tx_size, tx_type, qrow, dev(iscan),
                                  keep_coeff=True, want_sad=True)
    torch.cuda.synchronize()
    rec = d_recon.cpu().numpy()
    sb = np.stack([src[y:y + h, x:x + w] for y in ys for x in xs]); pb = np.stack([pred[y:y + h, x:x + w] for y in ys for x in xs])
    rco, rq, rdq, reob, rsad = oracle_chain(sb, pb, tx_size, tx_type, qrow)
    assert np.array_equal(out["qcoeff"].cpu().numpy(), rq) and np.array_equal(out["coeff"].cpu().numpy(), rco)
    assert np.array_equal(out["dqcoeff"].cpu().numpy(), rdq)
    assert np.array_equal(out["eob"].cpu().numpy().view(np.uint16), reob)
    assert np.array_equal(out["sad"].cpu().numpy().view(np.uint32), rsad)
    rrec = oracle_recon(pb, rdq, tx_size, tx_type)
    expect = pred.copy() if inplace else np.full_like(pred, 7)
    for i, (y, x) in enumerate([(y, x) for y in ys for x in xs]):
        expect[y:y + h, x:x + w] = rrec[i]
    assert np.array_equal(rec, expect)


@pytest.mark.parametrize("tx_size,tx_type", [(3, 0), (3, 9), (2, 0), (2, 7), (1, 0), (4, 0), (9, 0), (8, 2), (5, 0), (13, 0), (18, 0)])
@pytest.mark.parametrize("inplace", [False, True])
def test_encode_recon_on_planes_10bit(dsp, tx_size, tx_type, inplace):
    """BASELINE configs[4] shape: the fused chain on 10-bit planes (uint16 samples), every fused size."""
    import ctypes
    if not svtlibs.txfm_allowed(tx_size, tx_type):
        pytest.skip("type not defined for this size")
    O = svtlibs.oracle()
    w, h = TX_W[tx_size], TX_H[tx_size]
    rng = np.random.default_rng(1010 + tx_type + 17 * tx_size)
    PH, PW = 136, 200
    src = rng.integers(0, 1024, size=(PH, PW)).astype(np.uint16)
    pred = np.clip(src.astype(int) + rng.integers(-30, 31, size=src.shape), 0, 1023).astype(np.uint16)
    src[0:40, 0:40] = 1023; pred[0:40, 0:40] = 0                   # an extreme block
    xs = np.arange(5, PW - w + 1, w); ys = np.arange(2, PH - h + 1, h)
    xy = np.array([(y << 16) | x for y in ys for x in xs], np.uint32)
    qt = svtlibs.quant_tables(10)
    qrow = {k: v[120].copy() for k, v in qt.items()}
    _, iscan = svtlibs.scan_tables(tx_size, tx_type)
    d_pred = dev(pred.view(np.int16))
    d_recon = d_pred if inplace else torch.full_like(d_pred, 9)
    out = dsp.encode_recon_planes(dev(src.view(np.int16)), PW, d_pred, PW, d_recon, PW, dev(xy.view(np.int32)), tx_size, tx_type, qrow,
                                  dev(iscan), keep_coeff=True, bd=10)
    torch.cuda.synchronize()
    rec = d_recon.cpu().numpy().view(np.uint16)
    expect = pred.copy() if inplace else np.full_like(pred, 9)
    for i, (y, x) in enumerate([(y, x) for y in ys for x in xs]):
        nc = min(w, 32) * min(h, 32)
        rc = np.zeros(w * h, np.int32); rq = np.zeros(w * h, np.int32); rdq = np.zeros(w * h, np.int32); reob = np.zeros(1, np.uint16)
        sp = ctypes.c_void_p(int(src.ctypes.data) + int(y * PW + x) * 2); pp = ctypes.c_void_p(int(pred.ctypes.data) + int(y * PW + x) * 2)
        O.svt_oracle_fwd_quant_planes(sp, PW, pp, PW, 1, 10, tx_size, tx_type, ptr(qrow["zbin"]), ptr(qrow["round"]), ptr(qrow["quant"]),
                                      ptr(qrow["quant_shift"]), ptr(qrow["dequant"]), ptr(rc), ptr(rq), ptr(rdq), ptr(reob), None, None)
        blk = np.ascontiguousarray(pred[y:y + h, x:x + w])
        O.svt_oracle_inv_txfm2d_add(ptr(rdq), ptr(blk), w, tx_type, tx_size, 10)
        expect[y:y + h, x:x + w] = blk
        assert np.array_equal(out["qcoeff"][i].cpu().numpy(), rq[:nc]) and np.array_equal(out["coeff"][i].cpu().numpy(), rc[:nc]), i
        assert np.array_equal(out["dqcoeff"][i].cpu().numpy(), rdq[:nc]) and int(out["eob"][i].cpu().numpy().view(np.uint16)) == int(reob[0]), i
    assert np.array_equal(rec, expect)


@pytest.mark.parametrize("tx_size,n", [(1, 100003), (2, 50021), (4, 4099), (9, 20011), (16, 30011)])
def test_fused_chain_equals_two_kernel_path_at_scale(dsp, tx_size, n):
    """Size-independent property at batch sizes that are not multiples of any tile: the fused chain and the two
    separate kernels (each bit-exact against the oracle on small cases) must agree on every block."""
    w, h = TX_W[tx_size], TX_H[tx_size]
    g = torch.Generator(device="cuda:0"); g.manual_seed(n)
    src = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device="cuda:0", generator=g)
    pred = (src.to(torch.int16) + torch.randint(-25, 26, (n, h, w), dtype=torch.int16, device="cuda:0", generator=g)).clamp(0, 255).to(torch.uint8)
    qt = svtlibs.quant_tables(8)
    qrow = {k: v[55].copy() for k, v in qt.items()}
    _, iscan = svtlibs.scan_tables(tx_size, 0)
    isc = dev(iscan)
    a = dsp.encode_recon(src, pred, tx_size, 0, qrow, isc, keep_coeff=True)
    co, q, dq, eob, sad = dsp.fwd_quant_sad(src, pred, tx_size, 0, qrow, isc)
    rec2 = pred.clone()
    dsp.inv_txfm2d_add(dq, rec2, tx_size, 0, 8)
    torch.cuda.synchronize()
    assert torch.equal(a["recon"], rec2) and torch.equal(a["qcoeff"], q) and torch.equal(a["coeff"], co)
    assert torch.equal(a["dqcoeff"], dq) and torch.equal(a["eob"], eob) and torch.equal(a["sad"], sad)


@pytest.mark.parametrize("tx_type", range(16))
def test_encode_recon_4x4_every_type_fused_equals_two_stage(dsp, tx_type):
    """enc4_kernel (one lane per block, registers only) against the oracle AND against the two-kernel path it replaces,
    all 16 transform types incl. the flips, extreme residuals, several quantisers incl. a non-power-of-two quant_shift"""
    rng = np.random.default_rng(100 + tx_type)
    for kind, n in (("smooth", 130), ("extreme", 66), ("random", 515)):
        src, pred = make_pixels(rng, n, 4, 4, kind)
        for q in (0, 60, 255):
            check(dsp, src, pred, 0, tx_type, q, True)
            check(dsp, src, pred, 0, tx_type, q, False)
    src, pred = make_pixels(rng, 300, 4, 4, "random")
    qt = svtlibs.quant_tables(8)
    qrow = {k: v[90].copy() for k, v in qt.items()}
    qrow["quant_shift"][:] = 12345                     # not a power of two: the exact 64-bit quantiser form
    _, iscan = svtlibs.scan_tables(0, tx_type)
    a = dsp.encode_recon(dev(src), dev(pred), 0, tx_type, qrow, dev(iscan), keep_coeff=True)
    dsp.lib.svt_hip_tune(b"no_enc_staged", 1)
    try:
        b = dsp.encode_recon(dev(src), dev(pred), 0, tx_type, qrow, dev(iscan), keep_coeff=True)
    finally:
        dsp.lib.svt_hip_tune(b"no_enc_staged", 0)
    for k in ("coeff", "qcoeff", "dqcoeff", "eob", "recon", "sad"):
        assert torch.equal(a[k], b[k]), k
