"""GPU: BASELINE.json's FULL sizes (2^20 blocks) through size-independent properties plus a
sampled oracle comparison — the oracle alone would need minutes for a million blocks."""
import ctypes

import numpy as np
import pytest
import torch

import svtlibs
from svtlibs import ptr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = 1 << 20


def test_headline_chain_1m_blocks_properties_and_sample(dsp):
    O = svtlibs.oracle()
    g = torch.Generator(device=DEV); g.manual_seed(13596)
    src = torch.randint(0, 256, (N, 32, 32), dtype=torch.uint8, device=DEV, generator=g)
    # smooth-ish prediction so that eob varies: pred = src +- small noise on most blocks, random on some
    noise = torch.randint(-6, 7, (N, 32, 32), dtype=torch.int16, device=DEV, generator=g)
    pred = (src.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
    pred[::97] = torch.randint(0, 256, (len(range(0, N, 97)), 32, 32), dtype=torch.uint8, device=DEV, generator=g)
    pred[5] = src[5]                                           # identical -> everything zero
    qt = svtlibs.quant_tables(8)
    qrow = {k: v[100].copy() for k, v in qt.items()}
    scan, iscan = svtlibs.scan_tables(3, 0)
    co, q, dq, eob, sad = dsp.fwd_quant_sad(src, pred, 3, 0, qrow, torch.from_numpy(iscan).to(DEV))
    torch.cuda.synchronize()
    eobv = eob.to(torch.int32) & 0xffff
    # (1) SAD is the L1 norm of the residual (checksum of checksums over the whole batch)
    l1 = (src.to(torch.int32) - pred.to(torch.int32)).abs_().sum(dim=(1, 2))
    assert torch.equal(l1, sad)
    # (2) eob == 0  <=>  qcoeff all zero  <=> dqcoeff all zero
    qnz = (q != 0).any(dim=1)
    assert torch.equal(qnz, eobv > 0) and torch.equal((dq != 0).any(dim=1), qnz)
    # (3) eob is 1 + the largest scan position holding a non-zero level
    isc = torch.from_numpy(iscan.astype(np.int32)).to(DEV)
    pos = torch.where(q != 0, isc.unsqueeze(0) + 1, torch.zeros_like(q)).amax(dim=1)
    assert torch.equal(pos, eobv)
    # (4) dequantisation law of the 32x32 quantizer: |dq| = (|q| * dequant) >> 1, sign(dq) = sign(q)
    deq = torch.full((1024,), int(qrow["dequant"][1]), dtype=torch.int32, device=DEV); deq[0] = int(qrow["dequant"][0])
    assert torch.equal(dq.abs(), (q.abs() * deq) >> 1) and torch.equal(torch.sign(dq), torch.sign(q))
    # (5) DC coefficient of DCT_DCT 32x32 == (sum of residual) * 4 rounded twice; check linearity weakly: identical blocks -> 0
    assert int(eobv[5]) == 0 and not co[5].any()
    # (6) sampled bit-exact comparison with the oracle
    idx = np.random.default_rng(1).choice(N, size=1500, replace=False)
    s_h = src[idx].cpu().numpy(); p_h = pred[idx].cpu().numpy()
    co_h = co[idx].cpu().numpy(); q_h = q[idx].cpu().numpy(); dq_h = dq[idx].cpu().numpy()
    e_h = eobv[idx].cpu().numpy(); sad_h = sad[idx].cpu().numpy()
    for i in range(len(idx)):
        rc = np.zeros(1024, np.int32); rq = np.zeros(1024, np.int32); rdq = np.zeros(1024, np.int32)
        reob = np.zeros(1, np.uint16); rsad = np.zeros(1, np.uint32)
        O.svt_oracle_fwd_quant_sad(ptr(s_h[i]), 32, ptr(p_h[i]), 32, 3, 0, ptr(qrow["zbin"]), ptr(qrow["round"]), ptr(qrow["quant"]),
                                   ptr(qrow["quant_shift"]), ptr(qrow["dequant"]), ptr(rc), ptr(rq), ptr(rdq), ptr(reob), ptr(rsad))
        assert np.array_equal(co_h[i], rc) and np.array_equal(q_h[i], rq) and np.array_equal(dq_h[i], rdq)
        assert int(e_h[i]) == int(reob[0]) and int(sad_h[i]) == int(rsad[0])


def test_config3_sad_search_1m_blocks(dsp):
    """configs[2]: 16x16 SAD, 64-candidate search, 1M blocks, 1 % exact duplicates (ties)"""
    O = svtlibs.oracle()
    g = torch.Generator(device=DEV); g.manual_seed(13598)
    src = torch.randint(0, 256, (N, 16, 16), dtype=torch.uint8, device=DEV, generator=g)
    ref = torch.randint(0, 256, (N, 23, 23), dtype=torch.uint8, device=DEV, generator=g)
    dup = torch.arange(0, N, 100, device=DEV)
    ref[dup, 3:19, 5:21] = src[dup]                            # exact copy at (5, 3)
    ref[dup, 0:16, 0:16] = src[dup]                            # ... and at (0, 0): the first one must win
    best, x, y = dsp.sad_search(src, ref, 8, 8)
    torch.cuda.synchronize()
    assert torch.all(best[dup] == 0) and torch.all(x[dup] == 0) and torch.all(y[dup] == 0)
    # property: best_sad equals the SAD at the reported position, and no candidate is smaller (sampled)
    idx = np.random.default_rng(2).choice(N, size=1200, replace=False)
    s_h = src[idx].cpu().numpy(); r_h = ref[idx].cpu().numpy()
    b_h = best[idx].cpu().numpy(); x_h = x[idx].cpu().numpy(); y_h = y[idx].cpu().numpy()
    for i in range(len(idx)):
        rb = np.zeros(1, np.uint64); rx = np.zeros(1, np.int16); ry = np.zeros(1, np.int16)
        O.svt_oracle_sad_loop(ptr(s_h[i]), 16, ptr(r_h[i]), 23, 16, 16, ptr(rb), ptr(rx), ptr(ry), 23, ctypes.c_int16(8), ctypes.c_int16(8))
        assert (int(b_h[i]), int(x_h[i]), int(y_h[i])) == (int(rb[0]), int(rx[0]), int(ry[0]))


def test_roundtrip_idempotence_1m_8x8(dsp):
    """fwd -> inverse with no quantisation reproduces the residual exactly for small blocks at bd 8:
    recon(pred, fwd(src - pred)) == src (the 8x8 integer DCT pair is lossless on 9-bit residuals)."""
    n = 1 << 20
    g = torch.Generator(device=DEV); g.manual_seed(7)
    src = torch.randint(0, 256, (n, 8, 8), dtype=torch.uint8, device=DEV, generator=g)
    pred = torch.randint(0, 256, (n, 8, 8), dtype=torch.uint8, device=DEV, generator=g)
    res = dsp.residual(src, pred)
    co = dsp.fwd_txfm2d(res, 1, 0, 8)
    recon = pred.clone()
    dsp.inv_txfm2d_add(co, recon, 1, 0, 8)
    torch.cuda.synchronize()
    diff = (recon.to(torch.int16) - src.to(torch.int16)).abs()
    assert int(diff.max()) <= 1          # integer DCT pair: at most one LSB of rounding noise
    assert float((diff != 0).float().mean()) < 0.2


def test_wide_sad_search_full_size_properties(dsp):
    """sad_search_q16_kernel at bench scale (2^17 32x32 and 2^15 64x64 blocks, 16x16 search areas): equal to the
    4-candidate kernel, the reported SAD is the SAD at the reported candidate, and no candidate beats it."""
    g = torch.Generator(device=DEV); g.manual_seed(7)
    for (b, n) in ((32, 1 << 17), (64, 1 << 15)):
        sw = sh = 16
        src = torch.randint(0, 256, (n, b, b), dtype=torch.uint8, device=DEV, generator=g)
        ref = torch.randint(0, 256, (n, b + sh - 1, b + sw - 1), dtype=torch.uint8, device=DEV, generator=g)
        ref[::5, 3:3 + b, 9:9 + b] = src[::5]                    # planted exact matches
        best, x, y = dsp.sad_search(src, ref, sw, sh)
        assert dsp.lib.svt_hip_tune(b"no_q16", 1) == 0
        try:
            best2, x2, y2 = dsp.sad_search(src, ref, sw, sh)
        finally:
            dsp.lib.svt_hip_tune(b"no_q16", 0)
        assert torch.equal(best, best2) and torch.equal(x, x2) and torch.equal(y, y2)
        assert (best[::5] == 0).all() and (x[::5] == 9).all() and (y[::5] == 3).all()
        # SAD at the reported candidate, recomputed with torch on a slice of the batch
        m = 4096
        xi, yi = x[:m].long(), y[:m].long()
        rows = yi[:, None, None] + torch.arange(b, device=DEV)[None, :, None]
        cols = xi[:, None, None] + torch.arange(b, device=DEV)[None, None, :]
        win = ref[:m][torch.arange(m, device=DEV)[:, None, None], rows, cols]
        sad = (win.to(torch.int32) - src[:m].to(torch.int32)).abs().sum(dim=(1, 2))
        assert torch.equal(sad.to(torch.int64), best[:m])
        at00 = (ref[:m, :b, :b].to(torch.int32) - src[:m].to(torch.int32)).abs().sum(dim=(1, 2))
        assert (best[:m] <= at00.to(torch.int64)).all()


def test_cfl_and_levels_full_size_properties(dsp):
    n = 1 << 19
    g = torch.Generator(device=DEV); g.manual_seed(11)
    w = h = 16
    luma = torch.randint(0, 256, (n, 2 * h, 2 * w), dtype=torch.uint8, device=DEV, generator=g)
    q3 = dsp.cfl_luma_subsampling_420(luma, 2 * w, 2 * w, 2 * h, luma_block_pitch=4 * w * h, n=n)
    want = (luma.to(torch.int32).reshape(n, h, 2, w, 2).sum(dim=(2, 4)) * 2).to(torch.int16)
    assert torch.equal(q3[:, :h, :w], want)
    ac = dsp.cfl_luma_subsampling_420(luma, 2 * w, 2 * w, 2 * h, luma_block_pitch=4 * w * h, n=n, subtract_average=True)
    # the mean is removed to within the rounding of the average: |sum of AC| <= w*h/2 per block
    assert (ac[:, :h, :w].to(torch.int32).sum(dim=(1, 2)).abs() <= w * h // 2).all()
    pred = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=DEV, generator=g)
    zero = torch.zeros(n, dtype=torch.int32, device=DEV)
    out = torch.empty_like(pred)
    dsp.cfl_predict(ac, pred, w, out, w, zero, 8, w, h)
    assert torch.equal(out, pred)                                # alpha = 0: the DC prediction itself
    # antisymmetry away from the clip: pred + t and pred - t for alpha and -alpha
    mid = torch.full_like(pred, 128)
    a = torch.full((n,), 3, dtype=torch.int32, device=DEV)
    up = dsp.cfl_predict(ac, mid, w, torch.empty_like(mid), w, a, 8, w, h).to(torch.int32)
    dn = dsp.cfl_predict(ac, mid, w, torch.empty_like(mid), w, -a, 8, w, h).to(torch.int32)
    inside = (up > 0) & (up < 255) & (dn > 0) & (dn < 255)
    assert torch.equal((up - 128)[inside], (128 - dn)[inside])
    coeff = torch.randint(-400, 401, (n, w * h), dtype=torch.int32, device=DEV, generator=g)
    size = (w + 4) * (h + 6) + 16
    lv = dsp.txb_init_levels(coeff, w, h, torch.full((n, size + 8), 0x33, dtype=torch.uint8, device=DEV))
    body = lv[:, :(w + 4) * (h + 6)].reshape(n, h + 6, w + 4)
    assert torch.equal(body[:, 2:2 + h, :w], coeff.abs().clamp(max=127).to(torch.uint8).reshape(n, h, w))
    assert not body[:, :2].any() and not body[:, 2 + h:].any() and not body[:, :, w:].any()
    assert not lv[:, (w + 4) * (h + 6):size].any() and (lv[:, size:] == 0x33).all()


def test_ois_1080p_properties(dsp):
    """every 16x16 block of a 1080p picture: the best index is the first minimum of its distortion row, the DC
    candidate equals the SAD against the block's DC value, and a vertically constant picture is predicted
    exactly by V_PRED away from the top border."""
    W, H, pad, b = 1920, 1080, 64, 16
    g = torch.Generator(device=DEV); g.manual_seed(5)
    plane = torch.randint(0, 256, (H + 2 * pad, W + 2 * pad), dtype=torch.uint8, device=DEV, generator=g)
    plane[pad + 512:pad + 768] = plane[pad + 511:pad + 512]                      # rows 512..767 repeat row 511
    blocks = [(x, y) for y in range(0, H - b + 1, b) for x in range(0, W - b + 1, b)]
    xy = torch.from_numpy(np.array([(y << 16) | x for x, y in blocks], np.uint32).view(np.int32)).to(DEV)
    modes, deltas = dsp.ois_candidates(b)
    dist, best = dsp.ois_search(plane[pad:, pad:], W + 2 * pad, W, H, xy, b, modes, deltas)
    dist = dist.to(torch.int64)
    mn = dist.min(dim=1).values
    first = (dist == mn[:, None]).int().argmax(dim=1)
    assert torch.equal(first.to(torch.int8), best)
    iv = int(np.nonzero((modes == 1) & (deltas == 0))[0][0])                     # V_PRED, angle 90
    ys = torch.tensor([y for _, y in blocks], device=DEV)
    inside = (ys >= 512) & (ys + b <= 768)
    assert (dist[inside, iv] == 0).all() and inside.sum() > 1000
