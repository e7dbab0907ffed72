"""CPU: the oracle (oracle/libsvt_oracle.so) against the committed golden
fixtures, which are outputs of the reference's own kernels (tests/golden/
make_golden.py).  This is what pins the oracle when /root/reference is absent."""
import ctypes
import os

import numpy as np
import pytest

import svtlibs
from svtlibs import TX_H, TX_SIZES, TX_TYPES, TX_W, ptr, txfm_allowed

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def tabs():
    return np.load(os.path.join(G, "tables.npz"))


def test_scan_tables(tabs):
    for s in range(19):
        for t in range(16):
            sc, isc = svtlibs.scan_tables(s, t)
            assert np.array_equal(sc, tabs[f"scan_{s}_{t}"]), (TX_SIZES[s], TX_TYPES[t])
            assert np.array_equal(isc, tabs[f"iscan_{s}_{t}"]), (TX_SIZES[s], TX_TYPES[t])


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_quantizer_tables(tabs, bd):
    t = svtlibs.quant_tables(bd)
    for k in ("zbin", "round", "quant", "quant_shift", "dequant"):
        assert np.array_equal(t[k], tabs[f"{k}_{bd}"]), k


def test_fwd_txfm2d_golden():
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "fwd_txfm2d.npz"))
    n = 0
    for s in range(19):
        w, h = TX_W[s], TX_H[s]
        for t in range(16):
            if not txfm_allowed(s, t):
                continue
            for bd in (8, 10):
                xin, out = g[f"{s}_{t}_{bd}_in"], g[f"{s}_{t}_{bd}_out"]
                for i in range(xin.shape[0]):
                    x = np.ascontiguousarray(xin[i]); o = np.zeros(w * h, np.int32)
                    O.svt_oracle_fwd_txfm2d(ptr(x), ptr(o), ctypes.c_uint32(w), t, s, bd)
                    assert np.array_equal(o, out[i]), (TX_SIZES[s], TX_TYPES[t], bd, i)
                    n += 1
    assert n == 954


def test_inv_txfm2d_add_golden():
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "inv_txfm2d_add.npz"))
    for s in range(19):
        w = TX_W[s]
        for t in range(16):
            if not txfm_allowed(s, t):
                continue
            for bd in (8, 10):
                co, d0, d1 = (g[f"{s}_{t}_{bd}_{k}"] for k in ("coeff", "dst_in", "dst_out"))
                for i in range(co.shape[0]):
                    d = np.ascontiguousarray(d0[i])
                    O.svt_oracle_inv_txfm2d_add(ptr(np.ascontiguousarray(co[i])), ptr(d), w, t, s, bd)
                    assert np.array_equal(d, d1[i]), (TX_SIZES[s], TX_TYPES[t], bd, i)


def test_quantize_b_golden(tabs):
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "quantize_b.npz"))
    cases = 0
    for s, ls in ((0, 0), (2, 0), (3, 1), (4, 2)):
        sc, isc = tabs[f"scan_{s}_0"], tabs[f"iscan_{s}_0"]
        n = len(sc)
        for bd in (8, 10):
            for q in (0, 1, 100, 255):
                rows = [np.ascontiguousarray(tabs[f"{k}_{bd}"][q]) for k in ("zbin", "round", "quant", "quant_shift", "dequant")]
                for kind in ("uniform", "sparse", "small", "dc", "zero"):
                    key = f"{s}_{bd}_{q}_{kind}"
                    co = np.ascontiguousarray(g[key + "_coeff"])
                    for variant, tag in ((0, "hbd"), (1, "cII")):
                        qc = np.zeros(n, np.int32); dqc = np.zeros(n, np.int32); eob = np.zeros(1, np.uint16)
                        O.svt_oracle_quantize_b(ptr(co), ctypes.c_ssize_t(n), 0, ptr(rows[0]), ptr(rows[1]), ptr(rows[2]),
                                                ptr(rows[3]), ptr(qc), ptr(dqc), ptr(rows[4]), ptr(eob),
                                                ptr(np.ascontiguousarray(sc)), ptr(np.ascontiguousarray(isc)), ls, variant)
                        assert np.array_equal(qc, g[f"{key}_q_{tag}"]), (key, tag)
                        assert np.array_equal(dqc, g[f"{key}_dq_{tag}"]), (key, tag)
                        assert eob[0] == g[f"{key}_eob_{tag}"][0], (key, tag)
                        cases += 1
    assert cases == 4 * 2 * 4 * 5 * 2


def test_pixel_golden():
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "pixel.npz"))
    for (w, h) in ((4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (24, 16), (48, 64), (8, 32), (64, 16)):
        k = f"{w}x{h}"
        a, b = g[k + "_a"], g[k + "_b"]
        for i in range(a.shape[0]):
            ai, bi = np.ascontiguousarray(a[i]), np.ascontiguousarray(b[i])
            assert O.svt_oracle_sad(ptr(ai), w, ptr(bi), w, h, w) == int(g[k + "_sad"][i])
            assert O.svt_oracle_sse(ptr(ai), w, ptr(bi), w, w, h) == int(g[k + "_sse"][i])
            r = np.zeros((h, w), np.int16)
            O.svt_oracle_residual(ptr(ai), w, ptr(bi), w, ptr(r), w, w, h)
            assert np.array_equal(r, g[k + "_res"][i])
    for (w, h, sw, sh) in ((16, 16, 8, 8), (16, 16, 1, 1), (8, 8, 16, 7), (32, 32, 13, 5), (64, 64, 9, 9), (4, 4, 8, 8)):
        k = f"loop_{w}x{h}_{sw}x{sh}"
        src, rf = g[k + "_src"], g[k + "_ref"]
        rw = w + sw - 1
        for i in range(src.shape[0]):
            b = np.zeros(1, np.uint64); x = np.zeros(1, np.int16); y = np.zeros(1, np.int16)
            O.svt_oracle_sad_loop(ptr(np.ascontiguousarray(src[i])), w, ptr(np.ascontiguousarray(rf[i])), rw, h, w,
                                  ptr(b), ptr(x), ptr(y), rw, ctypes.c_int16(sw), ctypes.c_int16(sh))
            assert (int(b[0]), int(x[0]), int(y[0])) == (int(g[k + "_best"][i]), int(g[k + "_x"][i]), int(g[k + "_y"][i])), (k, i)


def test_intra_golden():
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "intra.npz"))
    S = ctypes.c_ssize_t
    checked = 0
    for key in g.files:
        parts = key.split("_")
        if len(parts) < 3 or parts[2] in ("above", "left"):
            continue
        s, bd = int(parts[0]), int(parts[1])
        bw, bh = TX_W[s], TX_H[s]
        a = np.ascontiguousarray(g[f"{s}_{bd}_above"]); l = np.ascontiguousarray(g[f"{s}_{bd}_left"])
        es = a.itemsize
        pa = ctypes.c_void_p(a.ctypes.data + 16 * es); pl = ctypes.c_void_p(l.ctypes.data + 16 * es)
        o = np.zeros((bh, bw), a.dtype)
        if parts[2].startswith("m"):
            m = int(parts[2][1:])
            if bd == 8: O.svt_oracle_intra_pred(m, ptr(o), S(bw), bw, bh, pa, pl)
            else: O.svt_oracle_intra_pred_hbd(m, ptr(o), S(bw), bw, bh, pa, pl, bd)
        else:
            zone, ang, ua, ul = int(parts[2][1:]), int(parts[3]), int(parts[4][0]), int(parts[4][1])
            from dr_deriv_data import DR_DERIV
            dx = DR_DERIV[ang] if zone in (1, 2) else 1
            dy = DR_DERIV[90 - ang] if zone == 2 else (DR_DERIV[ang] if zone == 3 else 1)
            if bd == 8: O.svt_oracle_dr_prediction(zone, ptr(o), S(bw), bw, bh, pa, pl, ua, ul, dx, dy)
            else: O.svt_oracle_dr_prediction_hbd(zone, ptr(o), S(bw), bw, bh, pa, pl, ua, ul, dx, dy, bd)
        assert np.array_equal(o, g[key]), key
        checked += 1
    assert checked > 500


def test_cfl_levels_golden():
    """oracle/cfl.c against the reference's scalar C outputs (tests/golden/make_golden.py gen_cfl_levels)."""
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "cfl_levels.npz"))
    c_int = ctypes.c_int
    n_cfl = n_lv = 0
    for key in g.files:
        if key.startswith("cfl_") and key.endswith("_luma"):
            base = key[:-5]
            wh, bd = base.split("_")[1], int(base.split("_")[2])
            w, h = (int(v) for v in wh.split("x"))
            luma = g[key]
            for i in range(luma.shape[0]):
                li = np.ascontiguousarray(luma[i])
                q3 = np.full((32, 32), 77, np.int16)
                O.svt_oracle_cfl_luma_subsampling_420(ptr(li), c_int(bd > 8), c_int(li.shape[1]), ptr(q3), c_int(2 * w), c_int(2 * h))
                assert np.array_equal(q3, g[base + "_q3"][i]), (base, i)
                O.svt_oracle_subtract_average(ptr(q3), c_int(w), c_int(h), c_int(w * h // 2), c_int(int(np.log2(w * h))))
                assert np.array_equal(q3, g[base + "_ac"][i]), (base, i)
                pred = np.ascontiguousarray(g[base + "_pred"][i])
                out = np.zeros_like(pred)
                O.svt_oracle_cfl_predict(ptr(q3), ptr(pred), c_int(pred.shape[1]), ptr(out), c_int(pred.shape[1]),
                                         c_int(int(g[base + "_alpha"][i])), c_int(bd), c_int(w), c_int(h), c_int(bd > 8))
                assert np.array_equal(out, g[base + "_dst"][i]), (base, i)
                n_cfl += 1
        if key.startswith("lv_") and key.endswith("_coeff"):
            base = key[:-6]
            w, h = (int(v) for v in base.split("_")[1].split("x"))
            coeff = g[key]
            for i in range(coeff.shape[0]):
                lv = np.full((w + 4) * (h + 6) + 16, 0xAA, np.uint8)
                O.svt_oracle_txb_init_levels(ptr(np.ascontiguousarray(coeff[i])), c_int(w), c_int(h),
                                             ctypes.c_void_p(lv.ctypes.data + 2 * (w + 4)))
                assert np.array_equal(lv, g[base + "_levels"][i]), (base, i)
                n_lv += 1
    assert n_cfl == 14 * 2 * 3 and n_lv == 14 * 3


def ois_md_scan():
    out = []

    def rec(x, y, s):
        out.append((x, y, s))
        if s > 8:
            h = s // 2
            for (dx, dy) in ((0, 0), (h, 0), (0, h), (h, h)):
                rec(x + dx, y + dy, h)
    rec(0, 0, 64)
    return out


def ois_raster_idx(x, y, s):
    return {64: 0, 32: 1, 16: 5, 8: 21}[s] + (y // s) * (64 // s) + (x // s)


def test_ois_golden():
    """oracle/ois.c against the reference's own open_loop_intra_search_sb outputs (tests/golden/ois.npz)."""
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "ois.npz"))
    assert all(int(g["dr_intra_derivative"][a]) == O.svt_oracle_dr_intra_derivative(a) for a in range(90))
    buf = np.ascontiguousarray(g["pic"]); W, H, pad = (int(v) for v in g["dims"])
    stride = buf.shape[1]
    pic = ctypes.c_void_p(buf.ctypes.data + pad * stride + pad)
    md = ois_md_scan()
    c_int = ctypes.c_int
    blocks = 0
    for k, (sx, sy, tl, ipm, isref) in enumerate(g["cases"].tolist()):
        for i, (x, y, s) in enumerate(md):
            if not g[f"c{k}_valid"][ois_raster_idx(x, y, s)]:
                assert g[f"c{k}_count"][i] == 0
                continue
            m = np.zeros(61, np.uint8); d = np.zeros(61, np.int8); ds = np.zeros(61, np.uint32)
            n = O.svt_oracle_ois_candidates(c_int(s), c_int(tl), c_int(ipm), c_int(isref), c_int(0), ptr(m), ptr(d))
            bi = O.svt_oracle_ois_block(pic, c_int(stride), c_int(W), c_int(H), c_int(sx + x), c_int(sy + y), c_int(s), c_int(n),
                                        ptr(m), ptr(d), ptr(ds))
            assert n == int(g[f"c{k}_count"][i]) and bi == int(g[f"c{k}_best"][i]), (k, i)
            assert np.array_equal(m[:n], g[f"c{k}_mode"][i, :n]) and np.array_equal(d[:n], g[f"c{k}_delta"][i, :n])
            assert np.array_equal(ds[:n], g[f"c{k}_dist"][i, :n]), (k, i)
            blocks += 1
    assert blocks > 400


# ---- round 2: K6 (reference's own full-pel search drivers) and the caller-level pins (oracle/ref_me.c, ref_pins.c) ----
ME_MAX_SAD = 128 * 128 * 255


def me_oracle(src, win, stride, sw, sh, xo, yo, flavour, nsq, chain):
    O = svtlibs.oracle()
    bs = np.full(209, ME_MAX_SAD, np.uint32); bm = np.zeros(209, np.uint32)
    O.svt_oracle_me_sb_search_full(ptr(src), 64, ptr(win), stride, sw, sh, xo, yo, flavour, nsq, ptr(bs), ptr(bm))
    if chain:
        O.svt_oracle_me_sb_search_full(ptr(src), 64, ctypes.c_void_p(win.ctypes.data + stride + 1), stride, max(1, sw - 1), sh,
                                       xo + 1, yo + 1, flavour, nsq, ptr(bs), ptr(bm))
    n = 209 if nsq else 85
    return bs[:n], bm[:n]


def test_me_fullpel_golden_both_flavours_and_nsq():
    """oracle K6 == the reference's FullPelSearch_LCU / open_loop_me_fullpel_search_sblock for asm_type 0 and 1"""
    g = np.load(os.path.join(G, "me.npz"))
    cases = g["cases"]
    assert int(g["asm_divergent_cases"][0]) > 0      # the AVX2 build really diverges from the C kernels
    for k, (sw, sh, xo, yo, _) in enumerate(cases):
        src = np.ascontiguousarray(g[f"c{k}_src"]); win = np.ascontiguousarray(g[f"c{k}_win"])
        for asm in (0, 1):
            for nsq in (0, 1):
                bs, bm = me_oracle(src, win, win.shape[1], int(sw), int(sh), int(xo), int(yo), asm, nsq, k >= len(cases) - 2)
                assert np.array_equal(bs, g[f"c{k}_sad_a{asm}_n{nsq}"]), (k, asm, nsq)
                assert np.array_equal(bm, g[f"c{k}_mv_a{asm}_n{nsq}"]), (k, asm, nsq)


def test_me_legacy_layout_equals_full_layout():
    """svt_oracle_me_sb_search (8x8 | 16x16 | 32x32 | 64x64 back to back) == flavour 0 of the pinned function"""
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "me.npz"))
    for k, (sw, sh, xo, yo, _) in enumerate(g["cases"][:10]):
        src = np.ascontiguousarray(g[f"c{k}_src"]); win = np.ascontiguousarray(g[f"c{k}_win"])
        bs = np.full(85, ME_MAX_SAD, np.uint32); bm = np.zeros(85, np.uint32)
        O.svt_oracle_me_sb_search(ptr(src), 64, ptr(win), win.shape[1], int(sw), int(sh), int(xo), int(yo), ptr(bs), ptr(bm))
        reorder = lambda a: np.concatenate([a[84:85], a[80:84], a[64:80], a[0:64]])
        assert np.array_equal(reorder(bs), g[f"c{k}_sad_a0_n0"]) and np.array_equal(reorder(bm), g[f"c{k}_mv_a0_n0"]), k


def test_estimate_transform_pack64_energy_golden():
    """a5: forward transform + 64-point re-pack + three_quad_energy == the reference's av1_estimate_transform"""
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "pins.npz"))
    n = 0
    for key in [k[:-3] for k in g.files if k.startswith("est_") and k.endswith("_in")]:
        _, s, t, bd = key.split("_"); s, t, bd = int(s), int(t), int(bd)
        w, h = TX_W[s], TX_H[s]
        for i in range(3):
            x = np.ascontiguousarray(g[key + "_in"][i]); full = np.zeros(w * h, np.int32)
            O.svt_oracle_fwd_txfm2d(ptr(x), ptr(full), ctypes.c_uint32(w), t, s, bd)
            e = O.svt_oracle_fwd_txfm2d_pack64(ptr(full), s)
            m = min(w, 32) * min(h, 32)
            assert np.array_equal(full[:m], g[key + "_coeff"][i]), key
            assert int(e) == int(g[key + "_energy"][i]), key
            n += 1
    assert n >= 60


def test_inv_txfm_add_u8_entry_golden():
    """a6: svt_oracle_inv_txfm2d_add_u8 == av1_inv_txfm_add_c == av1_inv_txfm_add_ssse3 == av1_inv_transform_recon8bit"""
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "pins.npz"))
    n = 0
    for s in range(19):
        w = TX_W[s]
        for t in range(16):
            if not txfm_allowed(s, t):
                continue
            co, d0, d1 = g[f"inv8_{s}_{t}_coeff"], g[f"inv8_{s}_{t}_dst_in"], g[f"inv8_{s}_{t}_dst_out"]
            for i in range(2):
                d = np.ascontiguousarray(d0[i])
                O.svt_oracle_inv_txfm2d_add_u8(ptr(np.ascontiguousarray(co[i])), ptr(d), ctypes.c_int(w), t, s)
                assert np.array_equal(d, d1[i]), (TX_SIZES[s], TX_TYPES[t], i)
                n += 1
    assert n == 2 * 159


def test_full_distortion32_golden_c_and_avx2_flavours():
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "pins.npz"))
    diverged = 0
    for key in [k[:-2] for k in g.files if k.startswith("dist_") and k.endswith("_a")]:
        w, h = (int(v) for v in key.split("_")[1].split("x"))
        a, b, out, out2 = g[key + "_a"], g[key + "_b"], g[key + "_out"], g[key + "_out_avx2"]
        for i in range(4):
            aa = np.ascontiguousarray(a[i]); bb = np.ascontiguousarray(b[i])
            r = np.zeros(2, np.uint64); r2 = np.zeros(2, np.uint64)
            O.svt_oracle_full_distortion32(ptr(aa), ctypes.c_uint32(w + 8), ptr(bb), ctypes.c_uint32(w + 8), ptr(r), ctypes.c_uint32(w), ctypes.c_uint32(h))
            O.svt_oracle_full_distortion32_avx2(ptr(aa), ctypes.c_uint32(w + 8), ptr(bb), ctypes.c_uint32(w + 8), ptr(r2), ctypes.c_uint32(w), ctypes.c_uint32(h))
            assert np.array_equal(r, out[i, 0]), (key, i)
            assert np.array_equal(r2, out_avx2 := out2[i, 0]), (key, i)
            assert r[1] == out[i, 1, 0] == out[i, 1, 1]                    # cbf_zero: both outputs = prediction energy
            diverged += int(not np.array_equal(out[i, 0], out2[i, 0]))
            if w == h:                                                     # picture_full_distortion32_bits (luma)
                k = min(w, 32)
                ca = np.ascontiguousarray(a[i][:k, :k]); cb = np.ascontiguousarray(b[i][:k, :k])
                y = np.zeros(2, np.uint64)
                O.svt_oracle_full_distortion32(ptr(ca), ctypes.c_uint32(k), ptr(cb), ctypes.c_uint32(k), ptr(y), ctypes.c_uint32(k), ctypes.c_uint32(k))
                assert np.array_equal(y, g[f"pdist_{w}_{i}_1"]), (key, i)
                assert y[1] == g[f"pdist_{w}_{i}_0"][0] == g[f"pdist_{w}_{i}_0"][1]
    assert diverged > 0      # the reference's AVX2 kernel loses carries; both flavours are pinned


def test_inv_txfm2d_add_at_clamp_limits_golden():
    """bd 8 / 10 / 12 coefficients at +-2^(bd+7) and beyond: av1_inv_txfm2d_add_*_c (saturation of the final add)"""
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "pins.npz"))
    n = 0
    for key in [k[:-6] for k in g.files if k.startswith("sat_") and k.endswith("_coeff")]:
        _, s, t, bd = key.split("_"); s, t, bd = int(s), int(t), int(bd)
        for i in range(5):
            d = np.ascontiguousarray(g[key + "_dst_in"][i])
            O.svt_oracle_inv_txfm2d_add(ptr(np.ascontiguousarray(g[key + "_coeff"][i])), ptr(d), ctypes.c_int(TX_W[s]), t, s, bd)
            assert np.array_equal(d, g[key + "_dst_out"][i]), (key, i)
            n += 1
    assert n == 195


def test_hme_levels_golden():
    """n1: oracle HME level arithmetic (search-area derivation, clipping, sub-sampled SAD search, scaling) == the
    reference's own HmeLevel0 / HmeLevel1 / HmeLevel2 on 3 x 56 SB / centre / region cases (tests/golden/hme.npz)"""
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "hme.npz"))
    hme_w, hme_h = g["hme_w"], g["hme_h"]
    for level in range(3):
        W, H, sb, pad, stride = (int(v) for v in g[f"l{level}_dims"])
        src = np.ascontiguousarray(g[f"l{level}_src"]); ref = np.ascontiguousarray(g[f"l{level}_ref"])
        ref00 = ctypes.c_void_p(ref.ctypes.data + pad * stride + pad)
        for case, exp in zip(g[f"l{level}_cases"].tolist(), g[f"l{level}_out"].tolist()):
            lv, sbw, sbh, ox, oy, xc, yc, rw, rh, mx, my = case
            p = svtlibs.hme_params(lv, hme_w, hme_h, rw, rh, mx, my, pad, W, H)
            b = np.zeros(1, np.uint64); x = np.zeros(1, np.int16); y = np.zeros(1, np.int16)
            O.svt_oracle_hme_level(ptr(src), W, ref00, stride, ox, oy, sbw, sbh, xc, yc, ctypes.byref(p), ptr(b), ptr(x), ptr(y))
            assert [int(b[0]), int(x[0]), int(y[0])] == exp, case


def test_intra_availability_golden():
    """has_top_right / has_bottom_left: every enumerated argument tuple against the reference's table look-ups (packed bits)"""
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "bip.npz"))
    for sb_mi in (16, 32):
        n = int(g[f"avail_count_sb{sb_mi}"][0])
        tr = np.unpackbits(g[f"has_tr_sb{sb_mi}"])[:n]
        bl = np.unpackbits(g[f"has_bl_sb{sb_mi}"])[:n]
        i = 0
        for args in svtlibs.availability_tuples(sb_mi):
            assert O.svt_oracle_has_top_right(sb_mi, *args) == tr[i], args
            assert O.svt_oracle_has_bottom_left(sb_mi, *args) == bl[i], args
            i += 1
        assert i == n


def test_build_intra_predictors_golden():
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "bip.npz"))
    n = 0
    for p, top, left, exp in svtlibs.bip_fixture_cases(g):
        es = top.dtype.itemsize
        h, w = exp.shape
        d = np.zeros((h, w), top.dtype)
        O.svt_oracle_build_intra_predictors(p["is16"], ctypes.c_void_p(top.ctypes.data + 16 * es), ctypes.c_void_p(left.ctypes.data + 16 * es),
                                            ptr(d), w, p["mode"], p["angle_delta"], p["tx"], p["disable_edge_filter"], p["n_top"], p["n_tr"],
                                            p["n_left"], p["n_bl"], p["filt_type"], p["bd"])
        assert np.array_equal(d, exp), p
        n += 1
    assert n == 700


def test_predict_intra_block_golden():
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "bip.npz"))
    n = 0
    for c, exp in svtlibs.pib_fixture_cases(g):
        got, out5 = svtlibs.oracle_predict_intra_block(O, c)
        assert np.array_equal(got, exp), ({k: v for k, v in c.items() if np.isscalar(v)}, out5)
        n += 1
    assert n == 300


def test_picture_input_golden():
    """n4: y4m header parsing == the reference application's read_y4m_header on files; pad_input_picture + generate_padding and
    Decimation2D + generate_padding == the reference library's (tests/golden/picture.npz)"""
    O = svtlibs.oracle()
    g = np.load(os.path.join(G, "picture.npz"))
    for line, exp in zip(g["y4m_lines"].tolist(), g["y4m_out"].tolist()):
        info = svtlibs.Y4mInfo()
        rc = O.svt_oracle_y4m_parse_header(line.encode(), ctypes.byref(info))
        assert exp[0] == 1 and (rc == 0) == (exp[1] == 0), line
        if rc == 0:
            assert [info.width, info.height, info.fr_n, info.fr_d, info.bit_depth, info.interlaced] == exp[2:8], line
    for k in g["pad_cases"].tolist():
        w, h, ox, oy, pr, pb, is16, stride = (int(v) for v in g[f"pad{k}_prm"])
        exp = g[f"pad{k}_out"]
        buf = np.full_like(exp, 0x55)
        buf[oy:oy + h, ox:ox + w] = g[f"pad{k}_frame"]
        es = 2 if is16 else 1
        O.svt_oracle_pad_input_picture(ctypes.c_void_p(buf.ctypes.data + (oy * stride + ox) * es), stride, w, h, pr, pb, es)
        O.svt_oracle_generate_padding(ptr(buf), stride, w + pr, h + pb, ox, oy, es)
        assert np.array_equal(buf, exp), k
    for k in g["dec_cases"].tolist():
        w, h, stride, qo, so = (int(v) for v in g[f"dec{k}_prm"])
        luma = np.ascontiguousarray(g[f"dec{k}_luma"])
        for step, o, key in ((2, qo, "q"), (4, so, "s")):
            exp = g[f"dec{k}_{key}"]
            dw, dh = (w + step - 1) // step, (h + step - 1) // step
            buf = np.full_like(exp, 0x33)
            ds = exp.shape[1]
            O.svt_oracle_decimation_2d(ptr(luma), stride, w, h, ctypes.c_void_p(buf.ctypes.data + o * ds + o), ds, step)
            O.svt_oracle_generate_padding(ptr(buf), ds, dw, dh, o, o, 1)
            assert np.array_equal(buf, exp), (k, key)


def test_motion_estimate_lcu_golden():
    """the oracle's per-SB motion estimation driver (oracle/me_lcu.c: HME levels -> best-of-regions centre -> CheckZeroZeroCenter ->
    search-area clipping -> 85 / 209-PU full-pel search -> bi-prediction -> me_results) against the reference's OWN MotionEstimateLcu
    on every SB of three small pictures under fifteen parameter sets (tests/golden/me_setup.npz, made by make_golden.gen_me_setup)"""
    O = svtlibs.oracle()
    g, pics = svtlibs.me_setup_fixture()
    assert g["prm"].shape[0] >= 180
    for i, prm in enumerate(g["prm"]):
        (ps, _), (p0, _), (p1, _) = pics[str(g["picture_of_set"][g["param_set"][i]])]
        o = svtlibs.run_me_lcu(O.svt_oracle_me_lcu, np.ascontiguousarray(prm), ps, p0, p1)
        for k, v in o.items():
            assert np.array_equal(v, g[k][i]), (i, prm[:27].tolist(), k)


def test_me_pu_raster_to_storage_map_is_a_bijection_and_matches_the_rectangles():
    """raster PU order (me_results, BiPredictionSearch) -> index in the SAD / vector rows; pinned to the reference through the
    me_results rows of the fixture above, checked here for shape"""
    O = svtlibs.oracle()
    m = [O.svt_oracle_me_raster_to_storage(p) for p in range(209)]
    assert sorted(m) == list(range(209))
    assert m[:5] == [0, 1, 2, 3, 4] and m[5:9] == [5, 6, 9, 10] and m[85:87] == [85, 86] and m[87:91] == [87, 89, 88, 90]
