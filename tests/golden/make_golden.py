#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE:
the reference's own C (and, where it exists, AVX2) kernels compiled from
/root/reference into oracle/_ref/libsvtref.so (oracle/Makefile) are called through
ctypes on seeded inputs; inputs and the reference's outputs are stored as
compressed .npz files.  The fixtures are data only (no reference source text).

The reference holds no stored vectors for this path: its pins are
the procedures of FwdTxfm2dAsmTest / InvTxfm2dAsmTest / QuantizeTest, which is
what the input recipes below follow (uniform +-(2^bd - 1) residuals, coefficients
from the forward transform, coefficients in +-2^(7+bd), q sweep).  SAD / SSE /
residual / intra have no reference unit test at all; their fixtures are the
outputs of the reference's scalar C functions.

Run here (needs /root/reference):   python tests/golden/make_golden.py
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from svtlibs import TX_H, TX_SIZES, TX_TYPES, TX_W, ptr, ref, txfm_allowed  # noqa: E402

c_int = ctypes.c_int
R = ref()
assert R is not None, "oracle/_ref/libsvtref.so missing: run `make -C oracle ref`"


def fwd_name(s):
    w, h = TX_W[s], TX_H[s]
    return f"Av1TransformTwoD_{w}x{h}_c" if w == h else f"av1_fwd_txfm2d_{w}x{h}_c"


def ref_fwd(s, t, bd, x):
    w, h = TX_W[s], TX_H[s]
    o = np.zeros(w * h, np.int32)
    getattr(R, fwd_name(s))(ptr(x), ptr(o), ctypes.c_uint32(x.shape[1]), c_int(t), ctypes.c_uint8(bd))
    return o


SQ = {0, 1, 2, 3, 4}
NO_EOB = {5, 6, 13, 14}


def ref_inv(s, t, bd, co, dst):
    w, h = TX_W[s], TX_H[s]
    f = getattr(R, f"av1_inv_txfm2d_add_{w}x{h}_c")
    stride = dst.shape[1]
    if s in SQ:
        f(ptr(co), ptr(dst), c_int(stride), c_int(t), c_int(bd))
    elif s in NO_EOB:
        f(ptr(co), ptr(dst), c_int(stride), c_int(t), c_int(s), c_int(bd))
    else:
        f(ptr(co), ptr(dst), c_int(stride), c_int(t), c_int(s), c_int(len(co)), c_int(bd))


def pack64(co, s):
    w, h = TX_W[s], TX_H[s]
    kw, kh = min(w, 32), min(h, 32)
    return np.ascontiguousarray(co.reshape(h, w)[:kh, :kw]).reshape(-1)


def gen_txfm():
    rng = np.random.default_rng(13596)
    fw = {}
    iv = {}
    for s in range(19):
        w, h = TX_W[s], TX_H[s]
        kw, kh = min(w, 32), min(h, 32)
        for t in range(16):
            if not txfm_allowed(s, t):
                continue
            for bd in (8, 10):
                m = (1 << bd) - 1
                xs = [rng.integers(-m, m + 1, size=(h, w)).astype(np.int16),
                      np.full((h, w), m, np.int16),
                      (np.indices((h, w)).sum(0) % 2 * 2 * m - m).astype(np.int16)]
                xin = np.stack(xs)
                out = np.stack([ref_fwd(s, t, bd, x) for x in xs])
                key = f"{s}_{t}_{bd}"
                fw[key + "_in"] = xin
                fw[key + "_out"] = out
                # inverse: coefficients of the forward transform (packed), eob-truncated variant, random dst
                cos = [pack64(out[0], s), pack64(out[2], s)]
                tr = cos[0].copy(); tr[max(1, len(tr) // 5):] = 0
                cos.append(tr)
                cos = np.stack(cos).astype(np.int32)
                dst0 = rng.integers(0, 1 << bd, size=(3, h, w)).astype(np.uint16)
                dst1 = dst0.copy()
                for i in range(3):
                    ref_inv(s, t, bd, cos[i], dst1[i])
                iv[key + "_coeff"] = cos
                iv[key + "_dst_in"] = dst0
                iv[key + "_dst_out"] = dst1
    np.savez_compressed(os.path.join(HERE, "fwd_txfm2d.npz"), **fw)
    np.savez_compressed(os.path.join(HERE, "inv_txfm2d_add.npz"), **iv)


def gen_tables():
    d = {}
    for s in range(19):
        n = min(TX_W[s], 32) * min(TX_H[s], 32)
        for t in range(16):
            d[f"scan_{s}_{t}"] = np.ctypeslib.as_array(R.ref_get_scan(s, t, 0), shape=(n,)).copy()
            d[f"iscan_{s}_{t}"] = np.ctypeslib.as_array(R.ref_get_scan(s, t, 1), shape=(n,)).copy()
    for bd in (8, 10, 12):
        q = np.zeros((18, 256, 8), np.int16)
        dq = np.zeros((6, 256, 8), np.int16)
        R.av1_build_quantizer(c_int(bd), 0, 0, 0, 0, 0, ptr(q), ptr(dq))   # y tables: Quants fields 0..3, Dequants 0
        d[f"quant_{bd}"] = q[0]; d[f"quant_shift_{bd}"] = q[1]; d[f"zbin_{bd}"] = q[2]; d[f"round_{bd}"] = q[3]
        d[f"dequant_{bd}"] = dq[0]
    cos = (ctypes.c_int32 * (7 * 64)).in_dll(R, "av1_cospi_arr_data")
    d["cospi"] = np.array(cos[:], np.int32).reshape(7, 64)
    sin = (ctypes.c_int32 * (7 * 5)).in_dll(R, "av1_sinpi_arr_data")
    d["sinpi"] = np.array(sin[:], np.int32).reshape(7, 5)
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **d)
    return d


QNAMES = {0: ("aom_highbd_quantize_b_c", "aom_quantize_b_c_II", "aom_highbd_quantize_b_avx2"),
          1: ("aom_highbd_quantize_b_32x32_c", "aom_quantize_b_32x32_c_II", "aom_highbd_quantize_b_32x32_avx2"),
          2: ("aom_highbd_quantize_b_64x64_c", "aom_quantize_b_64x64_c_II", "aom_highbd_quantize_b_64x64_avx2")}


def gen_quant(tabs):
    rng = np.random.default_rng(13597)
    d = {}
    for s, ls in ((0, 0), (2, 0), (3, 1), (4, 2)):
        n = min(TX_W[s], 32) * min(TX_H[s], 32)
        sc, isc = tabs[f"scan_{s}_0"], tabs[f"iscan_{s}_0"]
        for bd in (8, 10):
            for q in (0, 1, 100, 255):
                kinds = {"uniform": rng.integers(-(1 << (7 + bd)), (1 << (7 + bd)) + 1, size=n),
                         "sparse": np.where(rng.random(n) < 0.9, 0, rng.integers(-3000, 3001, size=n)),
                         "small": rng.integers(-30, 31, size=n),
                         "dc": np.r_[rng.integers(-30000, 30001), np.zeros(n - 1, int)],
                         "zero": np.zeros(n, int)}
                for kname, co in kinds.items():
                    co = np.ascontiguousarray(co.astype(np.int32))
                    rows = [np.ascontiguousarray(tabs[f"{k}_{bd}"][q]) for k in ("zbin", "round", "quant", "quant_shift", "dequant")]
                    key = f"{s}_{bd}_{q}_{kname}"
                    d[key + "_coeff"] = co
                    for vi, fn in enumerate(QNAMES[ls]):
                        qc = np.zeros(n, np.int32); dqc = np.zeros(n, np.int32); eob = np.zeros(1, np.uint16)
                        getattr(R, fn)(ptr(co), ctypes.c_ssize_t(n), c_int(0), ptr(rows[0]), ptr(rows[1]), ptr(rows[2]),
                                       ptr(rows[3]), ptr(qc), ptr(dqc), ptr(rows[4]), ptr(eob), ptr(sc), ptr(isc))
                        tag = ("hbd", "cII", "avx2")[vi]
                        if tag == "avx2":   # production path must equal the highbd C path
                            assert np.array_equal(qc, d[key + "_q_hbd"]) and np.array_equal(dqc, d[key + "_dq_hbd"]) and eob[0] == d[key + "_eob_hbd"][0], key
                            continue
                        d[key + "_q_" + tag] = qc; d[key + "_dq_" + tag] = dqc; d[key + "_eob_" + tag] = eob
    np.savez_compressed(os.path.join(HERE, "quantize_b.npz"), **d)


def gen_pixel():
    rng = np.random.default_rng(13598)
    d = {}
    R.fast_loop_nx_m_sad_kernel.restype = ctypes.c_uint32
    R.spatial_full_distortion_kernel.restype = ctypes.c_uint64
    for (w, h) in ((4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (24, 16), (48, 64), (8, 32), (64, 16)):
        a = rng.integers(0, 256, size=(4, h, w), dtype=np.uint8)
        b = rng.integers(0, 256, size=(4, h, w), dtype=np.uint8)
        a[0] = 255; b[0] = 0; b[1] = a[1]
        sad = np.array([R.fast_loop_nx_m_sad_kernel(ptr(a[i]), w, ptr(b[i]), w, h, w) for i in range(4)], np.uint32)
        sse = np.array([R.spatial_full_distortion_kernel(ptr(a[i]), w, ptr(b[i]), w, w, h) for i in range(4)], np.uint64)
        res = np.zeros((4, h, w), np.int16)
        for i in range(4):
            R.residual_kernel_c(ptr(a[i]), w, ptr(b[i]), w, ptr(res[i]), w, w, h)
        k = f"{w}x{h}"
        d[k + "_a"] = a; d[k + "_b"] = b; d[k + "_sad"] = sad; d[k + "_sse"] = sse; d[k + "_res"] = res
    # sad_loop_kernel: incl. ties, constant planes, 1x1 search, leftover widths
    for (w, h, sw, sh) in ((16, 16, 8, 8), (16, 16, 1, 1), (8, 8, 16, 7), (32, 32, 13, 5), (64, 64, 9, 9), (4, 4, 8, 8)):
        n = 6
        rw, rh = w + sw - 1, h + sh - 1
        src = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
        rf = rng.integers(0, 256, size=(n, rh, rw), dtype=np.uint8)
        rf[0] = 7; src[0] = 9
        if sw > 3 and sh > 2:
            rf[1, 2:2 + h, 3:3 + w] = src[1]
            rf[2, 0:h, 1:1 + w] = src[2]; rf[2, 1:1 + h, 0:w] = src[2]
        best = np.zeros(n, np.uint64); xs = np.zeros(n, np.int16); ys = np.zeros(n, np.int16)
        for i in range(n):
            R.sad_loop_kernel(ptr(src[i]), w, ptr(rf[i]), rw, h, w, ptr(best[i:i + 1]), ptr(xs[i:i + 1]), ptr(ys[i:i + 1]),
                              rw, ctypes.c_int16(sw), ctypes.c_int16(sh))
            # the production AVX2 kernel must agree wherever it is defined (w in {8,16,24,32,48,64}).
            # Its 4-wide path returns different SADs from the scalar C kernel on the same
            # inputs (probed here: 200/200 trials differ, an exact copy scores 360 not 0), so
            # w = 4 is pinned by the scalar C function alone - recorded in DESIGN.md.
            if w == 4:
                continue
            b2 = np.zeros(1, np.uint64); x2 = np.zeros(1, np.int16); y2 = np.zeros(1, np.int16)
            R.sad_loop_kernel_avx2_intrin(ptr(src[i]), w, ptr(rf[i]), rw, h, w, ptr(b2), ptr(x2), ptr(y2), rw,
                                          ctypes.c_int16(sw), ctypes.c_int16(sh))
            assert (best[i], xs[i], ys[i]) == (b2[0], x2[0], y2[0]), (w, h, sw, sh, i)
        k = f"loop_{w}x{h}_{sw}x{sh}"
        d[k + "_src"] = src; d[k + "_ref"] = rf; d[k + "_best"] = best; d[k + "_x"] = xs; d[k + "_y"] = ys
    np.savez_compressed(os.path.join(HERE, "pixel.npz"), **d)


INTRA_NAMES = ["dc", "v", "h", "smooth", "smooth_v", "smooth_h", "paeth", "dc_top", "dc_left", "dc_128"]
DR_DERIV = {3: 1023, 6: 547, 9: 372, 14: 273, 17: 215, 20: 178, 23: 151, 26: 132, 29: 116, 32: 102, 36: 90, 39: 80,
            42: 71, 45: 64, 48: 57, 51: 51, 54: 45, 58: 40, 61: 35, 64: 31, 67: 27, 70: 23, 73: 19, 76: 15, 81: 11,
            84: 7, 87: 3}


def gen_intra():
    """No reference unit test covers intra prediction: the fixtures are the outputs of the
    reference's scalar C predictors aom_*_predictor_WxH_c / av1_*dr_prediction_z*_c."""
    rng = np.random.default_rng(13599)
    d = {}
    S = ctypes.c_ssize_t
    for s in range(19):
        bw, bh = TX_W[s], TX_H[s]
        for bd in (8, 10):
            dt = np.uint8 if bd == 8 else np.uint16
            es = 1 if bd == 8 else 2
            a = rng.integers(0, 1 << bd, size=304).astype(dt)
            l = rng.integers(0, 1 << bd, size=304).astype(dt)
            pa = ctypes.c_void_p(a.ctypes.data + 16 * es); pl = ctypes.c_void_p(l.ctypes.data + 16 * es)
            key = f"{s}_{bd}"
            d[key + "_above"] = a; d[key + "_left"] = l
            for m, name in enumerate(INTRA_NAMES):
                o = np.zeros((bh, bw), dt)
                if bd == 8:
                    getattr(R, f"aom_{name}_predictor_{bw}x{bh}_c")(ptr(o), S(bw), pa, pl)
                else:
                    getattr(R, f"aom_highbd_{name}_predictor_{bw}x{bh}_c")(ptr(o), S(bw), pa, pl, c_int(bd))
                d[f"{key}_m{m}"] = o
            if s in (0, 1, 2, 3, 5, 8, 9, 16):
                for zone in (1, 2, 3):
                    for ang in (3, 23, 45, 67, 87):
                        for up in ((0, 0), (1, 1)) if bw + bh <= 16 else ((0, 0),):
                            dx = DR_DERIV[ang] if zone in (1, 2) else 1
                            dy = DR_DERIV[90 - ang] if zone == 2 else (DR_DERIV[ang] if zone == 3 else 1)
                            o = np.zeros((bh, bw), dt)
                            pre = "av1_" if bd == 8 else "av1_highbd_"
                            f = getattr(R, f"{pre}dr_prediction_z{zone}_c")
                            args = [ptr(o), S(bw), bw, bh, pa, pl] + ([up[0]] if zone == 1 else [up[1]] if zone == 3 else [up[0], up[1]]) + [dx, dy]
                            if bd != 8:
                                args.append(bd)
                            f(*args)
                            d[f"{key}_z{zone}_{ang}_{up[0]}{up[1]}"] = o
    np.savez_compressed(os.path.join(HERE, "intra.npz"), **d)


CFL_SHAPES = [(4, 4), (8, 8), (16, 16), (32, 32), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (4, 16), (16, 4), (8, 32), (32, 8)]


def gen_cfl_levels():
    """K11 chroma-from-luma helpers and av1_txb_init_levels: outputs of the reference's scalar C functions
    (no reference unit test covers them).  Shapes are CHROMA block sizes (the luma block is twice as large)."""
    rng = np.random.default_rng(13600)
    d = {}
    for (w, h) in CFL_SHAPES:
        for bd in (8, 10):
            dt = np.uint8 if bd == 8 else np.uint16
            ls = 2 * w + 5
            luma = rng.integers(0, 1 << bd, size=(3, 2 * h, ls)).astype(dt)
            luma[1] = (1 << bd) - 1
            q3 = np.full((3, 32, 32), 77, np.int16)
            ac = np.zeros_like(q3)
            for i in range(3):
                getattr(R, "cfl_luma_subsampling_420_lbd_c" if bd == 8 else "cfl_luma_subsampling_420_hbd_c")(
                    ptr(luma[i]), c_int(ls), ptr(q3[i]), c_int(2 * w), c_int(2 * h))
                ac[i] = q3[i]
                R.subtract_average_c(ptr(ac[i]), c_int(w), c_int(h), c_int(w * h // 2), c_int(int(np.log2(w * h))))
            key = f"cfl_{w}x{h}_{bd}"
            d[key + "_luma"] = luma; d[key + "_q3"] = q3; d[key + "_ac"] = ac
            ps = w + 3
            pred = rng.integers(0, 1 << bd, size=(3, h, ps)).astype(dt)
            pred[1] = (1 << bd) - 1; pred[2] = 0
            alphas = np.array([-16, 5, 16], np.int32)
            out = np.zeros((3, h, ps), dt)
            for i in range(3):
                getattr(R, "cfl_predict_lbd_c" if bd == 8 else "cfl_predict_hbd_c")(
                    ptr(ac[i]), ptr(pred[i]), c_int(ps), ptr(out[i]), c_int(ps), c_int(int(alphas[i])), c_int(bd), c_int(w), c_int(h))
            d[key + "_pred"] = pred; d[key + "_alpha"] = alphas; d[key + "_dst"] = out
    for (w, h) in ((4, 4), (8, 8), (16, 16), (32, 32), (4, 8), (8, 4), (16, 32), (32, 16), (4, 16), (16, 4), (8, 32), (32, 8), (8, 16), (16, 8)):
        coeff = rng.integers(-300, 301, size=(3, h, w)).astype(np.int32)
        coeff[1] = rng.integers(-(1 << 20), 1 << 20, size=(h, w)); coeff[1, 0, 0] = -(1 << 31) + 1; coeff[1, 0, 1] = (1 << 31) - 1; coeff[2] = 0
        size = (w + 4) * (h + 6) + 16
        lv = np.full((3, size), 0xAA, np.uint8)
        for i in range(3):
            R.av1_txb_init_levels_c(ptr(coeff[i]), c_int(w), c_int(h), ctypes.c_void_p(lv[i].ctypes.data + 2 * (w + 4)))
        d[f"lv_{w}x{h}_coeff"] = coeff; d[f"lv_{w}x{h}_levels"] = lv
    np.savez_compressed(os.path.join(HERE, "cfl_levels.npz"), **d)


def ois_md_scan():
    """(x, y, size) of the 85 blocks of a 64x64 SB in the reference's MD-scan (depth-first, z-order) order"""
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


OIS_CASES = [(0, 0, 0, 0, 1), (64, 64, 0, 0, 1), (192, 128, 0, 0, 1), (128, 64, 1, 0, 1), (64, 0, 0, 5, 1), (0, 64, 0, 4, 0),
             (128, 128, 0, 0, 1), (192, 0, 0, 0, 1)]


def gen_ois():
    """Open-loop intra search: the outputs of the reference's OWN open_loop_intra_search_sb
    (EbMotionEstimation.c:8694) run through oracle/ref_ois.c on a 200x136 picture (partial SBs at the right and
    bottom edges), for SBs at the picture corner / edges / interior and the temporal-layer / preset switches."""
    rng = np.random.default_rng(13601)
    W, H, pad = 200, 136, 64
    buf = rng.integers(0, 256, size=(H + 2 * pad, W + 2 * pad), dtype=np.uint8)
    buf[pad + 64:pad + 128, pad + 64:pad + 128] //= 8           # a smooth region: ties between candidates
    buf[pad:pad + 32, pad + 128:pad + 160] = 200
    md = ois_md_scan()
    d = {"pic": buf, "dims": np.array([W, H, pad], np.int32), "cases": np.array(OIS_CASES, np.int32)}
    for k, (sx, sy, tl, ipm, isref) in enumerate(OIS_CASES):
        valid = np.zeros(85, np.uint8)
        for (x, y, s) in md:
            valid[ois_raster_idx(x, y, s)] = sx + x + s <= W and sy + y + s <= H
        cnt = np.zeros(85, np.uint8); best = np.zeros(85, np.int8); mode = np.zeros((85, 61), np.uint8)
        delta = np.zeros((85, 61), np.int8); dist = np.zeros((85, 61), np.uint32)
        rc = R.ref_ois_sb(ptr(buf), c_int(W + 2 * pad), c_int(pad), c_int(pad), c_int(W), c_int(H), c_int(sx), c_int(sy), ptr(valid),
                          c_int(tl), c_int(ipm), c_int(isref), ptr(cnt), ptr(best), ptr(mode), ptr(delta), ptr(dist))
        assert rc == 0
        d[f"c{k}_valid"] = valid; d[f"c{k}_count"] = cnt; d[f"c{k}_best"] = best; d[f"c{k}_mode"] = mode
        d[f"c{k}_delta"] = delta; d[f"c{k}_dist"] = dist
    d["dr_intra_derivative"] = np.array((ctypes.c_uint16 * 90).in_dll(R, "dr_intra_derivative"), np.uint16)
    np.savez_compressed(os.path.join(HERE, "ois.npz"), **d)



def aligned(shape, dt, al=64):
    """numpy array whose data pointer is `al`-byte aligned (the reference's AVX2 kernels use aligned loads / stores,
    as the encoder's own buffers are aligned)"""
    n = int(np.prod(shape)); it = np.dtype(dt).itemsize
    raw = np.zeros(n * it + al, np.uint8)
    off = (-raw.ctypes.data) % al
    return raw[off:off + n * it].view(dt).reshape(shape)


ME_PUS = 209
ME_MAX_SAD = 128 * 128 * 255        # MAX_SAD_VALUE, EbMotionEstimation.h:79
# (search_w, search_h, x_origin, y_origin, content): content 0 random, 1 coarse (many ties), 2 very coarse, 3 constant (max SAD)
ME_CASES = [(8, 1, 0, 0, 0), (16, 3, -8, -1, 0), (24, 2, -12, -1, 1), (64, 4, -32, -2, 0), (8, 8, -4, -4, 2),
            (7, 3, -3, -1, 0), (5, 2, 2, 3, 1), (1, 1, 0, 0, 0), (21, 2, -10, 0, 0), (13, 3, -40, -20, 1),
            (16, 2, -8, -1, 3), (40, 5, -37, 6, 2), (3, 1, -1, 0, 3), (32, 2, 5, -9, 1)]


def gen_me():
    """K6: the reference's OWN full-pel search drivers FullPelSearch_LCU / open_loop_me_fullpel_search_sblock
    (EbMotionEstimation.c:3199, 3251), reached through oracle/ref_me.c, for asm_type 0 (C / SSE4.1 kernels) and
    asm_type 1 (the AVX2 8-search-point production kernels), square PUs only (nsq 0) and with the non-square PUs
    (nsq 1), in the reference's p_sb_best_sad / p_sb_best_mv layout (209 entries).  The last two cases chain a second
    search onto the running bests of the first (IN/OUT semantics).  Divergences between the two asm types are NOT
    asserted away here - they are what `flavour` in the oracle / product restates."""
    rng = np.random.default_rng(13602)
    d = {"cases": np.array(ME_CASES, np.int32)}
    ndiff = 0
    for k, (sw, sh, xo, yo, content) in enumerate(ME_CASES):
        stride = 64 + sw + 8 + (k % 5) * 3
        rows = 64 + sh + 4
        src = rng.integers(0, 256, size=(64, 64), dtype=np.uint8)
        win = rng.integers(0, 256, size=(rows, stride), dtype=np.uint8)
        if content == 1:
            src = (src >> 6) << 6; win = (win >> 6) << 6
        elif content == 2:
            src = (src >> 7) << 7; win = (win >> 7) << 7
        elif content == 3:
            src[:] = 0; win[:] = 255
        src = np.ascontiguousarray(src); win = np.ascontiguousarray(win)
        d[f"c{k}_src"] = src; d[f"c{k}_win"] = win
        res = {}
        for asm in (0, 1):
            for nsq in (0, 1):
                bs = np.zeros(ME_PUS, np.uint32); bm = np.zeros(ME_PUS, np.uint32)
                rc = R.ref_me_fullpel(ptr(src), 64, ptr(win), stride, xo, yo, sw, sh, asm, nsq, 1, ptr(bs), ptr(bm))
                assert rc == 0
                if k >= len(ME_CASES) - 2:      # chain: search again, shifted by one row / column, on the running bests
                    rc = R.ref_me_fullpel(ptr(src), 64, ctypes.c_void_p(win.ctypes.data + stride + 1), stride, xo + 1, yo + 1,
                                          max(1, sw - 1), sh, asm, nsq, 0, ptr(bs), ptr(bm))
                    assert rc == 0
                n = ME_PUS if nsq else 85
                d[f"c{k}_sad_a{asm}_n{nsq}"] = bs[:n].copy(); d[f"c{k}_mv_a{asm}_n{nsq}"] = bm[:n].copy()
                res[(asm, nsq)] = (bs[:n].copy(), bm[:n].copy())
        for nsq in (0, 1):
            ndiff += int(not (np.array_equal(res[(0, nsq)][0], res[(1, nsq)][0]) and np.array_equal(res[(0, nsq)][1], res[(1, nsq)][1])))
    d["asm_divergent_cases"] = np.array([ndiff], np.int32)
    np.savez_compressed(os.path.join(HERE, "me.npz"), **d)
    print("me.npz: (case, nsq) pairs on which asm_type 0 and 1 disagree:", ndiff)


def gen_pins():
    """Caller-level functions of the transform path, through oracle/ref_pins.c (the reference's own functions, its own
    RTCD dispatch):  av1_estimate_transform incl. the 64-point re-pack + three_quad_energy (EbTransforms.c:4918-5292,
    4351-4408); the 8-bit reconstruction entry av1_inv_txfm_add_c (:8882) == av1_inv_txfm_add_ssse3 (production slot)
    == av1_inv_transform_recon8bit (:8939); full_distortion_kernel32_bits / _cbf_zero32_bits (EbPictureOperators.c:283-346,
    C == AVX2 asserted) and picture_full_distortion32_bits (:349); av1_inv_txfm2d_add_*_c at the clamp limits (bd 8/10/12)."""
    rng = np.random.default_rng(13603)
    d = {}
    # ---- av1_estimate_transform
    for s in (4, 11, 12, 17, 18, 3, 2, 9, 16):           # the five 64-point sizes, 32x32, 16x16, 16x32, 32x8
        w, h = TX_W[s], TX_H[s]
        n = min(w, 32) * min(h, 32)
        for t in (0, 9):
            if not txfm_allowed(s, t):
                continue
            for bi in (0, 1):
                bd = 10 if bi else 8
                m = (1 << bd) - 1
                xs = np.stack([rng.integers(-m, m + 1, size=(h, w)), np.full((h, w), m),
                               (np.indices((h, w)).sum(0) % 2 * 2 * m - m)]).astype(np.int16)
                cos = np.zeros((3, n), np.int32); es = np.zeros(3, np.uint64)
                for i in range(3):
                    x = aligned((h, 64), np.int16); x[:, :w] = xs[i]
                    co = aligned(w * h + 64, np.int32); e = np.zeros(1, np.uint64)
                    rc = R.ref_estimate_transform(ptr(x), ctypes.c_uint32(64), ptr(co), c_int(s), c_int(t), c_int(bi), ptr(e))
                    assert rc == 0
                    cos[i] = co[:n]; es[i] = e[0]
                key = f"est_{s}_{t}_{bd}"
                d[key + "_in"] = xs; d[key + "_coeff"] = cos; d[key + "_energy"] = es
    # ---- 8-bit inverse entry
    for s in range(19):
        w, h = TX_W[s], TX_H[s]
        kw, kh = min(w, 32), min(h, 32)
        for t in range(16):
            if not txfm_allowed(s, t):
                continue
            x = aligned((h, 64), np.int16); x[:, :w] = rng.integers(-255, 256, size=(h, w))
            full = aligned(w * h + 64, np.int32); e = np.zeros(1, np.uint64)
            R.ref_estimate_transform(ptr(x), ctypes.c_uint32(64), ptr(full), c_int(s), c_int(t), c_int(0), ptr(e))
            cos = np.zeros((2, kw * kh), np.int32)
            cos[0] = full[:kw * kh]
            cos[1] = full[:kw * kh]; cos[1, max(1, kw * kh // 5):] = 0
            dst0 = rng.integers(0, 256, size=(2, h, w), dtype=np.uint8)
            dst1 = np.zeros_like(dst0)
            for i in range(2):
                outs = []
                for which in (0, 1, 2):
                    co = aligned(kw * kh + 64, np.int32); co[:kw * kh] = cos[i]
                    dd = aligned((h, 96), np.uint8); dd[:, :w] = dst0[i]
                    R.ref_inv_txfm_add_u8(ptr(co), ptr(dd), c_int(96), c_int(t), c_int(s), c_int(kw * kh), c_int(which))
                    outs.append(dd[:, :w].copy())
                # production (SSSE3 low-bit-depth code) == C on transform-consistent input (InvTxfm2dAsmTest lowbd sub-test)
                assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), (TX_SIZES[s], TX_TYPES[t], i)
                dst1[i] = outs[0]
            key = f"inv8_{s}_{t}"
            d[key + "_coeff"] = cos; d[key + "_dst_in"] = dst0; d[key + "_dst_out"] = dst1
    # ---- coefficient-domain distortion
    for (w, h) in ((4, 4), (8, 8), (16, 16), (32, 32), (8, 16), (32, 8), (16, 64), (64, 64)):
        mags = (8, 15, 20, 26)
        a = np.stack([rng.integers(-(1 << m), 1 << m, size=(h, w + 8)) for m in mags]).astype(np.int32)
        b = np.stack([rng.integers(-(1 << m), 1 << m, size=(h, w + 8)) for m in mags]).astype(np.int32)
        out = np.zeros((4, 2, 2), np.uint64)        # [case][cbf_zero][2]: the C kernels
        out_avx2 = np.zeros((4, 2, 2), np.uint64)   # the AVX2 kernels (production slot)
        for i in range(4):
            aa = aligned(a[i].shape, np.int32); aa[:] = a[i]
            bb = aligned(b[i].shape, np.int32); bb[:] = b[i]
            for z, names in enumerate((("full_distortion_kernel32_bits", "full_distortion_kernel32_bits_avx2"),
                                       ("full_distortion_kernel_cbf_zero32_bits", "full_distortion_kernel_cbf_zero32_bits_avx2"))):
                r = [np.zeros(2, np.uint64) for _ in names]
                for rr, nm in zip(r, names):
                    getattr(R, nm)(ptr(aa), ctypes.c_uint32(w + 8), ptr(bb), ctypes.c_uint32(w + 8), ptr(rr), ctypes.c_uint32(w), ctypes.c_uint32(h))
                # full_distortion_kernel32_bits_avx2 adds the squared differences with _mm256_add_epi32
                # (EbPictureOperators_Intrinsic_AVX2.c:1989): no carry from the low into the high half of its 64-bit lanes,
                # so DIST_CALC_RESIDUAL differs from the C kernel once a lane's low halves sum past 2^32.  Both recorded.
                if i == 0:
                    assert np.array_equal(r[0], r[1]), (w, h, i, names)
                if z == 1:
                    assert np.array_equal(r[0], r[1]), (w, h, i, names)    # the cbf_zero kernel adds in 64 bits
                out[i, z] = r[0]; out_avx2[i, z] = r[1]
            if w == h:          # picture_full_distortion32_bits, luma: clamps the area of a 64x64 block to 32x32, stride = width
                k = min(w, 32)
                ca = aligned((k, k), np.int32); ca[:] = a[i][:k, :k]
                cb = aligned((k, k), np.int32); cb[:] = b[i][:k, :k]
                for nz in (1, 0):
                    y = np.zeros(2, np.uint64)
                    rc = R.ref_picture_full_distortion32(ptr(ca), 0, ptr(cb), 0, c_int(w), c_int(h), c_int(nz), c_int(0), ptr(y))
                    assert rc == 0
                    d[f"pdist_{w}_{i}_{nz}"] = y
        d[f"dist_{w}x{h}_a"] = a; d[f"dist_{w}x{h}_b"] = b; d[f"dist_{w}x{h}_out"] = out; d[f"dist_{w}x{h}_out_avx2"] = out_avx2
    # ---- inverse transform at the clamp limits (the add runs on 16-bit lanes in the product)
    for (s, types) in ((3, (0, 9)), (9, (0, 9)), (15, (9,)), (2, (0, 3, 9, 12)), (4, (0,)), (1, (9, 0)), (11, (0,))):
        w, h = TX_W[s], TX_H[s]
        kw, kh = min(w, 32), min(h, 32)
        f = getattr(R, f"av1_inv_txfm2d_add_{w}x{h}_c")
        for t in types:
            for bd in (8, 10, 12):
                top = 1 << (bd + 7)
                cos = np.stack([np.full(kw * kh, top), np.full(kw * kh, -top),
                                rng.choice([-top, top], size=kw * kh), rng.integers(-(1 << 21), 1 << 21, size=kw * kh),
                                np.r_[top * 4, np.zeros(kw * kh - 1, int)]]).astype(np.int32)
                dst0 = rng.integers(0, 1 << bd, size=(5, h, w)).astype(np.uint16)
                dst0[0] = (1 << bd) - 1; dst0[1] = 0
                dst1 = dst0.copy()
                for i in range(5):
                    ref_inv(s, t, bd, np.ascontiguousarray(cos[i]), dst1[i])
                key = f"sat_{s}_{t}_{bd}"
                d[key + "_coeff"] = cos; d[key + "_dst_in"] = dst0; d[key + "_dst_out"] = dst1
    np.savez_compressed(os.path.join(HERE, "pins.npz"), **d)


HME_FIELDS = ("level", "sb_w", "sb_h", "origin_x", "origin_y", "x_center", "y_center", "region_w", "region_h", "mult_x", "mult_y")


def gen_hme():
    """Hierarchical ME levels 0 / 1 / 2: the reference's OWN HmeLevel0 / HmeLevel1 / HmeLevel2
    (EbMotionEstimation.c:5689, 5883, 6016) through oracle/ref_me.c, on one padded picture pair per level (1/16-, 1/4- and
    full-resolution stand-ins of a 416x240 picture), incl. SBs at every picture edge (search-area clipping), partial SBs,
    off-picture search centres, two search regions with area multipliers, ties (coarse content).  asm_type 0 (C / SSE4.1)
    is stored; asm_type 1 is asserted equal except for 4-sample-wide blocks, where the AVX2 kernels are known to return
    wrong SADs (same divergence as sad_loop_kernel_avx2_intrin at width 4)."""
    rng = np.random.default_rng(13604)
    d = {}
    dims = {0: (104, 60, 16, 24), 1: (208, 120, 32, 40), 2: (416, 240, 64, 72)}        # level: W, H, SB size, padding
    hme_w = np.array([40, 24], np.uint16); hme_h = np.array([20, 12], np.uint16)
    d["hme_w"] = hme_w; d["hme_h"] = hme_h
    for level, (W, H, sb, pad) in dims.items():
        stride = W + 2 * pad + 5
        ref_buf = rng.integers(0, 256, (H + 2 * pad, stride), dtype=np.uint8)
        src = rng.integers(0, 256, (H, W), dtype=np.uint8)
        ref_buf[pad + H // 2:, :] = (ref_buf[pad + H // 2:, :] >> 6) << 6          # lower half coarse: ties
        src[H // 2:, :] = (src[H // 2:, :] >> 6) << 6
        ref_buf[pad + 8:pad + 8 + sb, pad + 20:pad + 20 + sb] = src[4:4 + sb, 10:10 + sb]   # an exact match somewhere
        d[f"l{level}_src"] = src; d[f"l{level}_ref"] = ref_buf; d[f"l{level}_dims"] = np.array([W, H, sb, pad, stride], np.int32)
        cases = []
        xs = list(range(0, W - sb + 1, sb)) + ([W - (W % sb)] if W % sb else [])
        ys = list(range(0, H - sb + 1, sb)) + ([H - (H % sb)] if H % sb else [])
        for oy in ys:
            for ox in xs:
                sbw = min(sb, W - ox); sbh = min(sb, H - oy)
                for (xc, yc) in ((0, 0), (int(rng.integers(-60, 60)), int(rng.integers(-40, 40)))):
                    rw, rh = int(rng.integers(0, 2)), int(rng.integers(0, 2))
                    mx, my = (100, 100) if level else (int(rng.choice([100, 150])), int(rng.choice([100, 200])))
                    cases.append((level, sbw, sbh, ox, oy, xc, yc, rw, rh, mx, my))
        cases = np.array(cases, np.int32)
        out = np.zeros((len(cases), 3), np.int64)
        for i, (lv, sbw, sbh, ox, oy, xc, yc, rw, rh, mx, my) in enumerate(cases.tolist()):
            res = []
            for asm in (0, 1):
                b = np.zeros(1, np.uint64); x = np.zeros(1, np.int16); y = np.zeros(1, np.int16)
                rc = R.ref_hme_level(c_int(lv), ptr(src), c_int(W), ptr(ref_buf), c_int(stride), c_int(pad), c_int(pad), c_int(W), c_int(H),
                                     c_int(ox), c_int(oy), c_int(sbw), c_int(sbh), c_int(xc), c_int(yc), ptr(hme_w), ptr(hme_h), c_int(rw), c_int(rh),
                                     c_int(int(hme_w.sum())), c_int(int(hme_h.sum())), c_int(mx), c_int(my), c_int(asm), ptr(b), ptr(x), ptr(y))
                assert rc == 0
                res.append((int(b[0]), int(x[0]), int(y[0])))
            if sbw != 4:
                assert res[0] == res[1], (lv, sbw, sbh, ox, oy, res)
            out[i] = res[0]
        d[f"l{level}_cases"] = cases; d[f"l{level}_out"] = out
    np.savez_compressed(os.path.join(HERE, "hme.npz"), **d)


def gen_bip():
    """a14 glue (VERDICT r1 missing #4): the reference's OWN has_top_right / has_bottom_left (every enumerated argument tuple,
    packed bits), build_intra_predictors{,_high} and av1_predict_intra_block{,_16bit} (EbIntraPrediction.c:1567, 1755, 3667,
    3857, 4078, 4336) through oracle/ref_intra.c.  The mode-info grids of the block cases are the pattern svtlibs.mi_pattern
    builds from the stored key, so only the key travels."""
    import svtlibs
    from svtlibs import (BLOCK_W, BLOCK_H, TX_W, TX_H, availability_tuples, intra_block_case, mi_pattern, ref_predict_intra_block,
                         aligned_array)
    d = {}
    for sb_bsize, sb_mi in ((12, 16), (15, 32)):
        tr, bl = [], []
        for args in availability_tuples(sb_mi):
            tr.append(R.ref_has_top_right(sb_bsize, *args)); bl.append(R.ref_has_bottom_left(sb_bsize, *args))
        d[f"has_tr_sb{sb_mi}"] = np.packbits(np.array(tr, np.uint8)); d[f"has_bl_sb{sb_mi}"] = np.packbits(np.array(bl, np.uint8))
        d[f"avail_count_sb{sb_mi}"] = np.array([len(tr)], np.int64)
    # build_intra_predictors: params / edges / packed outputs
    rng = np.random.default_rng(3667)
    prm, tops, lefts, outs = [], [], [], []
    for trial in range(700):
        s = trial % 19; w, h = TX_W[s], TX_H[s]
        mode = (trial // 19) % 13 if trial < 19 * 13 * 2 else int(rng.integers(0, 13))
        ad = int(rng.integers(-3, 4)) if 1 <= mode <= 8 else 0
        is16 = int(rng.integers(0, 2)); bd = 10 if is16 else 8
        top = rng.integers(0, 1 << bd, 176).astype(np.uint16); left = rng.integers(0, 1 << bd, 176).astype(np.uint16)
        n_top = int(rng.choice([0, w])); n_left = int(rng.choice([0, h]))
        if trial % 5 == 0:
            n_top, n_left = w, h
        if trial % 7 == 0:
            n_top = int(rng.integers(1, w // 4 + 1)) * 4; n_left = int(rng.integers(1, h // 4 + 1)) * 4
        n_tr = int(rng.choice([0, h, int(rng.integers(0, h + 1))])) if n_top == w else 0
        n_bl = int(rng.choice([0, w, int(rng.integers(0, w + 1))])) if n_left == h else 0
        dis = int(rng.integers(0, 4) == 0); ft = int(rng.integers(0, 2))
        dt = np.uint16 if is16 else np.uint8; es = 2 if is16 else 1
        t2 = top.astype(dt); l2 = left.astype(dt)
        out = aligned_array((h, 128), dt)
        R.ref_build_intra_predictors(is16, ctypes.c_void_p(t2.ctypes.data + 16 * es), ctypes.c_void_p(l2.ctypes.data + 16 * es), ptr(out), 128,
                                     mode, ad, s, dis, n_top, n_tr, n_left, n_bl, ft, bd)
        prm.append((is16, mode, ad, s, dis, n_top, n_tr, n_left, n_bl, ft, bd)); tops.append(top); lefts.append(left)
        outs.append(out[:, :w].astype(np.uint16).ravel())
    d["bip_params"] = np.array(prm, np.int32); d["bip_top"] = np.array(tops); d["bip_left"] = np.array(lefts)
    d["bip_out"] = np.concatenate(outs)
    # av1_predict_intra_block{,_16bit}
    rng = np.random.default_rng(4078)
    prm, tops, lefts, outs = [], [], [], []
    trial = 0
    while len(prm) < 300:
        trial += 1
        c = intra_block_case(rng, trial)
        if c is None:
            continue
        key = int(rng.integers(0, 8))
        c["mi_mode"], c["mi_uv_mode"] = mi_pattern(c["mi_rows"], c["mi_cols"], key)
        got = ref_predict_intra_block(R, c)
        prm.append((c["is16"], c["mi_rows"], c["mi_cols"], c["plane"], c["bsize"], c["partition"], c["tx"], c["mirow"], c["micol"], c["col_off"],
                    c["row_off"], c["wpx"], c["hpx"], c["mode"], c["angle_delta"], *c["tile"].tolist(), key))
        tops.append(c["top"].astype(np.uint16)); lefts.append(c["left"].astype(np.uint16)); outs.append(got.astype(np.uint16).ravel())
    d["pib_params"] = np.array(prm, np.int32); d["pib_top"] = np.array(tops); d["pib_left"] = np.array(lefts)
    d["pib_out"] = np.concatenate(outs)
    np.savez_compressed(os.path.join(HERE, "bip.npz"), **d)


def gen_picture():
    """picture input (n4): the reference application's read_y4m_header on files, and the reference library's pad_input_picture ->
    generate_padding{,16_bit} sequence and Decimation2D + generate_padding on small frames"""
    import tempfile
    from svtlibs import Y4M_HEADERS, write_y4m
    rng = np.random.default_rng(4907)
    d = {}
    R.ref_y4m_header.restype = ctypes.c_long
    hdr = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, (line, _) in enumerate(Y4M_HEADERS):
            path = os.path.join(tmp, f"h{i}.y4m")
            write_y4m(path, line, [(np.zeros(4, np.uint8),) * 3])
            out = np.zeros(8, np.int32)
            R.ref_y4m_header(path.encode(), ptr(out))
            hdr.append(out.copy())
    d["y4m_lines"] = np.array([l for l, _ in Y4M_HEADERS])
    d["y4m_out"] = np.array(hdr)
    cases = []
    for k, (w, h, ox, oy, pr, pb, is16) in enumerate([(17, 9, 4, 3, 7, 7, 0), (40, 24, 16, 8, 0, 0, 0), (33, 31, 5, 5, 7, 1, 0), (64, 48, 68, 68, 0, 0, 0),
                                                      (20, 12, 6, 2, 4, 4, 1), (35, 17, 8, 8, 5, 7, 1)]):
        dt = np.uint16 if is16 else np.uint8
        frame = rng.integers(0, 1 << (10 if is16 else 8), (h, w)).astype(dt)
        W, H = w + pr, h + pb
        stride = W + 2 * ox + 3
        buf = np.full((H + 2 * oy, stride), 0x55, dt)
        buf[oy:oy + h, ox:ox + w] = frame
        if is16:
            # the reference holds a 10-bit input picture as an 8-bit plane + a bit-increment plane (2 bits at the top of a byte) and
            # runs pad_input_picture on each (EbPictureAnalysisProcess.c:4830-4876): done so here, then recombined into 16-bit samples
            hi = np.ascontiguousarray((buf >> 2).astype(np.uint8)); lo = np.ascontiguousarray(((buf & 3) << 6).astype(np.uint8))
            for pl in (hi, lo):
                R.ref_pad_input_picture(ctypes.c_void_p(pl.ctypes.data + (oy * stride + ox)), stride, w, h, pr, pb)
            buf[:] = (hi.astype(np.uint16) << 2) | (lo.astype(np.uint16) >> 6)
        else:
            R.ref_pad_input_picture(ctypes.c_void_p(buf.ctypes.data + (oy * stride + ox)), stride, w, h, pr, pb)
        R.ref_generate_padding(ptr(buf), stride, W, H, ox, oy, is16)
        d[f"pad{k}_prm"] = np.array([w, h, ox, oy, pr, pb, is16, stride], np.int32)
        d[f"pad{k}_frame"] = frame
        d[f"pad{k}_out"] = buf
        cases.append(k)
    d["pad_cases"] = np.array(cases, np.int32)
    for k, (w, h, qo, so) in enumerate([(64, 32, 8, 4), (50, 22, 6, 3), (33, 19, 4, 4)]):
        stride = w + 5
        luma = rng.integers(0, 256, (h, stride)).astype(np.uint8)
        outs = []
        for step, o in ((2, qo), (4, so)):
            dw, dh = (w + step - 1) // step, (h + step - 1) // step
            ds = dw + 2 * o + 1
            buf = np.full((dh + 2 * o, ds), 0x33, np.uint8)
            R.ref_decimation_2d(ptr(luma), stride, w, h, ctypes.c_void_p(buf.ctypes.data + o * ds + o), ds, step)
            R.ref_generate_padding(ptr(buf), ds, dw, dh, o, o, 0)
            outs.append(buf)
        d[f"dec{k}_prm"] = np.array([w, h, stride, qo, so], np.int32)
        d[f"dec{k}_luma"] = luma; d[f"dec{k}_q"] = outs[0]; d[f"dec{k}_s"] = outs[1]
    d["dec_cases"] = np.arange(3, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "picture.npz"), **d)


ME_SETUP_PARAM_SETS = [
    # (picture set, keyword overrides of svtlibs.ME_LCU_DEFAULTS)
    ("edge", dict()),                                                                 # P picture, 85 PUs, 2 x 2 regions, three HME levels
    ("edge", dict(slice_type=0, pic_depth_mode=0)),                                   # B picture, 209 PUs, bi-prediction on every PU
    ("edge", dict(slice_type=0, pic_depth_mode=0, search_area_width=7, search_area_height=5, regions_w=1, regions_h=1)),   # widths < 8 everywhere
    ("edge", dict(slice_type=0, pic_depth_mode=0, ref1_poc=8, temporal_layer_index=2)),                      # same-POC list 1: second-best region
    ("edge", dict(slice_type=0, pic_depth_mode=0, ref1_poc=8, temporal_layer_index=0)),                      # base layer: list 1 without HME
    ("edge", dict(slice_type=0, pic_depth_mode=2, cu8x8_mode=1, fractional_search_method=1, is_used_as_reference_flag=0)),   # bi-pred on PUs 0..20, full SAD, no (0,0) check
    ("edge", dict(slice_type=0, pic_depth_mode=0, hme_l1=0, search_area_width=24, search_area_height=16)),   # level 1 off: level 2 starts from (0, 0)
    ("edge", dict(slice_type=1, pic_depth_mode=0, hme_l2=0, hme_l0=0, temporal_layer_index=0, hierarchical_levels=4)),
    ("edge", dict(slice_type=0, pic_depth_mode=0, enable_hme_flag=0, search_area_width=64, search_area_height=16)),
    ("noise", dict(slice_type=0, pic_depth_mode=0, asm_type=1)),                      # the AVX2 build's flavour on whole SBs
    ("noise", dict(slice_type=0, pic_depth_mode=2, asm_type=1, search_area_width=30, search_area_height=9, temporal_layer_index=0, hierarchical_levels=5)),
    ("noise", dict(slice_type=1, pic_depth_mode=0, regions_w=1, regions_h=1, search_area_width=8, search_area_height=64)),
    # "noise_edge": 200 x 136 of noise - the HME vectors of the 8-wide last SB column are arbitrary, so its 8-wide search areas run over
    # the right picture edge and are clipped to 1 .. 7 columns: the single-search-point form of the 209-PU search (ExtSadCalculation's
    # 32x16_5 rule), of the 85-PU search, and (asm_type 1 is not used here: the AVX2 HME kernels are undefined on 2- / 4-wide blocks)
    ("noise_edge", dict(slice_type=0, pic_depth_mode=0, search_area_width=8, search_area_height=7, is_used_as_reference_flag=0)),
    ("noise_edge", dict(slice_type=0, pic_depth_mode=0, search_area_width=8, search_area_height=12, temporal_layer_index=0)),
    ("noise_edge", dict(slice_type=0, pic_depth_mode=2, search_area_width=6, search_area_height=9, is_used_as_reference_flag=0, regions_w=1, regions_h=1)),
]


def me_setup_pictures():
    """two picture triples (source, list-0 reference, list-1 reference).  "edge": 200 x 136 (an 8-wide last SB column, an 8-high last
    SB row), smooth content; the references are the source displaced by (+37, +6) and (-29, -11), so HME finds real motion and the
    search areas of the SBs at the right / left picture edge are clipped to a few columns.  "noise": 256 x 192 of noise with a
    coarse half (ties), whole SBs only."""
    rng = np.random.default_rng(7527)
    a = rng.integers(0, 256, (70, 120)).astype(np.float64)
    a = np.kron(a, np.ones((4, 4)))
    k = np.ones(7) / 7
    a = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, a); a = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 0, a)
    base = (a + rng.integers(-5, 6, a.shape)).clip(0, 255).astype(np.uint8)
    W, H = 200, 136
    edge = tuple(np.ascontiguousarray(base[64 + dy:64 + dy + H, 64 + dx:64 + dx + W]) for dx, dy in ((0, 0), (37, 6), (-29, -11)))
    n = [rng.integers(0, 256, (192, 256), dtype=np.uint8) for _ in range(3)]
    for m in n:
        m[96:, :] = (m[96:, :] >> 6) << 6
    n[1][10:74, 20:84] = n[0][64:128, 64:128]
    ne = tuple(rng.integers(0, 256, (136, 200), dtype=np.uint8) for _ in range(3))
    return {"edge": edge, "noise": tuple(n), "noise_edge": ne}


def gen_me_setup():
    """MotionEstimateLcu, whole: HME levels -> best-of-regions centre -> CheckZeroZeroCenter -> search-area clipping ->
    full-pel search -> BiPredictionSearch -> me_results, by the reference's OWN function through oracle/ref_me.c
    (ref_motion_estimate_lcu), for every SB of two small pictures under fifteen parameter sets."""
    import svtlibs
    pics = me_setup_pictures()
    d = {}
    prm_all, key_all, outs = [], [], {k: [] for k in ("best_sad", "best_mv", "area_origin", "bipred_sad", "results")}
    for name, planes in pics.items():
        for i, pl in enumerate(planes):
            d[f"{name}_pic{i}"] = pl
    for si, (name, kw) in enumerate(ME_SETUP_PARAM_SETS):
        src, r0, r1 = pics[name]
        H, W = src.shape
        ps, geo = svtlibs.me_pyramid(src); p0, _ = svtlibs.me_pyramid(r0); p1, _ = svtlibs.me_pyramid(r1)
        for sy in range(0, H, 64):
            for sx in range(0, W, 64):
                prm = svtlibs.me_lcu_params(W, H, sx, sy, geo, **kw)
                o = svtlibs.run_me_lcu(R.ref_motion_estimate_lcu, prm, ps, p0, p1)
                prm_all.append(prm); key_all.append(si)
                for k in outs:
                    outs[k].append(o[k])
    d["prm"] = np.array(prm_all); d["param_set"] = np.array(key_all, np.int32)
    d["picture_of_set"] = np.array([n for n, _ in ME_SETUP_PARAM_SETS])
    for k, v in outs.items():
        d[k] = np.array(v)
    np.savez_compressed(os.path.join(HERE, "me_setup.npz"), **d)


if __name__ == "__main__":
    if len(sys.argv) > 1:                    # one family only: python make_golden.py cfl_levels
        globals()["gen_" + sys.argv[1]]()
        sys.exit(0)
    gen_txfm()
    tabs = gen_tables()
    gen_quant(tabs)
    gen_pixel()
    gen_intra()
    gen_cfl_levels()
    gen_ois()
    gen_me()
    gen_pins()
    gen_hme()
    gen_bip()
    gen_picture()
    gen_me_setup()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
