"""GPU: MotionEstimateLcu's glue on the device - svt_hip_me_setup_batch (best-of-regions search centre,
CheckZeroZeroCenter, per-SB search-area clipping), svt_hip_me_fullpel_search_areas_batch (one area per SB, read on the device) and
svt_hip_me_bipred_batch (BiPredictionSearch + the me_results rows) - chained behind the HME levels exactly as the reference chains
them, against the outputs of the reference's OWN MotionEstimateLcu (tests/golden/me_setup.npz) on every SB of the fixture's pictures,
and against the oracle on synthetic HME results that drive the clipped widths through 1 .. 7 (the single-search-point form)."""
import ctypes
import os

import numpy as np
import pytest
import torch

import svtlibs
from svtlibs import ptr

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def pic00(d_plane, geo):
    """view whose data_ptr() is sample (0, 0) of a padded plane"""
    stride, ox, oy, _, _ = geo
    return d_plane.view(-1)[oy * stride + ox:]


def hme_multiplier(hl, tl):
    top3 = {3: (200, 140, 100, 70), 4: (350, 200, 100, 100), 5: (525, 350, 200, 100)}
    return top3[hl][tl] if hl in top3 and tl < 4 else 100


def device_me_lcu(dsp, pkg, prm_rows, planes, geo):
    """every SB of one picture under one parameter set, the whole chain on the device -> dict of numpy arrays shaped like the fixture's"""
    p0 = prm_rows[0]
    W, H = int(p0[0]), int(p0[1])
    nl = 1 if p0[4] == 1 else 2
    nsq = p0[5] <= 1
    tl, hl = int(p0[6]), int(p0[7])
    hme_on, lv_on, is_ref = int(p0[8]), [int(p0[9]), int(p0[10]), int(p0[11])], int(p0[12])
    saw, sah, rw_n, rh_n = int(p0[13]), int(p0[14]), int(p0[15]), int(p0[16])
    flavour, npus = int(p0[21]), int(p0[25])
    n = len(prm_rows)
    sbs = [(int(r[2]), int(r[3])) for r in prm_rows]
    size = [(min(64, W - x), min(64, H - y)) for x, y in sbs]
    d_pl = [[dev(p) for p in pl] for pl in planes]                      # [picture][level: full, 1/4, 1/16]
    hw = [np.array(p0[42 + 4 * lv:44 + 4 * lv], np.uint16) for lv in range(3)]
    hh = [np.array(p0[44 + 4 * lv:46 + 4 * lv], np.uint16) for lv in range(3)]
    mult = hme_multiplier(hl, tl)
    org_full = dev(np.array(sbs, np.int16)); size_full = dev(np.array(size, np.int16))
    stride = geo[0][0]
    soff = dev(np.array([y * stride + x for x, y in sbs], np.int32))
    out = dict(best_sad=np.zeros((n, 2, 209), np.uint32), best_mv=np.zeros((n, 2, 209), np.uint32), area_origin=np.zeros((n, 2, 2), np.int16),
               area=np.zeros((n, 2, 4), np.int16))
    rows = []
    for li in range(nl):
        hme_list = tl > 0 or li == 0 or (p0[19] != p0[20] and li == 1)
        hme_sad = hme_mv = None
        last = -1
        if hme_on and hme_list:
            centers = None
            for lv in range(3):
                if not lv_on[lv]:
                    centers = None                                   # a switched-off level hands on the initial centre (0, 0)
                    continue
                last = lv
                sh = 2 - lv
                g = geo[2 - lv]
                org = dev(np.array([(x >> sh, y >> sh) for x, y in sbs], np.int16))
                sz = dev(np.array([(w >> sh, h >> sh) for w, h in size], np.int16))
                prm = [dsp.hme_level_params(lv, hw[lv], hh[lv], rw, rh, int(p0[17]), int(p0[18]), mult, mult, g[1], g[2], g[3], g[4])
                       for rh in range(rh_n) for rw in range(rw_n)]  # region r = rh * regions_w + rw: the reference's visiting order
                hme_sad, hme_mv = dsp.hme_level_regions(pic00(d_pl[0][2 - lv], g), g[0], pic00(d_pl[1 + li][2 - lv], g), g[0], org, sz, centers,
                                                        1 if lv == 1 else 0, prm)
                centers = hme_mv
        sp = dsp.MeSetupParams(W, H, geo[0][3], geo[0][4], saw, sah, rw_n if last >= 0 else 0, rh_n if last >= 0 else 0,
                               int(last == 2 and p0[19] == p0[20] and li == 1), is_ref)
        center, area = dsp.me_setup(pic00(d_pl[0][0], geo[0]), stride, pic00(d_pl[1 + li][0], geo[0]), stride, org_full, size_full,
                                    hme_sad if last >= 0 else None, hme_mv if last >= 0 else None, sp)
        bs, bm = dsp.me_fullpel_search_areas(pic00(d_pl[0][0], geo[0]), stride, soff, pic00(d_pl[1 + li][0], geo[0]), stride, soff, area,
                                             (saw + 7) & ~7, sah, flavour=flavour, nsq=nsq)
        rows.append((bs, bm))
        a = area.cpu().numpy()
        out["area"][:, li] = a; out["area_origin"][:, li] = a[:, :2]
        out["best_sad"][:, li, :bs.shape[1]] = bs.cpu().numpy().view(np.uint32)
        out["best_mv"][:, li, :bs.shape[1]] = bm.cpu().numpy().view(np.uint32)
    bip, res = dsp.me_bipred(pic00(d_pl[0][0], geo[0]), stride, pic00(d_pl[1][0], geo[0]), stride, pic00(d_pl[2][0], geo[0]), stride, org_full,
                             rows[0][0], rows[0][1], rows[1][0] if nl == 2 else None, rows[1][1] if nl == 2 else None, npus=npus,
                             bipred_all_pus=bool(p0[23] == 0 or nsq), sub_sad=bool(p0[24] == 0))
    out["bipred_sad"] = np.zeros((n, 209), np.uint32)
    out["bipred_sad"][:, :bip.shape[1]] = bip.cpu().numpy().view(np.uint32)
    out["results"] = np.zeros((n, 209, 11), np.int64)
    out["results"][:, :npus] = dsp.me_results_as_rows(res)
    return out


def test_motion_estimate_lcu_chain_equals_the_references_own_function(dsp, pkg):
    """HME levels -> svt_hip_me_setup_batch -> svt_hip_me_fullpel_search_areas_batch -> svt_hip_me_bipred_batch for every SB of the
    fixture's pictures, one launch per stage and list, against ref_motion_estimate_lcu's outputs: result rows of both lists, search-area
    origins, bi-prediction SADs and the ordered me_results candidates"""
    g, pics = svtlibs.me_setup_fixture()
    nsets = int(g["param_set"].max()) + 1
    narrow = 0
    for si in range(nsets):
        idx = np.nonzero(g["param_set"] == si)[0]
        (ps, geo), (r0, _), (r1, _) = pics[str(g["picture_of_set"][si])]
        got = device_me_lcu(dsp, pkg, g["prm"][idx], (ps, r0, r1), geo)
        nl = 1 if g["prm"][idx[0]][4] == 1 else 2
        npus = int(g["prm"][idx[0]][25])
        for k in ("best_sad", "best_mv", "area_origin"):
            assert np.array_equal(got[k][:, :nl], g[k][idx][:, :nl]), (si, k, np.argwhere(got[k][:, :nl] != g[k][idx][:, :nl])[:4].tolist())
        assert np.array_equal(got["results"][:, :npus], g["results"][idx][:, :npus]), (si, np.argwhere(got["results"][:, :npus] != g["results"][idx][:, :npus])[:4].tolist())
        if nl == 2:
            assert np.array_equal(got["bipred_sad"], g["bipred_sad"][idx]), si
        narrow += int(((got["area"][:, :nl, 2] > 0) & (got["area"][:, :nl, 2] < 8)).sum())
    assert narrow >= 2                                 # the fixture holds search areas clipped to fewer than 8 columns


def test_me_setup_and_area_search_on_synthetic_hme_results_vs_oracle(dsp, pkg):
    """random HME region results (vectors far beyond the picture edges included) for every SB of a 456 x 200 picture (a 8-wide last
    column, an 8-high last row), x {second-best pick, CheckZeroZeroCenter on / off, 1 x 1 / 2 x 2 regions}: centres and areas equal
    svt_oracle_me_setup; then the search over those areas - 85 and 209 PUs, both result flavours - equals the oracle SB by SB.
    The clipped widths must have covered 1 .. 7: the narrow single-search-point areas no longer take a different (slower) launch."""
    O = svtlibs.oracle()
    rng = np.random.default_rng(8040)
    W, H = 456, 200
    src = svtlibs.smooth_picture(rng, H, W); ref = svtlibs.smooth_picture(rng, H, W)
    ref[:, :W // 2] = src[:, :W // 2]                                   # half the picture matches at (0, 0): the zero centre wins there
    (ps, geo), (pr, _) = svtlibs.me_pyramid(src), svtlibs.me_pyramid(ref)
    stride, pad = geo[0][0], geo[0][1]
    sbs = [(x, y) for y in range(0, H, 64) for x in range(0, W, 64)] * 6
    n = len(sbs)
    size = [(min(64, W - x), min(64, H - y)) for x, y in sbs]
    d_src, d_ref = dev(ps[0]), dev(pr[0])
    s00, r00 = pic00(d_src, geo[0]), pic00(d_ref, geo[0])
    org = dev(np.array(sbs, np.int16)); sz = dev(np.array(size, np.int16))
    soff = dev(np.array([y * stride + x for x, y in sbs], np.int32))
    src00 = ctypes.c_void_p(ps[0].ctypes.data + pad * stride + pad); ref00 = ctypes.c_void_p(pr[0].ctypes.data + pad * stride + pad)
    widths = set()
    for (rw_n, rh_n, second, zz, saw, sah) in ((2, 2, 0, 1, 8, 7), (2, 2, 1, 0, 8, 5), (1, 1, 0, 0, 16, 9), (2, 2, 1, 1, 24, 4), (1, 1, 0, 1, 6, 12)):
        nreg = rw_n * rh_n
        hsad = rng.integers(0, 1 << 20, (nreg, n)).astype(np.int64)
        hsad[:, ::5] = hsad[0, ::5]                                      # ties between regions: first strict minimum / the sort's order
        hmv = rng.integers(-90, 91, (nreg, n, 2)).astype(np.int16)
        hmv[:, ::7, 0] = rng.integers(-4, 12, (nreg, len(range(0, n, 7))))   # small positive x on some: the last column clips to 1 .. 7
        last_col = [i for i, (x, y) in enumerate(sbs) if x + 64 > W and y + 64 <= H]
        for k, i in enumerate(last_col):                                 # ... and on purpose: centre 5 .. 11 -> clipped width 7 .. 1 (8-wide area)
            hmv[:, i, 0] = 5 + k % 7
            hmv[:, i, 1] = rng.integers(-3, 4)
        sp = dsp.MeSetupParams(W, H, W, H, saw, sah, rw_n, rh_n, second, zz)
        center, area = dsp.me_setup(s00, stride, r00, stride, org, sz, dev(hsad), dev(hmv), sp)
        c_got, a_got = center.cpu().numpy(), area.cpu().numpy()
        for i, ((x, y), (w, h)) in enumerate(zip(sbs, size)):
            s4 = np.zeros((2, 2), np.uint64); x4 = np.zeros((2, 2), np.int16); y4 = np.zeros((2, 2), np.int16)
            for r in range(nreg):                                        # region r = rh * regions_w + rw -> [w][h]
                s4[r % rw_n, r // rw_n] = hsad[r, i]; x4[r % rw_n, r // rw_n] = hmv[r, i, 0]; y4[r % rw_n, r // rw_n] = hmv[r, i, 1]
            ce = np.zeros(2, np.int16); ar = np.zeros(4, np.int16)          # (no HME result is used on SBs that are not 64 rows high, :7678)
            O.svt_oracle_me_setup(src00, stride, ref00, stride, x, y, w, h, W, H, W, H, int(h == 64), ptr(s4), ptr(x4), ptr(y4), rw_n, rh_n, second, zz, saw,
                                  sah, ptr(ce), ptr(ar))
            assert c_got[i].tolist() == ce.tolist() and a_got[i].tolist() == ar.tolist(), (i, (x, y), (rw_n, rh_n, second, zz), c_got[i], ce, a_got[i], ar)
        widths |= set(int(v) for v in a_got[:, 2])
        for nsq in (False, True):
            for flavour in (0, 1):
                bs, bm = dsp.me_fullpel_search_areas(s00, stride, soff, r00, stride, soff, area, (saw + 7) & ~7, sah, flavour=flavour, nsq=nsq)
                gs, gm = bs.cpu().numpy().view(np.uint32), bm.cpu().numpy().view(np.uint32)
                npu = 209 if nsq else 85
                for i in range(0, n, 1 if saw <= 8 else 3):
                    x, y = sbs[i]
                    xo, yo, aw, ah = (int(v) for v in a_got[i])
                    es = np.full(209, 128 * 128 * 255, np.uint32); em = np.zeros(209, np.uint32)
                    O.svt_oracle_me_sb_search_full(ctypes.c_void_p(src00.value + y * stride + x), stride,
                                                   ctypes.c_void_p(ref00.value + (y + yo) * stride + x + xo), stride, aw, ah, xo, yo, flavour, int(nsq),
                                                   ptr(es), ptr(em))
                    assert np.array_equal(gs[i], es[:npu]) and np.array_equal(gm[i], em[:npu]), (i, (x, y), (xo, yo, aw, ah), nsq, flavour,
                                                                                                 np.nonzero(gs[i] != es[:npu])[0][:6].tolist())
    assert {1, 2, 3, 4, 5, 6, 7} <= widths, sorted(widths)


def test_me_bipred_and_result_rows_vs_oracle(dsp, pkg):
    """svt_hip_me_bipred_batch on random result rows (vectors inside +-40 full-pel, SADs with ties between the three candidates):
    bi-prediction SADs (every other row and full), the P-picture and the two-candidate forms, the 21-PU gating"""
    O = svtlibs.oracle()
    rng = np.random.default_rng(6639)
    W, H = 320, 192
    planes = [svtlibs.me_pyramid(svtlibs.smooth_picture(rng, H, W)) for _ in range(3)]
    geo = planes[0][1]
    stride, pad = geo[0][0], geo[0][1]
    sbs = [(x, y) for y in range(0, H, 64) for x in range(0, W, 64)]
    n = len(sbs)
    d = [dev(p[0][0]) for p in planes]
    p00 = [pic00(t, geo[0]) for t in d]
    c00 = [ctypes.c_void_p(p[0][0].ctypes.data + pad * stride + pad) for p in planes]
    org = dev(np.array(sbs, np.int16))
    for (npus, two, allpu, sub) in ((209, True, True, True), (209, True, True, False), (85, True, False, True), (85, False, True, True), (209, False, True, True)):
        sad = rng.integers(0, 5000, (2, n, 209)).astype(np.uint32)
        sad[1, :, ::3] = sad[0, :, ::3]                                  # ties between the lists
        mvx = rng.integers(-40, 41, (2, n, 209)); mvy = rng.integers(-40, 41, (2, n, 209))
        mv = ((mvy.astype(np.int64) << 2) & 0xffff) << 16 | ((mvx.astype(np.int64) << 2) & 0xffff)
        mv = mv.astype(np.uint32)
        bip, res = dsp.me_bipred(p00[0], stride, p00[1], stride, p00[2], stride, org, dev(sad[0].view(np.int32)), dev(mv[0].view(np.int32)),
                                 dev(sad[1].view(np.int32)) if two else None, dev(mv[1].view(np.int32)) if two else None, npus=npus,
                                 bipred_all_pus=allpu, sub_sad=sub)
        got_b = bip.cpu().numpy().view(np.uint32); got_r = dsp.me_results_as_rows(res)
        for i, (x, y) in enumerate(sbs):
            eb = np.zeros(209, np.uint32); er = np.zeros((209, 11), np.int32)
            bs2 = np.ascontiguousarray(np.stack([sad[0, i], sad[1, i]])); bm2 = np.ascontiguousarray(np.stack([mv[0, i], mv[1, i]]))
            O.svt_oracle_me_bipred_results(c00[0], stride, c00[1], stride, c00[2], stride, x, y, ptr(bs2), ptr(bm2), 2 if two else 1, npus, int(allpu),
                                           int(sub), ptr(eb), ptr(er))
            assert np.array_equal(got_r[i], er[:npus]), (npus, two, allpu, sub, i, np.argwhere(got_r[i] != er[:npus])[:4].tolist())
            assert np.array_equal(got_b[i], eb), (npus, two, allpu, sub, i)
    assert [dsp.lib.svt_hip_me_pu_storage_index(p) for p in range(209)] == [O.svt_oracle_me_raster_to_storage(p) for p in range(209)]
    assert dsp.lib.svt_hip_me_pu_storage_index(209) == -1 and dsp.lib.svt_hip_me_pu_storage_index(-1) == -1


def test_me_setup_argument_errors(dsp, pkg):
    z2 = torch.zeros((1, 2), dtype=torch.int16, device="cuda")
    pic = torch.zeros((256, 256), dtype=torch.uint8, device="cuda")
    good = dsp.MeSetupParams(64, 64, 64, 64, 16, 9, 0, 0, 0, 1)
    dsp.me_setup(pic.view(-1)[100 * 256 + 100:], 256, pic.view(-1)[100 * 256 + 100:], 256, z2, z2 + 64, None, None, good)
    for bad in (dsp.MeSetupParams(64, 64, 64, 64, 16, 9, 2, 1, 1, 1),          # second_best on a non-square region grid
                dsp.MeSetupParams(64, 64, 64, 64, 16, 9, 3, 3, 0, 1), dsp.MeSetupParams(64, 64, 64, 64, 0, 9, 0, 0, 0, 1),
                dsp.MeSetupParams(0, 64, 64, 64, 16, 9, 0, 0, 0, 1), dsp.MeSetupParams(64, 64, 64, 64, 16, 9, 1, 0, 0, 1)):
        with pytest.raises(pkg.SvtHipError):
            dsp.me_setup(pic, 256, pic, 256, z2, z2 + 64, None, None, bad)
    off = torch.zeros(1, dtype=torch.int32, device="cuda")
    area = torch.tensor([[0, 0, 8, 8]], dtype=torch.int16, device="cuda")
    with pytest.raises(pkg.SvtHipError):
        dsp.me_fullpel_search_areas(pic, 256, off, pic, 256, off, area, 128, 64)                       # more than 4096 points
    with pytest.raises(pkg.SvtHipError):
        dsp.me_fullpel_search_areas(pic, 256, off, pic, 256, off, area, 8, 8, flavour=3)
    # an area outside 1 .. max is skipped: the rows keep their incoming values
    bs = torch.full((1, 85), 777, dtype=torch.int32, device="cuda"); bm = torch.full((1, 85), 5, dtype=torch.int32, device="cuda")
    wide = torch.tensor([[0, 0, 16, 8]], dtype=torch.int16, device="cuda")
    dsp.me_fullpel_search_areas(pic.view(-1)[70 * 256 + 70:], 256, off, pic.view(-1)[70 * 256 + 70:], 256, off, wide, 8, 8, best_sad=bs, best_mv=bm)
    assert int(bs.min()) == 777 and int(bm.max()) == 5
