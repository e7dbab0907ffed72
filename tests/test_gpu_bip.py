"""GPU: svt_hip_build_intra_predictors_batch (the neighbour-availability glue of av1_predict_intra_block) against
the reference's own build_intra_predictors{,_high} and av1_predict_intra_block{,_16bit} outputs (tests/golden/bip.npz) and
against the oracle on random batches: every mode x size, angle deltas, availability patterns, edge-filter types, 8 / 10 bit,
dense and in-picture (unaligned) destinations."""
import ctypes
import os

import numpy as np
import pytest
import torch

import svtlibs
from svtlibs import TX_H, TX_W, ptr

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PITCH = 1 + 2 * 64 + 15          # corner + 2 * 64 samples, padded


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def neigh_rows(edges, is16):
    """[n, >= 16 + 129] arrays whose element 16 is above[0] -> the batch layout: element 0 = corner"""
    a = np.stack([e[15:15 + PITCH] for e in edges])
    return dev(a.astype(np.uint16).view(np.int16) if is16 else a.astype(np.uint8))


def run_groups(dsp, cases):
    """cases: list of (key=(tx, is16, bd), blk descriptor 8 bytes, top, left, expected); one launch per key"""
    n = 0
    for key in sorted({c[0] for c in cases}):
        sel = [c for c in cases if c[0] == key]
        tx, is16, bd = key
        blks = dev(np.array([c[1] for c in sel], np.uint8))
        got = dsp.build_intra_predictors(neigh_rows([c[2] for c in sel], is16), neigh_rows([c[3] for c in sel], is16), blks, tx, bd=bd)
        got = got.cpu().numpy()
        if is16:
            got = got.view(np.uint16)
        for i, c in enumerate(sel):
            assert np.array_equal(got[i], c[4]), (key, c[1], np.argwhere(got[i] != c[4])[:4])
            n += 1
    return n


def blk_bytes(mode, ad, ft, dis, n_top, n_tr, n_left, n_bl):
    return [mode, ad & 0xff, ft, dis, n_top, n_tr, n_left, n_bl]


def test_build_intra_predictors_reference_fixture(dsp):
    g = np.load(os.path.join(G, "bip.npz"))
    cases = []
    for p, top, left, exp in svtlibs.bip_fixture_cases(g):
        cases.append(((p["tx"], p["is16"], p["bd"]), blk_bytes(p["mode"], p["angle_delta"], p["filt_type"], p["disable_edge_filter"],
                                                              p["n_top"], p["n_tr"], p["n_left"], p["n_bl"]), top, left, exp))
    assert run_groups(dsp, cases) == 700


def test_predict_intra_block_reference_fixture(dsp, pkg):
    """picture position -> svt_hip_intra_neighbor_px (host) -> descriptor -> device; expected = the reference's av1_predict_intra_block"""
    g = np.load(os.path.join(G, "bip.npz"))
    O = svtlibs.oracle()
    cases = []
    for c, exp in svtlibs.pib_fixture_cases(g):
        pos = pkg.SvtHipDsp.IntraPos(c["is16"], 16, c["mi_rows"], c["mi_cols"], int(c["tile"][0]), int(c["tile"][1]), int(c["tile"][2]),
                                     int(c["tile"][3]), c["partition"], c["bsize"], c["tx"], c["plane"], c["micol"] * 4, c["mirow"] * 4,
                                     c["col_off"], c["row_off"], c["wpx"], c["hpx"])
        n_top, n_tr, n_left, n_bl = dsp.intra_neighbor_px(pos)
        _, out5 = svtlibs.oracle_predict_intra_block(O, c)            # filt_type is the caller's (get_filt_type reads the mode infos)
        cases.append(((c["tx"], c["is16"], c["bd"]), blk_bytes(c["mode"], c["angle_delta"], int(out5[4]), 0, n_top, n_tr, n_left, n_bl),
                      c["top"][1:], c["left"][1:], exp))             # the case arrays keep the corner at element 16, above[0] at 17
    assert run_groups(dsp, cases) == 300


@pytest.mark.parametrize("is16", [0, 1])
def test_build_intra_predictors_random_batches_vs_oracle(dsp, is16):
    """per transform size one batch of 13 modes x 40 draws with random availability, written INTO a picture at unaligned positions"""
    O = svtlibs.oracle()
    rng = np.random.default_rng(31 + is16)
    bd = 10 if is16 else 8
    dt = np.uint16 if is16 else np.uint8
    es = 2 if is16 else 1
    for tx in range(19):
        w, h = TX_W[tx], TX_H[tx]
        n = 13 * 40
        tops = rng.integers(0, 1 << bd, (n, 16 + 160)).astype(dt); lefts = rng.integers(0, 1 << bd, (n, 16 + 160)).astype(dt)
        if tx % 3 == 0:                                  # flat neighbourhoods: ties in PAETH, equal DC sums
            tops[: n // 4] = tops[: n // 4, :1]; lefts[: n // 4] = tops[: n // 4, :1]
        blks, exps = [], []
        for i in range(n):
            mode = i % 13
            ad = int(rng.integers(-3, 4)) if 1 <= mode <= 8 else 0
            n_top = int(rng.choice([0, w, int(rng.integers(1, w // 4 + 1)) * 4])); n_left = int(rng.choice([0, h, int(rng.integers(1, h // 4 + 1)) * 4]))
            n_tr = int(rng.choice([0, h, int(rng.integers(0, h + 1))])) if n_top == w else 0
            n_bl = int(rng.choice([0, w, int(rng.integers(0, w + 1))])) if n_left == h else 0
            dis = int(rng.integers(0, 5) == 0); ft = int(rng.integers(0, 2))
            d = np.zeros((h, w), dt)
            O.svt_oracle_build_intra_predictors(is16, ctypes.c_void_p(tops[i].ctypes.data + 16 * es), ctypes.c_void_p(lefts[i].ctypes.data + 16 * es),
                                                ptr(d), w, mode, ad, tx, dis, n_top, n_tr, n_left, n_bl, ft, bd)
            blks.append(blk_bytes(mode, ad, ft, dis, n_top, n_tr, n_left, n_bl)); exps.append(d)
        # destination: a picture, blocks on a grid with odd origins and a stride that is not a multiple of 4
        stride = 40 * (w + 3) + 1
        rows = (n + 39) // 40
        pic = torch.zeros((rows * (h + 2) + 2, stride), dtype=torch.int16 if is16 else torch.uint8, device="cuda")
        offs = np.array([(1 + (i // 40) * (h + 2)) * stride + 1 + (i % 40) * (w + 3) for i in range(n)], np.uint32)
        dsp.build_intra_predictors(neigh_rows(list(tops), is16), neigh_rows(list(lefts), is16), dev(np.array(blks, np.uint8)), tx, bd=bd,
                                   dst=pic, dst_stride=stride, dst_offsets=dev(offs.view(np.int32)))
        got = pic.cpu().numpy()
        if is16:
            got = got.view(np.uint16)
        mask = np.zeros(got.shape, bool)
        for i in range(n):
            y, x = divmod(int(offs[i]), stride)
            assert np.array_equal(got[y:y + h, x:x + w], exps[i]), (tx, is16, blks[i], np.argwhere(got[y:y + h, x:x + w] != exps[i])[:4])
            mask[y:y + h, x:x + w] = True
        assert not got[~mask].any(), "wrote outside the blocks"


def test_build_intra_predictors_argument_errors(dsp, pkg):
    t = torch.zeros((4, PITCH), dtype=torch.uint8, device="cuda")
    b = torch.zeros((4, 8), dtype=torch.uint8, device="cuda")
    with pytest.raises(pkg.SvtHipError):
        dsp.build_intra_predictors(t, t, b, 19)
    with pytest.raises(pkg.SvtHipError):
        dsp.build_intra_predictors(t[:, :100], t[:, :100], b, 4)          # 64x64 needs 129 samples per edge
    with pytest.raises(pkg.SvtHipError):
        dsp.build_intra_predictors(t, t, b, 0, bd=10)                      # 8-bit samples with bd 10


@pytest.mark.parametrize("is16", [0, 1])
def test_ordered_batch_equals_plain_batch_and_order_is_sorted_by_kind(dsp, is16):
    """svt_hip_intra_order_blocks_batch: a permutation of the batch's block indices, grouped by predictor kind inside tiles of 4 096 blocks (so that a
    wave runs one kind's code); svt_hip_build_intra_predictors_ordered_batch through it writes exactly what the plain call writes (every size)"""
    rng = np.random.default_rng(91 + is16)
    bd = 10 if is16 else 8
    dt = np.uint16 if is16 else np.uint8
    for tx in range(19):
        w, h = TX_W[tx], TX_H[tx]
        n = 3001 if tx % 5 else 9001
        tops = rng.integers(0, 1 << bd, (n, 16 + 160)).astype(dt); lefts = rng.integers(0, 1 << bd, (n, 16 + 160)).astype(dt)
        blks = np.zeros((n, 8), np.uint8)
        blks[:, 0] = rng.integers(0, 13, n)
        blks[:, 1] = (rng.integers(-3, 4, n) * ((blks[:, 0] >= 1) & (blks[:, 0] <= 8))).astype(np.int8).view(np.uint8)
        blks[:, 2] = rng.integers(0, 2, n); blks[:, 3] = rng.integers(0, 5, n) == 0
        blks[:, 4] = rng.choice([0, w], n); blks[:, 6] = rng.choice([0, h], n)
        blks[:, 5] = np.where(blks[:, 4] == w, rng.choice([0, h], n), 0); blks[:, 7] = np.where(blks[:, 6] == h, rng.choice([0, w], n), 0)
        d_blks = dev(blks)
        T, L = neigh_rows(list(tops), is16), neigh_rows(list(lefts), is16)
        plain = dsp.build_intra_predictors(T, L, d_blks, tx, bd=bd)
        order = dsp.intra_order_blocks(d_blks, tx)
        o = order.cpu().numpy()
        assert sorted(o.tolist()) == list(range(n))
        ordered = dsp.build_intra_predictors(T, L, d_blks, tx, bd=bd, order=order)
        assert torch.equal(plain, ordered), tx
        # grouped: the (coarse) kind of the blocks along the order changes at most 12 times
        def kind(b):
            m, ad, nt, nl = int(b[0]), int(np.int8(b[1])), int(b[4]), int(b[6])
            if 1 <= m <= 8:
                a = [0, 90, 180, 45, 135, 113, 157, 203, 67][m] + 3 * ad
                if (a >= 180 and nl == 0) or (a <= 90 and nt == 0):
                    return "c"
                return "v" if a == 90 else ("h" if a == 180 else ("z1" if a < 90 else ("z2" if a < 180 else "z3")))
            if m == 0:
                return ("dc" if nt else "dl") if nl else ("dt" if nt else "c")
            return "m%d" % m
        # grouped inside tiles of SVT_HIP_INTRA_ORDER_TILE blocks: every tile is a permutation of its own indices and the (coarse)
        # kind of the blocks along it changes at most 12 times
        for t0 in range(0, n, 4096):
            tile = o[t0:t0 + 4096]
            assert sorted(tile.tolist()) == list(range(t0, min(n, t0 + 4096))), (tx, t0)
            ks = [kind(blks[i]) for i in tile]
            assert sum(1 for a, b in zip(ks, ks[1:]) if a != b) <= 12, (tx, t0)
