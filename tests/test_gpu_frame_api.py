"""GPU: the frame-level entry point svt_hip_encode_recon_frame (every (plane, size) group of a picture in one call, issued
concurrently) - BASELINE.json configs[3] at full size and one full 3840x2160 10-bit frame of configs[4] - against the
per-group entry points (bit-exact) and against the oracle on sampled blocks; HIP-graph capture; digest additivity."""
import ctypes

import numpy as np
import pytest
import torch

import svtlibs
from svtlibs import ptr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def planes_420(rng, w, h, bits):
    hi = 1 << bits
    dt = np.uint8 if bits == 8 else np.uint16
    out_s, out_p = {}, {}
    for name, (ph, pw) in (("Y", (h, w)), ("U", (h // 2, w // 2)), ("V", (h // 2, w // 2))):
        s = rng.integers(0, hi, size=(ph, pw)).astype(dt)
        p = np.clip(s.astype(np.int32) + rng.integers(-48, 49, size=s.shape), 0, hi - 1).astype(dt)
        out_s[name], out_p[name] = s, p
    return out_s, out_p


def to_dev(planes, bits):
    return {k: torch.from_numpy(v if bits == 8 else v.view(np.int16)).to(DEV) for k, v in planes.items()}


def check_groups_against_oracle(fp, src, pred, qrow, bits, step_of_side):
    O = svtlibs.oracle()
    n = 0
    for g in fp.groups:
        side = {4: 64, 3: 32, 2: 16, 1: 8, 0: 4}[g["tx_size"]]
        s_np, p_np = src[g["name"]], pred[g["name"]]
        pw = s_np.shape[1]
        rec = g["recon"].cpu().numpy()
        rec = rec if bits == 8 else rec.view(np.uint16)
        q = g["qcoeff"].cpu().numpy(); eob = g["eob"].cpu().numpy().view(np.uint16)
        xy = g["xy"].cpu().numpy().view(np.uint32)
        for i in range(0, xy.size, step_of_side[side]):
            y, x = int(xy[i] >> 16), int(xy[i] & 0xffff)
            rc = np.zeros(1024, np.int32); rq = np.zeros(1024, np.int32); rdq = np.zeros(1024, np.int32); reob = np.zeros(1, np.uint16)
            sp = ctypes.c_void_p(s_np.ctypes.data + (y * pw + x) * s_np.itemsize)
            pp = ctypes.c_void_p(p_np.ctypes.data + (y * pw + x) * p_np.itemsize)
            O.svt_oracle_fwd_quant_planes(sp, pw, pp, pw, int(bits != 8), bits, g["tx_size"], 0, ptr(qrow["zbin"]), ptr(qrow["round"]),
                                          ptr(qrow["quant"]), ptr(qrow["quant_shift"]), ptr(qrow["dequant"]), ptr(rc), ptr(rq), ptr(rdq),
                                          ptr(reob), None, None)
            blk = np.ascontiguousarray(p_np[y:y + side, x:x + side]).astype(np.uint16)
            O.svt_oracle_inv_txfm2d_add(ptr(rdq), ptr(blk), side, 0, g["tx_size"], bits)
            assert np.array_equal(rec[y:y + side, x:x + side].astype(np.uint16), blk), (g["name"], side, x, y)
            assert np.array_equal(q[i], rq[:q.shape[1]]) and eob[i] == reob[0], (g["name"], side, x, y)
            n += 1
    return n


def test_config4_1080p_frame_in_one_call(dsp, pkg):
    """C4 at its full size through the frame entry point: five luma sizes + four chroma sizes x two planes = 13 groups, one call; equal to
    the per-group entry points bit for bit, equal to the oracle on sampled blocks, and replayable from a HIP graph"""
    from cidana_svt_av1_amd import frames
    rng = np.random.default_rng(13596)
    src, pred = planes_420(rng, 1920, 1080, 8)
    qt = pkg.tables.quant_tables(8); qrow = {k: v[100].copy() for k, v in qt.items()}
    d_src, d_pred = to_dev(src, 8), to_dev(pred, 8)
    fp = frames.FramePass(dsp, pkg, d_src, d_pred)
    assert len(fp.groups) == 13 and fp.blocks == 480 + 1980 + 8040 + 32400 + 129600 + 2 * (480 + 1980 + 8040 + 32400)
    fp.run(qrow)
    torch.cuda.synchronize()
    n = check_groups_against_oracle(fp, src, pred, qrow, 8, {64: 37, 32: 151, 16: 601, 8: 2399, 4: 9001})
    assert n >= 90
    got = [(g["qcoeff"].clone(), g["eob"].clone(), g["recon"].clone()) for g in fp.groups]
    dig = fp.digest().cpu().numpy()
    # the same groups one call after the other; the 4x4 groups through the TWO-STAGE path (forward + quantise kernel, then the
    # inverse kernel adding onto a copy of the prediction): an independent check of the one-lane-per-block fused 4x4 kernel
    fp2 = frames.FramePass(dsp, pkg, d_src, d_pred)
    for g in fp2.groups:
        if g["tx_size"] == 0:
            _, q, dq, eob, _, _ = dsp.fwd_quant_planes(g["src"], g["src_stride"], g["pred"], g["pred_stride"], g["xy"], 0, 0, qrow, g["iscan"])
            dsp.inv_txfm2d_add(dq, g["recon"], 0, 0, 8, dst_stride=g["recon_stride"], dst_block_pitch=0, offsets=g["offsets"])
            g["qcoeff"], g["eob"] = q, eob
        else:
            r = dsp.encode_recon_planes(g["src"], g["src_stride"], g["pred"], g["pred_stride"], g["recon"], g["recon_stride"], g["xy"], g["tx_size"],
                                        0, qrow, g["iscan"])
            g["qcoeff"], g["eob"] = r["qcoeff"], r["eob"]
    torch.cuda.synchronize()
    for (q, e, r), g in zip(got, fp2.groups):
        assert torch.equal(q, g["qcoeff"]) and torch.equal(e, g["eob"]) and torch.equal(r, g["recon"]), (g["name"], g["tx_size"])
    assert np.array_equal(dig, fp2.digest().cpu().numpy())
    # HIP-graph capture of the call (fork / join over the library's internal streams becomes graph branches)
    fp3 = frames.FramePass(dsp, pkg, d_src, d_pred)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fp3.run(qrow); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            fp3.run(qrow)
    for g in fp3.groups:
        g["qcoeff"].zero_(); g["recon"].copy_(g["pred"])
    torch.cuda.synchronize()
    gr.replay(); torch.cuda.synchronize()
    ref = {(g["name"], g["tx_size"]): g for g in fp.groups}
    bad = [(g["name"], g["tx_size"], bool(torch.equal(g["qcoeff"], ref[(g["name"], g["tx_size"])]["qcoeff"])),
            bool(torch.equal(g["recon"], ref[(g["name"], g["tx_size"])]["recon"]))) for g in fp3.groups]
    assert all(b[2] and b[3] for b in bad), [b for b in bad if not (b[2] and b[3])]


def test_config5_one_full_4k_10bit_frame_and_gop_digest(dsp, pkg):
    """configs[4] at its full picture size: one 3840x2160 yuv420p10 frame, every CU size, bd 10, in one call; sampled oracle
    blocks; the digest a rank would all-reduce is additive over frames (two frames = the sum of the two single-frame runs)"""
    from cidana_svt_av1_amd import frames, sharding
    rng = np.random.default_rng(13597)
    qt = pkg.tables.quant_tables(10); qrow = {k: v[120].copy() for k, v in qt.items()}
    digs = []
    for f in range(2):
        src, pred = planes_420(rng, 3840, 2160, 10)
        fp = frames.FramePass(dsp, pkg, to_dev(src, 10), to_dev(pred, 10), is_16bit=True)
        assert len(fp.groups) == 13
        fp.run(qrow)
        torch.cuda.synchronize()
        if f == 0:
            n = check_groups_against_oracle(fp, src, pred, qrow, 10, {64: 149, 32: 601, 16: 2399, 8: 9601, 4: 38401})
            assert n >= 90
            for g in fp.groups:
                assert int((g["recon"].to(torch.int32) & 0xffff).max()) <= 1023
        digs.append(fp.digest().cpu().numpy())
        del fp
    total = digs[0] + digs[1]; total[2] %= sharding.DIGEST_MOD
    assert total[0] == 2 * digs[0][0] and (digs[0] != digs[1]).any()
    # GOP -> rank map of the 240-frame run: 8 GOPs of 30 frames, one per GPU on an 8-GPU node
    assert [sharding.gop_owner(g, 8) for g in range(8)] == list(range(8))
    assert sharding.gops_of_rank(8, 1, 2) == [1, 3, 5, 7]


def test_frame_call_rejects_bad_groups_before_enqueuing(dsp, pkg):
    from cidana_svt_av1_amd import frames
    rng = np.random.default_rng(5)
    src, pred = planes_420(rng, 128, 64, 8)
    fp = frames.FramePass(dsp, pkg, to_dev(src, 8), to_dev(pred, 8), luma_sizes=(32, 4))
    qt = pkg.tables.quant_tables(8); qrow = {k: v[60].copy() for k, v in qt.items()}
    fp.array[0].d_iscan = None
    with pytest.raises(pkg.SvtHipError):
        fp.run(qrow)
    torch.cuda.synchronize()


@pytest.mark.parametrize("bd", [8, 10])
def test_stacked_gop_call_equals_per_picture_calls(dsp, pkg, bd):
    """frames.FramePass on a stack of pictures [F, H, W] (one call for a GOP) against one FramePass per picture: every group's
    qcoeff / eob / recon, and the digest, which must not depend on how pictures are grouped into calls"""
    from cidana_svt_av1_amd import frames, sharding
    dev_ = torch.device("cuda:0")
    W, H, F = 320, 192, 5
    qrow = {k: v[120].copy() for k, v in pkg.tables.quant_tables(bd).items()}
    g = torch.Generator(device=dev_); g.manual_seed(50 + bd)
    shapes = {"Y": (H, W), "U": (H // 2, W // 2), "V": (H // 2, W // 2)}
    hi = 1 << bd
    dt = torch.uint8 if bd == 8 else torch.int16
    src = {k: torch.randint(0, hi, (F,) + s, dtype=torch.int32, device=dev_, generator=g).to(dt) for k, s in shapes.items()}
    pred = {k: (src[k].to(torch.int32) + torch.randint(-40, 41, src[k].shape, dtype=torch.int32, device=dev_, generator=g)).clamp_(0, hi - 1).to(dt) for k in shapes}
    st = frames.FramePass(dsp, pkg, src, pred, is_16bit=bd > 8)
    st.run(qrow)
    torch.cuda.synchronize()
    total = torch.zeros(4, dtype=torch.int64, device=dev_)
    for f in range(F):
        one = frames.FramePass(dsp, pkg, {k: src[k][f] for k in shapes}, {k: pred[k][f] for k in shapes}, is_16bit=bd > 8)
        one.run(qrow)
        torch.cuda.synchronize()
        for gs, go in zip(st.groups, one.groups):
            n1 = go["xy"].numel()
            assert torch.equal(gs["qcoeff"][f * n1:(f + 1) * n1], go["qcoeff"]), (f, gs["name"], gs["luma_size"])
            assert torch.equal(gs["eob"][f * n1:(f + 1) * n1], go["eob"])
            assert torch.equal(gs["recon"][f], go["recon"])
        total += one.digest()
    total[2] %= sharding.DIGEST_MOD
    assert torch.equal(total, st.digest())


@pytest.mark.parametrize("bd", [8, 10])
def test_frame_single_launch_equals_per_size_launches(dsp, pkg, bd):
    """svt_hip_tune("frame_single_launch", 1): enc_frame_kernel runs the same bodies as the per-size kernels for every group of the
    call in ONE launch - qcoeff / eob / recon of every group must be identical; odd plane sizes so that the last workgroups of the
    groups are partly empty; also a call with a group the single launch does not cover (falls back, still correct)"""
    from cidana_svt_av1_amd import frames
    dev_ = torch.device("cuda:0")
    W, H = 448 + 8, 320 + 12
    qrow = {k: v[90].copy() for k, v in pkg.tables.quant_tables(bd).items()}
    g = torch.Generator(device=dev_); g.manual_seed(70 + bd)
    shapes = {"Y": (H, W), "U": (H // 2, W // 2), "V": (H // 2, W // 2)}
    hi = 1 << bd
    dt = torch.uint8 if bd == 8 else torch.int16
    src = {k: torch.randint(0, hi, s, dtype=torch.int32, device=dev_, generator=g).to(dt) for k, s in shapes.items()}
    pred = {k: (src[k].to(torch.int32) + torch.randint(-40, 41, src[k].shape, dtype=torch.int32, device=dev_, generator=g)).clamp_(0, hi - 1).to(dt) for k in shapes}
    outs = []
    types = {64: 0, 32: 9, 16: 3, 8: 11, 4: 6} if bd == 8 else None      # DCT_DCT, IDTX, ADST_ADST, H_DCT, FLIPADST_FLIPADST
    try:
        for knob in (0, 1, 2):               # per-size launches, one launch (class 3), one launch per register class
            assert dsp.lib.svt_hip_tune(b"frame_single_launch", knob) == 0
            fp = frames.FramePass(dsp, pkg, src, pred, is_16bit=bd > 8, tx_types=types)
            fp.run(qrow)
            torch.cuda.synchronize()
            outs.append(fp)
    finally:
        dsp.lib.svt_hip_tune(b"frame_single_launch", -1)
    assert len(outs[0].groups) == 13
    for other in outs[1:]:
        for ga, gb in zip(outs[0].groups, other.groups):
            assert torch.equal(ga["qcoeff"], gb["qcoeff"]), (ga["name"], ga["luma_size"])
            assert torch.equal(ga["eob"], gb["eob"]) and torch.equal(ga["recon"], gb["recon"])
            assert not torch.equal(gb["recon"], pred[gb["name"]])


@pytest.mark.parametrize("bd", [8, 10])
def test_frame_call_with_rectangular_groups_by_register_class(dsp, pkg, bd):
    """all 19 transform sizes in one svt_hip_encode_recon_frame call: 13 square groups + the 14 rectangular sizes on the luma plane.
    The default policy sends a call with rectangular groups to the launches by register class (<= 16 | 32 and the long rectangles |
    64x64); every group must equal the per-size launches (knob 0) and, sampled, the oracle's chain"""
    from cidana_svt_av1_amd import frames
    dev_ = torch.device("cuda:0")
    W, H = 256 + 8, 192 + 4
    qrow = {k: v[70].copy() for k, v in pkg.tables.quant_tables(bd).items()}
    g = torch.Generator(device=dev_); g.manual_seed(170 + bd)
    shapes = {"Y": (H, W), "U": (H // 2, W // 2), "V": (H // 2, W // 2)}
    hi = 1 << bd
    dt = torch.uint8 if bd == 8 else torch.int16
    src = {k: torch.randint(0, hi, s, dtype=torch.int32, device=dev_, generator=g).to(dt) for k, s in shapes.items()}
    pred = {k: (src[k].to(torch.int32) + torch.randint(-30, 31, src[k].shape, dtype=torch.int32, device=dev_, generator=g)).clamp_(0, hi - 1).to(dt) for k in shapes}
    outs = []
    try:
        for knob in (0, -1, 2):
            assert dsp.lib.svt_hip_tune(b"frame_single_launch", knob) == 0
            fp = frames.FramePass(dsp, pkg, src, pred, is_16bit=bd > 8, rect_tx_sizes=tuple(range(5, 19)))
            assert len(fp.groups) == 13 + 14 and len(fp.groups) > 16          # more groups than one class table holds: split by class
            fp.run(qrow)
            torch.cuda.synchronize()
            outs.append(fp)
    finally:
        dsp.lib.svt_hip_tune(b"frame_single_launch", -1)
    for other in outs[1:]:
        for ga, gb in zip(outs[0].groups, other.groups):
            assert torch.equal(ga["qcoeff"], gb["qcoeff"]), (ga["name"], ga["luma_size"])
            assert torch.equal(ga["eob"], gb["eob"]) and torch.equal(ga["recon"], gb["recon"]), (ga["name"], ga["luma_size"])
            assert not torch.equal(gb["recon"], pred[gb["name"]])


def test_plain_c_host_of_the_frame_call_equals_the_python_path(dsp, pkg, tmp_path):
    """tests/c/frame_host.c (gcc, -lsvt_hip_dsp, no Python on the data path) on a 208x144 picture: its digest == the digest of
    frames.FramePass on the same samples (same LCG), i.e. the C host tables, uploads, group array and the frame call agree with
    the ctypes mirror"""
    import subprocess
    from test_abi_and_host import build_frame_host
    from cidana_svt_av1_amd import frames
    W, H, q, seed = 208, 144, 100, 12345
    exe = build_frame_host(tmp_path)
    S_EX = 16
    pr = subprocess.run([exe, str(W), str(H), str(q), str(seed), str(S_EX)], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, (pr.returncode, pr.stdout, pr.stderr)
    lines = pr.stdout.strip().splitlines()
    tok = lines[0].split()
    got = [int(tok[1]), int(tok[3]), int(tok[5]), int(tok[7])]
    tok = lines[1].split()
    assert tok[0] == "ex"
    got_ex = [int(tok[k]) for k in (2, 4, 6, 8, 10, 12)]
    # the same picture in numpy: LCG state * 1664525 + 1013904223, sample = state >> 8
    s = np.uint64(seed)
    planes_src, planes_pred = {}, {}
    for name, (ph, pw) in (("Y", (H, W)), ("U", (H // 2, W // 2)), ("V", (H // 2, W // 2))):
        n = ph * pw
        st = np.zeros(2 * n, np.uint64)
        cur = int(s)
        for i in range(2 * n):
            cur = (cur * 1664525 + 1013904223) & 0xffffffff
            st[i] = cur >> 8
        s = np.uint64(cur)
        a = (st[0::2] & 255).astype(np.int32)
        b = np.clip(a + (st[1::2] % 49).astype(np.int32) - 24, 0, 255)
        planes_src[name] = torch.from_numpy(a.astype(np.uint8).reshape(ph, pw)).cuda()
        planes_pred[name] = torch.from_numpy(b.astype(np.uint8).reshape(ph, pw)).cuda()
    qrow = {k: v[q].copy() for k, v in pkg.tables.quant_tables(8).items()}
    fp = frames.FramePass(dsp, pkg, planes_src, planes_pred)
    fp.run(qrow)
    torch.cuda.synchronize()
    assert got == [int(v) for v in fp.digest().cpu().numpy()]
    # the second line: svt_hip_encode_recon_frame_ex from C (one pass at luma 16, chroma from luma on every chroma block, level maps)
    fx = frames.FramePass(dsp, pkg, planes_src, {k: v.clone() for k, v in planes_pred.items()}, luma_sizes=(S_EX,))
    nb = fx.groups[1]["xy"].numel()
    b = torch.arange(nb, dtype=torch.int64)
    fx.add_cfl(((b % 33) - 16).to(torch.int32).cuda(), (((7 * b) % 33) - 16).to(torch.int32).cuda())
    fx.add_levels()
    fx.run_ex(qrow)
    torch.cuda.synchronize()
    dg = [int(v) for v in fx.digest().cpu().numpy()]
    pred_sum = sum(int(g["pred"].to(torch.int64).sum()) for g in fx.groups)
    levels_sum = 0
    for g in fx.groups:
        k = min(16 if g["name"] == "Y" else 8, 32)
        levels_sum += int(g["levels"][:, :(k + 4) * (k + 6) + 16].to(torch.int64).sum())
    assert got_ex == dg + [pred_sum, levels_sum]
    assert pred_sum != sum(int(v.to(torch.int64).sum()) for v in planes_pred.values())          # the chroma predictions did change


@pytest.mark.parametrize("bd,S", [(8, 8), (8, 32), (10, 16), (10, 64)])
def test_frame_call_with_chroma_from_luma_and_level_maps(dsp, pkg, bd, S):
    """svt_hip_encode_recon_frame_ex on a 4:2:0 picture: luma groups -> cfl_luma_subsampling_420 of the luma RECONSTRUCTION +
    subtract_average + cfl_predict of both chroma predictions in place (oracle, per block; EbCodingLoop.c:736-846) -> chroma groups
    on that prediction -> av1_txb_init_levels of every group's qcoeff (oracle).  The encode phases themselves are checked against
    the plain frame call (itself pinned to the per-group entry points and the oracle above) on the oracle's chroma-from-luma prediction."""
    from cidana_svt_av1_amd import frames
    O = svtlibs.oracle()
    rng = np.random.default_rng(1000 * bd + S)
    W, H = 256 + 64, 128 + 64
    src, pred = planes_420(rng, W, H, bd)
    qrow = {k: v[60].copy() for k, v in pkg.tables.quant_tables(bd).items()}
    d_src, d_pred = to_dev(src, bd), to_dev(pred, bd)
    fp = frames.FramePass(dsp, pkg, d_src, d_pred, luma_sizes=(S,), is_16bit=bd > 8)
    assert [g["name"] for g in fp.groups] == ["Y", "U", "V"] and fp.n_luma == 1
    c = S // 2
    # chroma-from-luma on two thirds of the chroma blocks, in a shuffled order
    xy_all = fp.groups[1]["xy"].cpu().numpy().view(np.uint32)
    sel = rng.permutation(xy_all.size)[: max(1, 2 * xy_all.size // 3)]
    xy = xy_all[sel]
    a_cb = rng.integers(-16, 17, size=xy.size).astype(np.int32)
    a_cr = rng.integers(-16, 17, size=xy.size).astype(np.int32)
    fp.add_cfl(torch.from_numpy(a_cb).to(DEV), torch.from_numpy(a_cr).to(DEV), xy=torch.from_numpy(xy.view(np.int32)).to(DEV))
    fp.add_levels(fill=0x5a)
    fp.run_ex(qrow)
    torch.cuda.synchronize()
    dt = np.uint8 if bd == 8 else np.uint16
    as_np = lambda t: t.cpu().numpy() if bd == 8 else t.cpu().numpy().view(np.uint16)
    # 1. luma phase == the per-group entry point
    ref = frames.FramePass(dsp, pkg, d_src, to_dev(pred, bd), luma_sizes=(S,), is_16bit=bd > 8)
    ref.run(qrow)
    torch.cuda.synchronize()
    assert torch.equal(ref.groups[0]["recon"], fp.groups[0]["recon"]) and torch.equal(ref.groups[0]["qcoeff"], fp.groups[0]["qcoeff"])
    # 2. the chroma prediction after the call == oracle chroma-from-luma on the luma reconstruction
    luma_rec = as_np(fp.groups[0]["recon"])
    lw = luma_rec.shape[1]
    lg = int(np.log2(c * c))
    want = {"U": pred["U"].copy(), "V": pred["V"].copy()}
    cw = want["U"].shape[1]
    for i, q in enumerate(xy):
        x, y = int(q & 0xffff), int(q >> 16)
        q3 = np.zeros((32, 32), np.int16)
        O.svt_oracle_cfl_luma_subsampling_420(ctypes.c_void_p(luma_rec.ctypes.data + (2 * y * lw + 2 * x) * luma_rec.itemsize), ctypes.c_int(bd > 8),
                                              ctypes.c_int(lw), ptr(q3), ctypes.c_int(2 * c), ctypes.c_int(2 * c))
        O.svt_oracle_subtract_average(ptr(q3), ctypes.c_int(c), ctypes.c_int(c), ctypes.c_int(c * c // 2), ctypes.c_int(lg))
        for name, al in (("U", a_cb), ("V", a_cr)):
            at = ctypes.c_void_p(want[name].ctypes.data + (y * cw + x) * want[name].itemsize)
            O.svt_oracle_cfl_predict(ptr(q3), at, ctypes.c_int(cw), at, ctypes.c_int(cw), ctypes.c_int(int(al[i])), ctypes.c_int(bd), ctypes.c_int(c),
                                     ctypes.c_int(c), ctypes.c_int(bd > 8))
    for k, name in ((1, "U"), (2, "V")):
        assert np.array_equal(as_np(fp.groups[k]["pred"]), want[name]), name
        assert not np.array_equal(want[name], pred[name])
    # 3. chroma phase == the per-group entry point (and the oracle on sampled blocks) on that prediction
    ref2 = frames.FramePass(dsp, pkg, d_src, to_dev({"Y": pred["Y"], **want}, bd), luma_sizes=(S,), is_16bit=bd > 8)
    ref2.run(qrow)
    torch.cuda.synchronize()
    for k in (1, 2):
        assert torch.equal(ref2.groups[k]["recon"], fp.groups[k]["recon"]) and torch.equal(ref2.groups[k]["qcoeff"], fp.groups[k]["qcoeff"])
        assert torch.equal(ref2.groups[k]["eob"], fp.groups[k]["eob"])
    assert check_groups_against_oracle(fp, src, {"Y": pred["Y"], **want}, qrow, bd, {64: 1, 32: 2, 16: 5, 8: 17, 4: 61}) > 20
    # 4. level maps of every group == oracle on sampled blocks; interior property on all; slack untouched
    for g in fp.groups:
        w = h = min({4: 64, 3: 32, 2: 16, 1: 8, 0: 4}[g["tx_size"]], 32)
        size = (w + 4) * (h + 6) + 16
        lv = g["levels"].cpu().numpy()
        q = g["qcoeff"].cpu().numpy()
        assert (lv[:, size:] == 0x5a).all()
        body = lv[:, :(w + 4) * (h + 6)].reshape(-1, h + 6, w + 4)
        assert np.array_equal(body[:, 2:2 + h, :w], np.minimum(np.abs(q.astype(np.int64)), 127).astype(np.uint8).reshape(-1, h, w))
        assert not body[:, :2].any() and not body[:, 2 + h:].any() and not body[:, :, w:].any() and not lv[:, (w + 4) * (h + 6):size].any()
        for i in range(0, q.shape[0], 7):
            one = np.full(lv.shape[1], 0x5a, np.uint8)
            O.svt_oracle_txb_init_levels(ptr(np.ascontiguousarray(q[i])), ctypes.c_int(w), ctypes.c_int(h), ctypes.c_void_p(one.ctypes.data + 2 * (w + 4)))
            assert np.array_equal(lv[i], one), (g["name"], i)


def test_frame_ex_argument_errors_enqueue_nothing(dsp, pkg):
    from cidana_svt_av1_amd import frames
    rng = np.random.default_rng(5)
    src, pred = planes_420(rng, 128, 64, 8)
    qrow = {k: v[60].copy() for k, v in pkg.tables.quant_tables(8).items()}
    fp = frames.FramePass(dsp, pkg, to_dev(src, 8), to_dev(pred, 8), luma_sizes=(16,))
    n = fp.groups[1]["xy"].numel()
    z = torch.zeros(n, dtype=torch.int32, device=DEV)
    fp.add_cfl(z, z)
    before = [g["recon"].clone() for g in fp.groups]
    fp.cfl_groups[0]["width"] = 6                                    # not a chroma transform side
    fp.cfl_array = dsp.make_frame_cfl_groups(fp.cfl_groups)
    with pytest.raises(pkg.SvtHipError):
        fp.run_ex(qrow)
    fp.cfl_groups[0]["width"] = 8
    fp.cfl_array = dsp.make_frame_cfl_groups(fp.cfl_groups)
    fp.add_levels()
    fp.levels_array[2].levels_block_pitch = 64                       # smaller than the 8x8 block's 12 * 14 + 16 bytes
    with pytest.raises(pkg.SvtHipError):
        fp.run_ex(qrow)
    torch.cuda.synchronize()
    assert all(torch.equal(a, g["recon"]) for a, g in zip(before, fp.groups))      # nothing ran
    fp.levels_array[2].levels_block_pitch = fp.groups[2]["levels"].shape[1]
    fp.run_ex(qrow)                                                  # alpha 0 everywhere: the prediction does not change
    torch.cuda.synchronize()
    assert torch.equal(fp.groups[1]["pred"], to_dev(pred, 8)["U"]) and not torch.equal(before[0], fp.groups[0]["recon"])


def test_malloc_spread_allocates_usable_buffers_far_apart(dsp, pkg):
    """svt_hip_malloc_spread / SvtHipDsp.alloc_spread (arrays a kernel writes at the same time should lie 32 GiB apart):
    the buffers are distinct and usable; bad arguments are refused"""
    L = dsp.lib
    L.svt_hip_malloc_spread.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    nbytes = (ctypes.c_size_t * 3)(1 << 26, 1 << 26, 1 << 20)
    ptrs = (ctypes.c_void_p * 3)()
    assert L.svt_hip_malloc_spread(nbytes, 3, 0, ptrs) == 0
    p = [int(v) for v in ptrs]
    assert all(p) and len(set(p)) == 3          # (virtual addresses say nothing about the distance that matters: small buffers reuse holes)
    host = np.arange(1 << 18, dtype=np.int32)
    back = np.zeros_like(host)
    for q in p:                       # each buffer takes and returns data
        assert L.svt_hip_memcpy_h2d(ctypes.c_void_p(q), host.ctypes.data, host.nbytes, None) == 0
        assert L.svt_hip_memcpy_d2h(back.ctypes.data, ctypes.c_void_p(q), host.nbytes, None) == 0
        assert L.svt_hip_stream_sync(None) == 0 and np.array_equal(host, back)
        L.svt_hip_free(ctypes.c_void_p(q))
    assert L.svt_hip_malloc_spread(nbytes, 65, 0, ptrs) != 0 and L.svt_hip_malloc_spread(None, 3, 0, ptrs) != 0
    ts = dsp.alloc_spread([((1 << 20,), torch.int32), ((1 << 20,), torch.int32)], gap_bytes=1 << 30)
    assert len(ts) == 2 and ts[0].data_ptr() != ts[1].data_ptr() and ts[0].is_cuda
    ts[0].fill_(3); ts[1].fill_(5)
    assert int(ts[0].sum()) == 3 << 20 and int(ts[1].sum()) == 5 << 20
