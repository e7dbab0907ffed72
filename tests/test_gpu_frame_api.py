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
    pr = subprocess.run([exe, str(W), str(H), str(q), str(seed)], capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, (pr.returncode, pr.stdout, pr.stderr)
    tok = pr.stdout.split()
    got = [int(tok[1]), int(tok[3]), int(tok[5]), int(tok[7])]
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
