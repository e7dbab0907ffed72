"""GPU: the rows of SURVEY 8(f) chained on a y4m clip (tools/pipeline_y4m.py): picture input + decimation -> HME 0/1/2 -> full-pel ME
(209 PUs) -> open-loop intra search -> encode pass, each stage reading the previous stage's device buffers.  The stages are pinned
to the reference one by one elsewhere; this checks the GLUE - buffer geometry, strides, origins, vector units - with a clip whose
motion is known: every picture is the previous one displaced by (4, 4) samples (a multiple of 4, so that the 1/4 and 1/16 pictures
of consecutive frames are displaced copies as well), so interior SBs must find exactly that vector with SAD 0, at every PU shape."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_tool():
    spec = importlib.util.spec_from_file_location("pipeline_y4m", os.path.join(ROOT, "tools", "pipeline_y4m.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("bd", [8, 10])
def test_pipeline_finds_the_known_motion_and_is_deterministic(dsp, tmp_path, bd):
    tool = load_tool()
    path = str(tmp_path / "pan.y4m")
    w, h, nf = 448, 320, 4
    tool.synthetic_clip(path, w, h, nf, pan=(4, 4), bd=bd)
    digests = []
    for rep in range(2):
        p = tool.Pipeline(dsp, path)
        outs = []
        while True:
            ref_before = p.prev["pyr"][2].clone() if p.prev is not None else None      # the reference luma plane of the coming step
            o = p.step()
            if o is None:
                break
            if bd == 8 and rep == 0 and ref_before is not None:
                # every SB, corners included: the search area the device derived (best region, CheckZeroZeroCenter, clipping against the
                # picture) is the oracle's for the HME results the device produced
                check_areas_against_oracle(p, o, ref_before, w, h)
            outs.append(o)
        assert len(outs) == nf and "me_mv" not in outs[0] and "me_mv" in outs[1]
        for o in outs[1:]:
            sad = o["me_sad"].cpu().numpy().view(np.uint32)
            mv = o["me_mv"].cpu().numpy().view(np.uint32)
            sbs = p.sb_xy.cpu().numpy()
            # SBs whose 64x64 block and displaced match lie inside the picture (the borders are replicated samples, not the moving texture)
            inner = [i for i, (x, y) in enumerate(sbs) if x + 64 + 4 <= w and y + 64 + 4 <= h]
            assert len(inner) >= 12
            for i in inner:
                assert (sad[i] == 0).all(), (i, sad[i][:8])
                assert ((mv[i] & 0xffff) == 4 * 4).all() and ((mv[i] >> 16) == 4 * 4).all(), (i, mv[i][:4])
            # HME level 2 arrives at the same vector for those SBs
            hm = o["hme_mv"].cpu().numpy()
            assert all(tuple(hm[i]) == (4, 4) for i in inner), hm[inner[:6]]
            assert tool.digest_of(o)["enc_digest"][0] > 0
        digests.append([tool.digest_of(o) for o in outs])
        p.pi.close()
    assert digests[0] == digests[1]


def check_areas_against_oracle(p, o, ref_luma, w, h):
    import ctypes
    import svtlibs
    from svtlibs import ptr
    O = svtlibs.oracle()
    cur = o["_cur_luma"].cpu().numpy(); ref = ref_luma.cpu().numpy()
    stride, pad = cur.shape[1], p.pad
    c00 = ctypes.c_void_p(cur.ctypes.data + pad * stride + pad); r00 = ctypes.c_void_p(ref.ctypes.data + pad * stride + pad)
    hs = o["hme_sad"].cpu().numpy(); hm = o["hme_regions_mv"].cpu().numpy()
    area = o["me_area"].cpu().numpy(); centre = o["hme_mv"].cpu().numpy()
    narrow = 0
    for i, (x, y) in enumerate(p.sb_xy.cpu().numpy().tolist()):
        s4 = np.zeros((2, 2), np.uint64); x4 = np.zeros((2, 2), np.int16); y4 = np.zeros((2, 2), np.int16)
        for r in range(4):
            s4[r % 2, r // 2] = hs[r, i]; x4[r % 2, r // 2] = hm[r, i, 0]; y4[r % 2, r // 2] = hm[r, i, 1]
        ce = np.zeros(2, np.int16); ar = np.zeros(4, np.int16)
        sw, sh = min(64, w - x), min(64, h - y)
        O.svt_oracle_me_setup(c00, stride, r00, stride, x, y, sw, sh, w, h, w, h, int(sh == 64), ptr(s4), ptr(x4), ptr(y4), 2, 2, 0, 1, p.SW, p.SH,
                              ptr(ce), ptr(ar))
        assert area[i].tolist() == ar.tolist() and centre[i].tolist() == ce.tolist(), (i, (x, y), area[i], ar, centre[i], ce)
        narrow += int(ar[2] < p.SW or ar[3] < p.SH)
    return narrow                            # (areas clipped against the picture; a (4, 4) pan clips none - test_gpu_me_setup.py covers those)


def test_pipeline_as_a_hip_graph_equals_the_eager_chain(dsp, tmp_path):
    """the per-picture analysis captured once and replayed == issuing every launch from Python"""
    tool = load_tool()
    path = str(tmp_path / "pan.y4m")
    tool.synthetic_clip(path, 320, 192, 6, pan=(4, 4))
    got = []
    for use_graph in (False, True):
        p = tool.Pipeline(dsp, path, use_graph=use_graph)
        d = []
        while True:
            o = p.step()
            if o is None:
                break
            torch.cuda.synchronize()
            d.append(tool.digest_of(o))
        assert use_graph == (p.graph is not None)
        got.append(d)
        p.pi.close()
        del p
    assert got[0] == got[1] and len(got[0]) == 6
