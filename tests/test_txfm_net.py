"""CPU: the butterfly-network builder that the HIP device code is generated from
(cidana-svt-av1_amd/tools/txfm_net.py), evaluated in numpy, against the oracle's
independent loop-based 1-D transforms; the generated header must be current; and
configuration C1 of BASELINE.json (8x8 DCT_DCT, 1k random residual blocks on the
CPU path) including the reference's floating-point tolerance test
(test/FwdTxfm2dTest.cc:226-246)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import svtlibs
from svtlibs import ptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "cidana-svt-av1_amd", "tools")
sys.path.insert(0, TOOLS)
import txfm_net as T  # noqa: E402

KIND = {"dct": 0, "adst": 1, "idtx": 3}
CASES = [("dct", n) for n in (4, 8, 16, 32, 64)] + [("adst", n) for n in (4, 8, 16)] + [("idtx", n) for n in (4, 8, 16, 32, 64)]


def test_constant_tables_match_golden():
    g = np.load(os.path.join(ROOT, "tests", "golden", "tables.npz"))
    for b in range(10, 17):
        assert T.cospi_table(b) == list(g["cospi"][b - 10])
        assert T.SINPI[b] == list(g["sinpi"][b - 10])


@pytest.mark.parametrize("kind,n", CASES)
def test_forward_net_equals_oracle(kind, n):
    O = svtlibs.oracle()
    rng = np.random.default_rng(n)
    net = T.build_fwd(kind, n)
    for cos_bit in (10, 11, 12, 13):
        x = rng.integers(-(1 << 15), 1 << 15, size=(64, n)).astype(np.int32)
        x[0] = 0; x[1] = (1 << 15) - 1; x[2] = -(1 << 15)
        ref = np.zeros_like(x)
        for i in range(x.shape[0]):
            O.svt_oracle_fwd_txfm1d(KIND[kind], n, ptr(np.ascontiguousarray(x[i])), ptr(ref[i]), cos_bit)
        assert np.array_equal(T.evaluate(net, x, cos_bit), ref)


@pytest.mark.parametrize("kind,n", CASES)
def test_inverse_net_equals_oracle(kind, n):
    O = svtlibs.oracle()
    rng = np.random.default_rng(1000 + n)
    net = T.build_inv(kind, n)
    for bits in (16, 18, 20):
        x = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), size=(64, n)).astype(np.int32)
        x[:20] >>= 5
        ref = np.zeros_like(x)
        for i in range(x.shape[0]):
            O.svt_oracle_inv_txfm1d(KIND[kind], n, ptr(np.ascontiguousarray(x[i])), ptr(ref[i]), 12, bits)
        assert np.array_equal(T.evaluate(net, x, 12, {"stage": bits}), ref)


def test_idct64_low32_net_equals_oracle():
    """The zero-propagated 64-point inverse (inputs 32..63 = 0, the only way AV1 uses it) against the
    oracle's full av1_idct64_new restatement on such inputs, including extreme values."""
    O = svtlibs.oracle()
    rng = np.random.default_rng(6464)
    net = T.specialize_zero_inputs(T.build_inv("dct", 64), 32, "idct64_low32")
    assert sum(1 for op in net.ops if op[0] == "in") == 32
    for bits in (16, 18, 20):
        x = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), size=(96, 64)).astype(np.int32)
        x[:20] >>= 5
        x[20] = (1 << (bits - 1)) - 1; x[21] = -(1 << (bits - 1))
        x[:, 32:] = 0
        ref = np.zeros_like(x)
        for i in range(x.shape[0]):
            O.svt_oracle_inv_txfm1d(KIND["dct"], 64, ptr(np.ascontiguousarray(x[i])), ptr(ref[i]), 12, bits)
        assert np.array_equal(T.evaluate(net, x, 12, {"stage": bits}), ref)


def test_generated_header_is_current():
    """csrc/gen/txfm1d_gen.h must be exactly what gen_device.py emits."""
    cur = open(os.path.join(ROOT, "cidana-svt-av1_amd", "csrc", "gen", "txfm1d_gen.h")).read()
    import gen_device
    assert cur == gen_device.render()


def _ref_dct2d_float(x):
    """double-precision orthonormal-free DCT-II as test/ref/TxfmRef.cc:142-160"""
    n = x.shape[0]
    k = np.arange(n)
    m = np.cos(np.pi * (2 * k[None, :] + 1) * k[:, None] / (2 * n))
    m[0] *= 1 / np.sqrt(2)
    return m @ x @ m.T


def test_config1_8x8_dct_1k_blocks_cpu():
    """BASELINE.json configs[0]: FwdTxfm2d 8x8 DCT_DCT 8-bit, 1k random residual blocks,
    reference C path on the CPU: oracle bit-exact vs the golden-pinned path, and within
    max_error_ls[TX_8X8] = 5 (x scale 2) of the floating-point DCT."""
    O = svtlibs.oracle()
    R = svtlibs.ref()
    rng = np.random.default_rng(13596)
    worst = 0.0
    for _ in range(1000):
        x = rng.integers(-255, 256, size=(8, 8)).astype(np.int16)
        o = np.zeros(64, np.int32)
        O.svt_oracle_fwd_txfm2d(ptr(x), ptr(o), ctypes.c_uint32(8), 0, 1, 8)
        if R is not None:
            r = np.zeros(64, np.int32)
            R.Av1TransformTwoD_8x8_c(ptr(x), ptr(r), ctypes.c_uint32(8), ctypes.c_int(0), ctypes.c_uint8(8))
            assert np.array_equal(o, r)
        f = _ref_dct2d_float(x.astype(np.float64)) * 2.0     # 8x8: shifts 2,-1,0 -> net x2 (FwdTxfm2dTest.cc:108-118)
        worst = max(worst, float(np.abs(o.reshape(8, 8) - np.round(f)).max()) / 2.0)
    assert worst <= 5.0


def test_clamp_free_bound_makes_every_stage_clamp_a_no_op():
    """txfm_net.clamp_free_bound: while gain * L1(x) + slack <= the clamp bound, the inverse networks with and without their stage
    clamps agree.  Inputs AT the bound, with the mass on one, two, a few or all inputs and every sign pattern drawn at random
    (the kernels branch on exactly this L1 test, csrc/kernel_txfm.h inv1d)."""
    import math
    rng = np.random.default_rng(4242)
    nets = [T.build_inv(kind, n) for kind, sizes in (("dct", (4, 8, 16, 32, 64)), ("adst", (8, 16))) for n in sizes]
    nets.append(T.specialize_zero_inputs(T.build_inv("dct", 64), 32, "idct64_low32"))
    for net in nets:
        g, e = T.clamp_free_bound(net)
        assert g > 0
        live = 32 if net.name == "idct64_low32" else net.n_in
        for bits in (16, 18):
            hi = (1 << (bits - 1)) - 1
            l1 = int(((hi - (math.ceil(e) + 1)) << 10) // (math.ceil(g * 1024) + 1))          # svt_clamp_free_l1
            rows = []
            for i in range(live):                                    # all the mass on one input, both signs
                for sgn in (1, -1):
                    v = np.zeros(net.n_in, np.int64); v[i] = sgn * min(l1, hi); rows.append(v)
            for _ in range(600):                                     # random splits of exactly l1 over k inputs, random signs
                k = int(rng.choice([2, 3, 4, live // 2, live]))
                idx = rng.choice(live, size=min(k, live), replace=False)
                cuts = np.sort(rng.integers(0, l1 + 1, size=len(idx) - 1))
                mag = np.diff(np.concatenate([[0], cuts, [l1]]))
                v = np.zeros(net.n_in, np.int64); v[idx] = mag * rng.choice([-1, 1], size=len(idx)); rows.append(v)
            for _ in range(200):                                     # flat: the same magnitude everywhere
                v = np.zeros(net.n_in, np.int64); v[:live] = (l1 // live) * rng.choice([-1, 1], size=live); rows.append(v)
            x = np.stack(rows)
            assert np.abs(x).sum(axis=1).max() <= l1
            a = T.evaluate(net, x, 12, {"stage": bits})
            b = T.evaluate(net, x, 12, {"stage": 40})                # clamps that can never act
            assert np.array_equal(a, b), (net.name, bits)
    # and the bound is not vacuous: far above it the clamps do act
    net = T.build_inv("dct", 32)
    x = np.full((1, 32), 30000, np.int64)
    assert not np.array_equal(T.evaluate(net, x, 12, {"stage": 16}), T.evaluate(net, x, 12, {"stage": 40}))
