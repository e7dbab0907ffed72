#!/usr/bin/env python3
"""Butterfly-network builder for the AV1 integer 1-D transforms.

This is the single description of the 1-D transform dataflow from which the
straight-line, register-resident device code in ``csrc/gen/`` is emitted
(``gen_device.py``).  A network is an SSA list of integer ops; the only
rounding op is ``hb`` (the reference's ``half_btf``):

    hb(wa, a, wb, b) = (wa*a + wb*b + (1 << (bit-1))) >> bit

The networks are built from closed-form rules (recursive even/odd split,
bit-reversed output order, mirrored rotation pairs), NOT transcribed from the
reference's unrolled stage listings; they are proven equal to the reference's
1-D kernels (EbTransforms.c:1314-3660 forward, :5465-7748 inverse) by
tests/test_txfm_net.py on random vectors against oracle/_ref and against the
committed golden fixtures.

Weights are symbolic ``(sign, j)`` = sign * cospi[j] (j in 0..63 of the
``cos(pi*j/128)`` table, EbTransforms.c:1242) so one network serves every
cos_bit; ``('sin', sign, j)`` refers to the sinpi table used by ADST4.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np

COS_BIT_MIN = 10

# ---------------------------------------------------------------------------
# constant tables (AV1 spec constants; values checked against the reference's
# av1_cospi_arr_data / av1_sinpi_arr_data in tests)
# ---------------------------------------------------------------------------


def cospi_table(cos_bit: int) -> List[int]:
    """round(cos(pi*j/128) * 2^cos_bit), j = 0..63 (EbTransforms.c:1241)."""
    return [int(math.floor(math.cos(math.pi * j / 128.0) * (1 << cos_bit) + 0.5)) for j in range(64)]


# sinpi[j] = round(sqrt(2)*sin(j*pi/9)*2/3 * 2^bit), adjusted so that
# sinpi[1] + sinpi[2] == sinpi[4] (EbTransforms.c:1302-1303).  Stored as data:
# the adjustment is not a pure rounding rule.
SINPI = {
    10: [0, 330, 621, 836, 951],
    11: [0, 660, 1241, 1672, 1901],
    12: [0, 1321, 2482, 3344, 3803],
    13: [0, 2642, 4964, 6689, 7606],
    14: [0, 5283, 9929, 13377, 15212],
    15: [0, 10566, 19858, 26755, 30424],
    16: [0, 21133, 39716, 53510, 60849],
}

NEW_SQRT2_BITS = 12
NEW_SQRT2 = 5793      # 2^12 * sqrt(2)
NEW_INV_SQRT2 = 2896  # 2^12 / sqrt(2)


def bitrev(x: int, nbits: int) -> int:
    r = 0
    for _ in range(nbits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


# ---------------------------------------------------------------------------
# SSA network
# ---------------------------------------------------------------------------


@dataclass
class Net:
    n_in: int
    ops: List[tuple] = field(default_factory=list)   # op tuples, value id = index
    outs: List[int] = field(default_factory=list)
    name: str = ""

    def _emit(self, op: tuple) -> int:
        self.ops.append(op)
        return len(self.ops) - 1

    def inp(self, i: int) -> int:
        return self._emit(("in", i))

    def add(self, a: int, b: int) -> int:
        return self._emit(("add", a, b))

    def sub(self, a: int, b: int) -> int:
        return self._emit(("sub", a, b))

    def neg(self, a: int) -> int:
        return self._emit(("neg", a))

    def hb(self, wa, a: int, wb, b: int) -> int:
        """wa, wb: (sign, j) cos weights."""
        return self._emit(("hb", wa, a, wb, b))

    # --- ops used by ADST4 / identity / inverse clamps -------------------
    def mulc(self, w, a: int) -> int:
        """exact (non-rounded) int32 product with a symbolic constant."""
        return self._emit(("mulc", w, a))

    def rshift_round(self, a: int) -> int:
        """(a + 2^(bit-1)) >> bit with bit = the network's cos_bit."""
        return self._emit(("rsr", a))

    def scale_sqrt2(self, a: int, mult: int) -> int:
        """round_shift(a * mult * NewSqrt2, 12) (identity4/16/64)."""
        return self._emit(("sqrt2", a, mult))

    def shl(self, a: int, k: int) -> int:
        return self._emit(("shl", a, k))

    def clamp(self, a: int, which: str) -> int:
        """inverse-transform stage clamp; ``which`` names the runtime range."""
        return self._emit(("clamp", a, which))

    def count(self):
        c = {}
        for op in self.ops:
            c[op[0]] = c.get(op[0], 0) + 1
        return c


C = lambda j: (1, j)      # +cospi[j]
NC = lambda j: (-1, j)    # -cospi[j]


# ---------------------------------------------------------------------------
# forward DCT (EbTransforms.c:1314 fdct4 .. :1973 fdct64)
# ---------------------------------------------------------------------------


def _fdct_lines(net: Net, x: Sequence[int]) -> List[int]:
    """DCT-II butterfly network; returns outputs in natural frequency order."""
    n = len(x)
    if n == 2:
        return [net.hb(C(32), x[0], C(32), x[1]), net.hb(NC(32), x[1], C(32), x[0])]
    h = n // 2
    nb = n.bit_length() - 1
    s = [net.add(x[i], x[n - 1 - i]) for i in range(h)]
    # odd-half line j (absolute line h+j) carries x[h-1-j] - x[h+j]
    d = [net.sub(x[h - 1 - j], x[h + j]) for j in range(h)]
    even = _fdct_lines(net, s)
    odd = _fdct_odd(net, d, n)
    y = [None] * n
    for k in range(h):
        y[2 * k] = even[k]
    for j in range(h):
        y[bitrev(h + j, nb)] = odd[j]
    return y


def _fdct_odd(net: Net, l: List[int], n: int) -> List[int]:
    """Odd half of the N-point forward DCT on h = N/2 lines.

    levels m = 1..log2(h)-1: a mirrored-pair rotation stage followed by an
    add/sub stage on groups of h>>m lines (orientation alternating per group);
    then the final rotation with angle bitrev(line)*64/N.
    """
    h = len(l)
    nb = n.bit_length() - 1
    lh = h.bit_length() - 1
    l = list(l)
    for m in range(1, lh):
        u = h >> (m + 1)                 # unit size
        nprime_bits = m                  # N' = 2^m
        new = list(l)
        for t in range(1 << m):          # units of the first half
            kind = t & 3
            if kind not in (1, 2):
                continue
            g = t >> 2
            X = bitrev((1 << (m - 1)) + g, nprime_bits) * (64 >> m)
            for j in range(t * u, (t + 1) * u):
                p = h - 1 - j
                if kind == 1:
                    new[j] = net.hb(NC(X), l[j], C(64 - X), l[p])
                    new[p] = net.hb(C(X), l[p], C(64 - X), l[j])
                else:
                    new[j] = net.hb(NC(64 - X), l[j], NC(X), l[p])
                    new[p] = net.hb(C(64 - X), l[p], NC(X), l[j])
        l = new
        G = h >> m
        new = list(l)
        for q in range(h // G):
            b = q * G
            for i in range(G // 2):
                lo, hi = b + i, b + G - 1 - i
                if q % 2 == 0:
                    new[lo] = net.add(l[lo], l[hi])
                    new[hi] = net.sub(l[lo], l[hi])
                else:
                    new[lo] = net.sub(l[hi], l[lo])
                    new[hi] = net.add(l[hi], l[lo])
        l = new
    new = list(l)
    for j in range(h // 2):
        p = h - 1 - j
        th = bitrev(h + j, nb) * (64 // n)
        new[j] = net.hb(C(64 - th), l[j], C(th), l[p])
        new[p] = net.hb(C(64 - th), l[p], NC(th), l[j])
    return new


def build_fdct(n: int) -> Net:
    net = Net(n, name=f"fdct{n}")
    x = [net.inp(i) for i in range(n)]
    net.outs = _fdct_lines(net, x)
    return net


# ---------------------------------------------------------------------------
# forward ADST (EbTransforms.c:2764 fadst4, :2856 fadst8, :2970 fadst16)
# ---------------------------------------------------------------------------


def _adst_in_perm(n: int) -> List[int]:
    seq = [0, 1]
    while len(seq) < n:
        m = 2 * len(seq)
        seq = [v for a in seq for v in (a, m - 1 - a)]
    return seq


def build_fadst(n: int) -> Net:
    net = Net(n, name=f"fadst{n}")
    x = [net.inp(i) for i in range(n)]
    if n == 4:
        S = lambda j: ("sin", 1, j)
        s0 = net.mulc(S(1), x[0]); s1 = net.mulc(S(4), x[0])
        s2 = net.mulc(S(2), x[1]); s3 = net.mulc(S(1), x[1])
        s4 = net.mulc(S(3), x[2])
        s5 = net.mulc(S(4), x[3]); s6 = net.mulc(S(2), x[3])
        s7 = net.sub(net.add(x[0], x[1]), x[3])
        a0 = net.add(net.add(s0, s2), s5)          # x0
        a1 = net.mulc(S(3), s7)                    # x1
        a2 = net.add(net.sub(s1, s3), s6)          # x2
        a3 = s4                                    # x3
        o0 = net.add(a0, a3)
        o1 = a1
        o2 = net.sub(a2, a3)
        o3 = net.add(net.sub(a2, a0), a3)
        net.outs = [net.rshift_round(o) for o in (o0, o1, o2, o3)]
        return net
    # signed input permutation: sign follows popcount parity of the position
    perm = _adst_in_perm(n)
    l = []
    for pos, src in enumerate(perm):
        v = x[src]
        if bin(pos).count("1") & 1:
            v = net.neg(v)
        l.append(v)
    G = 4
    while G <= n:
        # rotation stage on the upper half of every group of G lines
        npairs = G // 4
        new = list(l)
        for b in range(0, n, G):
            for i in range(npairs):
                a, bb = b + G // 2 + 2 * i, b + G // 2 + 2 * i + 1
                nP = max(1, npairs // 2)
                if i < nP:
                    th = (4 * i + 1) * (128 // G)
                    new[a] = net.hb(C(th), l[a], C(64 - th), l[bb])
                    new[bb] = net.hb(C(64 - th), l[a], NC(th), l[bb])
                else:
                    th = (4 * (i - nP) + 1) * (128 // G)
                    new[a] = net.hb(NC(64 - th), l[a], C(th), l[bb])
                    new[bb] = net.hb(C(th), l[a], C(64 - th), l[bb])
        l = new
        # add/sub stage with stride G/2 inside each group
        new = list(l)
        for b in range(0, n, G):
            for i in range(G // 2):
                new[b + i] = net.add(l[b + i], l[b + i + G // 2])
                new[b + i + G // 2] = net.sub(l[b + i], l[b + i + G // 2])
        l = new
        G *= 2
    new = list(l)
    for i in range(n // 2):
        th = (4 * i + 1) * (32 // n)
        a, b = 2 * i, 2 * i + 1
        new[a] = net.hb(C(th), l[a], C(64 - th), l[b])
        new[b] = net.hb(C(64 - th), l[a], NC(th), l[b])
    l = new
    net.outs = [l[k + 1] if k % 2 == 0 else l[n - 1 - k] for k in range(n)]
    return net


# ---------------------------------------------------------------------------
# forward identity (EbTransforms.c:3620-3660)
# ---------------------------------------------------------------------------


def build_fidentity(n: int) -> Net:
    net = Net(n, name=f"fidentity{n}")
    x = [net.inp(i) for i in range(n)]
    if n == 4:
        net.outs = [net.scale_sqrt2(v, 1) for v in x]
    elif n == 8:
        net.outs = [net.shl(v, 1) for v in x]
    elif n == 16:
        net.outs = [net.scale_sqrt2(v, 2) for v in x]
    elif n == 32:
        net.outs = [net.shl(v, 2) for v in x]
    elif n == 64:
        net.outs = [net.scale_sqrt2(v, 4) for v in x]
    else:
        raise ValueError(n)
    return net


FWD_BUILDERS = {"dct": build_fdct, "adst": build_fadst, "idtx": build_fidentity}


def build_fwd(kind: str, n: int) -> Net:
    return FWD_BUILDERS[kind](n)


# ---------------------------------------------------------------------------
# inverse transforms (EbTransforms.c:5465-5746 idct4..32, :6938 idct64,
# :6097-6503 iadst, :7717-7748 iidentity).  Every forward stage matrix is
# symmetric, so the inverse network is the forward one run backwards; the
# reference clamps every add/sub result to the pass's stage range
# (clamp_value, EbTransforms.c:5458) - emitted here as 'clamp' ops.
# ---------------------------------------------------------------------------


def _idct_lines(net: Net, y: Sequence[int]) -> List[int]:
    n = len(y)
    if n == 2:
        return [net.hb(C(32), y[0], C(32), y[1]), net.hb(C(32), y[0], NC(32), y[1])]
    h = n // 2
    nb = n.bit_length() - 1
    s = _idct_lines(net, [y[2 * k] for k in range(h)])
    d = _idct_odd(net, [y[bitrev(h + j, nb)] for j in range(h)], n)
    out = [None] * n
    for i in range(h):
        out[i] = net.clamp(net.add(s[i], d[h - 1 - i]), "stage")
        out[n - 1 - i] = net.clamp(net.sub(s[i], d[h - 1 - i]), "stage")
    return out


def _idct_odd(net: Net, l: List[int], n: int) -> List[int]:
    h = len(l)
    nb = n.bit_length() - 1
    lh = h.bit_length() - 1
    new = list(l)
    for j in range(h // 2):
        p = h - 1 - j
        th = bitrev(h + j, nb) * (64 // n)
        new[j] = net.hb(C(64 - th), l[j], NC(th), l[p])
        new[p] = net.hb(C(th), l[j], C(64 - th), l[p])
    l = new
    for m in range(lh - 1, 0, -1):
        G = h >> m
        new = list(l)
        for q in range(h // G):
            b = q * G
            for i in range(G // 2):
                lo, hi = b + i, b + G - 1 - i
                if q % 2 == 0:
                    new[lo] = net.clamp(net.add(l[lo], l[hi]), "stage")
                    new[hi] = net.clamp(net.sub(l[lo], l[hi]), "stage")
                else:
                    new[lo] = net.clamp(net.sub(l[hi], l[lo]), "stage")
                    new[hi] = net.clamp(net.add(l[lo], l[hi]), "stage")
        l = new
        u = h >> (m + 1)
        new = list(l)
        for t in range(1 << m):
            kind = t & 3
            if kind not in (1, 2):
                continue
            g = t >> 2
            X = bitrev((1 << (m - 1)) + g, m) * (64 >> m)
            for j in range(t * u, (t + 1) * u):
                p = h - 1 - j
                if kind == 1:
                    new[j] = net.hb(NC(X), l[j], C(64 - X), l[p])
                    new[p] = net.hb(C(64 - X), l[j], C(X), l[p])
                else:
                    new[j] = net.hb(NC(64 - X), l[j], NC(X), l[p])
                    new[p] = net.hb(NC(X), l[j], C(64 - X), l[p])
        l = new
    return l


def build_idct(n: int) -> Net:
    net = Net(n, name=f"idct{n}")
    y = [net.inp(i) for i in range(n)]
    net.outs = _idct_lines(net, y)
    return net


def build_iadst(n: int) -> Net:
    net = Net(n, name=f"iadst{n}")
    x = [net.inp(i) for i in range(n)]
    if n == 4:
        S = lambda j: ("sin", 1, j)
        s0 = net.mulc(S(1), x[0]); s1 = net.mulc(S(2), x[0])
        s2 = net.mulc(S(3), x[1])
        s3 = net.mulc(S(4), x[2]); s4 = net.mulc(S(1), x[2])
        s5 = net.mulc(S(2), x[3]); s6 = net.mulc(S(4), x[3])
        s7 = net.add(net.sub(x[0], x[2]), x[3])
        a0 = net.add(net.add(s0, s3), s5)
        a1 = net.sub(net.sub(s1, s4), s6)
        a3 = s2
        a2 = net.mulc(S(3), s7)
        o0 = net.add(a0, a3)
        o1 = net.add(a1, a3)
        o2 = a2
        o3 = net.sub(net.add(a0, a1), a3)
        net.outs = [net.rshift_round(o) for o in (o0, o1, o2, o3)]
        return net
    # input permutation = inverse of the forward output permutation
    l = [None] * n
    for k in range(n):
        l[k + 1 if k % 2 == 0 else n - 1 - k] = x[k]
    new = list(l)
    for i in range(n // 2):
        th = (4 * i + 1) * (32 // n)
        a, b = 2 * i, 2 * i + 1
        new[a] = net.hb(C(th), l[a], C(64 - th), l[b])
        new[b] = net.hb(C(64 - th), l[a], NC(th), l[b])
    l = new
    G = n
    while G >= 4:
        new = list(l)
        for b in range(0, n, G):
            for i in range(G // 2):
                new[b + i] = net.clamp(net.add(l[b + i], l[b + i + G // 2]), "stage")
                new[b + i + G // 2] = net.clamp(net.sub(l[b + i], l[b + i + G // 2]), "stage")
        l = new
        npairs = G // 4
        new = list(l)
        for b in range(0, n, G):
            for i in range(npairs):
                a, bb = b + G // 2 + 2 * i, b + G // 2 + 2 * i + 1
                nP = max(1, npairs // 2)
                if i < nP:
                    th = (4 * i + 1) * (128 // G)
                    new[a] = net.hb(C(th), l[a], C(64 - th), l[bb])
                    new[bb] = net.hb(C(64 - th), l[a], NC(th), l[bb])
                else:
                    th = (4 * (i - nP) + 1) * (128 // G)
                    new[a] = net.hb(NC(64 - th), l[a], C(th), l[bb])
                    new[bb] = net.hb(C(th), l[a], C(64 - th), l[bb])
        l = new
        G //= 2
    perm = _adst_in_perm(n)
    out = [None] * n
    for pos, dst in enumerate(perm):
        v = l[pos]
        if bin(pos).count("1") & 1:
            v = net.neg(v)
        out[dst] = v
    net.outs = out
    return net


def build_iidentity(n: int) -> Net:
    net = build_fidentity(n)      # same scalings (EbTransforms.c:7717-7748)
    net.name = f"iidentity{n}"
    return net


INV_BUILDERS = {"dct": build_idct, "adst": build_iadst, "idtx": build_iidentity}


def build_inv(kind: str, n: int) -> Net:
    return INV_BUILDERS[kind](n)



def clamp_free_bound(net: Net, bit: int = 12):
    """(gain, slack) such that EVERY value the network clamps satisfies |v| <= gain * sum_i |x_i| + slack, by the triangle
    inequality carried through the ops (|a +- b| <= |a| + |b|; a half_btf adds at most 1 for its rounding).  While
    gain * L1(x) + slack <= the clamp bound, every clamp of the network returns its argument, so the network without its clamps
    computes the same values (by induction over the ops in order).  Used by the kernels to take a clamp-free copy of an
    inverse transform for the (wave-uniform) common case; tests/test_txfm_net.py checks the claim on random and extreme inputs."""
    import math

    def wabs(w):
        if w[0] == "sin":
            return abs(SINPI[bit][w[2]]) / float(1 << bit)
        return cospi_table(bit)[w[1]] / float(1 << bit)
    n = net.n_in
    G: List[List[float]] = []
    E: List[float] = []
    gmax, emax = 0.0, 0.0
    for op in net.ops:
        k = op[0]
        if k == "in":
            g = [0.0] * n; g[op[1]] = 1.0; e = 0.0
        elif k in ("add", "sub"):
            g = [a + b for a, b in zip(G[op[1]], G[op[2]])]; e = E[op[1]] + E[op[2]]
        elif k == "neg":
            g = list(G[op[1]]); e = E[op[1]]
        elif k == "hb":
            wa, wb = wabs(op[1]), wabs(op[3])
            g = [wa * a + wb * b for a, b in zip(G[op[2]], G[op[4]])]; e = wa * E[op[2]] + wb * E[op[4]] + 1.0
        elif k == "hb1":
            wa = wabs(op[1]); g = [wa * a for a in G[op[2]]]; e = wa * E[op[2]] + 1.0
        elif k == "mulc":
            wa = wabs(op[1]) * (1 << bit); g = [wa * a for a in G[op[2]]]; e = wa * E[op[2]]
        elif k == "rsr":
            g = [a / float(1 << bit) for a in G[op[1]]]; e = E[op[1]] / float(1 << bit) + 1.0
        elif k == "sqrt2":
            f = op[2] * 5793 / 4096.0; g = [f * a for a in G[op[1]]]; e = f * E[op[1]] + 1.0
        elif k == "shl":
            f = float(1 << op[2]); g = [f * a for a in G[op[1]]]; e = f * E[op[1]]
        elif k == "clamp":
            g = G[op[1]]; e = E[op[1]]
            gmax = max(gmax, max(g)); emax = max(emax, e)
        elif k == "zero":
            g = [0.0] * n; e = 0.0
        else:
            raise ValueError(k)
        G.append(g); E.append(e)
    return gmax, emax


def specialize_zero_inputs(net: Net, nz: int, name: str) -> Net:
    """Network for inputs whose elements >= nz are known to be zero (AV1 64-point
    inverse: only the first 32 coefficients of a row / column can be non-zero,
    EbTransforms.c:8226-8240).  Zeros are propagated and the ops they kill are
    dropped; a half_btf with one zero operand becomes the one-term form
    ('hb1', w, a) = (w*a + 2^(bit-1)) >> bit — bit-identical to the full network
    on such inputs (tests/test_txfm_net.py)."""
    Z = -1
    out = Net(n_in=net.n_in, name=name)
    m: List[int] = []
    clamp_of = {}                      # (new id, which) -> new id of its clamp
    for op in net.ops:
        k = op[0]
        if k == "in":
            r = Z if op[1] >= nz else out._emit(op)
        elif k == "add":
            a, b = m[op[1]], m[op[2]]
            r = Z if (a == Z and b == Z) else (a if b == Z else (b if a == Z else out._emit(("add", a, b))))
        elif k == "sub":
            a, b = m[op[1]], m[op[2]]
            r = Z if (a == Z and b == Z) else (a if b == Z else (out._emit(("neg", b)) if a == Z else out._emit(("sub", a, b))))
        elif k == "neg":
            r = Z if m[op[1]] == Z else out._emit(("neg", m[op[1]]))
        elif k == "hb":
            a, b = m[op[2]], m[op[4]]
            if a == Z and b == Z:
                r = Z
            elif b == Z:
                r = out._emit(("hb1", op[1], a))
            elif a == Z:
                r = out._emit(("hb1", op[3], b))
            else:
                r = out._emit(("hb", op[1], a, op[3], b))
        elif k in ("mulc",):
            r = Z if m[op[2]] == Z else out._emit((k, op[1], m[op[2]]))
        elif k in ("rsr",):
            r = Z if m[op[1]] == Z else out._emit((k, m[op[1]]))
        elif k in ("sqrt2", "shl"):
            r = Z if m[op[1]] == Z else out._emit((k, m[op[1]], op[2]))
        elif k == "clamp":
            a = m[op[1]]
            if a == Z:
                r = Z
            else:
                src = out.ops[a]
                if src[0] == "clamp" and src[2] == op[2]:
                    r = a                                   # clamp(clamp(x)) with the same range
                elif (a, op[2]) in clamp_of:
                    r = clamp_of[(a, op[2])]
                else:
                    r = out._emit(("clamp", a, op[2]))
                    clamp_of[(a, op[2])] = r
        else:
            raise ValueError(k)
        m.append(r)
    zero_id = None
    for o in net.outs:
        if m[o] == Z:
            if zero_id is None:
                zero_id = out._emit(("zero",))
            out.outs.append(zero_id)
        else:
            out.outs.append(m[o])
    return out



# ---------------------------------------------------------------------------
# numpy evaluator (reference semantics: int32 products, 64-bit sums)
# ---------------------------------------------------------------------------


def _w(w, cos_bit):
    if w[0] == "sin":
        return w[1] * SINPI[cos_bit][w[2]]
    return w[0] * cospi_table(cos_bit)[w[1]]


def _i32(a):
    return ((a + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)


def evaluate(net: Net, x: np.ndarray, cos_bit: int, clamp_ranges=None) -> np.ndarray:
    """x: int array [..., n_in] -> [..., n_out]; int64 math with the exact
    int32 wrap points of the reference."""
    x = np.asarray(x, dtype=np.int64)
    vals: List[np.ndarray] = []
    for op in net.ops:
        k = op[0]
        if k == "in":
            v = x[..., op[1]]
        elif k == "add":
            v = _i32(vals[op[1]] + vals[op[2]])
        elif k == "sub":
            v = _i32(vals[op[1]] - vals[op[2]])
        elif k == "neg":
            v = _i32(-vals[op[1]])
        elif k == "hb":
            p0 = _i32(_w(op[1], cos_bit) * vals[op[2]])
            p1 = _i32(_w(op[3], cos_bit) * vals[op[4]])
            v = _i32((p0 + p1 + (1 << (cos_bit - 1))) >> cos_bit)
        elif k == "hb1":
            v = _i32((_i32(_w(op[1], cos_bit) * vals[op[2]]) + (1 << (cos_bit - 1))) >> cos_bit)
        elif k == "zero":
            v = np.zeros(x.shape[:-1], dtype=np.int64)
        elif k == "mulc":
            v = _i32(_w(op[1], cos_bit) * vals[op[2]])
        elif k == "rsr":
            v = _i32((vals[op[1]] + (1 << (cos_bit - 1))) >> cos_bit)
        elif k == "sqrt2":
            v = _i32((vals[op[1]] * op[2] * NEW_SQRT2 + (1 << (NEW_SQRT2_BITS - 1))) >> NEW_SQRT2_BITS)
        elif k == "shl":
            v = _i32(vals[op[1]] * (1 << op[2]))
        elif k == "clamp":
            bits = clamp_ranges[op[2]]
            v = np.clip(vals[op[1]], -(1 << (bits - 1)), (1 << (bits - 1)) - 1)
        else:
            raise ValueError(k)
        vals.append(v)
    return np.stack([vals[o] for o in net.outs], axis=-1)


if __name__ == "__main__":
    for kind, sizes in (("dct", (4, 8, 16, 32, 64)), ("adst", (4, 8, 16)), ("idtx", (4, 8, 16, 32, 64))):
        for n in sizes:
            net = build_fwd(kind, n)
            print(net.name, len(net.ops), net.count())
