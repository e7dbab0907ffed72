/*
 * oracle/txfm.c — scalar restatement of the AV1 integer transforms.
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 *
 * The reference spells every butterfly stage out (EbTransforms.c:1314-3660
 * forward, :5465-7748 inverse).  Here the same dataflow is expressed by its
 * closed-form rules and evaluated with loops, so this file is an independent
 * statement of the algorithm rather than a transcription:
 *
 *   DCT-N  = input butterfly, then DCT-N/2 on the sums and an "odd half"
 *            network on the differences; outputs in bit-reversed line order.
 *   odd half (h = N/2 lines): for level m = 1..log2(h)-1 a rotation of
 *            mirrored line pairs (j, h-1-j) followed by add/sub butterflies on
 *            groups of h>>m lines with alternating orientation; finally a
 *            rotation by angle bitrev(line) * 64/N.
 *   ADST-N = signed input permutation, log2(N)-1 x {rotate upper half of each
 *            group, add/sub across half groups}, final rotations, output
 *            permutation.
 *   inverse = the forward stages run backwards (every stage matrix is
 *            symmetric) with a clamp on every add/sub result.
 *
 * Rounding is the reference's half_btf (EbTransforms.c:1292-1300): the two
 * products are formed in int32, summed in int64, rounded, shifted.
 */
#include "svt_oracle.h"
#include <string.h>
#include <stdlib.h>
#include <assert.h>

/* ---- constants ----------------------------------------------------------- */
/* cos(pi*j/128) in Q(cos_bit): EbTransforms.c:1242-1284 (AV1 spec table) */
#include "txfm_consts.h"

#define NEW_SQRT2_BITS 12
#define NEW_SQRT2 5793     /* EbTransforms.c:1237 */
#define NEW_INV_SQRT2 2896 /* EbTransforms.c:1239 */

static const int k_tx_wide[ORC_TX_SIZES_ALL] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64};
static const int k_tx_high[ORC_TX_SIZES_ALL] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};

int svt_oracle_tx_wide(int s) { return k_tx_wide[s]; }
int svt_oracle_tx_high(int s) { return k_tx_high[s]; }

/* vertical (column) / horizontal (row) 1-D kind per 2-D type: EbTransforms.h:87-98 */
static const uint8_t k_vkind[ORC_TX_TYPES] = {0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3};
static const uint8_t k_hkind[ORC_TX_TYPES] = {0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2};

int svt_oracle_txfm_allowed(int tx_size, int tx_type) {
    int w = k_tx_wide[tx_size], h = k_tx_high[tx_size];
    int m = w > h ? w : h;
    if (m == 64) return tx_type == ORC_DCT_DCT;
    if (m == 32) return tx_type == ORC_DCT_DCT || tx_type == ORC_IDTX;
    return 1;
}

static inline int32_t rshift_round64(int64_t v, int bit) { return (int32_t)((v + ((int64_t)1 << (bit - 1))) >> bit); }

/* half_btf, EbTransforms.c:1292: products wrap in int32, sum is 64-bit */
static inline int32_t hb(int32_t w0, int32_t a, int32_t w1, int32_t b, int bit) {
    int32_t p0 = (int32_t)((uint32_t)w0 * (uint32_t)a);
    int32_t p1 = (int32_t)((uint32_t)w1 * (uint32_t)b);
    return rshift_round64((int64_t)p0 + (int64_t)p1, bit);
}
static inline int32_t add32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t sub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

static inline int32_t clampb(int32_t v, int bits) { /* clamp_value, :5458 */
    if (bits <= 0) return v;
    int64_t hi = ((int64_t)1 << (bits - 1)) - 1, lo = -((int64_t)1 << (bits - 1));
    return (int32_t)(v < lo ? lo : (v > hi ? hi : v));
}

static int bitrev(int x, int nbits) {
    int r = 0;
    for (int i = 0; i < nbits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
static int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

/* ---- DCT ------------------------------------------------------------------ */
/* mirrored-pair rotation stage at level m of an odd half with h lines.
 * Symmetric, so the same routine serves forward and inverse. */
static void odd_rotate_level(int32_t *l, int h, int m, const int32_t *cs, int bit) {
    int u = h >> (m + 1);
    for (int t = 0; t < (1 << m); t++) {
        int kind = t & 3;
        if (kind != 1 && kind != 2) continue;
        int X = bitrev((1 << (m - 1)) + (t >> 2), m) * (64 >> m);
        int32_t cx = cs[X], cy = cs[64 - X];
        for (int j = t * u; j < (t + 1) * u; j++) {
            int p = h - 1 - j;
            int32_t a = l[j], b = l[p];
            if (kind == 1) { l[j] = hb(-cx, a, cy, b, bit); l[p] = hb(cx, b, cy, a, bit); }
            else           { l[j] = hb(-cy, a, -cx, b, bit); l[p] = hb(cy, b, -cx, a, bit); }
        }
    }
}
/* add/sub stage on groups of G lines, orientation alternating per group */
static void odd_addsub_level(int32_t *l, int h, int G, int clamp_bits) {
    for (int q = 0; q < h / G; q++) {
        int base = q * G;
        for (int i = 0; i < G / 2; i++) {
            int lo = base + i, hi = base + G - 1 - i;
            int32_t a = l[lo], b = l[hi];
            if ((q & 1) == 0) { l[lo] = clampb(add32(a, b), clamp_bits); l[hi] = clampb(sub32(a, b), clamp_bits); }
            else              { l[lo] = clampb(sub32(b, a), clamp_bits); l[hi] = clampb(add32(b, a), clamp_bits); }
        }
    }
}

static void fdct_rec(const int32_t *x, int32_t *y, int n, const int32_t *cs, int bit) {
    if (n == 2) {
        y[0] = hb(cs[32], x[0], cs[32], x[1], bit);
        y[1] = hb(-cs[32], x[1], cs[32], x[0], bit);
        return;
    }
    int h = n / 2, nb = ilog2(n), lh = ilog2(h);
    int32_t s[32] = {0}, d[32] = {0}, e[32] = {0};
    for (int i = 0; i < h; i++) s[i] = add32(x[i], x[n - 1 - i]);
    for (int j = 0; j < h; j++) d[j] = sub32(x[h - 1 - j], x[h + j]);
    fdct_rec(s, e, h, cs, bit);
    for (int m = 1; m < lh; m++) {
        odd_rotate_level(d, h, m, cs, bit);
        odd_addsub_level(d, h, h >> m, 0);
    }
    for (int k = 0; k < h; k++) y[2 * k] = e[k];
    for (int j = 0; j < h / 2; j++) {
        int p = h - 1 - j;
        int th = bitrev(h + j, nb) * (64 / n);
        int32_t a = d[j], b = d[p];
        y[bitrev(h + j, nb)] = hb(cs[64 - th], a, cs[th], b, bit);
        y[bitrev(h + p, nb)] = hb(cs[64 - th], b, -cs[th], a, bit);
    }
}

static void idct_rec(const int32_t *y, int32_t *x, int n, const int32_t *cs, int bit, int cb) {
    if (n == 2) {
        x[0] = hb(cs[32], y[0], cs[32], y[1], bit);
        x[1] = hb(cs[32], y[0], -cs[32], y[1], bit);
        return;
    }
    int h = n / 2, nb = ilog2(n), lh = ilog2(h);
    int32_t e[32] = {0}, s[32] = {0}, d[32] = {0};
    for (int k = 0; k < h; k++) e[k] = y[2 * k];
    idct_rec(e, s, h, cs, bit, cb);
    for (int j = 0; j < h / 2; j++) {
        int p = h - 1 - j;
        int th = bitrev(h + j, nb) * (64 / n);
        int32_t a = y[bitrev(h + j, nb)], b = y[bitrev(h + p, nb)];
        d[j] = hb(cs[64 - th], a, -cs[th], b, bit);
        d[p] = hb(cs[th], a, cs[64 - th], b, bit);
    }
    for (int m = lh - 1; m >= 1; m--) {
        odd_addsub_level(d, h, h >> m, cb);
        odd_rotate_level(d, h, m, cs, bit);
    }
    for (int i = 0; i < h; i++) {
        x[i] = clampb(add32(s[i], d[h - 1 - i]), cb);
        x[n - 1 - i] = clampb(sub32(s[i], d[h - 1 - i]), cb);
    }
}

/* ---- ADST ----------------------------------------------------------------- */
static void adst_in_perm(int n, int *perm) { /* interleave (a, m-1-a) recursively */
    int len = 2;
    perm[0] = 0; perm[1] = 1;
    while (len < n) {
        int m = 2 * len, tmp[16];
        for (int i = 0; i < len; i++) { tmp[2 * i] = perm[i]; tmp[2 * i + 1] = m - 1 - perm[i]; }
        memcpy(perm, tmp, sizeof(int) * m);
        len = m;
    }
}
static int parity(int x) { int p = 0; while (x) { p ^= x & 1; x >>= 1; } return p; }

/* rotation of the upper half of every group of G lines (symmetric matrices) */
static void adst_rotate_groups(int32_t *l, int n, int G, const int32_t *cs, int bit) {
    int npairs = G / 4, nP = npairs / 2 > 1 ? npairs / 2 : 1;
    for (int b = 0; b < n; b += G)
        for (int i = 0; i < npairs; i++) {
            int ia = b + G / 2 + 2 * i, ib = ia + 1;
            int32_t a = l[ia], c = l[ib];
            if (i < nP) {
                int th = (4 * i + 1) * (128 / G);
                l[ia] = hb(cs[th], a, cs[64 - th], c, bit);
                l[ib] = hb(cs[64 - th], a, -cs[th], c, bit);
            } else {
                int th = (4 * (i - nP) + 1) * (128 / G);
                l[ia] = hb(-cs[64 - th], a, cs[th], c, bit);
                l[ib] = hb(cs[th], a, cs[64 - th], c, bit);
            }
        }
}
static void adst_addsub_groups(int32_t *l, int n, int G, int cb) {
    for (int b = 0; b < n; b += G)
        for (int i = 0; i < G / 2; i++) {
            int32_t a = l[b + i], c = l[b + i + G / 2];
            l[b + i] = clampb(add32(a, c), cb);
            l[b + i + G / 2] = clampb(sub32(a, c), cb);
        }
}
static void adst_final_rot(int32_t *l, int n, const int32_t *cs, int bit) {
    for (int i = 0; i < n / 2; i++) {
        int th = (4 * i + 1) * (32 / n);
        int32_t a = l[2 * i], c = l[2 * i + 1];
        l[2 * i] = hb(cs[th], a, cs[64 - th], c, bit);
        l[2 * i + 1] = hb(cs[64 - th], a, -cs[th], c, bit);
    }
}

/* av1_fadst4_new (EbTransforms.c:2764): sinpi products are exact int32 */
static void fadst4(const int32_t *in, int32_t *out, int bit) {
    const int32_t *sp = k_sinpi[bit - 10];
    int32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
#define M32(a, b) ((int32_t)((uint32_t)(a) * (uint32_t)(b)))
    int32_t t0 = add32(add32(M32(sp[1], x0), M32(sp[2], x1)), M32(sp[4], x3));
    int32_t t1 = M32(sp[3], sub32(add32(x0, x1), x3));
    int32_t t2 = add32(sub32(M32(sp[4], x0), M32(sp[1], x1)), M32(sp[2], x3));
    int32_t t3 = M32(sp[3], x2);
    out[0] = rshift_round64(add32(t0, t3), bit);
    out[1] = rshift_round64(t1, bit);
    out[2] = rshift_round64(sub32(t2, t3), bit);
    out[3] = rshift_round64(add32(sub32(t2, t0), t3), bit);
}
/* av1_iadst4_new (EbTransforms.c:6097) */
static void iadst4(const int32_t *in, int32_t *out, int bit) {
    const int32_t *sp = k_sinpi[bit - 10];
    int32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
    int32_t a0 = add32(add32(M32(sp[1], x0), M32(sp[4], x2)), M32(sp[2], x3));
    int32_t a1 = sub32(sub32(M32(sp[2], x0), M32(sp[1], x2)), M32(sp[4], x3));
    int32_t a3 = M32(sp[3], x1);
    int32_t a2 = M32(sp[3], add32(sub32(x0, x2), x3));
    out[0] = rshift_round64(add32(a0, a3), bit);
    out[1] = rshift_round64(add32(a1, a3), bit);
    out[2] = rshift_round64(a2, bit);
    out[3] = rshift_round64(sub32(add32(a0, a1), a3), bit);
#undef M32
}

static void fadst(const int32_t *in, int32_t *out, int n, const int32_t *cs, int bit) {
    int perm[16];
    int32_t l[16];
    adst_in_perm(n, perm);
    for (int p = 0; p < n; p++) l[p] = parity(p) ? sub32(0, in[perm[p]]) : in[perm[p]];
    for (int G = 4; G <= n; G *= 2) {
        adst_rotate_groups(l, n, G, cs, bit);
        adst_addsub_groups(l, n, G, 0);
    }
    adst_final_rot(l, n, cs, bit);
    for (int k = 0; k < n; k++) out[k] = (k & 1) ? l[n - 1 - k] : l[k + 1];
}
static void iadst(const int32_t *in, int32_t *out, int n, const int32_t *cs, int bit, int cb) {
    int perm[16];
    int32_t l[16];
    for (int k = 0; k < n; k++) l[(k & 1) ? n - 1 - k : k + 1] = in[k];
    adst_final_rot(l, n, cs, bit);
    for (int G = n; G >= 4; G /= 2) {
        adst_addsub_groups(l, n, G, cb);
        adst_rotate_groups(l, n, G, cs, bit);
    }
    adst_in_perm(n, perm);
    for (int p = 0; p < n; p++) out[perm[p]] = parity(p) ? sub32(0, l[p]) : l[p];
}

/* av1_fidentity*_c / av1_iidentity*_c (EbTransforms.c:3620-3660, 7717-7748) */
static void identity(const int32_t *in, int32_t *out, int n) {
    for (int i = 0; i < n; i++) {
        int64_t v = in[i];
        switch (n) {
        case 4:  out[i] = rshift_round64(v * NEW_SQRT2, NEW_SQRT2_BITS); break;
        case 8:  out[i] = (int32_t)(v * 2); break;
        case 16: out[i] = rshift_round64(v * 2 * NEW_SQRT2, NEW_SQRT2_BITS); break;
        case 32: out[i] = (int32_t)(v * 4); break;
        default: out[i] = rshift_round64(v * 4 * NEW_SQRT2, NEW_SQRT2_BITS); break;
        }
    }
}

void svt_oracle_fwd_txfm1d(int kind, int n, const int32_t *in, int32_t *out, int cos_bit) {
    const int32_t *cs = k_cospi[cos_bit - 10];
    if (kind == ORC_1D_IDTX) identity(in, out, n);
    else if (kind == ORC_1D_DCT) fdct_rec(in, out, n, cs, cos_bit);
    else if (n == 4) fadst4(in, out, cos_bit);
    else fadst(in, out, n, cs, cos_bit);
}
void svt_oracle_inv_txfm1d(int kind, int n, const int32_t *in, int32_t *out, int cos_bit, int stage_bits) {
    const int32_t *cs = k_cospi[cos_bit - 10];
    if (kind == ORC_1D_IDTX) identity(in, out, n);
    else if (kind == ORC_1D_DCT) idct_rec(in, out, n, cs, cos_bit, stage_bits);
    else if (n == 4) iadst4(in, out, cos_bit);
    else iadst(in, out, n, cs, cos_bit, stage_bits);
}

/* ---- 2-D configuration (Av1TransformConfig, EbTransforms.c:4329-4349) ----- */
/* fwd shifts: EbTransforms.h:120-138 (col up-shift, mid round, row round) */
static const int8_t k_fwd_shift[ORC_TX_SIZES_ALL][3] = {
    {2, 0, 0}, {2, -1, 0}, {2, -2, 0}, {2, -4, 0}, {0, -2, -2}, {2, -1, 0}, {2, -1, 0},
    {2, -2, 0}, {2, -2, 0}, {2, -4, 0}, {2, -4, 0}, {0, -2, -2}, {2, -4, -2}, {2, -1, 0},
    {2, -1, 0}, {2, -2, 0}, {2, -2, 0}, {0, -2, 0}, {2, -4, 0}};
/* cos_bit by [log2(w)-2][log2(h)-2]: EbTransforms.h:141-156 */
static const int8_t k_fwd_cos_col[5][5] = {{13, 13, 13, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 13, 12, 13}, {0, 13, 13, 12, 13}, {0, 0, 13, 12, 13}};
static const int8_t k_fwd_cos_row[5][5] = {{13, 13, 12, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 12, 13, 12}, {0, 12, 13, 12, 11}, {0, 0, 12, 11, 10}};
/* inverse: cos_bit 12 everywhere (EbTransforms.h:252-267); shifts :268-286 */
static const int8_t k_inv_shift[ORC_TX_SIZES_ALL][2] = {
    {0, -4}, {-1, -4}, {-2, -4}, {-2, -4}, {-2, -4}, {0, -4}, {0, -4}, {-1, -4}, {-1, -4}, {-1, -4},
    {-1, -4}, {-1, -4}, {-1, -4}, {-1, -4}, {-1, -4}, {-2, -4}, {-2, -4}, {-2, -4}, {-2, -4}};

/* av1_round_shift_array_c (EbTransforms.c:3676-3700): bit>0 rounds down-shift,
 * bit<0 multiplies by 2^-bit */
static void shift_array(int32_t *a, int n, int bit) {
    if (bit == 0) return;
    if (bit > 0) for (int i = 0; i < n; i++) a[i] = rshift_round64(a[i], bit);
    else for (int i = 0; i < n; i++) a[i] = (int32_t)((uint32_t)a[i] * (1u << (-bit)));
}
static int rect_ratio_abs1(int w, int h) { return (w == 2 * h) || (h == 2 * w); }

void svt_oracle_fwd_txfm2d(const int16_t *in, int32_t *out, uint32_t stride, int tx_type,
                           int tx_size, int bd) {
    (void)bd; /* only feeds range asserts in the reference */
    const int w = k_tx_wide[tx_size], h = k_tx_high[tx_size];
    const int wi = ilog2(w) - 2, hi = ilog2(h) - 2;
    const int8_t *sh = k_fwd_shift[tx_size];
    const int cbc = k_fwd_cos_col[wi][hi], cbr = k_fwd_cos_row[wi][hi];
    int vk = k_vkind[tx_type], hk = k_hkind[tx_type];
    const int ud = (vk == ORC_1D_FLIPADST), lr = (hk == ORC_1D_FLIPADST);
    if (vk == ORC_1D_FLIPADST) vk = ORC_1D_ADST;
    if (hk == ORC_1D_FLIPADST) hk = ORC_1D_ADST;
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * w * h);
    int32_t tin[64], tout[64];
    for (int c = 0; c < w; c++) {
        for (int r = 0; r < h; r++) tin[r] = in[(ud ? h - 1 - r : r) * stride + c];
        shift_array(tin, h, -sh[0]);
        svt_oracle_fwd_txfm1d(vk, h, tin, tout, cbc);
        shift_array(tout, h, -sh[1]);
        for (int r = 0; r < h; r++) buf[r * w + (lr ? w - 1 - c : c)] = tout[r];
    }
    for (int r = 0; r < h; r++) {
        svt_oracle_fwd_txfm1d(hk, w, buf + r * w, tout, cbr);
        shift_array(tout, w, -sh[2]);
        if (rect_ratio_abs1(w, h))
            for (int c = 0; c < w; c++) tout[c] = rshift_round64((int64_t)tout[c] * NEW_SQRT2, NEW_SQRT2_BITS);
        memcpy(out + r * w, tout, sizeof(int32_t) * w);
    }
    free(buf);
}

uint64_t svt_oracle_fwd_txfm2d_pack64(int32_t *coeff, int tx_size) {
    const int w = k_tx_wide[tx_size], h = k_tx_high[tx_size];
    if (w != 64 && h != 64) return 0;
    const int kw = w > 32 ? 32 : w, kh = h > 32 ? 32 : h;
    uint64_t energy = 0;
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++)
            if (r >= kh || c >= kw) { int64_t v = coeff[r * w + c]; energy += (uint64_t)(v * v); }
    /* re-pack kept rows to stride kw, zero the tail (HandleTransform64x64_c + memcpy loop) */
    for (int r = 0; r < kh; r++) memmove(coeff + r * kw, coeff + r * w, sizeof(int32_t) * kw);
    memset(coeff + kh * kw, 0, sizeof(int32_t) * (w * h - kh * kw));
    return energy;
}

void svt_oracle_inv_txfm2d_add(const int32_t *in, uint16_t *dst, int32_t stride, int tx_type,
                               int tx_size, int bd) {
    const int w = k_tx_wide[tx_size], h = k_tx_high[tx_size];
    const int kw = w > 32 ? 32 : w, kh = h > 32 ? 32 : h;
    const int8_t *sh = k_inv_shift[tx_size];
    int vk = k_vkind[tx_type], hk = k_hkind[tx_type];
    const int ud = (vk == ORC_1D_FLIPADST), lr = (hk == ORC_1D_FLIPADST);
    if (vk == ORC_1D_FLIPADST) vk = ORC_1D_ADST;
    if (hk == ORC_1D_FLIPADST) hk = ORC_1D_ADST;
    /* av1_gen_inv_stage_range (EbTransforms.c:5404-5456) */
    const int row_bits = bd == 8 ? 16 : (bd == 10 ? 18 : 20);
    const int col_bits = bd == 12 ? 18 : 16;
    const int col_in_bits = bd + 6 > 16 ? bd + 6 : 16;
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * w * h);
    int32_t tin[64], tout[64];
    for (int r = 0; r < h; r++) {
        for (int c = 0; c < w; c++) {
            int32_t v = (r < kh && c < kw) ? in[r * kw + c] : 0; /* :8299-8315 zero re-expansion */
            if (rect_ratio_abs1(w, h)) v = rshift_round64((int64_t)v * NEW_INV_SQRT2, NEW_SQRT2_BITS);
            tin[c] = clampb(v, bd + 8);
        }
        svt_oracle_inv_txfm1d(hk, w, tin, buf + r * w, 12, row_bits);
        shift_array(buf + r * w, w, -sh[0]);
    }
    const int maxpix = (1 << bd) - 1;
    for (int c = 0; c < w; c++) {
        for (int r = 0; r < h; r++) tin[r] = clampb(buf[r * w + (lr ? w - 1 - c : c)], col_in_bits);
        svt_oracle_inv_txfm1d(vk, h, tin, tout, 12, col_bits);
        shift_array(tout, h, -sh[1]);
        for (int r = 0; r < h; r++) {
            /* highbd_clip_pixel_add: dst + residual in 32-bit, clipped to [0, 2^bd) */
            int32_t v = add32((int32_t)dst[r * stride + c], tout[ud ? h - 1 - r : r]);
            dst[r * stride + c] = (uint16_t)(v < 0 ? 0 : (v > maxpix ? maxpix : v));
        }
    }
    free(buf);
}

void svt_oracle_inv_txfm2d_add_u8(const int32_t *in, uint8_t *dst, int32_t stride, int tx_type,
                                  int tx_size) {
    const int w = k_tx_wide[tx_size], h = k_tx_high[tx_size];
    uint16_t tmp[64 * 64];
    for (int r = 0; r < h; r++) for (int c = 0; c < w; c++) tmp[r * 64 + c] = dst[r * stride + c];
    svt_oracle_inv_txfm2d_add(in, tmp, 64, tx_type, tx_size, 8);
    for (int r = 0; r < h; r++) for (int c = 0; c < w; c++) dst[r * stride + c] = (uint8_t)tmp[r * 64 + c];
}
