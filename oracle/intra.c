/*
 * oracle/intra.c — intra prediction (non-directional, directional, edge filter,
 * edge upsample), 8-bit and high bit depth.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows EbIntraPrediction.c: dc/v/h/smooth/paeth :1838-2260 (exact-division
 * DC for every size: the multiplier forms are commented out in the reference),
 * av1_dr_prediction_z1/z2/z3_c :370-477 (highbd twins :3394-3506),
 * av1_filter_intra_edge_c / av1_upsample_intra_edge_c :3539-3660.
 * The reference has NO unit test for any of these: parity is pinned
 * by running the reference's scalar C functions (tests/golden, test_oracle_vs_ref).
 */
#include "svt_oracle.h"
#include <string.h>

/* AV1 spec smooth-prediction weights Sm_Weights_Tx_{4..64} (reference:
 * sm_weight_arrays, ASM_AVX2/EbIntraPrediction_AVX2.h:19-38), indexed [bs + i] */
static const uint8_t k_sm_weights[128] = {
    0, 0, 255, 128, 255, 149, 85, 64, 255, 197, 146, 105, 73, 50, 37, 32,
    255, 225, 196, 170, 145, 123, 102, 84, 68, 54, 43, 33, 26, 20, 17, 16,
    255, 240, 225, 210, 196, 182, 169, 157, 145, 133, 122, 111, 101, 92, 83, 74,
    66, 59, 52, 45, 39, 34, 29, 25, 21, 17, 14, 12, 10, 9, 8, 8,
    255, 248, 240, 233, 225, 218, 210, 203, 196, 189, 182, 176, 169, 163, 156, 150,
    144, 138, 133, 127, 121, 116, 111, 106, 101, 96, 91, 86, 82, 77, 73, 69,
    65, 61, 57, 54, 50, 47, 44, 41, 38, 35, 32, 29, 27, 25, 22, 20,
    18, 16, 15, 13, 12, 10, 9, 8, 7, 6, 6, 5, 5, 4, 4, 4};
const uint8_t *svt_oracle_sm_weights(void) { return k_sm_weights; }

static int absd(int a, int b) { return a > b ? a - b : b - a; }
static int paeth1(int left, int top, int tl) { /* paeth_predictor_single :2036 */
    const int base = top + left - tl;
    const int pl = absd(base, left), pt = absd(base, top), ptl = absd(base, tl);
    return (pl <= pt && pl <= ptl) ? left : (pt <= ptl ? top : tl);
}

/* generic over the sample type via a macro-free trick: compute in int, store via callback */
#define INTRA_BODY(PIX, BD)                                                                         \
    int sum = 0, dc = 0;                                                                            \
    const uint8_t *ww = k_sm_weights + bw, *wh = k_sm_weights + bh;                                 \
    switch (mode) {                                                                                 \
    case ORC_DC_PRED: for (int i = 0; i < bw; i++) sum += above[i];                                 \
                      for (int i = 0; i < bh; i++) sum += left[i];                                  \
                      dc = (sum + ((bw + bh) >> 1)) / (bw + bh); break;                             \
    case ORC_DC_TOP_PRED: for (int i = 0; i < bw; i++) sum += above[i]; dc = (sum + (bw >> 1)) / bw; break; \
    case ORC_DC_LEFT_PRED: for (int i = 0; i < bh; i++) sum += left[i]; dc = (sum + (bh >> 1)) / bh; break; \
    case ORC_DC_128_PRED: dc = 128 << ((BD) - 8); break;                                            \
    default: break;                                                                                 \
    }                                                                                               \
    for (int r = 0; r < bh; r++)                                                                    \
        for (int c = 0; c < bw; c++) {                                                              \
            int v;                                                                                  \
            switch (mode) {                                                                         \
            case ORC_V_PRED: v = above[c]; break;                                                   \
            case ORC_H_PRED: v = left[r]; break;                                                    \
            case ORC_SMOOTH_PRED:                                                                   \
                v = (wh[r] * above[c] + (256 - wh[r]) * left[bh - 1] + ww[c] * left[r] +            \
                     (256 - ww[c]) * above[bw - 1] + 256) >> 9; break;                              \
            case ORC_SMOOTH_V_PRED: v = (wh[r] * above[c] + (256 - wh[r]) * left[bh - 1] + 128) >> 8; break; \
            case ORC_SMOOTH_H_PRED: v = (ww[c] * left[r] + (256 - ww[c]) * above[bw - 1] + 128) >> 8; break; \
            case ORC_PAETH_PRED: v = paeth1(left[r], above[c], above[-1]); break;                   \
            default: v = dc; break;                                                                 \
            }                                                                                       \
            dst[r * stride + c] = (PIX)v;                                                           \
        }

void svt_oracle_intra_pred(int mode, uint8_t *dst, ptrdiff_t stride, int bw, int bh,
                           const uint8_t *above, const uint8_t *left) {
    INTRA_BODY(uint8_t, 8)
}
void svt_oracle_intra_pred_hbd(int mode, uint16_t *dst, ptrdiff_t stride, int bw, int bh,
                               const uint16_t *above, const uint16_t *left, int bd) {
    INTRA_BODY(uint16_t, bd)
}

/* ---- directional prediction ------------------------------------------------
 * zone 1 (0 < angle < 90): along `above`; zone 3 (180..270): along `left`;
 * zone 2 (90..180): above where the projected position is >= -(1<<upsample_above),
 * else left.  2-tap: (p[b]*(32-s) + p[b+1]*s + 16) >> 5, clipped to the pixel range. */
#define DR_BODY(PIX, MAXV)                                                                          \
    if (zone == 1) {                                                                                \
        const int max_base = (bw + bh - 1) << up_above, fb = 6 - up_above, inc = 1 << up_above;      \
        int x = dx;                                                                                 \
        for (int r = 0; r < bh; r++, x += dx) {                                                     \
            int base = x >> fb;                                                                     \
            const int sh = ((x << up_above) & 0x3f) >> 1;                                           \
            for (int c = 0; c < bw; c++, base += inc) {                                             \
                int v;                                                                              \
                if (base < max_base) { v = (above[base] * (32 - sh) + above[base + 1] * sh + 16) >> 5; v = v < 0 ? 0 : (v > (MAXV) ? (MAXV) : v); } \
                else v = above[max_base];                                                           \
                dst[r * stride + c] = (PIX)v;                                                       \
            }                                                                                       \
        }                                                                                           \
    } else if (zone == 3) {                                                                         \
        const int max_base = (bw + bh - 1) << up_left, fb = 6 - up_left, inc = 1 << up_left;         \
        int y = dy;                                                                                 \
        for (int c = 0; c < bw; c++, y += dy) {                                                     \
            int base = y >> fb;                                                                     \
            const int sh = ((y << up_left) & 0x3f) >> 1;                                            \
            for (int r = 0; r < bh; r++, base += inc) {                                             \
                int v;                                                                              \
                if (base < max_base) { v = (left[base] * (32 - sh) + left[base + 1] * sh + 16) >> 5; v = v < 0 ? 0 : (v > (MAXV) ? (MAXV) : v); } \
                else v = left[max_base];                                                            \
                dst[r * stride + c] = (PIX)v;                                                       \
            }                                                                                       \
        }                                                                                           \
    } else {                                                                                        \
        const int min_base_x = -(1 << up_above), fbx = 6 - up_above, fby = 6 - up_left, incx = 1 << up_above; \
        int x = -dx;                                                                                \
        for (int r = 0; r < bh; r++, x -= dx) {                                                     \
            int base1 = x >> fbx, y = (r << 6) - dy;                                                \
            for (int c = 0; c < bw; c++, base1 += incx, y -= dy) {                                  \
                int v;                                                                              \
                if (base1 >= min_base_x) {                                                          \
                    const int s1 = ((x * (1 << up_above)) & 0x3f) >> 1;                             \
                    v = (above[base1] * (32 - s1) + above[base1 + 1] * s1 + 16) >> 5;               \
                } else {                                                                            \
                    const int base2 = y >> fby, s2 = ((y * (1 << up_left)) & 0x3f) >> 1;            \
                    v = (left[base2] * (32 - s2) + left[base2 + 1] * s2 + 16) >> 5;                 \
                }                                                                                   \
                v = v < 0 ? 0 : (v > (MAXV) ? (MAXV) : v);                                          \
                dst[r * stride + c] = (PIX)v;                                                       \
            }                                                                                       \
        }                                                                                           \
    }

void svt_oracle_dr_prediction(int zone, uint8_t *dst, ptrdiff_t stride, int bw, int bh,
                              const uint8_t *above, const uint8_t *left, int up_above, int up_left,
                              int dx, int dy) {
    DR_BODY(uint8_t, 255)
}
void svt_oracle_dr_prediction_hbd(int zone, uint16_t *dst, ptrdiff_t stride, int bw, int bh,
                                  const uint16_t *above, const uint16_t *left, int up_above,
                                  int up_left, int dx, int dy, int bd) {
    const int maxv = (1 << bd) - 1;
    DR_BODY(uint16_t, maxv)
}

/* av1_filter_intra_edge_c (EbIntraPrediction.c:3539-3565): 5-tap smoothing of an
 * edge of sz samples, p[0] untouched, taps by strength, clamped indices */
#define EDGE_FILTER_BODY(PIX)                                                                       \
    static const int k[3][5] = {{0, 4, 8, 4, 0}, {0, 5, 6, 5, 0}, {2, 4, 4, 4, 2}};                  \
    if (!strength) return;                                                                          \
    PIX edge[129];                                                                                  \
    memcpy(edge, p, sz * sizeof(*p));                                                               \
    for (int i = 1; i < sz; i++) {                                                                  \
        int s = 0;                                                                                  \
        for (int j = 0; j < 5; j++) {                                                               \
            int q = i - 2 + j;                                                                      \
            q = q < 0 ? 0 : (q > sz - 1 ? sz - 1 : q);                                              \
            s += edge[q] * k[strength - 1][j];                                                      \
        }                                                                                           \
        p[i] = (PIX)((s + 8) >> 4);                                                                 \
    }
void svt_oracle_filter_intra_edge(uint8_t *p, int sz, int strength) { EDGE_FILTER_BODY(uint8_t) }
void svt_oracle_filter_intra_edge_hbd(uint16_t *p, int sz, int strength) { EDGE_FILTER_BODY(uint16_t) }

/* av1_upsample_intra_edge_c (EbIntraPrediction.c:3597-3660): p[-2..2*sz-2] <- 2x
 * interpolation of p[-1..sz-1] with (-1, 9, 9, -1)/16 */
#define UPSAMPLE_BODY(PIX, MAXV)                                                                    \
    PIX in[16 + 3];                                                                                 \
    in[0] = p[-1]; in[1] = p[-1];                                                                   \
    for (int i = 0; i < sz; i++) in[i + 2] = p[i];                                                  \
    in[sz + 2] = p[sz - 1];                                                                         \
    p[-2] = in[0];                                                                                  \
    for (int i = 0; i < sz; i++) {                                                                  \
        int s = -in[i] + 9 * in[i + 1] + 9 * in[i + 2] - in[i + 3];                                 \
        s = (s + 8) >> 4;                                                                           \
        s = s < 0 ? 0 : (s > (MAXV) ? (MAXV) : s);                                                  \
        p[2 * i - 1] = (PIX)s;                                                                      \
        p[2 * i] = in[i + 2];                                                                       \
    }
void svt_oracle_upsample_intra_edge(uint8_t *p, int sz) { UPSAMPLE_BODY(uint8_t, 255) }
void svt_oracle_upsample_intra_edge_hbd(uint16_t *p, int sz, int bd) {
    const int maxv = (1 << bd) - 1;
    UPSAMPLE_BODY(uint16_t, maxv)
}
