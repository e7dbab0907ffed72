/* oracle/ois.c — TEST INFRASTRUCTURE ONLY (see svt_oracle.h): scalar restatement of the open-loop intra
 * search of one block: open_loop_intra_search_sb (EbMotionEstimation.c:8694-8850) with
 * update_neighbor_samples_array_open_loop (EbIntraPrediction.c:4707-4773), intra_prediction_open_loop
 * (:4778-4808) and dr_predictor (:3352-3383).  Pinned by running the reference's own
 * open_loop_intra_search_sb (oracle/ref_ois.c) on the same pictures: tests/golden/ois.npz. */
#include <stdlib.h>
#include <string.h>
#include "svt_oracle.h"

/* dr_intra_derivative, EbIntraPrediction.c:299 (AV1 spec 7.11.2.4 Dr_Intra_Derivative): non-zero entries only */
static int dr_derivative(int angle) {
    static const uint16_t at[][2] = {{3, 1023}, {6, 547}, {9, 372}, {14, 273}, {17, 215}, {20, 178}, {23, 151}, {26, 132},
                                     {29, 116}, {32, 102}, {36, 90}, {39, 80}, {42, 71}, {45, 64}, {48, 57}, {51, 51},
                                     {54, 45}, {58, 40}, {61, 35}, {64, 31}, {67, 27}, {70, 23}, {73, 19}, {76, 15},
                                     {81, 11}, {84, 7}, {87, 3}};
    for (unsigned i = 0; i < sizeof(at) / sizeof(at[0]); i++)
        if (at[i][0] == angle) return at[i][1];
    return 0;
}
int svt_oracle_dr_intra_derivative(int angle) { return dr_derivative(angle); }

/* AV1 PredictionMode numbering (EbDefinitions.h): DC 0, V 1, H 2, D45 3, D135 4, D113 5, D157 6, D203 7, D67 8,
 * SMOOTH 9, SMOOTH_V 10, SMOOTH_H 11, PAETH 12; mode_to_angle_map, EbCodingUnit.h:129 */
static const int k_mode_angle[13] = {0, 90, 180, 45, 135, 113, 157, 203, 67, 0, 0, 0, 0};

/* the candidate list of open_loop_intra_search_sb (:8747-8846) for one block size */
int svt_oracle_ois_candidates(int bsize, int temporal_layer_index, int intra_pred_mode, int is_used_as_reference,
                              int is_16bit, uint8_t *modes, int8_t *angle_deltas) {
    int last = is_16bit ? 11 : 12;                                         /* SMOOTH_H_PRED : PAETH_PRED */
    int nd = intra_pred_mode >= 5 ? 1 : (bsize >= 8 ? 5 : 1);              /* M8_OIS */
    const int no_angular = temporal_layer_index > 0 ? 1 : (bsize > 16);
    if (no_angular) nd = 1;
    if (!is_used_as_reference && intra_pred_mode >= 4) last = 0;           /* DC only */
    int n = 0;
    for (int m = 0; m <= last; m++) {
        if (m >= 1 && m <= 8) {                                            /* av1_is_directional_mode: V_PRED .. D67_PRED */
            if (no_angular) continue;
            for (int k = 0; k < nd; k++) { modes[n] = (uint8_t)m; angle_deltas[n] = (int8_t)(nd == 1 ? 0 : k - (nd >> 1)); n++; }
        } else { modes[n] = (uint8_t)m; angle_deltas[n] = 0; n++; }
    }
    return n;
}

/* update_neighbor_samples_array_open_loop + the copies at :8742-8752.  pic points at picture sample (0, 0).
 * above / left: 2 * bsize + 1 entries, [0] = top-left. */
void svt_oracle_ois_neighbors(const uint8_t *pic, uint32_t stride, uint32_t width, uint32_t height, uint32_t x,
                              uint32_t y, uint32_t bsize, uint8_t *above, uint8_t *left) {
    const uint32_t n2 = 2 * bsize;
    const uint8_t *src = pic + (size_t)y * stride + x;
    memset(above, 127, n2 + 1);
    memset(left, 129, n2 + 1);
    above[0] = left[0] = (x != 0 && y != 0) ? src[-(ptrdiff_t)stride - 1] : 128;
    if (x != 0) {
        const uint32_t cnt = y + n2 > height ? height - y : n2;
        for (uint32_t i = 0; i < cnt; i++) left[1 + i] = src[(size_t)i * stride - 1];
    }
    if (y != 0) {
        const uint32_t cnt = x + n2 > width ? width - x : n2;
        memcpy(above + 1, src - stride, cnt);
    }
}

/* one block: distortion of every candidate (SAD against the source block) and the index of the first strict
 * minimum below 64 * 64 * 255 (:8756, 8800-8803) */
int svt_oracle_ois_block(const uint8_t *pic, uint32_t stride, uint32_t width, uint32_t height, uint32_t x, uint32_t y,
                         uint32_t bsize, int ncand, const uint8_t *modes, const int8_t *angle_deltas,
                         uint32_t *distortion) {
    uint8_t top[129], lft[129];
    uint8_t above_data[16 + 128 + 32], left_data[16 + 128 + 32];
    uint8_t *above = above_data + 16, *left = left_data + 16;
    uint8_t pred[64 * 64];
    svt_oracle_ois_neighbors(pic, stride, width, height, x, y, bsize, top, lft);
    memset(above_data, 0, sizeof(above_data)); memset(left_data, 0, sizeof(left_data));
    memcpy(above, top + 1, 2 * bsize); memcpy(left, lft + 1, 2 * bsize);
    above[-1] = left[-1] = top[0];
    uint32_t best = 64 * 64 * 255;
    int best_i = 0;
    const int b = (int)bsize;
    for (int c = 0; c < ncand; c++) {
        const int m = modes[c];
        if (m >= 1 && m <= 8) {                                            /* dr_predictor */
            const int a = k_mode_angle[m] + 3 * angle_deltas[c];
            if (a == 90) svt_oracle_intra_pred(ORC_V_PRED, pred, b, b, b, above, left);
            else if (a == 180) svt_oracle_intra_pred(ORC_H_PRED, pred, b, b, b, above, left);
            else if (a < 90) svt_oracle_dr_prediction(1, pred, b, b, b, above, left, 0, 0, dr_derivative(a), 1);
            else if (a < 180) svt_oracle_dr_prediction(2, pred, b, b, b, above, left, 0, 0, dr_derivative(180 - a), dr_derivative(a - 90));
            else svt_oracle_dr_prediction(3, pred, b, b, b, above, left, 0, 0, 1, dr_derivative(270 - a));
        } else if (m == 0) {                                               /* dc_pred[x > 0][y > 0] */
            const int om = x > 0 ? (y > 0 ? ORC_DC_PRED : ORC_DC_LEFT_PRED) : (y > 0 ? ORC_DC_TOP_PRED : ORC_DC_128_PRED);
            svt_oracle_intra_pred(om, pred, b, b, b, above, left);
        } else {
            const int om = m == 9 ? ORC_SMOOTH_PRED : m == 10 ? ORC_SMOOTH_V_PRED : m == 11 ? ORC_SMOOTH_H_PRED : ORC_PAETH_PRED;
            svt_oracle_intra_pred(om, pred, b, b, b, above, left);
        }
        distortion[c] = svt_oracle_sad(pic + (size_t)y * stride + x, stride, pred, bsize, bsize, bsize);
        if (distortion[c] < best) { best = distortion[c]; best_i = c; }
    }
    return best_i;
}

/* ---- build_intra_predictors / build_intra_predictors_high (EbIntraPrediction.c:3667-3855, 3857-4076): the neighbour-
 * availability glue of av1_predict_intra_block.  `top` / `left` point at element 0 of the reference's neighbour arrays
 * (element -1 of `top` is the corner sample); n_*_px are the available sample counts the caller derived from the block's
 * position (av1_predict_intra_block :4078-4190).  Steps: which edges the mode needs (extend_modes, directional angle
 * classes); constant fill when the needed edge is wholly missing; edge extension (replicate the last available sample;
 * base +- 1 defaults, base = 128 << (bd - 8)); corner; for directional modes the corner / edge filters and up-sampling
 * by size, angle and the neighbours' smoothness (filt_type); DC by availability. */
static const int kModeAngle[13] = {0, 90, 180, 45, 135, 113, 157, 203, 67, 0, 0, 0, 0};     /* mode_to_angle_map, EbCodingUnit.h:129 */
/* extend_modes (:1410-1424): 1 = left, 2 = above, 4 = above-right, 8 = above-left, 16 = bottom-left */
static const int kExtend[13] = {1 | 2, 2, 1, 2 | 4, 1 | 2 | 8, 1 | 2 | 8, 1 | 2 | 8, 1 | 16, 2 | 4, 1 | 2, 1 | 2, 1 | 2, 1 | 2 | 8};

static int edge_upsample(int bs0, int bs1, int delta, int type) {           /* use_intra_edge_upsample, :167-172 */
    const int d = abs(delta), wh = bs0 + bs1;
    if (d <= 0 || d >= 40) return 0;
    return type ? (wh <= 8) : (wh <= 16);
}
static int edge_strength(int bs0, int bs1, int delta, int type) {           /* intra_edge_filter_strength, :225-268 */
    const int d = abs(delta), wh = bs0 + bs1;
    int s = 0;
    if (type == 0) {
        if (wh <= 8) { if (d >= 56) s = 1; }
        else if (wh <= 16) { if (d >= 40) s = 1; }
        else if (wh <= 24) { if (d >= 8) s = 1; if (d >= 16) s = 2; if (d >= 32) s = 3; }
        else if (wh <= 32) { if (d >= 1) s = 1; if (d >= 4) s = 2; if (d >= 32) s = 3; }
        else { if (d >= 1) s = 3; }
    } else {
        if (wh <= 8) { if (d >= 40) s = 1; if (d >= 64) s = 2; }
        else if (wh <= 16) { if (d >= 20) s = 1; if (d >= 48) s = 2; }
        else if (wh <= 24) { if (d >= 4) s = 3; }
        else { if (d >= 1) s = 3; }
    }
    return s;
}

void svt_oracle_build_intra_predictors(int is16, const void *top_v, const void *left_v, void *dst_v, int32_t dst_stride,
                                       int mode, int angle_delta, int tx_size, int disable_edge_filter, int n_top_px,
                                       int n_topright_px, int n_left_px, int n_bottomleft_px, int filt_type, int bd) {
    const int w = svt_oracle_tx_wide(tx_size), h = svt_oracle_tx_high(tx_size);
    /* work on 16-bit copies for both sample sizes; the arithmetic is the same, only the clip differs (bd) */
    uint16_t above_data[64 * 2 + 48], left_data[64 * 2 + 48];
    uint16_t *above_row = above_data + 16, *left_col = left_data + 16;
    memset(above_data, 0, sizeof(above_data)); memset(left_data, 0, sizeof(left_data));
#define TOP(i) (is16 ? ((const uint16_t *)top_v)[i] : (uint16_t)((const uint8_t *)top_v)[i])
#define LEFT(i) (is16 ? ((const uint16_t *)left_v)[i] : (uint16_t)((const uint8_t *)left_v)[i])
    const int base = 128 << (bd - 8);
    int need_left = kExtend[mode] & 1, need_above = (kExtend[mode] & 2) != 0, need_above_left = (kExtend[mode] & 8) != 0;
    const int is_dr = mode >= 1 && mode <= 8;
    int p_angle = 0;
    if (is_dr) {
        p_angle = kModeAngle[mode] + angle_delta * 3;
        if (p_angle <= 90) { need_above = 1; need_left = 0; need_above_left = 1; }
        else if (p_angle < 180) { need_above = 1; need_left = 1; need_above_left = 1; }
        else { need_above = 0; need_left = 1; need_above_left = 1; }
    }
    uint16_t out16[64 * 64];
    int done = 0;
    if ((!need_above && n_left_px == 0) || (!need_left && n_top_px == 0)) {
        int val;
        if (need_left) val = (n_top_px > 0) ? TOP(0) : base + 1;
        else val = (n_left_px > 0) ? LEFT(0) : base - 1;
        for (int i = 0; i < w * h; i++) out16[i] = (uint16_t)val;
        done = 1;
    }
    if (!done) {
        if (need_left) {
            int need_bottom = (kExtend[mode] & 16) != 0;
            if (is_dr) need_bottom = p_angle > 180;
            const int need = h + (need_bottom ? w : 0);
            int i = 0;
            if (n_left_px > 0) {
                for (; i < n_left_px; i++) left_col[i] = LEFT(i);
                if (need_bottom && n_bottomleft_px > 0) for (; i < h + n_bottomleft_px; i++) left_col[i] = LEFT(i);
                for (; i < need; i++) left_col[i] = left_col[i - 1];
            } else {
                const int v = n_top_px > 0 ? TOP(0) : base + 1;
                for (i = 0; i < need; i++) left_col[i] = (uint16_t)v;
            }
        }
        if (need_above) {
            int need_right = (kExtend[mode] & 4) != 0;
            if (is_dr) need_right = p_angle < 90;
            const int need = w + (need_right ? h : 0);
            if (n_top_px > 0) {
                int i;
                for (i = 0; i < n_top_px; i++) above_row[i] = TOP(i);
                if (need_right && n_topright_px > 0) { for (int k = 0; k < n_topright_px; k++) above_row[w + k] = TOP(w + k); i += n_topright_px; }
                for (; i < need; i++) above_row[i] = above_row[i - 1];
            } else {
                const int v = n_left_px > 0 ? LEFT(0) : base - 1;
                for (int i = 0; i < need; i++) above_row[i] = (uint16_t)v;
            }
        }
        if (need_above_left) {
            if (n_top_px > 0 && n_left_px > 0) above_row[-1] = TOP(-1);
            else if (n_top_px > 0) above_row[-1] = TOP(0);
            else if (n_left_px > 0) above_row[-1] = LEFT(0);
            else above_row[-1] = (uint16_t)base;
            left_col[-1] = above_row[-1];
        }
        if (is_dr) {
            int up_a = 0, up_l = 0;
            if (!disable_edge_filter) {
                const int need_right = p_angle < 90, need_bottom = p_angle > 180;
                if (p_angle != 90 && p_angle != 180) {
                    const int ab_le = need_above_left ? 1 : 0;
                    if (need_above && need_left && (w + h >= 24)) {        /* filter_intra_edge_corner(_high), :3383 / :3562 */
                        const int s = (left_col[0] * 5 + above_row[-1] * 6 + above_row[0] * 5 + 8) >> 4;
                        above_row[-1] = (uint16_t)s; left_col[-1] = (uint16_t)s;
                    }
                    if (need_above && n_top_px > 0)
                        svt_oracle_filter_intra_edge_hbd(above_row - ab_le, n_top_px + ab_le + (need_right ? h : 0), edge_strength(w, h, p_angle - 90, filt_type));
                    if (need_left && n_left_px > 0)
                        svt_oracle_filter_intra_edge_hbd(left_col - ab_le, n_left_px + ab_le + (need_bottom ? w : 0), edge_strength(h, w, p_angle - 180, filt_type));
                }
                up_a = edge_upsample(w, h, p_angle - 90, filt_type);
                if (need_above && up_a) svt_oracle_upsample_intra_edge_hbd(above_row, w + (need_right ? h : 0), bd);
                up_l = edge_upsample(h, w, p_angle - 180, filt_type);
                if (need_left && up_l) svt_oracle_upsample_intra_edge_hbd(left_col, h + (need_bottom ? w : 0), bd);
            }
            /* dr_predictor (:3352-3381) */
            if (p_angle > 0 && p_angle < 90) svt_oracle_dr_prediction_hbd(1, out16, w, w, h, above_row, left_col, up_a, up_l, dr_derivative(p_angle), 1, bd);
            else if (p_angle > 90 && p_angle < 180)
                svt_oracle_dr_prediction_hbd(2, out16, w, w, h, above_row, left_col, up_a, up_l, dr_derivative(180 - p_angle), dr_derivative(p_angle - 90), bd);
            else if (p_angle > 180 && p_angle < 270) svt_oracle_dr_prediction_hbd(3, out16, w, w, h, above_row, left_col, up_a, up_l, 1, dr_derivative(270 - p_angle), bd);
            else if (p_angle == 90) svt_oracle_intra_pred_hbd(ORC_V_PRED, out16, w, w, h, above_row, left_col, bd);
            else svt_oracle_intra_pred_hbd(ORC_H_PRED, out16, w, w, h, above_row, left_col, bd);
        } else if (mode == 0) {
            /* dc_pred[n_left_px > 0][n_top_px > 0] (:3851): [0][0] dc_128, [0][1] dc_top, [1][0] dc_left, [1][1] dc */
            const int m = n_left_px > 0 ? (n_top_px > 0 ? ORC_DC_PRED : ORC_DC_LEFT_PRED) : (n_top_px > 0 ? ORC_DC_TOP_PRED : ORC_DC_128_PRED);
            svt_oracle_intra_pred_hbd(m, out16, w, w, h, above_row, left_col, bd);
        } else {
            static const int map[13] = {0, ORC_V_PRED, ORC_H_PRED, 0, 0, 0, 0, 0, 0, ORC_SMOOTH_PRED, ORC_SMOOTH_V_PRED, ORC_SMOOTH_H_PRED, ORC_PAETH_PRED};
            svt_oracle_intra_pred_hbd(map[mode], out16, w, w, h, above_row, left_col, bd);
        }
    }
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            if (is16) ((uint16_t *)dst_v)[(size_t)r * dst_stride + c] = out16[r * w + c];
            else ((uint8_t *)dst_v)[(size_t)r * dst_stride + c] = (uint8_t)out16[r * w + c];
        }
#undef TOP
#undef LEFT
}
