/* oracle/ois.c — TEST INFRASTRUCTURE ONLY (see svt_oracle.h): scalar restatement of the open-loop intra
 * search of one block: open_loop_intra_search_sb (EbMotionEstimation.c:8694-8850) with
 * update_neighbor_samples_array_open_loop (EbIntraPrediction.c:4707-4773), intra_prediction_open_loop
 * (:4778-4808) and dr_predictor (:3352-3383).  Pinned by running the reference's own
 * open_loop_intra_search_sb (oracle/ref_ois.c) on the same pictures: tests/golden/ois.npz. */
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
