/* oracle/cfl.c — TEST INFRASTRUCTURE ONLY (see svt_oracle.h): scalar restatement of the encode
 * pass's chroma-from-luma helpers (K11) and of av1_txb_init_levels, the pieces SURVEY.md §8(f) n3
 * lists next to the residual -> transform -> quantise -> reconstruct chain.  Pinned against the
 * reference's own cfl_*_c / subtract_average_c / av1_txb_init_levels_c compiled into
 * oracle/_ref/libsvtref.so (tests/test_cfl_levels_oracle.py). */
#include <string.h>
#include "svt_oracle.h"

#define CFL_LINE 32        /* CFL_BUF_LINE, EbDefinitions.h:185 */

/* cfl_luma_subsampling_420_{lbd,hbd}_c, EbIntraPrediction.c:1303-1332: each output is the sum of a
 * 2x2 luma quad times two (a Q3 value of the quad's average); output rows are CFL_LINE apart. */
void svt_oracle_cfl_luma_subsampling_420(const void *luma, int is_16bit, int32_t luma_stride,
                                         int16_t *out_q3, int32_t width, int32_t height) {
    for (int32_t y = 0; y < height / 2; y++)
        for (int32_t x = 0; x < width / 2; x++) {
            int32_t s = 0;
            for (int dy = 0; dy < 2; dy++)
                for (int dx = 0; dx < 2; dx++) {
                    const size_t at = (size_t)(2 * y + dy) * luma_stride + 2 * x + dx;
                    s += is_16bit ? ((const uint16_t *)luma)[at] : ((const uint8_t *)luma)[at];
                }
            out_q3[y * CFL_LINE + x] = (int16_t)(s * 2);
        }
}

/* subtract_average_c, EbIntraPrediction.c:1333-1359 */
void svt_oracle_subtract_average(int16_t *q3, int32_t width, int32_t height, int32_t round_offset,
                                 int32_t num_pel_log2) {
    int32_t total = 0;
    for (int32_t y = 0; y < height; y++)
        for (int32_t x = 0; x < width; x++) total += q3[y * CFL_LINE + x];
    const int16_t mean = (int16_t)((total + round_offset) >> num_pel_log2);
    for (int32_t y = 0; y < height; y++)
        for (int32_t x = 0; x < width; x++) q3[y * CFL_LINE + x] = (int16_t)(q3[y * CFL_LINE + x] - mean);
}

/* cfl_predict_{lbd,hbd}_c, EbIntraPrediction.c:1361-1402 with get_scaled_luma_q0 /
 * ROUND_POWER_OF_TWO_SIGNED (EbIntraPrediction.h:573-580): the AC term is alpha * ac in Q6, rounded
 * to Q0 symmetrically around zero, added to the DC prediction and clipped to the bit depth. */
void svt_oracle_cfl_predict(const int16_t *ac_q3, const void *pred, int32_t pred_stride, void *dst,
                            int32_t dst_stride, int32_t alpha_q3, int32_t bit_depth, int32_t width,
                            int32_t height, int is_16bit) {
    const int32_t hi = (1 << bit_depth) - 1;
    for (int32_t y = 0; y < height; y++)
        for (int32_t x = 0; x < width; x++) {
            const int32_t q6 = alpha_q3 * ac_q3[y * CFL_LINE + x];
            const int32_t mag = ((q6 < 0 ? -q6 : q6) + 32) >> 6;
            const int32_t dc = is_16bit ? (int16_t)((const uint16_t *)pred)[(size_t)y * pred_stride + x]
                                        : ((const uint8_t *)pred)[(size_t)y * pred_stride + x];
            int32_t v = dc + (q6 < 0 ? -mag : mag);
            v = v < 0 ? 0 : (v > hi ? hi : v);
            if (is_16bit) ((uint16_t *)dst)[(size_t)y * dst_stride + x] = (uint16_t)v;
            else ((uint8_t *)dst)[(size_t)y * dst_stride + x] = (uint8_t)v;
        }
}

/* av1_txb_init_levels_c, EbRateDistortionCost.c:125-150 (TX_PAD_* in EbDefinitions.h:273-280): the
 * level map is min(|coeff|, 127) in rows of width + 4 bytes (4 zero bytes of right padding) with 2
 * zero rows above and 4 zero rows + 16 zero bytes below.  `levels` points at the first coefficient
 * row, i.e. buffer + 2 * (width + 4). */
void svt_oracle_txb_init_levels(const int32_t *coeff, int32_t width, int32_t height, uint8_t *levels) {
    const int32_t pitch = width + 4;
    memset(levels - 2 * pitch, 0, (size_t)2 * pitch);
    memset(levels + (size_t)height * pitch, 0, (size_t)4 * pitch + 16);
    for (int32_t y = 0; y < height; y++) {
        for (int32_t x = 0; x < width; x++) {
            const int64_t c = coeff[y * width + x];
            const int64_t a = c < 0 ? -c : c;
            levels[y * pitch + x] = (uint8_t)(a > 127 ? 127 : a);
        }
        memset(levels + y * pitch + width, 0, 4);
    }
}
