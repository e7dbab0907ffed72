/*
 * oracle/pixel.c — SAD / SAD search / SSE / residual + the fused headline chain.
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 */
#include "svt_oracle.h"
#include <stdlib.h>
#include <string.h>

/* fast_loop_nx_m_sad_kernel, C_DEFAULT/EbComputeSAD_C.c:48-70 */
uint32_t svt_oracle_sad(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                        uint32_t ref_stride, uint32_t height, uint32_t width) {
    uint32_t acc = 0;
    for (uint32_t y = 0; y < height; y++, src += src_stride, ref += ref_stride)
        for (uint32_t x = 0; x < width; x++) acc += (uint32_t)abs((int)src[x] - (int)ref[x]);
    return acc;
}

/* sad_loop_kernel, EbComputeSAD_C.c:72-120: raster search (y outer, x inner),
 * strict '<' keeps the FIRST minimum; best starts at 0xffffff; candidate rows
 * advance by src_stride_raw while rows inside a block advance by ref_stride. */
void svt_oracle_sad_loop(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                         uint32_t ref_stride, uint32_t height, uint32_t width,
                         uint64_t *best_sad, int16_t *xc, int16_t *yc, uint32_t src_stride_raw,
                         int16_t saw, int16_t sah) {
    *best_sad = 0xffffff;
    for (int16_t ys = 0; ys < sah; ys++, ref += src_stride_raw)
        for (int16_t xs = 0; xs < saw; xs++) {
            const uint32_t s = svt_oracle_sad(src, src_stride, ref + xs, ref_stride, height, width);
            if (s < *best_sad) { *best_sad = s; *xc = xs; *yc = ys; }
        }
}

/* spatial_full_distortion_kernel, C_DEFAULT/EbPictureOperators_C.c:40-65 */
uint64_t svt_oracle_sse(const uint8_t *a, uint32_t a_stride, const uint8_t *b, uint32_t b_stride,
                        uint32_t width, uint32_t height) {
    uint64_t acc = 0;
    for (uint32_t y = 0; y < height; y++, a += a_stride, b += b_stride)
        for (uint32_t x = 0; x < width; x++) { int64_t d = (int64_t)a[x] - b[x]; acc += (uint64_t)(d * d); }
    return acc;
}

/* full_distortion_kernel32_bits, EbPictureOperators.c:283-315 */
void svt_oracle_full_distortion32(const int32_t *coeff, uint32_t coeff_stride, const int32_t *recon,
                                  uint32_t recon_stride, uint64_t out[2], uint32_t width, uint32_t height) {
    uint64_t resid = 0, pred = 0;
    for (uint32_t y = 0; y < height; y++, coeff += coeff_stride, recon += recon_stride)
        for (uint32_t x = 0; x < width; x++) {
            int64_t d = (int64_t)coeff[x] - (int64_t)recon[x], c = coeff[x];
            resid += (uint64_t)(d * d);
            pred += (uint64_t)(c * c);
        }
    out[0] = resid; out[1] = pred;
}

/* residual_kernel_c, EbPictureOperators.c:166-193 */
void svt_oracle_residual(const uint8_t *src, uint32_t src_stride, const uint8_t *pred,
                         uint32_t pred_stride, int16_t *res, uint32_t res_stride, uint32_t width,
                         uint32_t height) {
    for (uint32_t y = 0; y < height; y++, src += src_stride, pred += pred_stride, res += res_stride)
        for (uint32_t x = 0; x < width; x++) res[x] = (int16_t)((int)src[x] - (int)pred[x]);
}

/* The headline unit of work: the encoder's Av1EncodeLoop sequence
 * EbCodingLoop.c:617 (residual) -> :655 (av1_estimate_transform) -> :673
 * (quantize, highbd semantics == AVX2 production path) plus the SAD of the
 * same (src, pred) pair (EbProductCodingLoop.c:1259). */
void svt_oracle_fwd_quant_sad(const uint8_t *src, uint32_t src_stride, const uint8_t *pred,
                              uint32_t pred_stride, int tx_size, int tx_type, const int16_t *zbin,
                              const int16_t *round, const int16_t *quant, const int16_t *quant_shift,
                              const int16_t *dequant, int32_t *coeff, int32_t *qcoeff,
                              int32_t *dqcoeff, uint16_t *eob, uint32_t *sad) {
    const int w = svt_oracle_tx_wide(tx_size), h = svt_oracle_tx_high(tx_size);
    int16_t *res = (int16_t *)malloc(sizeof(int16_t) * w * h);
    int16_t *scan = (int16_t *)malloc(sizeof(int16_t) * 1024 * 2), *iscan = scan + 1024;
    svt_oracle_residual(src, src_stride, pred, pred_stride, res, (uint32_t)w, (uint32_t)w, (uint32_t)h);
    svt_oracle_fwd_txfm2d(res, coeff, (uint32_t)w, tx_type, tx_size, 8);
    svt_oracle_fwd_txfm2d_pack64(coeff, tx_size);
    const int n = svt_oracle_get_scan(tx_size, tx_type, scan, iscan);
    /* av1_get_tx_scale (EbTransforms.h:317-329): 0 for <=256 px, 1 for <=1024, 2 above */
    const int pels = w * h;
    const int log_scale = pels > 1024 ? 2 : (pels > 256 ? 1 : 0);
    svt_oracle_quantize_b(coeff, n, 0, zbin, round, quant, quant_shift, qcoeff, dqcoeff, dequant, eob,
                          scan, iscan, log_scale, 0);
    *sad = svt_oracle_sad(src, src_stride, pred, pred_stride, (uint32_t)h, (uint32_t)w);
    free(res); free(scan);
}

/* ---- ME multi-size SAD of one 64x64 SB over a full-pel search area (K6) -------
 * Restates FullPelSearch_LCU (EbMotionEstimation.c:3199-3247) -> GetSearchPointResults
 * (:2932-3057) -> ext_sad_calculation_8x8_16x16 (:208-262) + ext_sad_calculation_32x32_64x64
 * (:267-311): per search point (raster order, y outer), the 64 8x8 SADs are taken on
 * EVERY OTHER ROW and doubled (Compute8x4SAD_Kernel with 2x strides, :121-143), summed to
 * 16x16 / 32x32 / 64x64; each of the 85 PUs keeps its running best with strict '<' and
 * the packed quarter-pel MV ((uint16)y << 18) | (uint16)(x << 2)  (:2958-2960).
 * Layout of best_sad / best_mv[85]: [0..63] 8x8 at 4*z + k (z = z-order index of the
 * parent 16x16, k = raster index inside it), [64..79] 16x16 in z-order, [80..83] 32x32,
 * [84] 64x64 — the reference's p_best_sad8x8/16x16/32x32/64x64 arrays back to back. */
void svt_oracle_me_sb_search(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                             uint32_t ref_stride, int search_w, int search_h, int x_origin,
                             int y_origin, uint32_t *best_sad, uint32_t *best_mv) {
    for (int ys = 0; ys < search_h; ys++)
        for (int xs = 0; xs < search_w; xs++) {
            const uint32_t mv = (uint32_t)(((uint32_t)(uint16_t)(ys + y_origin)) << 18) |
                                (uint32_t)(uint16_t)((xs + x_origin) << 2);
            const uint8_t *r0 = ref + xs + (size_t)ys * ref_stride;
            uint32_t s16[16], s32[4] = {0, 0, 0, 0}, s64 = 0;
            for (int by16 = 0; by16 < 4; by16++)
                for (int bx16 = 0; bx16 < 4; bx16++) {
                    const int z = ((by16 >> 1) * 2 + (bx16 >> 1)) * 4 + (by16 & 1) * 2 + (bx16 & 1);
                    uint32_t sum = 0;
                    for (int k = 0; k < 4; k++) {
                        const int x0 = bx16 * 16 + (k & 1) * 8, y0 = by16 * 16 + (k >> 1) * 8;
                        uint32_t s = 0;
                        for (int rr = 0; rr < 4; rr++)
                            for (int c = 0; c < 8; c++)
                                s += (uint32_t)abs((int)src[(size_t)(y0 + 2 * rr) * src_stride + x0 + c] -
                                                   (int)r0[(size_t)(y0 + 2 * rr) * ref_stride + x0 + c]);
                        s <<= 1;
                        if (s < best_sad[4 * z + k]) { best_sad[4 * z + k] = s; best_mv[4 * z + k] = mv; }
                        sum += s;
                    }
                    s16[z] = sum;
                    if (sum < best_sad[64 + z]) { best_sad[64 + z] = sum; best_mv[64 + z] = mv; }
                }
            for (int q = 0; q < 4; q++) {
                s32[q] = s16[4 * q] + s16[4 * q + 1] + s16[4 * q + 2] + s16[4 * q + 3];
                if (s32[q] < best_sad[80 + q]) { best_sad[80 + q] = s32[q]; best_mv[80 + q] = mv; }
                s64 += s32[q];
            }
            if (s64 < best_sad[84]) { best_sad[84] = s64; best_mv[84] = mv; }
        }
}

/* General form of the chain for any size / type / bit depth, planes of 8- or 16-bit
 * samples (Av1EncodeLoop EbCodingLoop.c:545 and Av1EncodeLoop16bit :1020: residual
 * (residual_kernel16bit EbPictureOperators.c:134) -> av1_estimate_transform (incl.
 * three_quad_energy for 64-pt) -> av1_quantize_inv_quantize with the high-bit-depth
 * quantizer).  sad is only defined for 8-bit input (may be NULL). */
void svt_oracle_fwd_quant_planes(const void *src, uint32_t src_stride, const void *pred,
                                 uint32_t pred_stride, int is_16bit, int bd, int tx_size, int tx_type,
                                 const int16_t *zbin, const int16_t *round, const int16_t *quant,
                                 const int16_t *quant_shift, const int16_t *dequant, int32_t *coeff,
                                 int32_t *qcoeff, int32_t *dqcoeff, uint16_t *eob, uint32_t *sad,
                                 uint64_t *energy) {
    const int w = svt_oracle_tx_wide(tx_size), h = svt_oracle_tx_high(tx_size);
    int16_t *res = (int16_t *)malloc(sizeof(int16_t) * w * h);
    int32_t *full = (int32_t *)malloc(sizeof(int32_t) * w * h);
    int16_t *scan = (int16_t *)malloc(sizeof(int16_t) * 1024 * 2), *iscan = scan + 1024;
    uint32_t acc = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int s = is_16bit ? ((const uint16_t *)src)[(size_t)y * src_stride + x] : ((const uint8_t *)src)[(size_t)y * src_stride + x];
            const int p = is_16bit ? ((const uint16_t *)pred)[(size_t)y * pred_stride + x] : ((const uint8_t *)pred)[(size_t)y * pred_stride + x];
            res[y * w + x] = (int16_t)(s - p);
            acc += (uint32_t)abs(s - p);
        }
    svt_oracle_fwd_txfm2d(res, full, (uint32_t)w, tx_type, tx_size, bd);
    const uint64_t e = svt_oracle_fwd_txfm2d_pack64(full, tx_size);
    const int n = svt_oracle_get_scan(tx_size, tx_type, scan, iscan);
    memcpy(coeff, full, sizeof(int32_t) * n);
    const int pels = w * h;
    const int log_scale = pels > 1024 ? 2 : (pels > 256 ? 1 : 0);
    svt_oracle_quantize_b(coeff, n, 0, zbin, round, quant, quant_shift, qcoeff, dqcoeff, dequant, eob,
                          scan, iscan, log_scale, 0);
    if (sad) *sad = acc;
    if (energy) *energy = e;
    free(res); free(full); free(scan);
}
