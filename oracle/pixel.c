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

/* combined_averaging_sad, C_DEFAULT/EbComputeSAD_C.c:13-40: SAD against the rounded average of two references */
uint32_t svt_oracle_sad_avg(const uint8_t *src, uint32_t src_stride, const uint8_t *ref1, uint32_t ref1_stride,
                            const uint8_t *ref2, uint32_t ref2_stride, uint32_t height, uint32_t width) {
    uint32_t sad = 0;
    for (uint32_t y = 0; y < height; y++, src += src_stride, ref1 += ref1_stride, ref2 += ref2_stride)
        for (uint32_t x = 0; x < width; x++) {
            const int avg = (ref1[x] + ref2[x] + 1) >> 1;
            sad += (uint32_t)abs((int)src[x] - avg);
        }
    return sad;
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

/* The same sums as the reference's AVX2 kernel forms them (full_distortion_kernel32_bits_avx2,
 * EbPictureOperators_Intrinsic_AVX2.c:1955-2011): four 64-bit lanes, one per column mod 4; the squared difference is
 * the signed product of the LOW 32 bits of the 64-bit difference (_mm256_mul_epi32) and is accumulated with
 * _mm256_add_epi32, i.e. the low and the high 32-bit halves of a lane add separately, without a carry between them.
 * The prediction sum (and the whole cbf_zero kernel) adds in 64 bits like the C code.  width % 4 == 0. */
void svt_oracle_full_distortion32_avx2(const int32_t *coeff, uint32_t coeff_stride, const int32_t *recon,
                                       uint32_t recon_stride, uint64_t out[2], uint32_t width, uint32_t height) {
    uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    uint64_t pred = 0, resid = 0;
    for (uint32_t y = 0; y < height; y++, coeff += coeff_stride, recon += recon_stride)
        for (uint32_t x = 0; x < width; x++) {
            const int64_t d = (int64_t)coeff[x] - (int64_t)recon[x], c = coeff[x];
            const int64_t dl = (int32_t)(uint32_t)(uint64_t)d;            /* low 32 bits, sign-extended */
            const uint64_t sq = (uint64_t)(dl * dl);
            lo[x & 3] += (uint32_t)sq;
            hi[x & 3] += (uint32_t)(sq >> 32);
            pred += (uint64_t)(c * c);
        }
    for (int l = 0; l < 4; l++) resid += ((uint64_t)hi[l] << 32) | lo[l];
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

/* ---- K6 in the reference's own result layout, square + non-square PUs, both production flavours ---------------
 * best_sad / best_mv: uint32[209] in EbMeTierZeroPu order (EbMotionEstimationContext.h:47-270), i.e. the reference's
 * p_sb_best_sad[list][ref][...] / p_sb_best_mv[...]: 64x64 [0], 32x32 [1..4], 16x16 [5..20], 8x8 [21..84] and, when
 * nsq != 0 (open_loop_me_fullpel_search_sblock, EbMotionEstimation.c:3251, used when nsq_search_level is between
 * LEVEL1 and FULL, :7629), 64x32 [85..86], 32x16 [87..94], 16x8 [95..126], 32x64 [127..128], 16x32 [129..136],
 * 8x16 [137..168], 32x8 [169..184], 8x32 [185..200], 64x16 [201..204], 16x64 [205..208]; sums as
 * ext_eigth_sad_calculation_nsq_c (:1455-2490) / ext_sad_calculation (single search point) form them from the 8x8 /
 * 16x16 / 32x32 SADs.  Search points in raster order, strict '<' against the running best.
 *
 * flavour 0 = the scalar C / SSE4.1 kernels (asm_type 0).  flavour 1 = what an AVX2 build made with GCC or clang
 * really computes (asm_type 1, the only value production accepts, EbEncHandle.c:2676): in
 * get_eight_horizontal_search_point_results_32x32_64x64_pu_avx2_intrin (EbComputeSAD_Intrinsic_AVX2.c:3696-4072) the
 * `#ifdef __GNUC__` branch (:3989-4001) swaps the two 128-bit halves, so for the four 32x32 PUs, inside every full group
 * of eight search points of FullPelSearch_LCU (:3220), the search point p of the group is taken as p ^ 4 - for the
 * tie-break between equal SADs and for the motion vector that is stored (the SAD value stays the true minimum).
 * The remainder points of a row (GetSearchPointResults, :2932) and every other PU size are unaffected.  The NSQ path's
 * 8-point AVX2 kernels agree with the C ones; its single-search-point form (remainder points: production rounds the
 * search width down to a multiple of 8 unless it is below 8, :8016-8021) has two more quirks, restated where they apply
 * below: ExtSadCalculation's stale-`sad` test for 32x16_5 (both flavours) and the AVX2 kernel's source-stride row fetch. */
static void me_update(uint32_t *bs, uint32_t *bm, int idx, uint32_t sad, uint32_t mv) {
    if (sad < bs[idx]) { bs[idx] = sad; bm[idx] = mv; }
}

/* SADs of one search point: 64 8x8 (every other row, doubled: Compute8x4SAD_Kernel with 2x strides, :121-143, 208-262),
 * 16 16x16 in the reference's z-order, 4 32x32, 64x64 */
static void me_point_sads(const uint8_t *src, uint32_t src_stride, const uint8_t *r0, uint32_t ref_stride,
                          uint32_t s8[64], uint32_t s16[16], uint32_t s32[4], uint32_t *s64, int avx2_row_bug) {
    for (int by16 = 0; by16 < 4; by16++)
        for (int bx16 = 0; bx16 < 4; bx16++) {
            const int z = ((by16 >> 1) * 2 + (bx16 >> 1)) * 4 + (by16 & 1) * 2 + (bx16 & 1);
            s16[z] = 0;
            for (int k = 0; k < 4; k++) {
                const int x0 = bx16 * 16 + (k & 1) * 8, y0 = by16 * 16 + (k >> 1) * 8;
                uint32_t s = 0;
                for (int rr = 0; rr < 4; rr++) {
                    /* ext_sad_calculation_8x8_16x16_avx2_intrin (EbComputeSAD_Intrinsic_AVX2.c:50-52) fetches the first
                     * sampled reference row of the lower 8x8 pair at `ref + 4 * src_stride` (strides already doubled):
                     * 8 SOURCE strides below the 16x16 block's reference origin instead of 8 reference strides */
                    const uint8_t *rrow = (avx2_row_bug && k >= 2 && rr == 0)
                        ? r0 + (size_t)(by16 * 16) * ref_stride + (size_t)8 * src_stride
                        : r0 + (size_t)(y0 + 2 * rr) * ref_stride;
                    for (int c = 0; c < 8; c++)
                        s += (uint32_t)abs((int)src[(size_t)(y0 + 2 * rr) * src_stride + x0 + c] - (int)rrow[x0 + c]);
                }
                s8[4 * z + k] = s << 1;
                s16[z] += s8[4 * z + k];
            }
        }
    *s64 = 0;
    for (int q = 0; q < 4; q++) { s32[q] = s16[4 * q] + s16[4 * q + 1] + s16[4 * q + 2] + s16[4 * q + 3]; *s64 += s32[q]; }
}

static void me_nsq_update(const uint32_t s8[64], const uint32_t s16[16], const uint32_t s32[4], uint32_t mv,
                          uint32_t *best_sad, uint32_t *best_mv, int single_point) {
    uint32_t s32x16[8], s16x8[32], s16x32[8], s8x16[32];
    for (int i = 0; i < 2; i++) me_update(best_sad, best_mv, 85 + i, s32[2 * i] + s32[2 * i + 1], mv);
    for (int i = 0; i < 8; i++) {
        s32x16[i] = s16[2 * i] + s16[2 * i + 1];
        if (single_point && i == 5) {
            /* ExtSadCalculation (EbMotionEstimation.c:732-736, the single-search-point form) tests the stale `sad`
             * of the 64x32_1 sum here, not sad_32x16[5], and then stores sad_32x16[5] */
            if (s32[2] + s32[3] < best_sad[87 + 5]) { best_sad[87 + 5] = s32x16[5]; best_mv[87 + 5] = mv; }
        } else {
            me_update(best_sad, best_mv, 87 + i, s32x16[i], mv);
        }
    }
    for (int i = 0; i < 32; i++) { s16x8[i] = s8[2 * i] + s8[2 * i + 1]; me_update(best_sad, best_mv, 95 + i, s16x8[i], mv); }
    for (int i = 0; i < 2; i++) me_update(best_sad, best_mv, 127 + i, s32[i] + s32[i + 2], mv);
    for (int i = 0; i < 8; i++) {
        const int b = (i >> 1) * 4 + (i & 1);
        s16x32[i] = s16[b] + s16[b + 2];
        me_update(best_sad, best_mv, 129 + i, s16x32[i], mv);
    }
    for (int i = 0; i < 32; i++) {
        const int b = (i >> 1) * 4 + (i & 1);
        s8x16[i] = s8[b] + s8[b + 2];
        me_update(best_sad, best_mv, 137 + i, s8x16[i], mv);
    }
    for (int i = 0; i < 16; i++) {
        const int b = (i >> 1) * 4 + (i & 1);
        me_update(best_sad, best_mv, 169 + i, s16x8[b] + s16x8[b + 2], mv);
    }
    for (int i = 0; i < 16; i++) {
        const int b = (i >> 2) * 8 + (i & 3);
        me_update(best_sad, best_mv, 185 + i, s8x16[b] + s8x16[b + 4], mv);
    }
    for (int i = 0; i < 4; i++) {
        const int b = (i >> 1) * 4 + (i & 1);
        me_update(best_sad, best_mv, 201 + i, s32x16[b] + s32x16[b + 2], mv);
    }
    for (int i = 0; i < 4; i++) me_update(best_sad, best_mv, 205 + i, s16x32[i] + s16x32[i + 4], mv);
}

void svt_oracle_me_sb_search_full(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                                  int search_w, int search_h, int x_origin, int y_origin, int flavour, int nsq,
                                  uint32_t *best_sad, uint32_t *best_mv) {
    const int w8 = search_w & ~7;
    const int quirk = flavour == 1 && !nsq;
    for (int ys = 0; ys < search_h; ys++) {
        const uint32_t mvy = ((uint32_t)(uint16_t)(ys + y_origin)) << 18;
        for (int xg = 0; xg < search_w; xg += 8) {
            const int np = search_w - xg < 8 ? search_w - xg : 8;
            uint32_t g32[8][4];
            for (int p = 0; p < np; p++) {
                const int xs = xg + p;
                const uint32_t mv = mvy | (uint32_t)(uint16_t)((xs + x_origin) << 2);
                uint32_t s8[64], s16[16], s64;
                const int single = xg >= w8;      /* GetSearchPointResults / open_loop_me_get_search_point_results_block */
                me_point_sads(src, src_stride, ref + xs + (size_t)ys * ref_stride, ref_stride, s8, s16, g32[p], &s64,
                              flavour == 1 && nsq && single);
                me_update(best_sad, best_mv, 0, s64, mv);
                if (!(quirk && xg < w8))
                    for (int q = 0; q < 4; q++) me_update(best_sad, best_mv, 1 + q, g32[p][q], mv);
                for (int i = 0; i < 16; i++) me_update(best_sad, best_mv, 5 + i, s16[i], mv);
                for (int i = 0; i < 64; i++) me_update(best_sad, best_mv, 21 + i, s8[i], mv);
                if (nsq) me_nsq_update(s8, s16, g32[p], mv, best_sad, best_mv, single);
            }
            if (quirk && xg < w8)
                /* the swapped halves: the group's points are ranked, and reported, as p ^ 4 */
                for (int pa = 0; pa < 8; pa++)
                    for (int q = 0; q < 4; q++)
                        me_update(best_sad, best_mv, 1 + q, g32[pa ^ 4][q],
                                  mvy | (uint32_t)(uint16_t)((xg + pa + x_origin) << 2));
        }
    }
}

/* ---- hierarchical ME levels 0 / 1 / 2 (HmeLevel0 / HmeLevel1 / HmeLevel2, EbMotionEstimation.c:5689-6150) -----------
 * One arithmetic serves the three levels; what differs is the parameter set (svt_oracle_hme_params_for_level):
 *   search area before clipping  L0: width ((w[region] * mult / 100) + 15) & ~15, height h[region] * mult / 100
 *                                L1 / L2: width (w + 7) & ~7, height h
 *   origin = offset + centre     L0: -((total * mult / 100) >> 1) + sum of the preceding regions' widths; L1 / L2: -(area >> 1)
 *   clipping against the reference picture with its padding (pad = origin - 1; level 2: BLOCK_SIZE_64 - 1), left / right,
 *   top / bottom exactly in the reference's order (:5745-5798), then width rounded DOWN to a multiple of 16 (L0) / 8
 *   unless it is smaller; search = sad_loop_kernel on EVERY OTHER ROW of the block (source and reference strides doubled,
 *   candidate rows one reference row apart); results: SAD x 2, (x + origin) << mv_shift (x4, x2, x1). */
typedef struct svt_oracle_hme_params {
    int32_t search_area_width, search_area_height, x_origin_offset, y_origin_offset, pad_width, pad_height, ref_width,
        ref_height, round_down, mv_shift;
} svt_oracle_hme_params;

void svt_oracle_hme_params_for_level(int level, const uint16_t *hme_w, const uint16_t *hme_h, uint32_t region_w,
                                     uint32_t region_h, uint32_t total_w, uint32_t total_h, uint32_t mult_x,
                                     uint32_t mult_y, uint32_t ref_origin_x, uint32_t ref_origin_y, uint32_t ref_width,
                                     uint32_t ref_height, svt_oracle_hme_params *p) {
    memset(p, 0, sizeof(*p));
    p->ref_width = (int32_t)ref_width; p->ref_height = (int32_t)ref_height;
    if (level == 0) {
        p->search_area_width = (int16_t)((((hme_w[region_w] * mult_x) / 100) + 15) & ~0x0F);
        p->search_area_height = (int16_t)((hme_h[region_h] * mult_y) / 100);
        int xd = 0, yd = 0;
        for (uint32_t i = 0; i < region_w; i++) xd += (int16_t)((hme_w[i] * mult_x) / 100);
        for (uint32_t i = 0; i < region_h; i++) yd += (int16_t)((hme_h[i] * mult_y) / 100);
        p->x_origin_offset = -(int16_t)(((total_w * mult_x) / 100) >> 1) + xd;
        p->y_origin_offset = -(int16_t)(((total_h * mult_y) / 100) >> 1) + yd;
        p->pad_width = (int32_t)ref_origin_x - 1; p->pad_height = (int32_t)ref_origin_y - 1;
        p->round_down = 16; p->mv_shift = 2;
    } else {
        p->search_area_width = (int16_t)((hme_w[region_w] + 7) & ~0x07);
        p->search_area_height = (int16_t)hme_h[region_h];
        p->x_origin_offset = -(p->search_area_width >> 1);
        p->y_origin_offset = -(p->search_area_height >> 1);
        if (level == 1) { p->pad_width = (int32_t)ref_origin_x - 1; p->pad_height = (int32_t)ref_origin_y - 1; p->mv_shift = 1; }
        else { p->pad_width = 63; p->pad_height = 63; p->mv_shift = 0; }
        p->round_down = 8;
    }
}

/* src_pic / ref_pic point at sample (0, 0) of the level's source / reference picture (the reference may be read from
 * -pad .. size + search margins: the caller's buffer is padded as the encoder's pictures are). */
void svt_oracle_hme_level(const uint8_t *src_pic, uint32_t src_stride, const uint8_t *ref_pic, uint32_t ref_stride,
                          int origin_x, int origin_y, uint32_t sb_width, uint32_t sb_height, int x_center, int y_center,
                          const svt_oracle_hme_params *p, uint64_t *best_sad, int16_t *x_out, int16_t *y_out) {
    int saw = p->search_area_width, sah = p->search_area_height;
    int xo = p->x_origin_offset + x_center, yo = p->y_origin_offset + y_center;
    const int padw = p->pad_width, padh = p->pad_height, W = p->ref_width, H = p->ref_height;
    /* the reference's statements, in its order (:5745-5798).  NB: the width / height statement of the left / top clip tests the
     * ALREADY CORRECTED origin, so it never fires - the area keeps its size and slides; restated as written */
    xo = (origin_x + xo < -padw) ? -padw - origin_x : xo;
    saw = (origin_x + xo < -padw) ? saw - (-padw - (origin_x + xo)) : saw;
    xo = (origin_x + xo > W - 1) ? xo - ((origin_x + xo) - (W - 1)) : xo;
    if (origin_x + xo + saw > W) { const int v = saw - ((origin_x + xo + saw) - W); saw = v > 1 ? v : 1; }
    if (saw >= p->round_down) saw &= ~(p->round_down - 1);
    yo = (origin_y + yo < -padh) ? -padh - origin_y : yo;
    sah = (origin_y + yo < -padh) ? sah - (-padh - (origin_y + yo)) : sah;
    yo = (origin_y + yo > H - 1) ? yo - ((origin_y + yo) - (H - 1)) : yo;
    if (origin_y + yo + sah > H) { const int v = sah - ((origin_y + yo + sah) - H); sah = v > 1 ? v : 1; }
    const uint8_t *src = src_pic + (ptrdiff_t)origin_y * (ptrdiff_t)src_stride + origin_x;
    const uint8_t *ref = ref_pic + (ptrdiff_t)(origin_y + yo) * (ptrdiff_t)ref_stride + (origin_x + xo);
    int16_t bx = 0, by = 0;
    uint64_t best;
    svt_oracle_sad_loop(src, src_stride * 2, ref, ref_stride * 2, sb_height >> 1, sb_width, &best, &bx, &by, ref_stride, (int16_t)saw,
                        (int16_t)sah);
    *best_sad = best * 2;
    *x_out = (int16_t)((bx + xo) << p->mv_shift);
    *y_out = (int16_t)((by + yo) << p->mv_shift);
}
