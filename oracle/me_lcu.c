/*
 * oracle/me_lcu.c — CPU restatement of the reference's per-SB motion estimation driver, MotionEstimateLcu
 * (Source/Lib/Common/Codec/EbMotionEstimation.c:7527-8440), for full-pel vectors (use_subpel_flag = 0).
 * TEST INFRASTRUCTURE ONLY: nothing under oracle/ is part of the product; tests compare the HIP path with it.
 * Pinned to the reference's own MotionEstimateLcu (oracle/ref_me.c: ref_motion_estimate_lcu) by tests/golden/me_setup.npz and the
 * randomised comparison in tests/test_oracle_vs_ref.py.
 *
 * The stages, each citing what it follows:
 *   1  HME (:7672-7847)            levels 0 / 1 / 2 per search region, region centres chained level to level; a level that is
 *                                  switched off hands on the INITIAL centre (0, 0), not the previous level's (as written)
 *   2  search centre (:7849-7941)  first strict minimum of the last enabled level's SADs over the regions visited as
 *                                  [w][h] = [0][0], [1][0], [0][1], [1][1]; with level 2 on, list 1 and equal reference POCs the
 *                                  regions are sorted by SAD and the SECOND one is taken (:7906-7936)
 *   3  CheckZeroZeroCenter (:6844) the HME centre, clipped into the picture, survives only if its sub-sampled SB SAD is
 *                                  strictly below the (0, 0) SAD (hmeMvdRate is 0 in this snapshot and the rounding term
 *                                  MD_OFFSET >> MD_SHIFT is 0, so the two costs are the SADs << COST_PRECISION)
 *   4  search area (:7955-8040)    width rounded up to 8, centred, clipped left / right / top / bottom in the reference's
 *                                  statement order, width rounded down to 8 unless below 8
 *   5  full-pel search (:8064-8165) 209 PUs when pic_depth_mode <= PIC_ALL_C_DEPTH_MODE, else 85  (oracle/pixel.c)
 *   6  bi-prediction (:6639, :6457) per PU, the SAD of the source against the rounded average of the two lists' best blocks,
 *                                  on every other row and doubled when fractionalSearchMethod == SUB_SAD_SEARCH
 *   7  candidate order (:8330-8440) me_results: distortions sorted as Sort3Elements (:6809) / the two-candidate rule do
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "svt_oracle.h"

/* oracle/pixel.c */
typedef struct svt_oracle_hme_params {
    int32_t search_area_width, search_area_height, x_origin_offset, y_origin_offset, pad_width, pad_height, ref_width,
        ref_height, round_down, mv_shift;
} svt_oracle_hme_params;
void svt_oracle_hme_params_for_level(int level, const uint16_t *hme_w, const uint16_t *hme_h, uint32_t region_w,
                                     uint32_t region_h, uint32_t total_w, uint32_t total_h, uint32_t mult_x,
                                     uint32_t mult_y, uint32_t ref_origin_x, uint32_t ref_origin_y, uint32_t ref_width,
                                     uint32_t ref_height, svt_oracle_hme_params *p);
void svt_oracle_hme_level(const uint8_t *src_pic, uint32_t src_stride, const uint8_t *ref_pic, uint32_t ref_stride,
                          int origin_x, int origin_y, uint32_t sb_width, uint32_t sb_height, int x_center, int y_center,
                          const svt_oracle_hme_params *p, uint64_t *best_sad, int16_t *x_out, int16_t *y_out);

/* HME_LEVEL_0_SEARCH_AREA_MULTIPLIER_X / _Y [hierarchical_levels][temporal_layer_index] (EbDefinitions.h:3019-3035; the two
 * tables hold the same numbers): percent by which level 0 widens its area in the lower temporal layers */
static int hme_level0_multiplier(int hierarchical_levels, int temporal_layer) {
    static const int16_t top3[3][4] = {{200, 140, 100, 70}, {350, 200, 100, 100}, {525, 350, 200, 100}};
    if (hierarchical_levels < 3 || temporal_layer > hierarchical_levels) return 100;
    return temporal_layer < 4 ? top3[hierarchical_levels - 3][temporal_layer] : 100;
}

/* geometry of PU `n` of the reference's result rows (EbMeTierZeroPu storage order; 16x16 / 8x8 and the shapes built from them
 * sit in z-order there, oracle/pixel.c) and of PU `p` in RASTER order (partitionWidth / partitionHeight / puSearchIndexMap,
 * EbMotionEstimation.h:178-315: the order of me_results and of BiPredictionSearch's pu_index) */
static const uint8_t kGroupW[14] = {64, 32, 16, 8, 64, 32, 16, 32, 16, 8, 32, 8, 64, 16};
static const uint8_t kGroupH[14] = {64, 32, 16, 8, 32, 16, 8, 64, 32, 16, 8, 32, 16, 64};
static void raster_pu_rect(int p, int *x, int *y, int *w, int *h, int *group_base) {
    int base = 0;
    for (int g = 0; g < 14; g++) {
        const int cols = 64 / kGroupW[g], n = cols * (64 / kGroupH[g]);
        if (p < base + n) {
            const int i = p - base;
            *w = kGroupW[g]; *h = kGroupH[g]; *x = (i % cols) * kGroupW[g]; *y = (i / cols) * kGroupH[g]; *group_base = base;
            return;
        }
        base += n;
    }
    *x = *y = *w = *h = 0; *group_base = 0;
}
/* storage index of the PU with that rectangle: the SAD rows keep every group in the order its sums are formed from the
 * z-ordered 8x8 / 16x16 SADs (me_nsq_update in oracle/pixel.c); found by forming the index the same way */
static int z4(int bx, int by) { return ((by >> 1) * 2 + (bx >> 1)) * 4 + (by & 1) * 2 + (bx & 1); }      /* 4x4 grid of 16x16 */
static int storage_index(int x, int y, int w, int h, int base) {
    if (w == 64 && h == 64) return 0;
    if (w == 32 && h == 32) return base + (y >> 5) * 2 + (x >> 5);
    if (w == 16 && h == 16) return base + z4(x >> 4, y >> 4);
    if (w == 8 && h == 8) return base + 4 * z4(x >> 4, y >> 4) + ((y >> 3) & 1) * 2 + ((x >> 3) & 1);
    if (w == 64 && h == 32) return base + (y >> 5);
    if (w == 32 && h == 16) return base + z4(x >> 4, y >> 4) / 2;                        /* s16[2i] + s16[2i + 1] */
    if (w == 16 && h == 8) return base + (4 * z4(x >> 4, y >> 4) + ((y >> 3) & 1) * 2) / 2;   /* s8[2i] + s8[2i + 1] */
    if (w == 32 && h == 64) return base + (x >> 5);
    if (w == 16 && h == 32) { const int b = z4(x >> 4, y >> 4); return base + (b >> 2) * 2 + (b & 1); }        /* b = (i >> 1) * 4 + (i & 1) */
    if (w == 8 && h == 16) { const int b = 4 * z4(x >> 4, y >> 4) + ((x >> 3) & 1); return base + (b >> 2) * 2 + (b & 1); }
    if (w == 32 && h == 8) {                                                             /* s16x8[b] + s16x8[b + 2], b = (i >> 1) * 4 + (i & 1) */
        const int b = (4 * z4(x >> 4, y >> 4) + ((y >> 3) & 1) * 2) / 2;
        return base + (b >> 2) * 2 + (b & 1);
    }
    if (w == 8 && h == 32) {                                                             /* s8x16[b] + s8x16[b + 4], b = (i >> 2) * 8 + (i & 3) */
        const int b8 = 4 * z4(x >> 4, y >> 4) + ((x >> 3) & 1), b = (b8 >> 2) * 2 + (b8 & 1);
        return base + (b >> 3) * 4 + (b & 3);
    }
    if (w == 64 && h == 16) { const int b = z4(0, y >> 4) / 2; return base + (b >> 2) * 2 + (b & 1); }         /* s32x16[b] + s32x16[b + 2] */
    if (w == 16 && h == 64) { const int b16 = z4(x >> 4, 0), b = (b16 >> 2) * 2 + (b16 & 1); return base + b; }  /* s16x32[i] + s16x32[i + 4] */
    return -1;
}
int svt_oracle_me_raster_to_storage(int pu_index) {
    int x, y, w, h, base;
    raster_pu_rect(pu_index, &x, &y, &w, &h, &base);
    return storage_index(x, y, w, h, base);
}

/* stages 2 - 4 for one SB and one reference list.  hme_sad / hme_x / hme_y: [w][h] region arrays of the LAST enabled level
 * (ignored when !hme_used).  src00 / ref00: sample (0, 0) of the padded source / reference planes. */
void svt_oracle_me_setup(const uint8_t *src00, uint32_t src_stride, const uint8_t *ref00, uint32_t ref_stride, int sb_x, int sb_y,
                         int sb_w, int sb_h, int pic_w, int pic_h, int ref_w, int ref_h, int hme_used, const uint64_t hme_sad[2][2],
                         const int16_t hme_x[2][2], const int16_t hme_y[2][2], int regions_w, int regions_h, int second_best,
                         int zz_check, int search_area_width, int search_area_height, int16_t center_out[2], int16_t area_out[4]) {
    int xc = 0, yc = 0;
    if (hme_used) {
        uint64_t s[2][2];
        int16_t cx[2][2], cy[2][2];
        memcpy(s, hme_sad, sizeof(s)); memcpy(cx, hme_x, sizeof(cx)); memcpy(cy, hme_y, sizeof(cy));
        uint64_t best = s[0][0];
        xc = cx[0][0]; yc = cy[0][0];
        for (int rh = 0; rh < regions_h; rh++)
            for (int rw = (rh == 0 ? 1 : 0); rw < regions_w; rw++)
                if (s[rw][rh] < best) { best = s[rw][rh]; xc = cx[rw][rh]; yc = cy[rw][rh]; }
        const int total = regions_w * regions_h;
        if (second_best && total > 1) {
            /* the reference indexes its [width][height] arrays as [q / regions_w][q % regions_w] here (:7912-7930) */
            for (int q = 0; q < total - 1; q++)
                for (int n = q + 1; n < total; n++) {
                    const int qa = q / regions_w, qb = q % regions_w, na = n / regions_w, nb = n % regions_w;
                    if (s[qa][qb] > s[na][nb]) {
                        const uint64_t ts = s[qa][qb]; const int16_t tx = cx[qa][qb], ty = cy[qa][qb];
                        s[qa][qb] = s[na][nb]; cx[qa][qb] = cx[na][nb]; cy[qa][qb] = cy[na][nb];
                        s[na][nb] = ts; cx[na][nb] = tx; cy[na][nb] = ty;
                    }
                }
            xc = cx[0][1]; yc = cy[0][1];
        }
    }
    if ((xc != 0 || yc != 0) && zz_check) {
        const int pad = 63;
        const uint8_t *s = src00 + (ptrdiff_t)sb_y * (ptrdiff_t)src_stride + sb_x;
        const uint32_t zero = svt_oracle_sad(s, src_stride * 2, ref00 + (ptrdiff_t)sb_y * (ptrdiff_t)ref_stride + sb_x, ref_stride * 2,
                                             (uint32_t)sb_h >> 1, (uint32_t)sb_w) << 1;
        if (sb_x + xc < -pad) xc = -pad - sb_x;
        if (sb_x + xc > ref_w - 1) xc -= (sb_x + xc) - (ref_w - 1);
        if (sb_y + yc < -pad) yc = -pad - sb_y;
        if (sb_y + yc > ref_h - 1) yc -= (sb_y + yc) - (ref_h - 1);
        const uint32_t hme = svt_oracle_sad(s, src_stride * 2, ref00 + (ptrdiff_t)(sb_y + yc) * (ptrdiff_t)ref_stride + sb_x + xc,
                                            ref_stride * 2, (uint32_t)sb_h >> 1, (uint32_t)sb_w) << 1;
        if (zero <= hme) { xc = 0; yc = 0; }               /* MIN(zero cost, hme cost) == zero cost */
    }
    center_out[0] = (int16_t)xc; center_out[1] = (int16_t)yc;
    const int pad = 63;
    int saw = (search_area_width + 7) & ~7, sah = search_area_height;
    int xo = xc - (saw >> 1), yo = yc - (sah >> 1);
    if (sb_x + xo < -pad) xo = -pad - sb_x;
    if (sb_x + xo < -pad) saw -= -pad - (sb_x + xo);       /* tests the corrected origin: never true (as in the reference) */
    if (sb_x + xo > pic_w - 1) xo -= (sb_x + xo) - (pic_w - 1);
    if (sb_x + xo + saw > pic_w) { const int v = saw - ((sb_x + xo + saw) - pic_w); saw = v > 1 ? v : 1; }
    if (saw >= 8) saw &= ~7;
    if (sb_y + yo < -pad) yo = -pad - sb_y;
    if (sb_y + yo < -pad) sah -= -pad - (sb_y + yo);
    if (sb_y + yo > pic_h - 1) yo -= (sb_y + yo) - (pic_h - 1);
    if (sb_y + yo + sah > pic_h) { const int v = sah - ((sb_y + yo + sah) - pic_h); sah = v > 1 ? v : 1; }
    area_out[0] = (int16_t)xo; area_out[1] = (int16_t)yo; area_out[2] = (int16_t)saw; area_out[3] = (int16_t)sah;
}

/* stages 6 - 7 for one SB.  best_sad / best_mv: [2][209] rows of the two lists in storage order; nlists 1 (P) or 2 (B).
 * bipred_sad[209] (storage order), results[209][11] (raster PU order) as ref_motion_estimate_lcu lays them out. */
void svt_oracle_me_bipred_results(const uint8_t *src00, uint32_t src_stride, const uint8_t *ref0, uint32_t ref0_stride, const uint8_t *ref1,
                                  uint32_t ref1_stride, int sb_x, int sb_y, const uint32_t *best_sad, const uint32_t *best_mv, int nlists,
                                  int npus, int bipred_all_pus, int sub_sad, uint32_t *bipred_sad, int32_t *results) {
    for (int p = 0; p < npus; p++) {
        int x, y, w, h, base;
        raster_pu_rect(p, &x, &y, &w, &h, &base);
        const int n = storage_index(x, y, w, h, base);
        int32_t *o = results + 11 * p;
        memset(o, 0, 11 * sizeof(int32_t));
        int total = nlists;
        if (nlists == 2 && (bipred_all_pus || p < 21)) {
            const uint32_t mv0 = best_mv[n], mv1 = best_mv[209 + n];
            const int x0 = (int16_t)(mv0 & 0xffff) >> 2, y0 = (int16_t)(mv0 >> 16) >> 2, x1 = (int16_t)(mv1 & 0xffff) >> 2, y1 = (int16_t)(mv1 >> 16) >> 2;
            const uint8_t *s = src00 + (ptrdiff_t)(sb_y + y) * (ptrdiff_t)src_stride + sb_x + x;
            const uint8_t *a = ref0 + (ptrdiff_t)(sb_y + y + y0) * (ptrdiff_t)ref0_stride + sb_x + x + x0;
            const uint8_t *b = ref1 + (ptrdiff_t)(sb_y + y + y1) * (ptrdiff_t)ref1_stride + sb_x + x + x1;
            bipred_sad[n] = sub_sad ? svt_oracle_sad_avg(s, src_stride * 2, a, ref0_stride * 2, b, ref1_stride * 2, (uint32_t)h >> 1, (uint32_t)w) << 1
                                    : svt_oracle_sad_avg(s, src_stride, a, ref0_stride, b, ref1_stride, (uint32_t)h, (uint32_t)w);
            total = 3;
        }
        const uint32_t l0 = best_sad[n], l1 = nlists == 2 ? best_sad[209 + n] : 0, bi = bipred_sad[n];
        o[0] = (int16_t)(best_mv[n] & 0xffff); o[1] = (int16_t)(best_mv[n] >> 16);
        o[2] = nlists == 2 ? (int16_t)(best_mv[209 + n] & 0xffff) : 0; o[3] = nlists == 2 ? (int16_t)(best_mv[209 + n] >> 16) : 0;
        o[10] = total;
        uint32_t d[3] = {l0, l1, bi};
        int dir[3] = {0, 1, 2}, cnt = total;               /* UNI_PRED_LIST_0, UNI_PRED_LIST_1, BI_PRED */
        /* ascending, a candidate moves ahead of another only when strictly smaller ... except that Sort3Elements' last two
         * branches (:6829-6834) put the bi-prediction first whenever neither list is the smallest-or-equal of the three */
        if (cnt == 3) {
            int ord[3];
            if (l0 <= l1 && l0 <= bi) { ord[0] = 0; ord[1] = l1 <= bi ? 1 : 2; ord[2] = l1 <= bi ? 2 : 1; }
            else if (l1 <= l0 && l1 <= bi) { ord[0] = 1; ord[1] = l0 <= bi ? 0 : 2; ord[2] = l0 <= bi ? 2 : 0; }
            else if (l0 <= l1) { ord[0] = 2; ord[1] = 0; ord[2] = 1; }
            else { ord[0] = 2; ord[1] = 1; ord[2] = 0; }
            for (int k = 0; k < 3; k++) { o[4 + 2 * k] = (int32_t)d[ord[k]]; o[5 + 2 * k] = dir[ord[k]]; }
        } else if (cnt == 2) {
            const int first = l0 <= l1 ? 0 : 1;
            o[4] = (int32_t)d[first]; o[5] = first; o[6] = (int32_t)d[1 - first]; o[7] = 1 - first;
        } else {
            o[4] = (int32_t)l0; o[5] = 0;
        }
    }
}

/* The whole driver for one SB, with the parameter block and the nine picture planes of ref_motion_estimate_lcu
 * (oracle/ref_me.c documents the slots); flavour = asm_type. */
int svt_oracle_me_lcu_ex(const int32_t *prm, uint8_t *const *bufs, uint32_t *best_sad, uint32_t *best_mv, int16_t *area_origin,
                         uint32_t *bipred_sad, int32_t *results, int16_t *centers_out /* [2][2] or NULL */, int16_t *areas_out /* [2][4] or NULL */,
                         uint64_t *hme_sad_out /* [2 lists][3 levels][2][2] or NULL */, int16_t *hme_mv_out /* [2][3][2][2][2] (x, y) or NULL */) {
    const int pic_w = prm[0], pic_h = prm[1], sb_x = prm[2], sb_y = prm[3];
    const int nlists = prm[4] == 1 ? 1 : 2;                 /* P_SLICE 1: list 0 only */
    const int nsq = prm[5] <= 1, tl = prm[6], hl = prm[7];
    const int hme_on = prm[8], l0 = prm[9], l1 = prm[10], l2 = prm[11], is_ref = prm[12];
    const int regions_w = prm[15], regions_h = prm[16], flavour = prm[21], npus = prm[25];
    const int sb_w = pic_w - sb_x < 64 ? pic_w - sb_x : 64, sb_h = pic_h - sb_y < 64 ? pic_h - sb_y : 64;
    const int32_t *geo = prm + 27;                          /* [level][stride, origin_x, origin_y, width, height] */
    const uint8_t *p00[9];
    for (int k = 0; k < 9; k++) { const int32_t *g = geo + 5 * (k % 3); p00[k] = bufs[k] + (size_t)g[2] * (size_t)g[0] + g[1]; }
    uint16_t hw[3][2], hh[3][2];
    for (int lv = 0; lv < 3; lv++)
        for (int i = 0; i < 2; i++) { hw[lv][i] = (uint16_t)prm[42 + 4 * lv + i]; hh[lv][i] = (uint16_t)prm[44 + 4 * lv + i]; }
    memset(best_sad, 0, 2 * 209 * sizeof(uint32_t)); memset(best_mv, 0, 2 * 209 * sizeof(uint32_t));
    memset(area_origin, 0, 4 * sizeof(int16_t)); memset(bipred_sad, 0, 209 * sizeof(uint32_t));
    const int mult = hme_level0_multiplier(hl, tl);
    for (int list = 0; list < nlists; list++) {
        /* list 1 of a base-layer picture whose two references are the same picture gets no HME: its search is centred on (0, 0)
         * (BASE_LAYER_REF, :7655-7660 with the else at :7948) */
        const int hme_list = tl > 0 || list == 0 || (prm[19] != prm[20] && list == 1);
        const uint8_t *const *rp = p00 + 3 * (list + 1);
        uint64_t sad[3][2][2];
        int16_t cx[3][2][2], cy[3][2][2];
        memset(sad, 0, sizeof(sad)); memset(cx, 0, sizeof(cx)); memset(cy, 0, sizeof(cy));
        const int hme_used = hme_list && hme_on && sb_h == 64;          /* "no HME in boundaries" (:7678) */
        int last = -1;
        if (hme_used) {
            for (int lv = 0; lv < 3; lv++) {
                if (!(lv == 0 ? l0 : (lv == 1 ? l1 : l2))) continue;
                last = lv;
                const int sh = 2 - lv;
                const int32_t *g = geo + 5 * (2 - lv);      /* level 0 searches the 1/16 picture */
                for (int rh = 0; rh < regions_h; rh++)
                    for (int rw = 0; rw < regions_w; rw++) {
                        svt_oracle_hme_params hp;
                        svt_oracle_hme_params_for_level(lv, hw[lv], hh[lv], (uint32_t)rw, (uint32_t)rh, (uint32_t)prm[17], (uint32_t)prm[18],
                                                        (uint32_t)mult, (uint32_t)mult, (uint32_t)g[1], (uint32_t)g[2], (uint32_t)g[3], (uint32_t)g[4], &hp);
                        int xin = 0, yin = 0;                /* level 0 starts from (0, 0); level 1 takes level 0's vector >> 1 */
                        if (lv == 1) { xin = cx[0][rw][rh] >> 1; yin = cy[0][rw][rh] >> 1; }
                        if (lv == 2) { xin = cx[1][rw][rh]; yin = cy[1][rw][rh]; }
                        svt_oracle_hme_level(p00[2 - lv], (uint32_t)g[0], rp[2 - lv], (uint32_t)g[0], sb_x >> sh, sb_y >> sh, (uint32_t)sb_w >> sh,
                                             (uint32_t)sb_h >> sh, xin, yin, &hp, &sad[lv][rw][rh], &cx[lv][rw][rh], &cy[lv][rw][rh]);
                    }
            }
        }
        int16_t center[2], area[4];
        const int second = last == 2 && prm[19] == prm[20] && list == 1;
        svt_oracle_me_setup(p00[0], (uint32_t)geo[0], rp[0], (uint32_t)geo[0], sb_x, sb_y, sb_w, sb_h, pic_w, pic_h, geo[3], geo[4],
                            hme_used && last >= 0, last >= 0 ? sad[last] : sad[0], last >= 0 ? cx[last] : cx[0], last >= 0 ? cy[last] : cy[0],
                            regions_w, regions_h, second, is_ref, prm[13], prm[14], center, area);
        area_origin[2 * list] = area[0]; area_origin[2 * list + 1] = area[1];
        if (centers_out) memcpy(centers_out + 2 * list, center, sizeof(center));
        if (areas_out) memcpy(areas_out + 4 * list, area, sizeof(area));
        if (hme_sad_out) memcpy(hme_sad_out + 12 * list, sad, sizeof(sad));
        if (hme_mv_out)
            for (int lv = 0; lv < 3; lv++)
                for (int a = 0; a < 2; a++)
                    for (int b = 0; b < 2; b++) {
                        hme_mv_out[((12 * list + 4 * lv + 2 * a + b) * 2)] = cx[lv][a][b];
                        hme_mv_out[((12 * list + 4 * lv + 2 * a + b) * 2) + 1] = cy[lv][a][b];
                    }
        uint32_t *bs = best_sad + 209 * list, *bm = best_mv + 209 * list;
        for (int i = 0; i < (nsq ? 209 : 85); i++) bs[i] = 128 * 128 * 255;      /* MAX_SAD_VALUE */
        svt_oracle_me_sb_search_full(p00[0] + (ptrdiff_t)sb_y * geo[0] + sb_x, (uint32_t)geo[0],
                                     rp[0] + (ptrdiff_t)(sb_y + area[1]) * geo[0] + sb_x + area[0], (uint32_t)geo[0], area[2], area[3], area[0],
                                     area[1], flavour, nsq, bs, bm);
    }
    svt_oracle_me_bipred_results(p00[0], (uint32_t)geo[0], p00[3], (uint32_t)geo[0], p00[6], (uint32_t)geo[0], sb_x, sb_y, best_sad, best_mv, nlists,
                                 npus, prm[23] == 0 || nsq, prm[24] == 0, bipred_sad, results);
    return 0;
}

int svt_oracle_me_lcu(const int32_t *prm, uint8_t *const *bufs, uint32_t *best_sad, uint32_t *best_mv, int16_t *area_origin,
                      uint32_t *bipred_sad, int32_t *results) {
    return svt_oracle_me_lcu_ex(prm, bufs, best_sad, best_mv, area_origin, bipred_sad, results, NULL, NULL, NULL, NULL);
}
