/*
 * oracle/ref_me.c — OUR harness around the reference's own full-pel motion search of one 64x64 SB
 * (EbMotionEstimation.c); compiled into oracle/_ref/libsvtref.so (oracle/Makefile).  TEST INFRASTRUCTURE ONLY.
 *
 * FullPelSearch_LCU (:3199) and open_loop_me_fullpel_search_sblock (:3251) are `static` in the reference, so this
 * translation unit COMPILES THE REFERENCE SOURCE WHERE IT LIES (#include of the .c file, found through the
 * Makefile's -I path) and calls them as they are: the 8-search-point production kernels behind
 * GetEightHorizontalSearchPointResultsAll85PUs (:3064 -> EbComputeSAD_Intrinsic_AVX2.c:3554, 3696 or the SSE4.1
 * twins), the single-point remainder GetSearchPointResults (:2932), and the NSQ path
 * open_loop_me_get_eight_search_point_results_block (:2678) / ..._search_point_results_block (:2762) with
 * ext_all_sad_calculation_8x8_16x16 / ext_eight_sad_calculation_32x32_64x64 / ext_eigth_sad_calculation_nsq
 * (C at asm_type 0, AVX2 at 1).  Nothing of the reference is re-implemented here: the MeContext_t is the reference's
 * own struct, allocated zeroed, and only the fields the two functions read are filled, exactly as MotionEstimateLcu
 * does at :8040-8150.  The object is weakened like the other reference objects (its duplicate definitions of
 * EbMotionEstimation.c's global functions must not clash with Codec_EbMotionEstimation.o).
 */
#include "EbMotionEstimation.c"

/* best_sad / best_mv: uint32[MAX_ME_PU_COUNT = 209] in the reference's own p_sb_best_sad / p_sb_best_mv order
 * (EbMeTierZeroPu, EbMotionEstimationContext.h:47-270: 64x64, 32x32 x4, 16x16 x16, 8x8 x64, then the NSQ shapes).
 * init != 0: start from MAX_SAD_VALUE as the reference does (InitializeBuffer_32bits, :8131); else the arrays are
 * IN/OUT running bests.  ref_origin points at the top-left sample of the search area (search point 0, 0). */
int ref_me_fullpel(const uint8_t *src, uint32_t src_stride, const uint8_t *ref_origin, uint32_t ref_stride,
                   int x_search_area_origin, int y_search_area_origin, uint32_t search_area_width,
                   uint32_t search_area_height, int asm_type, int nsq, int init, uint32_t *best_sad,
                   uint32_t *best_mv) {
    MeContext_t *c = (MeContext_t *)calloc(1, sizeof(MeContext_t));
    uint16_t *eight = NULL;
    if (!c || posix_memalign((void **)&eight, 64, sizeof(uint16_t) * 8 * 16)) { free(c); return -1; }
    const uint32_t li = 0;
    c->sb_src_ptr = (uint8_t *)src;
    c->sb_src_stride = src_stride;
    /* :8048-8052: integer_buffer_ptr is the search region's top-left minus the interpolation margin */
    c->integer_buffer_ptr[li][0] = (uint8_t *)ref_origin - (ME_FILTER_TAP >> 1) - (ME_FILTER_TAP >> 1) * (ptrdiff_t)ref_stride;
    c->interpolated_full_stride[li][0] = ref_stride;
    c->p_eight_pos_sad16x16 = eight;
    uint32_t *S = c->p_sb_best_sad[li][0], *M = c->p_sb_best_mv[li][0];
    for (int i = 0; i < MAX_ME_PU_COUNT; i++) {
        S[i] = init ? (uint32_t)MAX_SAD_VALUE : best_sad[i];
        M[i] = init ? 0 : best_mv[i];
    }
    c->p_best_sad64x64 = &S[ME_TIER_ZERO_PU_64x64];   c->p_best_mv64x64 = &M[ME_TIER_ZERO_PU_64x64];
    c->p_best_sad32x32 = &S[ME_TIER_ZERO_PU_32x32_0]; c->p_best_mv32x32 = &M[ME_TIER_ZERO_PU_32x32_0];
    c->p_best_sad16x16 = &S[ME_TIER_ZERO_PU_16x16_0]; c->p_best_mv16x16 = &M[ME_TIER_ZERO_PU_16x16_0];
    c->p_best_sad8x8 = &S[ME_TIER_ZERO_PU_8x8_0];     c->p_best_mv8x8 = &M[ME_TIER_ZERO_PU_8x8_0];
    c->p_best_sad64x32 = &S[ME_TIER_ZERO_PU_64x32_0]; c->p_best_mv64x32 = &M[ME_TIER_ZERO_PU_64x32_0];
    c->p_best_sad32x16 = &S[ME_TIER_ZERO_PU_32x16_0]; c->p_best_mv32x16 = &M[ME_TIER_ZERO_PU_32x16_0];
    c->p_best_sad16x8 = &S[ME_TIER_ZERO_PU_16x8_0];   c->p_best_mv16x8 = &M[ME_TIER_ZERO_PU_16x8_0];
    c->p_best_sad32x64 = &S[ME_TIER_ZERO_PU_32x64_0]; c->p_best_mv32x64 = &M[ME_TIER_ZERO_PU_32x64_0];
    c->p_best_sad16x32 = &S[ME_TIER_ZERO_PU_16x32_0]; c->p_best_mv16x32 = &M[ME_TIER_ZERO_PU_16x32_0];
    c->p_best_sad8x16 = &S[ME_TIER_ZERO_PU_8x16_0];   c->p_best_mv8x16 = &M[ME_TIER_ZERO_PU_8x16_0];
    c->p_best_sad32x8 = &S[ME_TIER_ZERO_PU_32x8_0];   c->p_best_mv32x8 = &M[ME_TIER_ZERO_PU_32x8_0];
    c->p_best_sad8x32 = &S[ME_TIER_ZERO_PU_8x32_0];   c->p_best_mv8x32 = &M[ME_TIER_ZERO_PU_8x32_0];
    c->p_best_sad64x16 = &S[ME_TIER_ZERO_PU_64x16_0]; c->p_best_mv64x16 = &M[ME_TIER_ZERO_PU_64x16_0];
    c->p_best_sad16x64 = &S[ME_TIER_ZERO_PU_16x64_0]; c->p_best_mv16x64 = &M[ME_TIER_ZERO_PU_16x64_0];
    if (nsq)
        open_loop_me_fullpel_search_sblock(c, li, (int16_t)x_search_area_origin, (int16_t)y_search_area_origin,
                                           search_area_width, search_area_height, (EbAsm)asm_type);
    else
        FullPelSearch_LCU(c, li, (int16_t)x_search_area_origin, (int16_t)y_search_area_origin, search_area_width,
                          search_area_height, (EbAsm)asm_type);
    for (int i = 0; i < MAX_ME_PU_COUNT; i++) { best_sad[i] = S[i]; best_mv[i] = M[i]; }
    free(eight); free(c);
    return 0;
}

/* ---- hierarchical motion estimation levels 0 / 1 / 2: the reference's own HmeLevel0 / HmeLevel1 / HmeLevel2
 * (EbMotionEstimation.c:5689, 5883, 6016) on one SB.  The context's intermediate SB buffers are filled exactly as
 * MotionEstimationKernel does (EbMotionEstimationProcess.c:500-560): the 1/16 buffer holds every other row (stride 16),
 * the 1/4 buffer and the full-resolution SB buffer hold every row (strides 32 / 64) and are read with doubled strides.
 * src_pic points at sample (0, 0) of the level's SOURCE picture (origin_x / origin_y index it); ref_buffer is the level's
 * padded REFERENCE buffer_y with its origin / size.  hme_w / hme_h: the context's per-region search-area arrays of the
 * level (level 1: entry [region] too).  Outputs as the reference leaves them (SAD doubled, MV scaled to full resolution). */
int ref_hme_level(int level, const uint8_t *src_pic, uint32_t src_stride, uint8_t *ref_buffer, uint32_t ref_stride,
                  uint32_t ref_origin_x, uint32_t ref_origin_y, uint32_t ref_width, uint32_t ref_height, int origin_x,
                  int origin_y, uint32_t sb_width, uint32_t sb_height, int x_center, int y_center, const uint16_t *hme_w,
                  const uint16_t *hme_h, uint32_t region_w, uint32_t region_h, uint32_t total_w, uint32_t total_h,
                  uint32_t mult_x, uint32_t mult_y, int asm_type, uint64_t *best_sad, int16_t *x_out, int16_t *y_out) {
    MeContext_t *c = (MeContext_t *)calloc(1, sizeof(MeContext_t));
    EbPictureBufferDesc_t *pic = (EbPictureBufferDesc_t *)calloc(1, sizeof(EbPictureBufferDesc_t));
    uint8_t *sbuf = NULL;
    if (!c || !pic || posix_memalign((void **)&sbuf, 64, 64 * 64)) { free(c); free(pic); return -1; }
    memset(sbuf, 0, 64 * 64);
    pic->buffer_y = ref_buffer;
    pic->stride_y = (uint16_t)ref_stride;
    pic->origin_x = (uint16_t)ref_origin_x;
    pic->origin_y = (uint16_t)ref_origin_y;
    pic->width = (uint16_t)ref_width;
    pic->height = (uint16_t)ref_height;
    const uint8_t *blk = src_pic + (ptrdiff_t)origin_y * (ptrdiff_t)src_stride + origin_x;
    for (int i = 0; i < EB_HME_SEARCH_AREA_COLUMN_MAX_COUNT && i <= (int)region_w; i++) {
        c->hme_level0_search_area_in_width_array[i] = hme_w[i];
        c->hme_level1_search_area_in_width_array[i] = hme_w[i];
        c->hme_level2_search_area_in_width_array[i] = hme_w[i];
    }
    for (int i = 0; i < EB_HME_SEARCH_AREA_ROW_MAX_COUNT && i <= (int)region_h; i++) {
        c->hme_level0_search_area_in_height_array[i] = hme_h[i];
        c->hme_level1_search_area_in_height_array[i] = hme_h[i];
        c->hme_level2_search_area_in_height_array[i] = hme_h[i];
    }
    c->hme_level0_total_search_area_width = (uint16_t)total_w;
    c->hme_level0_total_search_area_height = (uint16_t)total_h;
    *best_sad = 0; *x_out = 0; *y_out = 0;
    if (level == 0) {
        c->sixteenth_sb_buffer = sbuf;
        c->sixteenth_sb_buffer_stride = 16;
        uint8_t *l = sbuf;
        const uint8_t *f = blk;
        for (uint32_t r = 0; r < sb_height; r += 2) { memcpy(l, f, sb_width); l += 16; f += (size_t)src_stride << 1; }
        HmeLevel0(NULL, c, (int16_t)origin_x, (int16_t)origin_y, sb_width, sb_height, (int16_t)x_center, (int16_t)y_center, pic, region_w,
                  region_h, best_sad, x_out, y_out, mult_x, mult_y, (EbAsm)asm_type);
    } else if (level == 1) {
        c->quarter_sb_buffer = sbuf;
        c->quarter_sb_buffer_stride = 32;
        for (uint32_t r = 0; r < sb_height; r++) memcpy(sbuf + 32 * r, blk + (size_t)r * src_stride, sb_width);
        HmeLevel1(c, (int16_t)origin_x, (int16_t)origin_y, sb_width, sb_height, pic, (int16_t)hme_w[region_w], (int16_t)hme_h[region_h],
                  (int16_t)x_center, (int16_t)y_center, best_sad, x_out, y_out, (EbAsm)asm_type);
    } else {
        c->sb_buffer = sbuf;
        c->sb_buffer_stride = 64;
        c->sb_src_ptr = sbuf;
        c->sb_src_stride = 64;
        for (uint32_t r = 0; r < sb_height; r++) memcpy(sbuf + 64 * r, blk + (size_t)r * src_stride, sb_width);
        HmeLevel2(NULL, c, (int16_t)origin_x, (int16_t)origin_y, sb_width, sb_height, pic, region_w, region_h, (int16_t)x_center,
                  (int16_t)y_center, best_sad, x_out, y_out, (EbAsm)asm_type);
    }
    free(sbuf); free(pic); free(c);
    return 0;
}

/* ---- the reference's WHOLE per-SB motion estimation: MotionEstimateLcu (EbMotionEstimation.c:7527) as it is ---------------
 * HME levels 0 / 1 / 2 over the search regions, the best-of-regions search centre (:7849-7941, incl. the same-POC
 * second-best pick of list 1), CheckZeroZeroCenter (:6844, called :7964), the round-up / clip / round-down-to-8 of the search
 * area (:7955-8040), the full-pel search (85 or 209 PUs), and - for B pictures - BiPredictionSearch (:6639) and the candidate
 * ordering into me_results (:8308-8420).  use_subpel_flag is 0: the half / quarter-pel refinement needs the interpolation
 * kernels that sit outside this path, so the vectors stay full-pel and BiPredAverging takes its integer branch.
 * The control-set objects are the reference's own structs, zero-allocated, with exactly the fields MotionEstimateLcu reads
 * filled as MotionEstimationKernel (EbMotionEstimationProcess.c:374-560) fills them.
 *
 * prm[]: 0 luma_width 1 luma_height 2 sb_origin_x 3 sb_origin_y 4 slice_type (B 0 / P 1) 5 pic_depth_mode 6 temporal_layer_index
 *        7 hierarchical_levels 8 enable_hme_flag 9 / 10 / 11 enable_hme_level0 / 1 / 2 12 is_used_as_reference_flag
 *        13 search_area_width 14 search_area_height 15 / 16 number_hme_search_region_in_width / height
 *        17 / 18 hme_level0_total_search_area_width / height 19 / 20 ref_pic_poc_array[0 / 1] 21 asm_type 22 input_resolution
 *        23 cu8x8_mode 24 fractionalSearchMethod 25 max_number_of_pus_per_sb 26 nsq_search_level
 *        27.. geometry of the three picture levels (full, 1/4, 1/16): stride, origin_x, origin_y, width, height (15 values)
 *        42.. hme_level{0,1,2}_search_area_in_width_array[2], ..._in_height_array[2] (12 values)
 * bufs[]: buffer_y of source full / quarter / sixteenth, list-0 reference full / quarter / sixteenth, list-1 reference ...
 * out: best_sad / best_mv [2][209], area_origin [2][2] (x, y per list), bipred_sad [209],
 *      results [209][11] = xMvL0 yMvL0 xMvL1 yMvL1 dist0 dir0 dist1 dir1 dist2 dir2 totalMeCandidateIndex */
#include "EbSequenceControlSet.h"
#include "EbReferenceObject.h"
int ref_motion_estimate_lcu(const int32_t *prm, uint8_t *const *bufs, uint32_t *best_sad, uint32_t *best_mv, int16_t *area_origin,
                            uint32_t *bipred_sad, int32_t *results) {
    PictureParentControlSet_t *pcs = (PictureParentControlSet_t *)calloc(1, sizeof(PictureParentControlSet_t));
    SequenceControlSet *scs = (SequenceControlSet *)calloc(1, sizeof(SequenceControlSet));
    EncodeContext_t *ectx = (EncodeContext_t *)calloc(1, sizeof(EncodeContext_t));
    MeContext_t *c = (MeContext_t *)calloc(1, sizeof(MeContext_t));
    EbObjectWrapper *w = (EbObjectWrapper *)calloc(3, sizeof(EbObjectWrapper));
    EbPaReferenceObject *ro = (EbPaReferenceObject *)calloc(2, sizeof(EbPaReferenceObject));
    EbPictureBufferDesc_t *pic = (EbPictureBufferDesc_t *)calloc(9, sizeof(EbPictureBufferDesc_t));
    MeCuResults_t *mer = (MeCuResults_t *)calloc(MAX_ME_PU_COUNT, sizeof(MeCuResults_t));
    MeCuResults_t *mer_rows[1] = {mer};
    uint16_t *eight = NULL;
    uint8_t *sb = NULL, *dummy = (uint8_t *)calloc(1, 1 << 16);
    if (!pcs || !scs || !ectx || !c || !w || !ro || !pic || !mer || !dummy || posix_memalign((void **)&eight, 64, sizeof(uint16_t) * 8 * 16) ||
        posix_memalign((void **)&sb, 64, 64 * 64 + 32 * 32 + 16 * 16))
        return -1;
    for (int k = 0; k < 9; k++) {                       /* k = 3 * picture (src, ref0, ref1) + level (full, 1/4, 1/16) */
        const int32_t *g = prm + 27 + 5 * (k % 3);
        pic[k].buffer_y = bufs[k];
        pic[k].stride_y = (uint16_t)g[0]; pic[k].origin_x = (uint16_t)g[1]; pic[k].origin_y = (uint16_t)g[2];
        pic[k].width = (uint16_t)g[3]; pic[k].height = (uint16_t)g[4];
    }
    for (int l = 0; l < 2; l++) {
        ro[l].input_padded_picture_ptr = &pic[3 * (l + 1)];
        ro[l].quarter_decimated_picture_ptr = &pic[3 * (l + 1) + 1];
        ro[l].sixteenth_decimated_picture_ptr = &pic[3 * (l + 1) + 2];
        w[l].object_ptr = &ro[l];
        pcs->ref_pa_pic_ptr_array[l] = &w[l];
        pcs->ref_pic_poc_array[l] = (uint64_t)prm[19 + l];
    }
    scs->luma_width = (uint16_t)prm[0]; scs->luma_height = (uint16_t)prm[1];
    scs->input_resolution = (uint8_t)prm[22];
    scs->encode_context_ptr = ectx;
    ectx->asm_type = (EbAsm)prm[21];
    w[2].object_ptr = scs;
    pcs->sequence_control_set_wrapper_ptr = &w[2];
    pcs->slice_type = (EB_SLICE)prm[4];
    pcs->pic_depth_mode = (uint8_t)prm[5];
    pcs->temporal_layer_index = (uint8_t)prm[6];
    pcs->hierarchical_levels = (uint8_t)prm[7];
    pcs->enable_hme_flag = (EbBool)prm[8];
    pcs->enable_hme_level0_flag = (EbBool)prm[9]; pcs->enable_hme_level1_flag = (EbBool)prm[10]; pcs->enable_hme_level2_flag = (EbBool)prm[11];
    pcs->is_used_as_reference_flag = (EbBool)prm[12];
    pcs->use_subpel_flag = 0;
    pcs->cu8x8_mode = (uint8_t)prm[23];
    pcs->max_number_of_pus_per_sb = (uint16_t)prm[25];
    pcs->nsq_search_level = (uint8_t)prm[26];
    pcs->me_results = mer_rows;
    c->search_area_width = (uint16_t)prm[13]; c->search_area_height = (uint16_t)prm[14];
    c->number_hme_search_region_in_width = (uint16_t)prm[15]; c->number_hme_search_region_in_height = (uint16_t)prm[16];
    c->hme_level0_total_search_area_width = (uint16_t)prm[17]; c->hme_level0_total_search_area_height = (uint16_t)prm[18];
    for (int i = 0; i < 2; i++) {
        c->hme_level0_search_area_in_width_array[i] = (uint16_t)prm[42 + i]; c->hme_level0_search_area_in_height_array[i] = (uint16_t)prm[44 + i];
        c->hme_level1_search_area_in_width_array[i] = (uint16_t)prm[46 + i]; c->hme_level1_search_area_in_height_array[i] = (uint16_t)prm[48 + i];
        c->hme_level2_search_area_in_width_array[i] = (uint16_t)prm[50 + i]; c->hme_level2_search_area_in_height_array[i] = (uint16_t)prm[52 + i];
    }
    c->update_hme_search_center_flag = 0;
    c->fractionalSearchMethod = (uint8_t)prm[24];
    c->p_eight_pos_sad16x16 = eight;
    c->interpolated_stride = 256;                        /* only enters addresses of the sub-pel planes, which full-pel vectors never read */
    for (int l = 0; l < 2; l++) { c->pos_b_buffer[l][0] = c->pos_h_buffer[l][0] = c->pos_j_buffer[l][0] = dummy + (1 << 15); }
    c->one_d_intermediate_results_buf0 = c->one_d_intermediate_results_buf1 = dummy;
    /* the SB buffers as MotionEstimationKernel loads them (EbMotionEstimationProcess.c:504-556) */
    const uint32_t ox = (uint32_t)prm[2], oy = (uint32_t)prm[3];
    const uint32_t sbw = (uint32_t)prm[0] - ox < 64 ? (uint32_t)prm[0] - ox : 64, sbh = (uint32_t)prm[1] - oy < 64 ? (uint32_t)prm[1] - oy : 64;
    c->sb_buffer = sb; c->sb_buffer_stride = 64;
    c->quarter_sb_buffer = sb + 64 * 64; c->quarter_sb_buffer_stride = 32;
    c->sixteenth_sb_buffer = sb + 64 * 64 + 32 * 32; c->sixteenth_sb_buffer_stride = 16;
    size_t bi = (size_t)(pic[0].origin_y + oy) * pic[0].stride_y + pic[0].origin_x + ox;
    for (int r = 0; r < 64; r++) memcpy(c->sb_buffer + 64 * r, pic[0].buffer_y + bi + (size_t)r * pic[0].stride_y, 64);
    c->sb_src_ptr = pic[0].buffer_y + bi;
    c->sb_src_stride = pic[0].stride_y;
    if (pcs->enable_hme_level1_flag) {
        bi = (size_t)(pic[1].origin_y + (oy >> 1)) * pic[1].stride_y + pic[1].origin_x + (ox >> 1);
        for (uint32_t r = 0; r < (sbh >> 1); r++) memcpy(c->quarter_sb_buffer + 32 * r, pic[1].buffer_y + bi + (size_t)r * pic[1].stride_y, sbw >> 1);
    }
    if (pcs->enable_hme_level0_flag) {
        bi = (size_t)(pic[2].origin_y + (oy >> 2)) * pic[2].stride_y + pic[2].origin_x + (ox >> 2);
        uint8_t *l = c->sixteenth_sb_buffer;
        const uint8_t *f = pic[2].buffer_y + bi;
        for (uint32_t r = 0; r < (sbh >> 2); r += 2) { memcpy(l, f, sbw >> 2); l += 16; f += (size_t)pic[2].stride_y << 1; }
    }
    EbPictureBufferDesc_t input = pic[0];                /* input_ptr: width / height are read for the SB size */
    input.width = (uint16_t)prm[0]; input.height = (uint16_t)prm[1];
    const EbErrorType rc = MotionEstimateLcu(pcs, 0, ox, oy, c, &input);
    for (int l = 0; l < 2; l++) {
        memcpy(best_sad + l * MAX_ME_PU_COUNT, c->p_sb_best_sad[l][0], sizeof(uint32_t) * MAX_ME_PU_COUNT);
        memcpy(best_mv + l * MAX_ME_PU_COUNT, c->p_sb_best_mv[l][0], sizeof(uint32_t) * MAX_ME_PU_COUNT);
        area_origin[2 * l] = c->x_search_area_origin[l][0]; area_origin[2 * l + 1] = c->y_search_area_origin[l][0];
    }
    memcpy(bipred_sad, c->p_sb_bipred_sad, sizeof(uint32_t) * MAX_ME_PU_COUNT);
    for (int p = 0; p < MAX_ME_PU_COUNT; p++) {
        int32_t *o = results + 11 * p;
        o[0] = mer[p].xMvL0; o[1] = mer[p].yMvL0; o[2] = mer[p].xMvL1; o[3] = mer[p].yMvL1;
        for (int k = 0; k < 3; k++) { o[4 + 2 * k] = (int32_t)mer[p].distortionDirection[k].distortion; o[5 + 2 * k] = (int32_t)mer[p].distortionDirection[k].direction; }
        o[10] = mer[p].totalMeCandidateIndex;
    }
    free(eight); free(sb); free(dummy); free(mer); free(pic); free(ro); free(w); free(c); free(ectx); free(scs); free(pcs);
    return rc == EB_ErrorNone ? 0 : -2;
}
