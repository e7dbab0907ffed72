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
