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
