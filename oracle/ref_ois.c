/*
 * oracle/ref_ois.c — OUR harness around the reference's own open_loop_intra_search_sb
 * (EbMotionEstimation.c:8694); compiled into oracle/_ref/libsvtref.so with the reference sources
 * (oracle/Makefile).  TEST INFRASTRUCTURE ONLY.
 *
 * The function under test reads a handful of fields from the encoder's control sets.  They are
 * allocated here with the reference's OWN struct definitions (its headers) and only those fields are
 * filled in; the RTCD dispatch pointers are defined and initialised by the reference's own
 * aom_dsp_rtcd.h (RTCD_C section, exactly as EbEncHandle.c:117 does) and its own
 * init_intra_predictors_internal().  Nothing of the reference is re-implemented here.
 */
#define RTCD_C
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "EbDefinitions.h"
#include "aom_dsp_rtcd.h"
#include "EbPictureControlSet.h"
#include "EbSequenceControlSet.h"
#include "EbMotionEstimationProcess.h"
#include "EbMotionEstimationContext.h"
#include "EbPictureBufferDesc.h"
#include "EbCodingUnit.h"

void init_intra_predictors_internal(void);
EbErrorType open_loop_intra_search_sb(PictureParentControlSet_t *picture_control_set_ptr, uint32_t sb_index,
                                      MotionEstimationContext_t *context_ptr, EbPictureBufferDesc_t *input_ptr,
                                      EbAsm asm_type);

static int g_ready;

void ref_ois_setup(void) {
    if (!g_ready) {
        setup_rtcd_internal(ASM_AVX2);
        /* setup_rtcd_internal picks the NASM kernels of intrapred_sse2.asm for these slots; there is no
         * assembler in this image, so those slots are pointed at the reference's OWN scalar C kernels
         * instead (what its "to use C: flags = 0" switch selects, aom_dsp_rtcd.h:2393). */
/* volatile read: the compiler otherwise folds `slot == NULL` to false right after `slot = &function` */
#define USE_C(slot) do { if (!*(void *volatile *)&slot) slot = slot##_c; } while (0)
        USE_C(aom_dc_predictor_8x8); USE_C(aom_dc_predictor_16x16);
        USE_C(aom_dc_top_predictor_8x8); USE_C(aom_dc_top_predictor_16x16);
        USE_C(aom_dc_left_predictor_8x8); USE_C(aom_dc_left_predictor_16x16);
        USE_C(aom_dc_128_predictor_8x8); USE_C(aom_dc_128_predictor_16x16);
        USE_C(aom_v_predictor_8x8); USE_C(aom_v_predictor_16x16);
        USE_C(aom_h_predictor_8x8); USE_C(aom_h_predictor_16x16);
        USE_C(aom_paeth_predictor_8x8);
        USE_C(aom_dc_predictor_4x4); USE_C(aom_dc_top_predictor_4x4); USE_C(aom_dc_left_predictor_4x4);
        USE_C(aom_dc_128_predictor_4x4); USE_C(aom_v_predictor_4x4); USE_C(aom_h_predictor_4x4);
        USE_C(aom_highbd_dc_predictor_4x4); USE_C(aom_highbd_dc_predictor_8x8);
        USE_C(aom_highbd_v_predictor_4x4); USE_C(aom_highbd_v_predictor_8x8);
#undef USE_C
        init_intra_predictors_internal();
        g_ready = 1;
    }
}

/* one SB.  validity: CU_MAX_COUNT flags in RASTER order (SbParams_t.raster_scan_cu_validity).  Outputs per
 * MD-scan block index (85): count, best index, and per candidate mode / angle delta / distortion. */
int ref_ois_sb(uint8_t *buffer_y, uint32_t stride_y, uint32_t origin_x, uint32_t origin_y, uint32_t width, uint32_t height,
               uint32_t sb_origin_x, uint32_t sb_origin_y, const uint8_t *validity, int temporal_layer_index,
               int intra_pred_mode, int is_used_as_reference, uint8_t *out_count, int8_t *out_best,
               uint8_t *out_mode /*[85][61]*/, int8_t *out_delta /*[85][61]*/, uint32_t *out_dist /*[85][61]*/) {
    ref_ois_setup();
    SequenceControlSet *scs = calloc(1, sizeof(*scs));
    SbParams_t *sbp = calloc(1, sizeof(*sbp));
    EbObjectWrapper *wrap = calloc(1, sizeof(*wrap));
    PictureParentControlSet_t *pcs = calloc(1, sizeof(*pcs));
    ois_sb_results_t *res = calloc(1, sizeof(*res));
    ois_sb_results_t *res_arr[1] = {res};
    MotionEstimationContext_t *ctx = calloc(1, sizeof(*ctx));
    MeContext_t *me = calloc(1, sizeof(*me));
    EbPictureBufferDesc_t *pic = calloc(1, sizeof(*pic));
    if (!scs || !sbp || !wrap || !pcs || !res || !ctx || !me || !pic) return -1;
    scs->static_config.encoder_bit_depth = EB_8BIT;
    scs->sb_params_array = sbp;
    sbp->origin_x = (uint16_t)sb_origin_x;
    sbp->origin_y = (uint16_t)sb_origin_y;
    for (int i = 0; i < CU_MAX_COUNT; i++) sbp->raster_scan_cu_validity[i] = validity[i];
    wrap->object_ptr = scs;
    pcs->sequence_control_set_wrapper_ptr = wrap;
    pcs->ois_sb_results = res_arr;
    pcs->temporal_layer_index = (uint8_t)temporal_layer_index;
    pcs->intra_pred_mode = (uint8_t)intra_pred_mode;
    pcs->is_used_as_reference_flag = (EbBool)is_used_as_reference;
    for (int i = 0; i < CU_MAX_COUNT; i++) res->ois_candidate_array[i] = calloc(MAX_OIS_CANDIDATES, sizeof(ois_candidate_t));
    ctx->me_context_ptr = me;
    me->sb_buffer = calloc(64 * 64, 1);
    me->sb_buffer_stride = 64;
    pic->buffer_y = buffer_y;
    pic->stride_y = (uint16_t)stride_y;
    pic->origin_x = (uint16_t)origin_x;
    pic->origin_y = (uint16_t)origin_y;
    pic->width = (uint16_t)width;
    pic->height = (uint16_t)height;
    open_loop_intra_search_sb(pcs, 0, ctx, pic, ASM_AVX2);
    for (int b = 0; b < CU_MAX_COUNT; b++) {
        out_count[b] = res->total_ois_intra_candidate[b];
        out_best[b] = res->best_distortion_index[b];
        for (int c = 0; c < MAX_OIS_CANDIDATES; c++) {
            const ois_candidate_t *o = &res->ois_candidate_array[b][c];
            out_mode[b * MAX_OIS_CANDIDATES + c] = (uint8_t)o->intra_mode;
            out_delta[b * MAX_OIS_CANDIDATES + c] = (int8_t)o->angle_delta;
            out_dist[b * MAX_OIS_CANDIDATES + c] = o->distortion;
        }
        free(res->ois_candidate_array[b]);
    }
    free(me->sb_buffer); free(me); free(ctx); free(res); free(pcs); free(wrap); free(sbp); free(scs); free(pic);
    return 0;
}
