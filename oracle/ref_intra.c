/*
 * oracle/ref_intra.c — OUR harness around the reference's own build_intra_predictors / build_intra_predictors_high
 * (EbIntraPrediction.c:3667, 3857: the neighbour-availability glue of av1_predict_intra_block, :4078); compiled into
 * oracle/_ref/libsvtref.so (oracle/Makefile).  TEST INFRASTRUCTURE ONLY.
 *
 * Both functions are `static`, so this translation unit compiles the reference source where it lies (#include of the .c
 * file through the Makefile's -I path).  The file-static predictor tables pred[][] / dc_pred[][][] they dispatch through are
 * per translation unit: this unit's copy is filled by ITS OWN init_intra_predictors_internal(), renamed below so that the
 * call cannot bind to the twin in Codec_EbIntraPrediction.o.  The RTCD pointers (av1_filter_intra_edge, av1_dr_prediction_z*,
 * the per-size predictors, ...) are the ones ref_ois.c defines and fills with the reference's setup_rtcd_internal.
 * get_filt_type() reads the neighbours' modes from MacroBlockD: an above neighbour with mode SMOOTH_PRED gives type 1.
 */
#define init_intra_predictors_internal ref_intra_unit_init_predictors
#define init_intra_dc_predictors_c_internal ref_intra_unit_init_dc_predictors
#include "EbIntraPrediction.c"

void ref_ois_setup(void);
static int g_intra_ready;

/* top / left point at element 0 of the reference's topNeighArray + 1 / leftNeighArray + 1 (element -1 = the corner sample) */
int ref_build_intra_predictors(int is16, void *top, void *left, void *dst, int32_t dst_stride, int mode, int angle_delta,
                               int tx_size, int disable_edge_filter, int n_top_px, int n_topright_px, int n_left_px,
                               int n_bottomleft_px, int filt_type, int bd) {
    ref_ois_setup();
    if (!g_intra_ready) {
        ref_intra_unit_init_dc_predictors();
        ref_intra_unit_init_predictors();
        g_intra_ready = 1;
    }
    MacroBlockD xd;
    MbModeInfo ab;
    memset(&xd, 0, sizeof(xd));
    memset(&ab, 0, sizeof(ab));
    ab.mode = SMOOTH_PRED;
    xd.above_mbmi = filt_type ? &ab : NULL;
    xd.left_mbmi = NULL;
    if (is16)
        build_intra_predictors_high(&xd, (uint16_t *)top, (uint16_t *)left, (uint16_t *)dst, dst_stride, (PredictionMode)mode, angle_delta,
                                    FILTER_INTRA_MODES, (TxSize)tx_size, disable_edge_filter, n_top_px, n_topright_px, n_left_px,
                                    n_bottomleft_px, 0, bd);
    else
        build_intra_predictors(&xd, (uint8_t *)top, (uint8_t *)left, (uint8_t *)dst, dst_stride, (PredictionMode)mode, angle_delta,
                               FILTER_INTRA_MODES, (TxSize)tx_size, disable_edge_filter, n_top_px, n_topright_px, n_left_px,
                               n_bottomleft_px, 0);
    return 0;
}

/* ---- has_top_right / has_bottom_left (:1567, :1755) as they are; sb_bsize = the sequence's sb_size (BLOCK_64X64 / BLOCK_128X128) */
static void ref_intra_fake_cm(Av1Common *cm, PictureParentControlSet_t *ppcs, SequenceControlSet *scs, int sb_bsize) {
    memset(cm, 0, sizeof(*cm)); memset(ppcs, 0, sizeof(*ppcs)); memset(scs, 0, sizeof(*scs));
    scs->sb_size = (block_size)sb_bsize;
    ppcs->sequence_control_set_ptr = scs;
    cm->p_pcs_ptr = ppcs;
}
int ref_has_top_right(int sb_bsize, int bsize, int mi_row, int mi_col, int top_available, int right_available, int partition,
                      int txsz, int row_off, int col_off, int ss_x, int ss_y) {
    Av1Common cm; PictureParentControlSet_t *ppcs = malloc(sizeof(*ppcs)); SequenceControlSet *scs = malloc(sizeof(*scs));
    ref_intra_fake_cm(&cm, ppcs, scs, sb_bsize);
    const int r = has_top_right(&cm, (block_size)bsize, mi_row, mi_col, top_available, right_available, (PartitionType)partition,
                                (TxSize)txsz, row_off, col_off, ss_x, ss_y);
    free(ppcs); free(scs);
    return r;
}
int ref_has_bottom_left(int sb_bsize, int bsize, int mi_row, int mi_col, int bottom_available, int left_available, int partition,
                        int txsz, int row_off, int col_off, int ss_x, int ss_y) {
    Av1Common cm; PictureParentControlSet_t *ppcs = malloc(sizeof(*ppcs)); SequenceControlSet *scs = malloc(sizeof(*scs));
    ref_intra_fake_cm(&cm, ppcs, scs, sb_bsize);
    const int r = has_bottom_left(&cm, (block_size)bsize, mi_row, mi_col, bottom_available, left_available, (PartitionType)partition,
                                  (TxSize)txsz, row_off, col_off, ss_x, ss_y);
    free(ppcs); free(scs);
    return r;
}

/* ---- av1_predict_intra_block (:4078) / av1_predict_intra_block_16bit (:4336) as they are, on a picture described by its mode-info
 * grid.  mi_mode / mi_uv_mode: [mi_rows * mi_cols] prediction modes of the already coded blocks (every block intra: ref_frame[0] =
 * INTRA_FRAME).  tile: {mi_row_start, mi_row_end, mi_col_start, mi_col_end} (the 8-bit function ignores it except for the chroma
 * sub-8x8 availability).  top_neigh / left_neigh: the caller's topNeighArray + 1 / leftNeighArray + 1 (EbCodingLoop.c:2902).
 * recon: plane buffer the prediction is written into (stride / origin of THAT plane). */
int ref_predict_intra_block(int is16, int sb_bsize, int mi_rows, int mi_cols, const uint8_t *mi_mode, const uint8_t *mi_uv_mode,
                            const int32_t *tile4, int shape, int bsize, int tx_size, int mode, int angle_delta, int plane,
                            int bl_org_x_pict, int bl_org_y_pict, int col_off, int row_off, int wpx, int hpx, void *top_neigh,
                            void *left_neigh, void *recon, int recon_stride, int recon_origin_x, int recon_origin_y) {
    ref_ois_setup();
    if (!g_intra_ready) {
        ref_intra_unit_init_dc_predictors();
        ref_intra_unit_init_predictors();
        g_intra_ready = 1;
    }
    Av1Common cm;
    PictureParentControlSet_t *ppcs = malloc(sizeof(*ppcs));
    SequenceControlSet *scs = malloc(sizeof(*scs));
    PictureControlSet_t *pcs = calloc(1, sizeof(*pcs));
    ref_intra_fake_cm(&cm, ppcs, scs, sb_bsize);
    cm.mi_rows = mi_rows; cm.mi_cols = mi_cols; cm.mi_stride = mi_cols;
    cm.pcs_ptr = pcs;
    const int n = mi_rows * mi_cols;
    ModeInfo *mip = calloc((size_t)n, sizeof(ModeInfo));
    ModeInfo **grid = calloc((size_t)n, sizeof(ModeInfo *));
    for (int i = 0; i < n; i++) {
        mip[i].mbmi.mode = (PredictionMode)mi_mode[i];
        mip[i].mbmi.uv_mode = (UV_PredictionMode)mi_uv_mode[i];
        mip[i].mbmi.ref_frame[0] = INTRA_FRAME;
        grid[i] = &mip[i];
    }
    pcs->mi_grid_base = grid;
    TileInfo tile; memset(&tile, 0, sizeof(tile));
    tile.mi_row_start = tile4[0]; tile.mi_row_end = tile4[1]; tile.mi_col_start = tile4[2]; tile.mi_col_end = tile4[3];
    BlockGeom geom; memset(&geom, 0, sizeof(geom));
    geom.shape = (PART)shape;
    EbPictureBufferDesc_t rb; memset(&rb, 0, sizeof(rb));
    /* the function derives the chroma origin as origin_x / 2: hand it the doubled plane origin for planes 1 / 2 */
    rb.origin_x = (uint16_t)(plane ? recon_origin_x * 2 : recon_origin_x);
    rb.origin_y = (uint16_t)(plane ? recon_origin_y * 2 : recon_origin_y);
    rb.buffer_y = rb.bufferCb = rb.bufferCr = (uint8_t *)recon;
    rb.stride_y = rb.strideCb = rb.strideCr = (uint16_t)recon_stride;
    if (is16) {
        EncDecContext_t *ctx = calloc(1, sizeof(*ctx));
        ctx->blk_geom = &geom;
        av1_predict_intra_block_16bit(&tile, ctx, &cm, wpx, hpx, (TxSize)tx_size, (PredictionMode)mode, angle_delta, 0, FILTER_INTRA_MODES,
                                      (uint16_t *)top_neigh, (uint16_t *)left_neigh, &rb, col_off, row_off, plane, (block_size)bsize,
                                      (uint32_t)bl_org_x_pict, (uint32_t)bl_org_y_pict);
        free(ctx);
    } else {
        av1_predict_intra_block(&tile, ED_STAGE, &geom, &cm, wpx, hpx, (TxSize)tx_size, (PredictionMode)mode, angle_delta, 0,
                                FILTER_INTRA_MODES, (uint8_t *)top_neigh, (uint8_t *)left_neigh, &rb, col_off, row_off, plane,
                                (block_size)bsize, (uint32_t)bl_org_x_pict, (uint32_t)bl_org_y_pict, 0, 0);
    }
    free(grid); free(mip); free(pcs); free(ppcs); free(scs);
    return 0;
}
