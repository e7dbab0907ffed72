/*
 * oracle/ref_pins.c — OUR harness around the reference's own caller-level functions of the transform path that go
 * through its RTCD dispatch pointers; compiled into oracle/_ref/libsvtref.so (oracle/Makefile).
 * TEST INFRASTRUCTURE ONLY.  Nothing of the reference is re-implemented: the dispatch pointers are the ones
 * ref_ois.c defines and fills with the reference's own setup_rtcd_internal(ASM_AVX2) (ref_ois_setup()).
 *
 *  ref_estimate_transform    av1_estimate_transform (EbTransforms.c:4918-5292): size switch -> RTCD forward transform,
 *                            and for 64-point sizes HandleTransform64x64_c & co (:4377-4408, 4580-4731) = energy of
 *                            the discarded region + zeroing + re-pack to stride 32.
 *  ref_inv_txfm_add_u8       the 8-bit reconstruction entry: which = 0 av1_inv_txfm_add_c (:8882-8903),
 *                            1 = av1_inv_txfm_add_ssse3 (the production slot, av1_inv_txfm_ssse3.c:2903),
 *                            2 = av1_inv_transform_recon8bit (:8939, through the av1_inv_txfm_add pointer).
 *  ref_picture_full_distortion32   picture_full_distortion32_bits (EbPictureOperators.c:349-457), luma component, on
 *                            two caller buffers wrapped in the reference's own EbPictureBufferDesc_t.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "EbDefinitions.h"
#include "aom_dsp_rtcd.h"
#include "EbTransforms.h"
#include "EbPictureBufferDesc.h"
#include "EbPictureOperators.h"

void ref_ois_setup(void);

int ref_estimate_transform(int16_t *residual, uint32_t residual_stride, int32_t *coeff, int tx_size, int tx_type,
                           int bit_increment, uint64_t *three_quad_energy) {
    ref_ois_setup();
    *three_quad_energy = 0;
    return (int)av1_estimate_transform(residual, residual_stride, coeff, 0 /* coeff_stride: unused */, (TxSize)tx_size,
                                       three_quad_energy, NULL, (uint32_t)bit_increment, (TxType)tx_type, ASM_AVX2,
                                       PLANE_TYPE_Y, DEFAULT_SHAPE);
}

void ref_inv_txfm_add_u8(const int32_t *dqcoeff, uint8_t *dst, int32_t stride, int tx_type, int tx_size, int eob,
                         int which) {
    ref_ois_setup();
    TxfmParam p;
    memset(&p, 0, sizeof(p));
    p.tx_type = (TxType)tx_type;
    p.tx_size = (TxSize)tx_size;
    p.eob = eob;
    p.lossless = 0;
    p.bd = 8;
    p.is_hbd = 1;
    if (which == 0) av1_inv_txfm_add_c(dqcoeff, dst, stride, &p);
    else if (which == 1) av1_inv_txfm_add_ssse3(dqcoeff, dst, stride, &p);
    else av1_inv_transform_recon8bit((int32_t *)dqcoeff, dst, (uint32_t)stride, (TxSize)tx_size, (TxType)tx_type,
                                     PLANE_TYPE_Y, (uint32_t)eob);
}

int ref_picture_full_distortion32(int32_t *coeff, uint32_t coeff_origin, int32_t *recon, uint32_t recon_origin,
                                  uint32_t bwidth, uint32_t bheight, uint32_t count_non_zero, int asm_type,
                                  uint64_t y_distortion[2]) {
    EbPictureBufferDesc_t a, b;
    uint64_t cb[2], cr[2];
    memset(&a, 0, sizeof(a));
    memset(&b, 0, sizeof(b));
    a.buffer_y = (EbByte)coeff;
    b.buffer_y = (EbByte)recon;
    return (int)picture_full_distortion32_bits(&a, coeff_origin, 0, &b, recon_origin, 0, bwidth, bheight, 0, 0,
                                               y_distortion, cb, cr, count_non_zero, 0, 0, COMPONENT_LUMA,
                                               (EbAsm)asm_type);
}
