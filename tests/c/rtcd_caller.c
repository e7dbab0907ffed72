/*
 * tests/c/rtcd_caller.c — a plain-C host of libsvt_hip_dsp.so that dispatches the way the reference does: a block of
 * global function pointers with the reference's own slot names and signatures (aom_dsp_rtcd.h), filled first with local
 * scalar stand-ins (the role of setup_rtcd_internal's C defaults), then overridden BY NAME through
 * svt_hip_rtcd_override_slot, then called through the pointers.  Each result is compared with the stand-in's.
 * Exit codes: 0 ok, 3 the library refused every override (no usable device) and left the host's pointers alone, 4.. a mismatch.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "svt_hip_dsp.h"

/* ---- the host's dispatch block (names and signatures of Source/Lib/Common/Codec/aom_dsp_rtcd.h) ---- */
static void (*aom_dc_predictor_16x16)(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
static void (*aom_v_predictor_8x8)(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
static void (*aom_highbd_h_predictor_4x8)(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
static unsigned int (*aom_sad16x16)(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride);
static void (*aom_sad8x8x4d)(const uint8_t *src, int src_stride, const uint8_t *const ref[], int ref_stride, uint32_t *sad_array);
static void (*ResidualKernel)(uint8_t *input, uint32_t input_stride, uint8_t *pred, uint32_t pred_stride, int16_t *residual,
                              uint32_t residual_stride, uint32_t area_width, uint32_t area_height);
static void (*subtract_average)(int16_t *pred_buf_q3, int32_t width, int32_t height, int32_t round_offset, int32_t num_pel_log2);
static void (*av1_txb_init_levels)(const int32_t *const coeff, const int32_t width, const int32_t height, uint8_t *const levels);

/* ---- scalar stand-ins (what the pointers hold before the override) ---- */
static void dc16_c(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left) {
    int sum = 0;
    for (int i = 0; i < 16; i++) sum += above[i] + left[i];
    const uint8_t v = (uint8_t)((sum + 16) / 32);
    for (int r = 0; r < 16; r++) memset(dst + r * stride, v, 16);
}
static void v8_c(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left) {
    (void)left;
    for (int r = 0; r < 8; r++) memcpy(dst + r * stride, above, 8);
}
static void hbd_h4x8_c(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd) {
    (void)above; (void)bd;
    for (int r = 0; r < 8; r++) for (int c = 0; c < 4; c++) dst[r * stride + c] = left[r];
}
static unsigned int sad_c(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h) {
    unsigned int s = 0;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) s += (unsigned)abs((int)a[y * as + x] - (int)b[y * bs + x]);
    return s;
}
static unsigned int sad16_c(const uint8_t *a, int as, const uint8_t *b, int bs) { return sad_c(a, as, b, bs, 16, 16); }
static void sad8x4d_c(const uint8_t *a, int as, const uint8_t *const r[], int rs, uint32_t *out) {
    for (int i = 0; i < 4; i++) out[i] = sad_c(a, as, r[i], rs, 8, 8);
}
static void residual_c(uint8_t *in, uint32_t is, uint8_t *pr, uint32_t ps, int16_t *res, uint32_t rs, uint32_t w, uint32_t h) {
    for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) res[y * rs + x] = (int16_t)((int)in[y * is + x] - (int)pr[y * ps + x]);
}
static void sub_avg_c(int16_t *q3, int32_t w, int32_t h, int32_t round_offset, int32_t num_pel_log2) {
    int sum = round_offset;
    for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) sum += q3[j * 32 + i];
    const int avg = sum >> num_pel_log2;
    for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) q3[j * 32 + i] = (int16_t)(q3[j * 32 + i] - avg);
}
static void levels_c(const int32_t *const coeff, const int32_t w, const int32_t h, uint8_t *const levels) {
    const int stride = w + 4;
    uint8_t *ls = levels;
    memset(levels - 2 * stride, 0, (size_t)2 * stride);
    memset(levels + stride * h, 0, (size_t)4 * stride + 16);
    for (int i = 0; i < h; i++) {
        for (int j = 0; j < w; j++) { int v = abs(coeff[i * w + j]); *ls++ = (uint8_t)(v > 127 ? 127 : v); }
        for (int j = 0; j < 4; j++) *ls++ = 0;
    }
}

static uint32_t rnd_state = 13596u;
static uint32_t rnd(void) { rnd_state = rnd_state * 1664525u + 1013904223u; return rnd_state >> 8; }

/* A refused override (no usable device) must leave the host's pointer as it was: the host then keeps the kernel it already
 * had - in the encoder its AVX2 one (SURVEY 8b "Errors").  Exit 3 = refused AND every pointer still the stand-in AND still
 * callable; 6 = a refused override touched a pointer. */
static int refused;
#define OVERRIDE(slot) do { if (svt_hip_rtcd_override_slot(#slot, (void **)&slot) != SVT_HIP_OK) { \
        fprintf(stderr, "override of %s refused: %s\n", #slot, svt_hip_last_error()); refused++; } } while (0)

int main(void) {
    aom_dc_predictor_16x16 = dc16_c; aom_v_predictor_8x8 = v8_c; aom_highbd_h_predictor_4x8 = hbd_h4x8_c; aom_sad16x16 = sad16_c;
    aom_sad8x8x4d = sad8x4d_c; ResidualKernel = residual_c; subtract_average = sub_avg_c; av1_txb_init_levels = levels_c;
    if (svt_hip_rtcd_slot_count() < 480) { fprintf(stderr, "registry too small\n"); return 4; }
    OVERRIDE(aom_dc_predictor_16x16); OVERRIDE(aom_v_predictor_8x8); OVERRIDE(aom_highbd_h_predictor_4x8); OVERRIDE(aom_sad16x16);
    OVERRIDE(aom_sad8x8x4d); OVERRIDE(ResidualKernel); OVERRIDE(subtract_average); OVERRIDE(av1_txb_init_levels);
    if (refused) {
        if ((void *)aom_dc_predictor_16x16 != (void *)dc16_c || (void *)aom_v_predictor_8x8 != (void *)v8_c ||
            (void *)aom_highbd_h_predictor_4x8 != (void *)hbd_h4x8_c || (void *)aom_sad16x16 != (void *)sad16_c ||
            (void *)aom_sad8x8x4d != (void *)sad8x4d_c || (void *)ResidualKernel != (void *)residual_c ||
            (void *)subtract_average != (void *)sub_avg_c || (void *)av1_txb_init_levels != (void *)levels_c || refused != 8)
            return 6;
        uint8_t x[256], y[256];
        for (int i = 0; i < 256; i++) { x[i] = (uint8_t)rnd(); y[i] = (uint8_t)rnd(); }
        if (aom_sad16x16(x, 16, y, 16) != sad_c(x, 16, y, 16, 16, 16)) return 6;      /* the host's own kernel still answers */
        return 3;
    }
    if ((void *)aom_dc_predictor_16x16 == (void *)dc16_c) return 5;

    uint8_t nb_a[96], nb_l[96], a[64 * 24], b[4][64 * 24];
    for (int i = 0; i < 96; i++) { nb_a[i] = (uint8_t)rnd(); nb_l[i] = (uint8_t)rnd(); }
    for (int i = 0; i < 64 * 24; i++) { a[i] = (uint8_t)rnd(); for (int k = 0; k < 4; k++) b[k][i] = (uint8_t)rnd(); }

    uint8_t d1[16 * 40], d2[16 * 40];
    memset(d1, 7, sizeof d1); memset(d2, 7, sizeof d2);
    aom_dc_predictor_16x16(d1, 40, nb_a + 16, nb_l + 16); dc16_c(d2, 40, nb_a + 16, nb_l + 16);
    if (memcmp(d1, d2, sizeof d1)) return 10;
    aom_v_predictor_8x8(d1, 40, nb_a + 16, nb_l + 16); v8_c(d2, 40, nb_a + 16, nb_l + 16);
    if (memcmp(d1, d2, sizeof d1)) return 11;
    uint16_t ha[32], hl[32], h1[8 * 12], h2[8 * 12];
    for (int i = 0; i < 32; i++) { ha[i] = (uint16_t)(rnd() & 1023); hl[i] = (uint16_t)(rnd() & 1023); }
    memset(h1, 0, sizeof h1); memset(h2, 0, sizeof h2);
    aom_highbd_h_predictor_4x8(h1, 12, ha + 8, hl + 8, 10); hbd_h4x8_c(h2, 12, ha + 8, hl + 8, 10);
    if (memcmp(h1, h2, sizeof h1)) return 12;
    if (aom_sad16x16(a, 24, b[0], 24) != sad16_c(a, 24, b[0], 24)) return 13;
    const uint8_t *refs[4] = {b[0] + 3, b[1] + 24, b[2], b[3] + 5};
    uint32_t s1[4], s2[4];
    aom_sad8x8x4d(a, 24, refs, 24, s1); sad8x4d_c(a, 24, refs, 24, s2);
    if (memcmp(s1, s2, sizeof s1)) return 14;
    int16_t r1[16 * 20], r2[16 * 20];
    memset(r1, 0, sizeof r1); memset(r2, 0, sizeof r2);
    ResidualKernel(a, 24, b[1], 24, r1, 20, 16, 16); residual_c(a, 24, b[1], 24, r2, 20, 16, 16);
    if (memcmp(r1, r2, sizeof r1)) return 15;
    int16_t q1[32 * 8], q2[32 * 8];
    for (int i = 0; i < 32 * 8; i++) q1[i] = q2[i] = (int16_t)(rnd() & 2047);
    subtract_average(q1, 8, 8, 32, 6); sub_avg_c(q2, 8, 8, 32, 6);
    if (memcmp(q1, q2, sizeof q1)) return 16;
    int32_t co[8 * 8];
    for (int i = 0; i < 64; i++) co[i] = (int32_t)(rnd() % 600) - 300;
    uint8_t l1[12 * 14 + 16], l2[12 * 14 + 16];
    memset(l1, 0xAA, sizeof l1); memset(l2, 0xAA, sizeof l2);
    av1_txb_init_levels(co, 8, 8, l1 + 2 * 12); levels_c(co, 8, 8, l2 + 2 * 12);
    if (memcmp(l1, l2, sizeof l1)) return 17;
    printf("rtcd_caller: 8 slots overridden by name and called through the host's own pointers: all results equal\n");
    return 0;
}
