/*
 * oracle/quant.c — scan orders, quantizer tables, quantize_b.
 * TEST INFRASTRUCTURE ONLY (see svt_oracle.h).
 */
#include "svt_oracle.h"
#include "qlookup_data.h"
#include <string.h>
#include <stdlib.h>

/* ---- scan orders -----------------------------------------------------------
 * The reference lists every table (EbTransforms.h:329-1115) and maps them in
 * av1_scan_orders[19][16] (:3349-3870).  They follow three rules, restated
 * here and checked entry-by-entry against the reference's tables in
 * tests/test_oracle_vs_ref.py:
 *   - 1-D column types (V_*) -> row-major ("mrow") scan; 1-D row types (H_*)
 *     -> column-major ("mcol") scan; everything else (2-D types, IDTX) ->
 *     diagonal "default" scan;
 *   - default scan: anti-diagonals d = r + c in increasing order.  Square
 *     blocks alternate direction (odd d: r ascending, even d: r descending);
 *     tall blocks (h > w) always walk r ascending; wide blocks r descending;
 *   - 64-pt sizes scan only the kept min(W,32) x min(H,32) region, using the
 *     scan of that smaller size (:50-54).
 */
static int scan_class(int tx_type) { /* 0 default, 1 mrow, 2 mcol */
    switch (tx_type) {
    case ORC_V_DCT: case ORC_V_ADST: case ORC_V_FLIPADST: return 1;
    case ORC_H_DCT: case ORC_H_ADST: case ORC_H_FLIPADST: return 2;
    default: return 0;
    }
}

int svt_oracle_get_scan(int tx_size, int tx_type, int16_t *scan, int16_t *iscan) {
    int w = svt_oracle_tx_wide(tx_size), h = svt_oracle_tx_high(tx_size);
    if (w > 32) w = 32;
    if (h > 32) h = 32;
    const int n = w * h;
    int16_t *sc = (int16_t *)malloc(sizeof(int16_t) * n);
    int cls = scan_class(tx_type), k = 0;
    if (cls == 1) {
        for (int i = 0; i < n; i++) sc[k++] = (int16_t)i;
    } else if (cls == 2) {
        for (int c = 0; c < w; c++) for (int r = 0; r < h; r++) sc[k++] = (int16_t)(r * w + c);
    } else {
        for (int d = 0; d < w + h - 1; d++) {
            int r_lo = d - (w - 1) > 0 ? d - (w - 1) : 0;
            int r_hi = d < h - 1 ? d : h - 1;
            int ascending = (h > w) ? 1 : (w > h) ? 0 : (d & 1);
            if (ascending) for (int r = r_lo; r <= r_hi; r++) sc[k++] = (int16_t)(r * w + (d - r));
            else           for (int r = r_hi; r >= r_lo; r--) sc[k++] = (int16_t)(r * w + (d - r));
        }
    }
    if (scan) memcpy(scan, sc, sizeof(int16_t) * n);
    if (iscan) for (int i = 0; i < n; i++) iscan[sc[i]] = (int16_t)i;
    free(sc);
    return n;
}

/* ---- quantizer tables (EbModeDecisionConfigurationProcess.c:301-330, 429-520) */
static int msb(unsigned v) { int l = 0; while (v > 1) { v >>= 1; l++; } return l; }

void svt_oracle_build_quantizer(int bd, int16_t *zbin, int16_t *round, int16_t *quant,
                                int16_t *quant_shift, int16_t *dequant) {
    const int bi = bd == 8 ? 0 : (bd == 10 ? 1 : 2);
    const int thr = bd == 8 ? 148 : (bd == 10 ? 592 : 2368);
    for (int q = 0; q < 256; q++) {
        const int dcq = k_dc_qlookup[bi][q];
        const int zbin_factor = q == 0 ? 64 : (dcq < thr ? 84 : 80); /* get_qzbin_factor */
        const int round_factor = q == 0 ? 64 : 48;
        for (int i = 0; i < 8; i++) {
            const int d = (i == 0) ? dcq : k_ac_qlookup[bi][q];
            const int l = msb((unsigned)d);                           /* invert_quant */
            const int m = 1 + (1 << (16 + l)) / d;
            quant[q * 8 + i] = (int16_t)(m - (1 << 16));
            quant_shift[q * 8 + i] = (int16_t)(1 << (16 - l));
            zbin[q * 8 + i] = (int16_t)((zbin_factor * d + 64) >> 7);
            round[q * 8 + i] = (int16_t)((round_factor * d) >> 7);
            dequant[q * 8 + i] = (int16_t)d;
        }
    }
}

/* ---- quantize_b ------------------------------------------------------------ */
static int rpot(int v, int n) { return n == 0 ? v : ((v + (1 << (n - 1))) >> n); } /* ROUND_POWER_OF_TWO */

void svt_oracle_quantize_b(const int32_t *coeff, intptr_t n, int skip_block, const int16_t *zbin,
                           const int16_t *round, const int16_t *quant, const int16_t *quant_shift,
                           int32_t *qcoeff, int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob,
                           const int16_t *scan, const int16_t *iscan, int log_scale, int variant) {
    (void)iscan;
    memset(qcoeff, 0, n * sizeof(*qcoeff));
    memset(dqcoeff, 0, n * sizeof(*dqcoeff));
    int last = -1;
    if (!skip_block) {
        const int zb[2] = {rpot(zbin[0], log_scale), rpot(zbin[1], log_scale)};
        intptr_t limit = n;
        if (variant == 1) {
            /* _c_II pre-scan (EbFullLoop.c:64-76): trailing in-deadzone run is skipped */
            while (limit > 0) {
                const int rc = scan[limit - 1];
                const int32_t c = coeff[rc];
                if (c < zb[rc != 0] && c > -zb[rc != 0]) limit--; else break;
            }
        }
        for (intptr_t i = 0; i < limit; i++) {
            const int rc = scan[i], ac = rc != 0;
            const int32_t c = coeff[rc];
            const int32_t sign = c >> 31;
            const int32_t a = (int32_t)(((uint32_t)c ^ (uint32_t)sign) - (uint32_t)sign);
            int keep;
            if (variant == 0) keep = (c >= zb[ac]) || (c <= -zb[ac]);   /* :266-268 */
            else              keep = a >= zb[ac];                       /* :88 */
            if (!keep) continue;
            int64_t t1 = (int64_t)a + rpot(round[ac], log_scale);
            if (variant == 1) t1 = t1 < INT16_MIN ? INT16_MIN : (t1 > INT16_MAX ? INT16_MAX : t1); /* :85-87 */
            /* qm_ptr == NULL -> weight 1 << AOM_QM_BITS (= 32) is still folded in
             * before the >>16, exactly as :279-283 / :89-93 do */
            const int64_t tw = t1 * 32;
            const int64_t t2 = ((tw * quant[ac]) >> 16) + tw;
            const int32_t aq = (int32_t)((t2 * quant_shift[ac]) >> (16 - log_scale + 5));
            qcoeff[rc] = (aq ^ sign) - sign;
            const int32_t adq = (int32_t)((uint32_t)aq * (uint32_t)(int32_t)dequant[ac]) >> log_scale;
            dqcoeff[rc] = (adq ^ sign) - sign;
            if (aq) last = (int)i;
        }
    }
    *eob = (uint16_t)(last + 1);
}
