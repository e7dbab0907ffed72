/*
 * oracle/intra_avail.c — TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into the product).
 *
 * Neighbour availability of one intra prediction block: the first half of av1_predict_intra_block /
 * av1_predict_intra_block_16bit (EbIntraPrediction.c:4078-4333, 4336-4566): up / left availability from the block's
 * mode-info position, has_top_right (:1567-1623), has_bottom_left (:1755-1826), the sample counts handed to
 * build_intra_predictors (:4318-4333) and get_filt_type (:146-163).
 *
 * The reference answers "is the top-right / bottom-left block already coded" from bit tables (has_tr_* / has_bl_*,
 * :1435-1550, 1626-1738).  This restatement does not hold those tables: it replays the coding order they encode — a
 * 128x128 superblock split recursively into quadrants (Z order; for PARTITION_VERT_A / VERT_B the last split is visited
 * TL, BL, TR, BR, the has_*_vert_* tables) down to the square that holds the block, rectangular blocks of that square
 * left to right / top to bottom — stamps an order number on every 4x4 unit, and compares the two numbers.  It is pinned to
 * the reference's tables through oracle/ref_intra.c (tests/test_oracle_vs_ref.py: every block size, position and partition).
 */
#include <string.h>
#include "svt_oracle.h"

/* block_size order of the AV1 enum (EbDefinitions.h): 4X4 4X8 8X4 8X8 8X16 16X8 16X16 16X32 32X16 32X32 32X64 64X32 64X64
 * 64X128 128X64 128X128 4X16 16X4 8X32 32X8 16X64 64X16 */
static const uint8_t kBw[22] = {4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64, 128, 128, 4, 16, 8, 32, 16, 64};
static const uint8_t kBh[22] = {4, 8, 4, 8, 16, 8, 16, 32, 16, 32, 64, 32, 64, 128, 64, 128, 16, 4, 32, 8, 64, 16};
int svt_oracle_block_wide(int bsize) { return kBw[bsize]; }
int svt_oracle_block_high(int bsize) { return kBh[bsize]; }

enum { P_NONE, P_HORZ, P_VERT, P_SPLIT, P_HORZ_A, P_HORZ_B, P_VERT_A, P_VERT_B, P_HORZ_4, P_VERT_4 };

/* order numbers of the 32x32 4x4-units of a 128x128 superblock for blocks of bw x bh units */
static void stamp(uint16_t *map, int x, int y, int size, int sq, int bw, int bh, int vert, int *ctr) {
    if (size > sq) {
        const int h = size >> 1;
        static const int zo[4][2] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}}, vo[4][2] = {{0, 0}, {0, 1}, {1, 0}, {1, 1}};
        const int(*o)[2] = (h == sq && vert) ? vo : zo;
        for (int q = 0; q < 4; q++) stamp(map, x + o[q][0] * h, y + o[q][1] * h, h, sq, bw, bh, vert, ctr);
        return;
    }
    for (int py = 0; py < sq; py += bh)
        for (int px = 0; px < sq; px += bw) {
            for (int r = 0; r < bh; r++)
                for (int c = 0; c < bw; c++) map[(y + py + r) * 32 + x + px + c] = (uint16_t)*ctr;
            (*ctr)++;
        }
}
static void order_map(uint16_t *map, int bsize, int partition) {
    const int bw = kBw[bsize] >> 2, bh = kBh[bsize] >> 2, sq = bw > bh ? bw : bh;
    int ctr = 0;
    /* vertical rectangles of a VERT_A / VERT_B partition use the ordinary order (has_*_vert_tables, :1536-1550) */
    stamp(map, 0, 0, 32, sq, bw, bh, (partition == P_VERT_A || partition == P_VERT_B) && bw == bh, &ctr);
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) l++; return l; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* has_top_right, :1567 */
int svt_oracle_has_top_right(int sb_mi, int bsize, int mi_row, int mi_col, int top_available, int right_available, int partition,
                             int txsz, int row_off, int col_off, int ss_x, int ss_y) {
    if (!top_available || !right_available) return 0;
    const int bw_unit = kBw[bsize] >> 2, plane_bw_unit = imax(bw_unit >> ss_x, 1), tr_units = svt_oracle_tx_wide(txsz) >> 2;
    if (row_off > 0) {
        if (kBw[bsize] > 64) {
            if (row_off == 16 >> ss_y && col_off + tr_units == 16 >> ss_x) return 1;
            const int u64 = 16 >> ss_x;
            return col_off % u64 + tr_units < u64;
        }
        return col_off + tr_units < plane_bw_unit;
    }
    if (col_off + tr_units < plane_bw_unit) return 1;
    const int bwl = ilog2(kBw[bsize] >> 2), bhl = ilog2(kBh[bsize] >> 2);
    const int brow = (mi_row & (sb_mi - 1)) >> bhl, bcol = (mi_col & (sb_mi - 1)) >> bwl;
    if (brow == 0) return 1;
    if (((bcol + 1) << bwl) >= sb_mi) return 0;
    uint16_t map[32 * 32];
    order_map(map, bsize, partition);
    const int x = bcol << bwl, y = brow << bhl;
    return map[(y - 1) * 32 + x + (1 << bwl)] < map[y * 32 + x];
}

/* has_bottom_left, :1755 */
int svt_oracle_has_bottom_left(int sb_mi, int bsize, int mi_row, int mi_col, int bottom_available, int left_available, int partition,
                               int txsz, int row_off, int col_off, int ss_x, int ss_y) {
    if (!bottom_available || !left_available) return 0;
    const int th_units = svt_oracle_tx_high(txsz) >> 2;
    if (kBw[bsize] > 64 && col_off > 0) {
        const int u64 = 16 >> ss_x;
        if (col_off % u64 == 0) {
            const int h64 = 16 >> ss_y;
            return row_off % h64 + th_units < imin((kBh[bsize] >> 2) >> ss_y, h64);
        }
    }
    if (col_off > 0) return 0;
    const int plane_bh_unit = imax((kBh[bsize] >> 2) >> ss_y, 1);
    if (row_off + th_units < plane_bh_unit) return 1;
    const int bwl = ilog2(kBw[bsize] >> 2), bhl = ilog2(kBh[bsize] >> 2);
    const int brow = (mi_row & (sb_mi - 1)) >> bhl, bcol = (mi_col & (sb_mi - 1)) >> bwl;
    if (bcol == 0) {
        const int start = (brow << bhl) >> ss_y;          /* bh_in_mi_log2 + MI_SIZE_LOG2 - tx_size_wide_log2[0] = bhl */
        return start + row_off + th_units < (sb_mi >> ss_y);
    }
    if (((brow + 1) << bhl) >= sb_mi) return 0;
    uint16_t map[32 * 32];
    order_map(map, bsize, partition);
    const int x = bcol << bwl, y = brow << bhl;
    return map[(y + (1 << bhl)) * 32 + x - 1] < map[y * 32 + x];
}

static int scale_chroma(int bsize, int ss) {           /* scale_chroma_bsize, :3619, 4:2:0 */
    if (!ss) return bsize;
    switch (bsize) {
    case 0: case 1: case 2: return 3;      /* 4x4, 4x8, 8x4 -> 8x8 */
    case 16: return 4;                     /* 4x16 -> 8x16 */
    case 17: return 5;                     /* 16x4 -> 16x8 */
    default: return bsize;
    }
}
static int smooth_luma(int m) { return m == 9 || m == 10 || m == 11; }     /* SMOOTH_PRED, SMOOTH_V_PRED, SMOOTH_H_PRED */

/* av1_predict_intra_block{,_16bit}: availability, sample counts, edge-filter type.  mi_mode / mi_uv_mode: [mi_rows * mi_cols]
 * (all blocks intra).  tile4: mi_row_start, mi_row_end, mi_col_start, mi_col_end.  out5: n_top_px, n_topright_px, n_left_px,
 * n_bottomleft_px, filt_type. */
void svt_oracle_intra_neighbor_px(int is16, int sb_mi, int mi_rows, int mi_cols, const uint8_t *mi_mode, const uint8_t *mi_uv_mode,
                                  const int32_t *tile4, int partition, int bsize, int tx_size, int plane, int bl_org_x_pict,
                                  int bl_org_y_pict, int col_off, int row_off, int wpx, int hpx, int32_t *out5) {
    const int mirow = bl_org_y_pict >> 2, micol = bl_org_x_pict >> 2;
    /* the 8-bit function works on a one-tile picture (:4124-4137); the 16-bit one on the given tile (:4371-4383) */
    const int t_r0 = is16 ? tile4[0] : 0, t_r1 = is16 ? tile4[1] : mi_rows, t_c0 = is16 ? tile4[2] : 0, t_c1 = is16 ? tile4[3] : mi_cols;
    const int up = mirow > t_r0, lf = micol > t_c0;
    const int bw = kBw[bsize] >> 2, bh = kBh[bsize] >> 2;
    const int to_bottom = (mi_rows - bh - mirow) * 4 * 8, to_right = (mi_cols - bw - micol) * 4 * 8;
    const int ss = plane ? 1 : 0;
    int c_up = up, c_lf = lf;
    if (ss && bw < 2) c_lf = (micol - 1) > tile4[2];
    if (ss && bh < 2) c_up = (mirow - 1) > tile4[0];
    const int txwpx = svt_oracle_tx_wide(tx_size), txhpx = svt_oracle_tx_high(tx_size);
    const int x = col_off << 2, y = row_off << 2;
    const int txw = txwpx >> 2, txh = txhpx >> 2;
    const int have_top = row_off || (ss ? c_up : up), have_left = col_off || (ss ? c_lf : lf);
    const int xr = (to_right >> (3 + ss)) + (wpx - x - txwpx), yd = (to_bottom >> (3 + ss)) + (hpx - y - txhpx);
    const int right_av = micol + ((col_off + txw) << ss) < t_c1;
    const int bottom_av = (yd > 0) && (mirow + ((row_off + txh) << ss) < t_r1);
    const int bs = scale_chroma(bsize, ss);
    const int tr = svt_oracle_has_top_right(sb_mi, bs, mirow, micol, have_top, right_av, partition, tx_size, row_off, col_off, ss, ss);
    const int bl = svt_oracle_has_bottom_left(sb_mi, bs, mirow, micol, bottom_av, have_left, partition, tx_size, row_off, col_off, ss, ss);
    out5[0] = have_top ? imin(txwpx, xr + txwpx) : 0;
    out5[1] = tr ? imin(txwpx, xr) : 0;
    out5[2] = have_left ? imin(txhpx, yd + txhpx) : 0;
    out5[3] = bl ? imin(txhpx, yd) : 0;
    /* get_filt_type (:146): luma from the above / left mode infos, chroma from the chroma reference blocks (:4200-4220) */
    int ab = 0, le = 0;
    if (!plane) {
        if (up) ab = smooth_luma(mi_mode[(mirow - 1) * mi_cols + micol]);
        if (lf) le = smooth_luma(mi_mode[mirow * mi_cols + micol - 1]);
    } else {
        const int br = mirow - (mirow & 1), bc = micol - (micol & 1);
        if (c_up) ab = smooth_luma(mi_uv_mode[(br - 1) * mi_cols + bc + 1]);        /* UV_SMOOTH* share the luma values 9..11 */
        if (c_lf) le = smooth_luma(mi_uv_mode[(br + 1) * mi_cols + bc - 1]);
    }
    out5[4] = ab || le;
}
