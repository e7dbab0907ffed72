/*
 * tests/c/frame_host.c — a plain-C host of the BATCHED interface (no Python, no torch): the encode pass of one yuv420p picture
 * through svt_hip_encode_recon_frame, the way an encoder thread would drive it - tables from the library's own host builders
 * (svt_hip_build_quantizer, svt_hip_get_scan), planes and origin tables uploaded with svt_hip_memcpy_h2d, one call for every
 * (plane, transform size) group, results fetched with svt_hip_memcpy_d2h.
 *
 * usage: frame_host <width> <height> <qindex> <seed>      (8-bit; luma sizes 64/32/16/8/4, chroma at half the side)
 * prints one line:  blocks <n> eob_sum <s> qcoeff_checksum <c> recon_sum <r>      - the digest frames.FramePass.digest() computes,
 * so the GPU test compares this program with the Python path on the same picture (same LCG for the samples).
 * Exit codes: 0 ok, 2 bad arguments, 3 no usable device (there is no CPU fallback), 4 a library call failed.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "svt_hip_dsp.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        const int rc_ = (call);                                                          \
        if (rc_ != SVT_HIP_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, svt_hip_last_error()); return 4; } \
    } while (0)

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

static void *upload(const void *h, size_t bytes) {
    void *d = svt_hip_malloc(bytes);
    if (d && svt_hip_memcpy_h2d(d, h, bytes, NULL) != SVT_HIP_OK) return NULL;
    return d;
}

/* A second call on the same planes (argv[5] = luma size S): svt_hip_encode_recon_frame_ex on ONE pass (Y at S, Cb / Cr at S / 2) with
 * chroma from luma on every chroma block (alpha_cb = (b % 33) - 16, alpha_cr = ((7 b) % 33) - 16 for chroma block b) and a level map
 * for every block of the three groups.  Prints:  ex blocks <n> eob_sum <s> qcoeff_checksum <c> recon_sum <r> pred_sum <p> levels_sum <l> */
static int run_ex(int S, const int *pw, const int *ph, void *const *d_src, uint8_t *const *pred, int16_t (*zb)[8], int16_t (*rn)[8], int16_t (*qu)[8],
                  int16_t (*qs)[8], int16_t (*dq)[8], int qindex) {
    static const int sides[5] = {64, 32, 16, 8, 4};
    static const int txs[5] = {SVT_TX_64X64, SVT_TX_32X32, SVT_TX_16X16, SVT_TX_8X8, SVT_TX_4X4};
    svt_hip_frame_group g[3];
    svt_hip_frame_levels lv[3];
    size_t nblk[3], lpitch[3];
    int nc[3];
    void *d_pred2[3];
    for (int p = 0; p < 3; p++) {
        const int side = p ? S / 2 : S;
        int ti = 0;
        while (ti < 5 && sides[ti] != side) ti++;
        if (ti == 5) return 2;
        const int k = side > 32 ? 32 : side;
        nc[p] = k * k;
        const int bx = pw[p] / side, by = ph[p] / side;
        nblk[p] = (size_t)bx * by;
        uint32_t *xy = malloc(nblk[p] * sizeof(uint32_t));
        for (int y = 0; y < by; y++)
            for (int x = 0; x < bx; x++) xy[(size_t)y * bx + x] = (uint32_t)(x * side) | ((uint32_t)(y * side) << 16);
        int16_t scan[1024], iscan[1024];
        if (svt_hip_get_scan(txs[ti], SVT_DCT_DCT, scan, iscan) != nc[p]) return 4;
        const size_t pn = (size_t)pw[p] * ph[p];
        d_pred2[p] = upload(pred[p], pn);                    /* the call predicts the chroma planes in place: its own copies */
        memset(&g[p], 0, sizeof(g[p]));
        g[p].d_src = d_src[p]; g[p].src_stride = (uint32_t)pw[p];
        g[p].d_pred = d_pred2[p]; g[p].pred_stride = (uint32_t)pw[p];
        g[p].d_recon = upload(pred[p], pn); g[p].recon_stride = (uint32_t)pw[p];
        g[p].d_xy = upload(xy, nblk[p] * sizeof(uint32_t));
        g[p].nblocks = (uint32_t)nblk[p]; g[p].tx_size = txs[ti]; g[p].tx_type = SVT_DCT_DCT;
        g[p].d_iscan = upload(iscan, (size_t)nc[p] * sizeof(int16_t));
        g[p].d_qcoeff = svt_hip_malloc(nblk[p] * (size_t)nc[p] * sizeof(int32_t));
        g[p].d_eob = svt_hip_malloc(nblk[p] * sizeof(uint16_t));
        lpitch[p] = (((size_t)(k + 4) * (k + 6) + 16) + 15) & ~(size_t)15;
        lv[p].d_levels_buf = svt_hip_malloc(nblk[p] * lpitch[p]); lv[p].levels_block_pitch = lpitch[p];
        if (!d_pred2[p] || !g[p].d_recon || !g[p].d_xy || !g[p].d_iscan || !g[p].d_qcoeff || !g[p].d_eob || !lv[p].d_levels_buf) return 4;
        free(xy);
    }
    int32_t *a_cb = malloc(nblk[1] * sizeof(int32_t)), *a_cr = malloc(nblk[1] * sizeof(int32_t));
    for (size_t b = 0; b < nblk[1]; b++) { a_cb[b] = (int32_t)(b % 33) - 16; a_cr[b] = (int32_t)((7 * b) % 33) - 16; }
    svt_hip_frame_cfl_group c;
    memset(&c, 0, sizeof(c));
    c.d_luma_recon = g[0].d_recon; c.luma_stride = g[0].recon_stride;
    c.d_pred_cb = d_pred2[1]; c.pred_stride_cb = (uint32_t)pw[1];
    c.d_pred_cr = d_pred2[2]; c.pred_stride_cr = (uint32_t)pw[2];
    c.d_xy = g[1].d_xy;
    c.d_alpha_q3_cb = upload(a_cb, nblk[1] * sizeof(int32_t)); c.d_alpha_q3_cr = upload(a_cr, nblk[1] * sizeof(int32_t));
    c.width = c.height = (uint32_t)(S / 2); c.nblocks = (uint32_t)nblk[1];
    if (!c.d_alpha_q3_cb || !c.d_alpha_q3_cr) return 4;
    CHECK(svt_hip_stream_sync(NULL));
    CHECK(svt_hip_encode_recon_frame_ex(g, 3, 1, &c, 1, lv, 0, 8, zb[qindex], rn[qindex], qu[qindex], qs[qindex], dq[qindex], NULL));
    CHECK(svt_hip_stream_sync(NULL));
    const int64_t M = 2147483647;
    int64_t blocks = 0, eob_sum = 0, qchk = 0, recon_sum = 0, pred_sum = 0, levels_sum = 0;
    for (int p = 0; p < 3; p++) {
        const size_t n = nblk[p], pn = (size_t)pw[p] * ph[p];
        uint16_t *eob = malloc(n * sizeof(uint16_t));
        int32_t *q = malloc(n * (size_t)nc[p] * sizeof(int32_t));
        uint8_t *rec = malloc(pn), *prd = malloc(pn), *lev = malloc(n * lpitch[p]);
        CHECK(svt_hip_memcpy_d2h(eob, g[p].d_eob, n * sizeof(uint16_t), NULL));
        CHECK(svt_hip_memcpy_d2h(q, g[p].d_qcoeff, n * (size_t)nc[p] * sizeof(int32_t), NULL));
        CHECK(svt_hip_memcpy_d2h(rec, g[p].d_recon, pn, NULL));
        CHECK(svt_hip_memcpy_d2h(prd, d_pred2[p], pn, NULL));
        CHECK(svt_hip_memcpy_d2h(lev, lv[p].d_levels_buf, n * lpitch[p], NULL));
        CHECK(svt_hip_stream_sync(NULL));
        blocks += (int64_t)n;
        for (size_t b = 0; b < n; b++) eob_sum += eob[b];
        int64_t s = 0;
        for (size_t b = 0; b < n; b++)
            for (int k = 0; k < nc[p]; k++) s += (int64_t)q[b * nc[p] + k] * (int64_t)((k % 8191) + 1);
        s %= M; if (s < 0) s += M;
        qchk += s;
        for (size_t k = 0; k < pn; k++) { recon_sum += rec[k]; pred_sum += prd[k]; }
        const int kk = nc[p] == 1024 ? 32 : (nc[p] == 256 ? 16 : (nc[p] == 64 ? 8 : 4));
        const size_t used = (size_t)(kk + 4) * (kk + 6) + 16;
        for (size_t b = 0; b < n; b++)
            for (size_t k = 0; k < used; k++) levels_sum += lev[b * lpitch[p] + k];
        free(eob); free(q); free(rec); free(prd); free(lev);
    }
    qchk %= M;
    printf("ex blocks %lld eob_sum %lld qcoeff_checksum %lld recon_sum %lld pred_sum %lld levels_sum %lld\n", (long long)blocks, (long long)eob_sum,
           (long long)qchk, (long long)recon_sum, (long long)pred_sum, (long long)levels_sum);
    return 0;
}

int main(int argc, char **argv) {
    if (argc != 5 && argc != 6) { fprintf(stderr, "usage: %s width height qindex seed [luma size of the _ex pass]\n", argv[0]); return 2; }
    const int W = atoi(argv[1]), H = atoi(argv[2]), qindex = atoi(argv[3]);
    uint32_t seed = (uint32_t)strtoul(argv[4], NULL, 0);
    if (W < 64 || H < 64 || (W & 1) || (H & 1) || qindex < 0 || qindex > 255) return 2;
    if (svt_hip_init(0) != SVT_HIP_OK) { fprintf(stderr, "svt_hip_init: %s\n", svt_hip_last_error()); return 3; }

    /* planes: src random, pred = src + noise in [-24, 24], clipped */
    const int pw[3] = {W, W / 2, W / 2}, ph[3] = {H, H / 2, H / 2};
    uint8_t *src[3], *pred[3];
    void *d_src[3], *d_pred[3], *d_recon[3];
    for (int p = 0; p < 3; p++) {
        const size_t n = (size_t)pw[p] * ph[p];
        src[p] = malloc(n); pred[p] = malloc(n);
        for (size_t i = 0; i < n; i++) {
            src[p][i] = (uint8_t)(lcg(&seed) & 255);
            int v = (int)src[p][i] + (int)(lcg(&seed) % 49) - 24;
            pred[p][i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
        d_src[p] = upload(src[p], n); d_pred[p] = upload(pred[p], n); d_recon[p] = upload(pred[p], n);
        if (!d_src[p] || !d_pred[p] || !d_recon[p]) { fprintf(stderr, "upload: %s\n", svt_hip_last_error()); return 4; }
    }
    /* quantiser rows of this qindex (8-bit) */
    static int16_t zbin[256][8], rnd[256][8], quant[256][8], qshift[256][8], deq[256][8];
    CHECK(svt_hip_build_quantizer(8, zbin, rnd, quant, qshift, deq));

    /* one group per (luma size, plane) */
    static const int luma_sizes[5] = {64, 32, 16, 8, 4};
    static const int tx_of_side[5] = {SVT_TX_64X64, SVT_TX_32X32, SVT_TX_16X16, SVT_TX_8X8, SVT_TX_4X4};
    svt_hip_frame_group groups[15];
    size_t gblocks[15];
    int gnc[15], gplane[15];
    int ng = 0;
    for (int si = 0; si < 5; si++)
        for (int p = 0; p < 3; p++) {
            const int side = p ? luma_sizes[si] / 2 : luma_sizes[si];
            if (side < 4) continue;
            int ti = 0;
            while (luma_sizes[ti] != side) ti++;
            const int tx = tx_of_side[ti], nc = (side > 32 ? 32 : side) * (side > 32 ? 32 : side);
            const int bx = pw[p] / side, by = ph[p] / side;
            const size_t n = (size_t)bx * by;
            if (!n) continue;
            uint32_t *xy = malloc(n * sizeof(uint32_t));
            for (int y = 0; y < by; y++)
                for (int x = 0; x < bx; x++) xy[(size_t)y * bx + x] = (uint32_t)(x * side) | ((uint32_t)(y * side) << 16);
            int16_t scan[1024], iscan[1024];
            if (svt_hip_get_scan(tx, SVT_DCT_DCT, scan, iscan) != nc) { fprintf(stderr, "svt_hip_get_scan\n"); return 4; }   /* returns the entry count */
            svt_hip_frame_group *g = &groups[ng];
            memset(g, 0, sizeof(*g));
            g->d_src = d_src[p]; g->src_stride = (uint32_t)pw[p];
            g->d_pred = d_pred[p]; g->pred_stride = (uint32_t)pw[p];
            g->d_recon = svt_hip_malloc((size_t)pw[p] * ph[p]);               /* one reconstruction per pass, as FramePass does */
            if (!g->d_recon) return 4;
            CHECK(svt_hip_memcpy_h2d(g->d_recon, pred[p], (size_t)pw[p] * ph[p], NULL));
            g->recon_stride = (uint32_t)pw[p];
            g->d_xy = upload(xy, n * sizeof(uint32_t));
            g->nblocks = (uint32_t)n; g->tx_size = tx; g->tx_type = SVT_DCT_DCT;
            g->d_iscan = upload(iscan, (size_t)nc * sizeof(int16_t));
            g->d_qcoeff = svt_hip_malloc(n * (size_t)nc * sizeof(int32_t));
            g->d_eob = svt_hip_malloc(n * sizeof(uint16_t));
            if (!g->d_xy || !g->d_iscan || !g->d_qcoeff || !g->d_eob) return 4;
            gblocks[ng] = n; gnc[ng] = nc; gplane[ng] = p;
            free(xy);
            ng++;
        }
    CHECK(svt_hip_stream_sync(NULL));
    CHECK(svt_hip_encode_recon_frame(groups, ng, 0, 8, zbin[qindex], rnd[qindex], quant[qindex], qshift[qindex], deq[qindex], NULL));
    CHECK(svt_hip_stream_sync(NULL));

    /* digest: blocks, sum eob, sum qcoeff * (1 + index mod 8191) mod 2^31 - 1 (per group, then summed), sum of reconstructed samples */
    const int64_t M = 2147483647;
    int64_t blocks = 0, eob_sum = 0, qchk = 0, recon_sum = 0;
    for (int i = 0; i < ng; i++) {
        const size_t n = gblocks[i];
        uint16_t *eob = malloc(n * sizeof(uint16_t));
        int32_t *q = malloc(n * (size_t)gnc[i] * sizeof(int32_t));
        const size_t pn = (size_t)pw[gplane[i]] * ph[gplane[i]];
        uint8_t *rec = malloc(pn);
        CHECK(svt_hip_memcpy_d2h(eob, groups[i].d_eob, n * sizeof(uint16_t), NULL));
        CHECK(svt_hip_memcpy_d2h(q, groups[i].d_qcoeff, n * (size_t)gnc[i] * sizeof(int32_t), NULL));
        CHECK(svt_hip_memcpy_d2h(rec, groups[i].d_recon, pn, NULL));
        CHECK(svt_hip_stream_sync(NULL));
        blocks += (int64_t)n;
        for (size_t b = 0; b < n; b++) eob_sum += eob[b];
        int64_t s = 0;
        for (size_t b = 0; b < n; b++)
            for (int c = 0; c < gnc[i]; c++) s += (int64_t)q[b * gnc[i] + c] * (int64_t)((c % 8191) + 1);
        s %= M; if (s < 0) s += M;
        qchk += s;
        for (size_t k = 0; k < pn; k++) recon_sum += rec[k];
        free(eob); free(q); free(rec);
    }
    qchk %= M;
    printf("blocks %lld eob_sum %lld qcoeff_checksum %lld recon_sum %lld\n", (long long)blocks, (long long)eob_sum, (long long)qchk, (long long)recon_sum);
    if (argc == 6) return run_ex(atoi(argv[5]), pw, ph, d_src, pred, zbin, rnd, quant, qshift, deq, qindex);
    return 0;
}
