/*
 * oracle/ref_bench.c — OUR harness around the reference's own kernels; it is
 * compiled into oracle/_ref/libsvtref.so together with the reference sources
 * (oracle/Makefile).  TEST INFRASTRUCTURE ONLY.
 *
 *  - ref_get_scan(): hands out the reference's scan/iscan tables
 *    (av1_scan_orders, EbTransforms.h:3349) so tests can pin the oracle's
 *    rule-generated scans against them.
 *  - ref_bench_fwd_quant_sad(): the CPU baseline of the headline metric —
 *    per block, from a pthread pool, exactly the production call sequence
 *    (BASELINE.md §4): ResidualKernel_avx2 -> av1_fwd_txfm2d_32x32_avx2 ->
 *    aom_highbd_quantize_b_32x32_avx2 -> compute32x_m_sad_avx2_intrin
 *    (or the scalar-C column when avx2 == 0).
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <pthread.h>
#include <time.h>
#include "EbDefinitions.h"
#include "EbTransforms.h"

const int16_t *ref_get_scan(int tx_size, int tx_type, int want_iscan) {
    const SCAN_ORDER *so = &av1_scan_orders[tx_size][tx_type];
    return want_iscan ? so->iscan : so->scan;
}

/* reference prototypes (called directly, never through the RTCD pointers) */
void ResidualKernel_avx2(uint8_t *input, uint32_t input_stride, uint8_t *pred, uint32_t pred_stride,
                         int16_t *residual, uint32_t residual_stride, uint32_t area_width, uint32_t area_height);
void residual_kernel_c(uint8_t *input, uint32_t input_stride, uint8_t *pred, uint32_t pred_stride,
                       int16_t *residual, uint32_t residual_stride, uint32_t area_width, uint32_t area_height);
void av1_fwd_txfm2d_32x32_avx2(int16_t *input, int32_t *output, uint32_t stride, TxType tx_type, uint8_t bd);
void Av1TransformTwoD_32x32_c(int16_t *input, int32_t *output, uint32_t stride, TxType tx_type, uint8_t bd);
void aom_highbd_quantize_b_32x32_avx2(const tran_low_t *coeff_ptr, intptr_t n_coeffs, int skip_block,
    const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr, const int16_t *quant_shift_ptr,
    tran_low_t *qcoeff_ptr, tran_low_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr,
    const int16_t *scan, const int16_t *iscan);
void aom_highbd_quantize_b_32x32_c(const tran_low_t *coeff_ptr, intptr_t n_coeffs, int32_t skip_block,
    const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr, const int16_t *quant_shift_ptr,
    tran_low_t *qcoeff_ptr, tran_low_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr,
    const int16_t *scan, const int16_t *iscan);
uint32_t compute32x_m_sad_avx2_intrin(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                                      uint32_t ref_stride, uint32_t height, uint32_t width);
uint32_t fast_loop_nx_m_sad_kernel(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                                   uint32_t ref_stride, uint32_t height, uint32_t width);

typedef struct {
    const uint8_t *src, *pred;       /* n blocks x 1024 bytes, block-major */
    int32_t *coeff, *qcoeff, *dqcoeff; /* n x 1024 (may be NULL: per-thread scratch) */
    uint16_t *eob; uint32_t *sad;
    const int16_t *zbin, *round, *quant, *quant_shift, *dequant; /* int16[8], 16-B aligned */
    size_t begin, end; int avx2; int failed;
} ChainJob;

static void *chain_worker(void *arg) {
    ChainJob *j = (ChainJob *)arg;
    const int16_t *scan = av1_scan_orders[TX_32X32][DCT_DCT].scan;
    const int16_t *iscan = av1_scan_orders[TX_32X32][DCT_DCT].iscan;
    int16_t *res = NULL; int32_t *sc = NULL;
    if (posix_memalign((void **)&res, 32, 1024 * sizeof(int16_t)) ||
        posix_memalign((void **)&sc, 32, 3 * 1024 * sizeof(int32_t))) {
        /* fail loudly: the caller sees failed != 0 and reports no number */
        fprintf(stderr, "ref_bench: posix_memalign failed in a worker (blocks %zu..%zu not computed)\n", j->begin, j->end);
        free(res);
        j->failed = 1;
        return NULL;
    }
    for (size_t b = j->begin; b < j->end; b++) {
        uint8_t *s = (uint8_t *)j->src + b * 1024, *p = (uint8_t *)j->pred + b * 1024;
        /* always compute into 32-B aligned scratch (the AVX2 kernels use aligned
         * stores, as the encoder's own buffers are aligned); copy out if asked */
        int32_t *co = sc, *qc = sc + 1024, *dq = sc + 2048;
        uint16_t eob;
        if (j->avx2) {
            ResidualKernel_avx2(s, 32, p, 32, res, 32, 32, 32);
            av1_fwd_txfm2d_32x32_avx2(res, co, 32, DCT_DCT, 8);
            aom_highbd_quantize_b_32x32_avx2(co, 1024, 0, j->zbin, j->round, j->quant, j->quant_shift,
                                             qc, dq, j->dequant, &eob, scan, iscan);
            j->sad[b] = compute32x_m_sad_avx2_intrin(s, 32, p, 32, 32, 32);
        } else {
            residual_kernel_c(s, 32, p, 32, res, 32, 32, 32);
            Av1TransformTwoD_32x32_c(res, co, 32, DCT_DCT, 8);
            aom_highbd_quantize_b_32x32_c(co, 1024, 0, j->zbin, j->round, j->quant, j->quant_shift,
                                          qc, dq, j->dequant, &eob, scan, iscan);
            j->sad[b] = fast_loop_nx_m_sad_kernel(s, 32, p, 32, 32, 32);
        }
        j->eob[b] = eob;
        if (j->coeff) memcpy(j->coeff + b * 1024, co, 4096);
        if (j->qcoeff) memcpy(j->qcoeff + b * 1024, qc, 4096);
        if (j->dqcoeff) memcpy(j->dqcoeff + b * 1024, dq, 4096);
    }
    free(res); free(sc);
    return NULL;
}

/* Threads the last ref_bench_fwd_quant_sad call really ran on (pthread_create can fail with EAGAIN on a box
 * that limits processes / threads: round 1's first hardware run asked for 256 threads beside torch's own pools
 * and joined pthread_t slots that had never been created). */
static int g_threads_used;
int ref_bench_threads_used(void) { return g_threads_used; }

/* returns elapsed seconds for n blocks on up to `threads` pthreads (the calling thread works too, and takes
 * over the share of every thread that could not be created); a negative value = a worker failed, no number. */
double ref_bench_fwd_quant_sad(const uint8_t *src, const uint8_t *pred, size_t n, int threads, int avx2,
                               const int16_t *zbin, const int16_t *round, const int16_t *quant,
                               const int16_t *quant_shift, const int16_t *dequant,
                               int32_t *coeff, int32_t *qcoeff, int32_t *dqcoeff,
                               uint16_t *eob, uint32_t *sad) {
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    char *created = (char *)calloc((size_t)threads, 1);
    ChainJob *jobs = (ChainJob *)calloc((size_t)threads, sizeof(ChainJob));
    if (!th || !created || !jobs) {
        fprintf(stderr, "ref_bench: out of memory for %d job slots\n", threads);
        free(th); free(created); free(jobs);
        return -1.0;
    }
    struct timespec t0, t1;
    int used = 1, failed = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        ChainJob j = {src, pred, coeff, qcoeff, dqcoeff, eob, sad, zbin, round, quant, quant_shift, dequant,
                      n * (size_t)t / (size_t)threads, n * (size_t)(t + 1) / (size_t)threads, avx2, 0};
        jobs[t] = j;
    }
    /* slot 0 is the calling thread's own share */
    for (int t = 1; t < threads; t++) {
        int rc = pthread_create(&th[t], NULL, chain_worker, &jobs[t]);
        if (rc == 0) { created[t] = 1; used++; }
    }
    chain_worker(&jobs[0]);
    for (int t = 1; t < threads; t++)
        if (!created[t]) chain_worker(&jobs[t]);      /* shares of threads that never started */
    for (int t = 1; t < threads; t++)
        if (created[t]) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int t = 0; t < threads; t++) failed |= jobs[t].failed;
    if (used < threads)
        fprintf(stderr, "ref_bench: only %d of %d threads could be created; the rest of the work ran on the caller\n",
                used, threads);
    g_threads_used = used;
    free(th); free(created); free(jobs);
    if (failed) return -1.0;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
