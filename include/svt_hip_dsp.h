/*
 * svt_hip_dsp.h — C ABI of libsvt_hip_dsp.so: the MI355X (gfx950) implementation
 * of SVT-AV1's block-DSP hot path.  Plain C, plain pointers and sizes; no HIP or
 * torch types appear in any signature (streams travel as void*).
 *
 * Two layers:
 *
 *  (A) DROP-IN entry points — the exact signatures of the reference's dispatch
 *      slots (RTCD globals in Source/Lib/Common/Codec/aom_dsp_rtcd.h and the
 *      *_funcPtrArray[asm_type] tables).  HOST pointers, synchronous, re-entrant
 *      (per-thread stream + staging).  They exist so that the reference's own
 *      call sites and unit tests can run unchanged against this library; one
 *      block per call cannot be fast on a GPU.
 *
 *  (B) BATCHED entry points — arrays of blocks per launch on DEVICE-resident
 *      buffers with an explicit stream; this is where the throughput is.
 *
 * There is no CPU fallback anywhere: when the HIP runtime / device is not
 * usable, (B) returns SVT_HIP_ERR_* and (A), whose reference signatures have no
 * error channel, print the error to stderr and abort().
 *
 * Numeric contract: every output is bit-exact with the reference's C / AVX2
 * kernels (all arithmetic on this path is integer).  8-bit quantisation follows
 * the reference's PRODUCTION slot, i.e. the high-bit-depth algorithm that
 * setup_rtcd_internal installs under AVX2 (aom_dsp_rtcd.h:3330-3334), not the
 * INT16-clamping aom_quantize_b_c_II.
 */
#ifndef SVT_HIP_DSP_H
#define SVT_HIP_DSP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference types (Source/Lib/Common/Codec/EbDefinitions.h) ------------- */
typedef uint8_t svt_tx_size_t;   /* TxSize, packed enum  (EbDefinitions.h:615-650) */
typedef uint8_t svt_tx_type_t;   /* TxType, packed enum  (EbDefinitions.h:727-746) */
typedef int32_t svt_tran_low_t;  /* tran_low_t           (EbDefinitions.h:662)     */

enum { SVT_TX_4X4, SVT_TX_8X8, SVT_TX_16X16, SVT_TX_32X32, SVT_TX_64X64, SVT_TX_4X8, SVT_TX_8X4,
       SVT_TX_8X16, SVT_TX_16X8, SVT_TX_16X32, SVT_TX_32X16, SVT_TX_32X64, SVT_TX_64X32,
       SVT_TX_4X16, SVT_TX_16X4, SVT_TX_8X32, SVT_TX_32X8, SVT_TX_16X64, SVT_TX_64X16, SVT_TX_SIZES_ALL };
enum { SVT_DCT_DCT, SVT_ADST_DCT, SVT_DCT_ADST, SVT_ADST_ADST, SVT_FLIPADST_DCT, SVT_DCT_FLIPADST,
       SVT_FLIPADST_FLIPADST, SVT_ADST_FLIPADST, SVT_FLIPADST_ADST, SVT_IDTX, SVT_V_DCT, SVT_H_DCT,
       SVT_V_ADST, SVT_H_ADST, SVT_V_FLIPADST, SVT_H_FLIPADST, SVT_TX_TYPES };

/* TxfmParam (EbDefinitions.h:764-776), same layout */
typedef struct svt_txfm_param {
    svt_tx_type_t tx_type;
    svt_tx_size_t tx_size;
    int32_t lossless;
    int32_t bd;
    int32_t is_hbd;
    uint8_t tx_set_type;
    int32_t eob;
} svt_txfm_param;

/* ---- status ---------------------------------------------------------------- */
enum {
    SVT_HIP_OK = 0,
    SVT_HIP_ERR_NO_DEVICE = -1,     /* HIP runtime or gfx950 device unavailable */
    SVT_HIP_ERR_INVALID = -2,       /* bad argument (size/type not defined by AV1, NULL, ...) */
    SVT_HIP_ERR_UNSUPPORTED = -3,   /* defined by the reference but not built yet */
    SVT_HIP_ERR_RUNTIME = -4        /* a HIP call failed; see svt_hip_last_error() */
};

/* Initialise on HIP device `device` (call once per process, before any worker
 * thread uses the library — mirrors eb_init_encoder's single-threaded RTCD fill,
 * EbEncHandle.c:917).  Idempotent. */
int svt_hip_init(int device);
void svt_hip_shutdown(void);
/* thread-local, NUL-terminated, never NULL */
const char *svt_hip_last_error(void);
/* "gfx950 ..." description of the device in use (empty string before init) */
const char *svt_hip_device_name(void);

/* Performance-tuning / A-B knobs; they never change results, only which kernel variant runs
 * (process-wide, not thread-safe: set them before the worker threads start).  Keys:
 *   "f32_wg_per_cu" (persistent grid = CUs x
 *   this; 0 = one-shot grid), "f32_nt" (non-temporal stores), "f32_qmode1" (general 24-bit quantiser form),
 *   "no_staged", "no_f32p", "no_enc_staged", "no_inv_planes" (1 = take the general un-staged / two-kernel path),
 *   "no_qsad", "no_q2", "no_q16" (1 = take the first-generation search kernels), "me_exact" (1 = svt_hip_me_fullpel_search_batch always takes its general search-point-by-search-point kernel), "q2_su4" (1 = two-step window staging), "ois_no_fold" (1 = directional predictions through scratch),
 *   "ois_no_dir3" (1 = the open-loop search's three directional zones as three launches instead of one),
 *   "ois_no_nd_multi" (1 = svt_hip_ois_search_frame: one non-directional launch per block size instead of one for the picture),
 *   "frame_single_launch" (svt_hip_encode_recon_frame: 0 = per-size launches, 1 = one launch, 2 = one launch per register class, -1 = by call size),
 *   "inv32_waves", "inv32_var" (probe variants of the inverse 32x32 kernel, tools/tune_inv32.py).
 * Unknown keys return SVT_HIP_ERR_INVALID. */
int svt_hip_tune(const char *key, int value);

/* device memory helpers for C hosts that do not link the HIP runtime */
void *svt_hip_malloc(size_t bytes);
void svt_hip_free(void *dptr);
int svt_hip_memcpy_h2d(void *dptr, const void *hptr, size_t bytes, void *stream);
int svt_hip_memcpy_d2h(void *hptr, const void *dptr, size_t bytes, void *stream);
int svt_hip_stream_sync(void *stream);
/* Allocates n device buffers as SEPARATE allocations with a temporary spacer of `gap_bytes` (0 = the default, 32 GiB) between
 * consecutive ones, freed again before the call returns.  Why: a kernel that writes several large arrays at once (the fused chain:
 * coeff, qcoeff, dqcoeff, 4 GiB each at 2^20 blocks) runs 20 - 25 % slower when those arrays are slices of ONE allocation - one
 * contiguous block of device memory - than when at least one of them lies in another block (MI355X: 350 - 375 against 420 - 460 M
 * blocks/s; no skew or offset inside the block changes it; DESIGN.md 5, profiles/r03_placement_probe*.log).  Separate allocations
 * are in the fast band either way (420 - 455 measured for three plain svt_hip_malloc calls and for this call alike); with torch's
 * allocator 32 GiB spacers were the reliable recipe.  If a spacer cannot be allocated the buffers simply follow each other.
 * Each pointer is freed with svt_hip_free. */
int svt_hip_malloc_spread(const size_t *bytes, int n, size_t gap_bytes, void **ptrs);
/* Box calibration for the roofline report (bench.py): streams `bytes` through HBM in the access shape the kernels of this
 * library use (one 16-byte access per lane, a grid as large as the job, non-temporal stores) so that a measured kernel rate
 * can be put next to what THIS device delivers today.  mode 0 = fill `bytes` of dst; 1 = copy `bytes` from src to dst;
 * 2 = the headline kernel's 1 : 6 read / write mix (reads `bytes` from src, writes 6 x `bytes` to dst).  `bytes` is a
 * multiple of 16, pointers 16-byte aligned.  Enqueues one kernel on `stream`. */
int svt_hip_membw_probe(int mode, void *dst, const void *src, size_t bytes, void *stream);
/* ... and the fused 32x32 chain's own traffic, nothing else: per block 1 KiB read from each of d_in0 / d_in1 (nblocks x 1 KiB each) and
 * 4 KiB written to each of d_out0 / d_out1 / d_out2 (nblocks x 4 KiB each) as 1 KiB stores of one wave - the time of this launch on
 * the caller's own arrays is what the memory system gives that stream pattern in that placement. */
int svt_hip_membw_probe_chain(const void *d_in0, const void *d_in1, void *d_out0, void *d_out1, void *d_out2, size_t nblocks,
                              void *stream);

/* ---- host-side tables for callers outside the encoder (csrc/host_tables.cpp; no device involved) ------------------
 * y-plane quantiser rows of av1_build_quantizer(bit_depth, 0, 0, 0, 0, 0) (EbModeDecisionConfigurationProcess.c:429-520):
 * five int16 [256][8] tables indexed [qindex][0 = DC, 1..7 = AC], the rows the quantiser entry points take. */
int svt_hip_build_quantizer(int bit_depth, int16_t (*zbin)[8], int16_t (*round)[8], int16_t (*quant)[8],
                            int16_t (*quant_shift)[8], int16_t (*dequant)[8]);
/* av1_scan_orders[tx_size][tx_type] (EbTransforms.h:3349-3870): scan / iscan over the kept min(W,32) x min(H,32)
 * coefficients (either pointer may be NULL); returns their number (<= 1024) or SVT_HIP_ERR_INVALID. */
int svt_hip_get_scan(int tx_size, int tx_type, int16_t *scan, int16_t *iscan);
/* the candidate list open_loop_intra_search_sb enumerates for one block size (EbMotionEstimation.c:8747-8846): modes in AV1
 * PredictionMode numbering, angle deltas -2..2; arrays of SVT_HIP_OIS_MAX_CANDIDATES; returns the count. */
#define SVT_HIP_OIS_MAX_CANDIDATES 61
int svt_hip_ois_candidates(uint32_t bsize, int temporal_layer_index, int intra_pred_mode, int is_used_as_reference,
                           int is_16bit, uint8_t *modes, int8_t *angle_deltas);

/* ============================================================================
 * (B) batched API — device pointers, `stream` is a hipStream_t (NULL = default)
 * Every call only enqueues work; it returns before the GPU finishes.
 * ==========================================================================*/

/* K1 forward 2-D transform.  Replaces av1_fwd_txfm2d_WxH
 * (aom_dsp_rtcd.h:160-255; C: EbTransforms.c:4410-4910; AVX2:
 * highbd_fwd_txfm_avx2.c:762-5814) over `nblocks` blocks.
 * d_in: int16 residual, block b at d_in + b*in_block_pitch, row stride in_stride
 * (elements).  d_out: dense W*H int32 per block, row-major, full size (64-pt
 * outputs are NOT packed; see svt_hip_pack64_batch). */
int svt_hip_fwd_txfm2d_batch(const int16_t *d_in, uint32_t in_stride, size_t in_block_pitch,
                             int32_t *d_out, size_t nblocks, int tx_size, int tx_type, int bd,
                             void *stream);

/* 64-pt handling of av1_estimate_transform (EbTransforms.c:4377-4408,
 * 4580-4731, 5157-5175): per block, energy of the discarded region -> d_energy
 * (may be NULL), then re-pack the kept min(W,32) x min(H,32) region in place and
 * zero the tail.  No-op (energy 0) for sizes without a 64 dimension. */
int svt_hip_pack64_batch(int32_t *d_coeff, uint64_t *d_energy, size_t nblocks, int tx_size,
                         void *stream);

/* K2 inverse 2-D transform + add.  Replaces av1_inv_txfm2d_add_WxH
 * (aom_dsp_rtcd.h:350-417; C: EbTransforms.c:8180-8458) and, with
 * dst_is_16bit = 0, the 8-bit recon entry av1_inv_txfm_add (:8882-8903).
 * d_coeff: dense packed min(W,32)*min(H,32) int32 per block.
 * d_dst: uint16 (dst_is_16bit) or uint8 samples; block b lives at
 * d_dst + (d_dst_offsets ? d_dst_offsets[b] : b*dst_block_pitch), row stride
 * dst_stride (all in elements).  Blocks must not overlap. */
int svt_hip_inv_txfm2d_add_batch(const int32_t *d_coeff, void *d_dst, int dst_is_16bit,
                                 int32_t dst_stride, size_t dst_block_pitch,
                                 const uint32_t *d_dst_offsets, size_t nblocks, int tx_size,
                                 int tx_type, int bd, void *stream);

/* K3 quantize + dequantize + eob.  Replaces aom_highbd_quantize_b{,_32x32,_64x64}
 * (aom_dsp_rtcd.h:323-342; C: EbFullLoop.c:239-333; AVX2:
 * highbd_quantize_intrin_avx2.c:127-484) over nblocks dense blocks of n_coeffs.
 * zbin/round/quant/quant_shift/dequant: HOST pointers to the reference's
 * int16[8] table rows ([0] = DC, [1] = AC).  d_iscan: DEVICE int16[n_coeffs].
 * log_scale: 0 (aom_highbd_quantize_b), 1 (_32x32), 2 (_64x64). */
int svt_hip_quantize_b_batch(const int32_t *d_coeff, size_t n_coeffs, int skip_block,
                             const int16_t *zbin, const int16_t *round, const int16_t *quant,
                             const int16_t *quant_shift, int32_t *d_qcoeff, int32_t *d_dqcoeff,
                             const int16_t *dequant, uint16_t *d_eob, const int16_t *d_iscan,
                             int log_scale, size_t nblocks, void *stream);

/* Headline chain (BASELINE.json metric), fused in one kernel: per block
 *   residual = src - pred            ResidualKernel   (EbCodingLoop.c:617)
 *   coeff    = FwdTxfm2d(residual)   av1_estimate_transform (EbFullLoop.c:763)
 *   q,dq,eob = quantize_b(coeff)     av1_quantize_inv_quantize (EbFullLoop.c:780)
 *   sad      = SAD(src, pred)        NxMSadKernel     (EbProductCodingLoop.c:1259)
 * d_src/d_pred: dense W*H uint8 per block.  Outputs dense per block (packed
 * min(W,32)*min(H,32) for 64-pt sizes); d_sad may be NULL.  TX_32X32/DCT_DCT runs the
 * tuned fused kernel, every other size/type the generic fused kernel. */
int svt_hip_fwd_quant_sad_batch(const uint8_t *d_src, const uint8_t *d_pred, size_t nblocks,
                                int tx_size, int tx_type, const int16_t *zbin,
                                const int16_t *round, const int16_t *quant,
                                const int16_t *quant_shift, const int16_t *dequant,
                                const int16_t *d_iscan, int32_t *d_coeff, int32_t *d_qcoeff,
                                int32_t *d_dqcoeff, uint16_t *d_eob, uint32_t *d_sad, void *stream);

/* Encode-pass chain of the transform blocks of one prediction, as the reference runs it per block
 * in Av1EncodeLoop (EbCodingLoop.c:545-950):
 *   residual = src - pred              ResidualKernel                 (EbCodingLoop.c:617)
 *   coeff    = FwdTxfm2d(residual)     av1_estimate_transform         (EbCodingLoop.c:632)
 *   q,dq,eob = quantize_b(coeff)       av1_quantize_inv_quantize      (EbCodingLoop.c:647)
 *   recon    = pred + InvTxfm2d(dq)    av1_inv_transform_recon8bit    (EbCodingLoop.c:753)
 * (+ sad = SAD(src, pred) when d_sad != NULL).  8-bit, dense W*H blocks; d_recon must not alias
 * d_src / d_pred.  TX_32X32 with DCT_DCT or IDTX runs ONE fused kernel in which coefficients and
 * residual never leave the CU (7 174 B of HBM traffic per block instead of 20 486); every other size
 * except 4x4 runs the generic fused kernel of the same structure.  d_coeff and d_dqcoeff are optional in
 * the fused kernels (both NULL = not written).  TX_4X4 runs one lane per block with the whole block in registers (any
 * quantiser table).  For the other sizes, quantiser tables whose quant_shift is not a power of two and buffers that are
 * not 16-B aligned run svt_hip_fwd_quant_sad_batch + a device copy + svt_hip_inv_txfm2d_add_batch and need both buffers. */
int svt_hip_encode_recon_batch(const uint8_t *d_src, const uint8_t *d_pred, size_t nblocks, int tx_size,
                               int tx_type, const int16_t *zbin, const int16_t *round,
                               const int16_t *quant, const int16_t *quant_shift, const int16_t *dequant,
                               const int16_t *d_iscan, int32_t *d_coeff, int32_t *d_qcoeff,
                               int32_t *d_dqcoeff, uint16_t *d_eob, uint32_t *d_sad, uint8_t *d_recon,
                               void *stream);

/* The same chain on picture planes: block b has its top-left sample at (x, y) = (d_xy[b] & 0xffff,
 * d_xy[b] >> 16) of the source, prediction and reconstruction planes (row strides in samples); the
 * reconstruction may be written in place into the prediction plane (d_recon == d_pred with equal
 * strides), as the reference does after pic_copy_kernel (EbCodingLoop.c:741-753).  is_16bit / bd: uint8
 * samples (bd 8) or uint16 samples (bd 10; d_sad must be NULL).  Fused kernels only: for sizes other than 4x4,
 * non-standard quantiser tables return SVT_HIP_ERR_INVALID (use svt_hip_fwd_quant_planes_batch +
 * svt_hip_inv_txfm2d_add_batch there). */
int svt_hip_encode_recon_planes_batch(const void *d_src, uint32_t src_stride, const void *d_pred,
                                      uint32_t pred_stride, void *d_recon, uint32_t recon_stride,
                                      const uint32_t *d_xy, size_t nblocks, int is_16bit, int bd,
                                      int tx_size, int tx_type,
                                      const int16_t *zbin, const int16_t *round, const int16_t *quant,
                                      const int16_t *quant_shift, const int16_t *dequant,
                                      const int16_t *d_iscan, int32_t *d_coeff, int32_t *d_qcoeff,
                                      int32_t *d_dqcoeff, uint16_t *d_eob, uint32_t *d_sad, void *stream);

/* The encode-pass chain for a whole FRAME (or many frames) in ONE call: every (plane, transform size) group of blocks the
 * frame is cut into - BASELINE.json configs[3] / [4], the per-SB loop of AV1EncodePass (EbCodingLoop.c:2249) turned inside
 * out.  A group is what svt_hip_encode_recon_planes_batch takes; the groups of a call are independent (disjoint blocks), so
 * the library issues them CONCURRENTLY on internal streams forked from, and joined back into, `stream` (a lone 1080p frame
 * gives each size's kernel less than one workgroup per CU: run one after the other the frame is launch-latency-bound).
 * The call only enqueues work and can be captured into a HIP graph by the caller (the fork / join becomes graph branches).
 * One quantiser row set per call (one qindex), as in the per-plane entry point. */
typedef struct svt_hip_frame_group {
    const void *d_src;  uint32_t src_stride;     /* planes: pointer to sample (0, 0), stride in samples */
    const void *d_pred; uint32_t pred_stride;
    void *d_recon;      uint32_t recon_stride;   /* may be d_pred with the same stride: reconstruct in place */
    const uint32_t *d_xy;                        /* block origins x | y << 16 */
    const uint32_t *d_offsets;                   /* unused (NULL) unless svt_hip_tune("no_enc_staged", 1): y * recon_stride + x */
    uint32_t nblocks;
    int32_t tx_size, tx_type;
    const int16_t *d_iscan;                      /* DEVICE int16 [min(W,32) * min(H,32)] for this size / type */
    int32_t *d_qcoeff;  uint16_t *d_eob;         /* dense per-group outputs */
    int32_t *d_coeff, *d_dqcoeff;                /* optional, both or neither */
} svt_hip_frame_group;
int svt_hip_encode_recon_frame(const svt_hip_frame_group *groups, int ngroups, int is_16bit, int bd,
                               const int16_t *zbin, const int16_t *round, const int16_t *quant,
                               const int16_t *quant_shift, const int16_t *dequant, void *stream);

/* The frame call with the two encode-pass pieces that sit beside the transform chain:
 *
 *   chroma from luma  (Av1EncodeLoop, EbCodingLoop.c:736-846; 16-bit :1198-1250).  For a chroma block predicted with
 *       UV_CFL_PRED the reference, after the luma transform block's reconstruction, calls cfl_luma_subsampling_420_{lbd,hbd}
 *       on the luma RECONSTRUCTION, subtract_average, and cfl_predict_{lbd,hbd} on the Cb and the Cr prediction IN PLACE with
 *       the block's two alphas; the chroma residual / transform / quantisation then runs on that prediction.  Here: groups
 *       [0, first_chroma_group) are encoded first (the luma groups), then ONE launch does the three steps for every block of
 *       every cfl group (Q3 values stay in registers), then groups [first_chroma_group, ngroups) are encoded.  With ncfl == 0
 *       the split is ignored and all groups go out together.
 *   av1_txb_init_levels  (EbRateDistortionCost.c:125-150, called per coded block by the coefficient cost / entropy stage,
 *       :450) of every group's quantised coefficients: levels[g] (NULL, or one entry per group; an entry with a NULL buffer
 *       skips its group) names one whole padded level buffer per block, (w + 4) * (h + 6) + 16 bytes with w, h the packed
 *       coefficient block's sides (min(.., 32)), as in svt_hip_txb_init_levels_batch.  ONE launch after the encode launches.
 *
 * Everything is enqueued on `stream`, in the order above (stream order carries the dependencies); arguments are validated
 * before the first launch. */
typedef struct svt_hip_frame_cfl_group {
    const void *d_luma_recon; uint32_t luma_stride;     /* the plane the luma groups reconstruct into; samples as the groups' */
    void *d_pred_cb; uint32_t pred_stride_cb;           /* chroma prediction planes (the chroma groups' d_pred): the DC prediction */
    void *d_pred_cr; uint32_t pred_stride_cr;           /* on entry, the chroma-from-luma prediction on return */
    const uint32_t *d_xy;                               /* chroma block origins x | y << 16; the luma area starts at (2x, 2y) */
    const int32_t *d_alpha_q3_cb, *d_alpha_q3_cr;       /* cfl_idx_to_alpha(.., CFL_PRED_U / CFL_PRED_V) per block */
    uint32_t width, height;                             /* chroma block (tx_width_uv x tx_height_uv): 4 .. 32 */
    uint32_t nblocks;
} svt_hip_frame_cfl_group;
typedef struct svt_hip_frame_levels {
    uint8_t *d_levels_buf; size_t levels_block_pitch;   /* >= (w + 4) * (h + 6) + 16, a multiple of 4; 16-byte alignment is faster */
} svt_hip_frame_levels;
int svt_hip_encode_recon_frame_ex(const svt_hip_frame_group *groups, int ngroups, int first_chroma_group,
                                  const svt_hip_frame_cfl_group *cfl, int ncfl, const svt_hip_frame_levels *levels,
                                  int is_16bit, int bd, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                                  const int16_t *quant_shift, const int16_t *dequant, void *stream);

/* BASELINE.json configs[1]: FwdTxfm2d + quantize on a batch of int16 residual blocks
 * (dense W*H per block) — av1_estimate_transform + av1_quantize_inv_quantize
 * (EbFullLoop.c:763, 780).  TX_32X32 8-bit runs the tuned fused kernel. */
int svt_hip_fwd_quant_batch(const int16_t *d_residual, size_t nblocks, int tx_size, int tx_type,
                            int bd, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                            const int16_t *quant_shift, const int16_t *dequant,
                            const int16_t *d_iscan, int32_t *d_coeff, int32_t *d_qcoeff,
                            int32_t *d_dqcoeff, uint16_t *d_eob, void *stream);

/* The same chain for EVERY transform size / type and for 8- or 16-bit sample planes
 * (Av1EncodeLoop EbCodingLoop.c:545-950, Av1EncodeLoop16bit :1020-1351): blocks are
 * addressed inside planes by d_xy[b] = (y << 16) | x (or, when d_xy == NULL, block b
 * starts at b * W*H elements with row stride W: the dense layout above).  Outputs are
 * dense, PACKED min(W,32)*min(H,32) coefficients per block; for 64-pt sizes d_energy
 * (may be NULL) receives av1_estimate_transform's three_quad_energy.  d_sad (may be
 * NULL) is defined for 8-bit planes only. */
int svt_hip_fwd_quant_planes_batch(const void *d_src, uint32_t src_stride, const void *d_pred,
                                   uint32_t pred_stride, const uint32_t *d_xy, size_t nblocks,
                                   int is_16bit, int bd, int tx_size, int tx_type,
                                   const int16_t *zbin, const int16_t *round, const int16_t *quant,
                                   const int16_t *quant_shift, const int16_t *dequant,
                                   const int16_t *d_iscan, int32_t *d_coeff, int32_t *d_qcoeff,
                                   int32_t *d_dqcoeff, uint16_t *d_eob, uint32_t *d_sad,
                                   uint64_t *d_energy, void *stream);

/* K4 NxM SAD (NxMSadKernel_funcPtrArray, EbComputeSAD.h:105-160; C:
 * EbComputeSAD_C.c:48) and K7 SSE (spatial_full_distortion_kernel,
 * EbPictureOperators_C.c:40) over nblocks block pairs; d_out: uint32 / uint64. */
int svt_hip_sad_batch(const uint8_t *d_src, uint32_t src_stride, size_t src_block_pitch,
                      const uint8_t *d_ref, uint32_t ref_stride, size_t ref_block_pitch,
                      uint32_t width, uint32_t height, uint32_t *d_out, size_t nblocks, void *stream);
int svt_hip_sse_batch(const uint8_t *d_a, uint32_t a_stride, size_t a_block_pitch,
                      const uint8_t *d_b, uint32_t b_stride, size_t b_block_pitch, uint32_t width,
                      uint32_t height, uint64_t *d_out, size_t nblocks, void *stream);
/* K8 residual (ResidualKernel, aom_dsp_rtcd.h:2372; C: EbPictureOperators.c:166) */
int svt_hip_residual_batch(const uint8_t *d_src, uint32_t src_stride, size_t src_block_pitch,
                           const uint8_t *d_pred, uint32_t pred_stride, size_t pred_block_pitch,
                           int16_t *d_res, uint32_t res_stride, size_t res_block_pitch,
                           uint32_t width, uint32_t height, size_t nblocks, void *stream);

/* residual_kernel16bit (EbPictureOperators.c:134-164): the same on 16-bit samples (Av1EncodeLoop16bit, EbCodingLoop.c:1085) */
int svt_hip_residual16_batch(const uint16_t *d_src, uint32_t src_stride, size_t src_block_pitch,
                             const uint16_t *d_pred, uint32_t pred_stride, size_t pred_block_pitch,
                             int16_t *d_res, uint32_t res_stride, size_t res_block_pitch,
                             uint32_t width, uint32_t height, size_t nblocks, void *stream);
/* K4 on picture planes (block b at plane + d_*_offsets[b], byte offsets) */
int svt_hip_sad_planes_batch(const uint8_t *d_src_plane, uint32_t src_stride, const uint32_t *d_src_offsets,
                             const uint8_t *d_ref_plane, uint32_t ref_stride, const uint32_t *d_ref_offsets,
                             uint32_t width, uint32_t height, uint32_t *d_out, size_t nblocks, void *stream);
/* aom_sadMxNx4d (aom_dsp_rtcd.h:1328-1500; C: C_DEFAULT/EbComputeSAD_C.c:151-160): per source block, SADs against FOUR
 * reference positions, d_ref_plane + d_ref_offsets[4*b + i]; d_out uint32 [nblocks][4].  Source blocks dense
 * (d_src_offsets == NULL: b * src_block_pitch) or at d_src + d_src_offsets[b]. */
int svt_hip_sad_x4d_batch(const uint8_t *d_src, uint32_t src_stride, size_t src_block_pitch,
                          const uint32_t *d_src_offsets, const uint8_t *d_ref_plane, uint32_t ref_stride,
                          const uint32_t *d_ref_offsets, uint32_t width, uint32_t height, uint32_t *d_out,
                          size_t nblocks, void *stream);
/* combined_averaging_sad (NxMSadAveragingKernel_funcPtrArray, EbComputeSAD.h:162-197; C: C_DEFAULT/EbComputeSAD_C.c:13-40;
 * called by BiPredictionSearch, EbMotionEstimation.c:6639): SAD(src, (ref1 + ref2 + 1) >> 1) */
int svt_hip_sad_avg_batch(const uint8_t *d_src, uint32_t src_stride, size_t src_block_pitch, const uint8_t *d_ref1,
                          uint32_t ref1_stride, size_t ref1_block_pitch, const uint8_t *d_ref2,
                          uint32_t ref2_stride, size_t ref2_block_pitch, uint32_t width, uint32_t height,
                          uint32_t *d_out, size_t nblocks, void *stream);

/* K5 SAD search (NxMSadLoopKernel_funcPtrArray, EbComputeSAD.h:199; C:
 * sad_loop_kernel, EbComputeSAD_C.c:72-120).  Per block b: source block at
 * d_src + b*src_block_pitch, reference window origin at d_ref + b*ref_block_pitch.
 * Outputs: best_sad (uint64), x/y search centre (int16), first minimum in raster
 * order.  Limits: width, height <= 64; window must fit the LDS budget
 * (else SVT_HIP_ERR_INVALID). */
int svt_hip_sad_search_batch(const uint8_t *d_src, uint32_t src_stride, size_t src_block_pitch,
                             const uint8_t *d_ref, uint32_t ref_stride, uint32_t ref_stride_raw,
                             size_t ref_block_pitch, uint32_t width, uint32_t height,
                             int16_t search_area_width, int16_t search_area_height,
                             uint64_t *d_best_sad, int16_t *d_x, int16_t *d_y, size_t nblocks,
                             void *stream);

/* The same search addressed on picture planes (frame-level HME): block b's source
 * block starts at d_src_plane + d_src_offsets[b] and its search window at d_ref_plane +
 * d_ref_offsets[b] (uint32 byte offsets; the caller clips windows to the padded picture exactly as
 * HmeLevel0 does, EbMotionEstimation.c:5720-5798). */
int svt_hip_sad_search_planes_batch(const uint8_t *d_src_plane, uint32_t src_stride,
                                    const uint32_t *d_src_offsets, const uint8_t *d_ref_plane,
                                    uint32_t ref_stride, uint32_t ref_stride_raw,
                                    const uint32_t *d_ref_offsets, uint32_t width, uint32_t height,
                                    int16_t search_area_width, int16_t search_area_height,
                                    uint64_t *d_best_sad, int16_t *d_x, int16_t *d_y, size_t nblocks,
                                    void *stream);

/* K6 ME multi-size SAD: full-pel search of 64x64 superblocks for all 85 PUs.
 * Replaces FullPelSearch_LCU's inner calls (EbMotionEstimation.c:3199-3247 ->
 * GetSearchPointResults :2932 -> SadCalculation_8x8_16x16 / _32x32_64x64 tables :145-199;
 * C: :208-311).  Per SB b: source d_src + b*src_block_pitch (64x64, src_stride),
 * reference window origin d_ref + b*ref_block_pitch (ref_stride), search area
 * search_w x search_h (<= 4096 points), area origin (x, y) from d_origins[b][2] or,
 * when NULL, the uniform x_origin / y_origin.  d_best_sad / d_best_mv: uint32[n][85],
 * IN/OUT running bests exactly like the reference's p_best_sad8x8[64] | 16x16[16] |
 * 32x32[4] | 64x64[1] (and p_best_mv*) arrays back to back; the caller initialises
 * them (reference: MAX_SAD_VALUE, EbMotionEstimation.h:79). */
#define SVT_HIP_ME_PUS 85
int svt_hip_me_sb_search_batch(const uint8_t *d_src, uint32_t src_stride, size_t src_block_pitch,
                               const uint8_t *d_ref, uint32_t ref_stride, size_t ref_block_pitch,
                               int search_w, int search_h, const int16_t *d_origins, int x_origin,
                               int y_origin, uint32_t *d_best_sad, uint32_t *d_best_mv,
                               size_t nblocks, void *stream);

/* K6 on picture planes: SB b's source at d_src_plane + d_src_offsets[b], window origin at d_ref_plane +
 * d_ref_offsets[b] (all SBs x reference pictures of a segment in one launch; FullPelSearch_LCU's
 * pointer arithmetic, EbMotionEstimation.c:3210-3225, done once on the host). */
int svt_hip_me_sb_search_planes_batch(const uint8_t *d_src_plane, uint32_t src_stride,
                                      const uint32_t *d_src_offsets, const uint8_t *d_ref_plane,
                                      uint32_t ref_stride, const uint32_t *d_ref_offsets, int search_w,
                                      int search_h, const int16_t *d_origins, int x_origin, int y_origin,
                                      uint32_t *d_best_sad, uint32_t *d_best_mv, size_t nblocks,
                                      void *stream);

/* Hierarchical motion estimation, one LEVEL per call for all SBs (x references x search regions, as separate tasks) of a
 * picture: HmeLevel0 / HmeLevel1 / HmeLevel2 (EbMotionEstimation.c:5689, 5883, 6016; called per SB at :7739-7830) INCLUDING
 * the search-area placement and clipping the reference does per SB on the host (:5729-5798).  Per task t:
 *   d_sb_origin[t] = (x, y) of the SB and d_sb_size[t] = (width, height), at the level's resolution (origin >> 2 / >> 1 / >> 0);
 *   width 1..64, height 2..64 (the reference's SBs); an entry outside that range gets best_sad = 2^64 - 1 and mv = (0, 0);
 *   search centre  = d_centers[t] >> center_shift (NULL: (0, 0)): level 0 takes the caller's centre, level 1 the level-0
 *                    result >> 1, level 2 the level-1 result (the reference's call sites), so the levels chain ON THE DEVICE;
 *   d_src_pic / d_ref_pic point at sample (0, 0) of the level's source / padded reference picture (the reference buffer
 *                    must be readable from -pad to size + pad, as the encoder's padded pictures are);
 *   outputs        d_best_sad[t] (uint64: SAD on every other row x 2) and d_mv[t] = (x, y) scaled to full resolution -
 *                    exactly hmeLevelNSad[][] and x/yHmeLevelNSearchCenter[][] of MotionEstimateLcu.
 * params: svt_hip_hme_level_params() derives them from the encoder's context values. */
typedef struct svt_hip_hme_params {
    int32_t search_area_width, search_area_height;    /* before clipping: L0 ((w * mult / 100) + 15) & ~15, h * mult / 100; L1 / L2 (w + 7) & ~7, h */
    int32_t x_origin_offset, y_origin_offset;         /* origin = offset + centre: L0 -(total * mult / 100 >> 1) + preceding regions; L1 / L2 -(area >> 1) */
    int32_t pad_width, pad_height;                    /* reference origin - 1 (L0, L1), BLOCK_SIZE_64 - 1 (L2) */
    int32_t ref_width, ref_height;                    /* reference picture size at this level */
    int32_t round_down;                               /* 16 (L0) or 8: width rounded down after clipping unless smaller */
    int32_t mv_shift;                                 /* 2, 1, 0 */
} svt_hip_hme_params;
int svt_hip_hme_level_params(int level, const uint16_t *hme_search_area_in_width_array,
                             const uint16_t *hme_search_area_in_height_array, uint32_t region_in_width,
                             uint32_t region_in_height, uint32_t level0_total_search_area_width,
                             uint32_t level0_total_search_area_height, uint32_t search_area_multiplier_x,
                             uint32_t search_area_multiplier_y, uint32_t ref_origin_x, uint32_t ref_origin_y,
                             uint32_t ref_width, uint32_t ref_height, svt_hip_hme_params *params);
int svt_hip_hme_level_batch(const uint8_t *d_src_pic, uint32_t src_stride, const uint8_t *d_ref_pic, uint32_t ref_stride,
                            const int16_t *d_sb_origin, const uint16_t *d_sb_size, const int16_t *d_centers,
                            int center_shift, const svt_hip_hme_params *params, uint64_t *d_best_sad, int16_t *d_mv,
                            size_t ntasks, void *stream);
/* The same for 1 .. 4 search regions of the level in ONE launch (the reference walks number_hme_search_region_in_width x
 * _in_height regions per SB and carries each region's vector through the next levels, EbMotionEstimation.c:7700-7950):
 * params[nregions]; d_centers / d_best_sad / d_mv are [region][task] planes (d_centers may be NULL at level 0). */
int svt_hip_hme_level_regions_batch(const uint8_t *d_src_pic, uint32_t src_stride, const uint8_t *d_ref_pic,
                                    uint32_t ref_stride, const int16_t *d_sb_origin, const uint16_t *d_sb_size,
                                    const int16_t *d_centers, int center_shift, const svt_hip_hme_params *params, int nregions,
                                    uint64_t *d_best_sad, int16_t *d_mv, size_t ntasks, void *stream);

/* K6 with the encoder's own result rows: d_best_sad / d_best_mv hold, per SB, pu_pitch uint32 whose first 85 (nsq = 0)
 * or 209 (nsq != 0) entries are MeContext_t.p_sb_best_sad[list][ref][..] / p_sb_best_mv[..] in EbMeTierZeroPu order
 * (EbMotionEstimationContext.h:47-270): 64x64, 32x32 x4, 16x16 x16, 8x8 x64 and, for the non-square search
 * open_loop_me_fullpel_search_sblock (EbMotionEstimation.c:3251; picked at :8114 when nsq_search_level is between LEVEL1 and
 * FULL), 64x32 x2, 32x16 x8, 16x8 x32, 32x64 x2, 16x32 x8, 8x16 x32, 32x8 x16, 8x32 x16, 64x16 x4, 16x64 x4.  IN/OUT running
 * bests as in svt_hip_me_sb_search_batch (initialise with MAX_SAD_VALUE = 128*128*255).  Sources / windows are dense
 * (d_*_offsets == NULL: block b at b * *_block_pitch) or addressed on planes by byte offsets.
 *
 * flavour: which of the reference's kernel sets the results are bit-identical to.
 *   SVT_HIP_FLAVOUR_C     its scalar C / SSE4.1 kernels (asm_type 0) - the parity target BASELINE.json names;
 *   SVT_HIP_FLAVOUR_AVX2  its AVX2 kernels as GCC / clang compile them (asm_type 1, the only value the encoder accepts,
 *                         EbEncHandle.c:2676).  They differ from the C kernels in three places, all restated exactly
 *                         (oracle/pixel.c, pinned by tests/golden/me.npz): the 32x32 motion vectors of the 8-search-point
 *                         kernel (lane swap under __GNUC__, EbComputeSAD_Intrinsic_AVX2.c:3989-4001) and, for the
 *                         non-square search on widths that are not a multiple of 8, ExtSadCalculation's stale-`sad` test
 *                         (both flavours, EbMotionEstimation.c:732-736) and the AVX2 single-point kernel's source-stride
 *                         row fetch (EbComputeSAD_Intrinsic_AVX2.c:50-52).  With the last one the kernel reads reference
 *                         bytes up to 8 * src_stride + 64 past a 16x16 block's window origin: the caller's reference
 *                         buffer must cover that, as the encoder's padded pictures do. */
enum { SVT_HIP_FLAVOUR_C = 0, SVT_HIP_FLAVOUR_AVX2 = 1 };
#define SVT_HIP_ME_PUS_ALL 209
int svt_hip_me_fullpel_search_batch(const uint8_t *d_src, uint32_t src_stride, size_t src_block_pitch,
                                    const uint32_t *d_src_offsets, const uint8_t *d_ref, uint32_t ref_stride,
                                    size_t ref_block_pitch, const uint32_t *d_ref_offsets, int search_w,
                                    int search_h, const int16_t *d_origins, int x_origin, int y_origin,
                                    int flavour, int nsq, uint32_t *d_best_sad, uint32_t *d_best_mv,
                                    uint32_t pu_pitch, size_t nblocks, void *stream);

/* ---- MotionEstimateLcu's per-SB glue around the full-pel search (EbMotionEstimation.c:7527) ---------------------
 * (1) svt_hip_me_setup_batch: between the HME levels and the search - per task (SB x reference picture) the search centre
 *     (first strict minimum of the last enabled HME level's SADs over its search regions, :7849-7941; for list 1 of a picture
 *     whose two references are the same picture the second entry of the reference's region sort, :7906-7936; SBs that are not
 *     64 rows high take no HME result, :7678), CheckZeroZeroCenter (:6844-6930, run when zz_check = is_used_as_reference_flag)
 *     and the search area: width rounded up to 8, centred, clipped against the picture in the reference's statement order,
 *     width rounded down to 8 unless below 8 (:7955-8040).
 *       d_src_pic / d_ref_pic    sample (0, 0) of the padded 8-bit source / reference luma planes (>= 63 + 64 samples of padding)
 *       d_sb_origin / d_sb_size  [n][2] (x, y) / (width, height) of the SBs
 *       d_hme_sad / d_hme_mv     [region][n] / [region][n][2] outputs of the last enabled level (svt_hip_hme_level_regions_batch),
 *                                region r = rh * regions_w + rw; NULL (or regions 0 x 0): no HME, every centre is (0, 0)
 *       d_center (may be NULL)   [n][2] the centre after CheckZeroZeroCenter;  d_area  [n][4] x_origin, y_origin, width, height
 * (2) svt_hip_me_fullpel_search_areas_batch: svt_hip_me_fullpel_search_batch with ONE AREA PER SB, read on the device from
 *     d_areas (so a picture's interior and clipped edge SBs share one launch and nothing returns to the host in between).
 *     d_ref_offsets[b] is the byte offset of the SB's CO-LOCATED position in the reference plane; the window starts at that
 *     position + the area's origin.  max_search_w / _h bound every area of the call (they size the LDS window: the nominal
 *     area, width rounded up to 8); an area outside 1 .. max is skipped (its rows keep their incoming values).
 * (3) svt_hip_me_bipred_batch: BiPredictionSearch (:6639, integer vectors: the SAD of each PU against the rounded average of
 *     its two lists' best blocks; sub_sad = fractionalSearchMethod == SUB_SAD_SEARCH: every other row, doubled) and the
 *     me_results rows MotionEstimateLcu writes (:8308-8440): vectors, up to three (distortion, direction) candidates in the
 *     reference's order (Sort3Elements :6809), candidate count.  Result rows are in RASTER PU order (partitionWidth /
 *     puSearchIndexMap, EbMotionEstimation.h:178-315) and read the SAD / vector rows at the storage index of the same rectangle
 *     (svt_hip_me_pu_storage_index).  d_best_sad1 / d_best_mv1 NULL: P picture, one candidate.  bipred_all_pus = (cu8x8_mode ==
 *     CU_8x8_MODE_0 || pic_depth_mode <= PIC_ALL_C_DEPTH_MODE): otherwise only PUs 0 .. 20 are bi-predicted (:8297).  npus 85 or
 *     209.  me_nsq[] of the reference's rows is not produced (it is written by the sub-pel stage, which is outside this path). */
typedef struct svt_hip_me_setup_params {
    int32_t picture_width, picture_height;        /* SequenceControlSet luma_width / luma_height: the search area's clip */
    int32_t ref_width, ref_height;                /* refPicPtr->width / height: the clip of the HME centre in CheckZeroZeroCenter */
    int32_t search_area_width, search_area_height;/* MeContext_t values, before the round-up to 8 */
    int32_t regions_w, regions_h;                 /* number_hme_search_region_in_width / _height (1 or 2 each; 0: no HME) */
    int32_t second_best;                          /* HME level 2 on && list 1 && ref_pic_poc_array[0] == [1] (needs regions_w == regions_h) */
    int32_t zz_check;                             /* is_used_as_reference_flag */
} svt_hip_me_setup_params;
int svt_hip_me_setup_batch(const uint8_t *d_src_pic, uint32_t src_stride, const uint8_t *d_ref_pic, uint32_t ref_stride,
                           const int16_t *d_sb_origin, const uint16_t *d_sb_size, const uint64_t *d_hme_sad,
                           const int16_t *d_hme_mv, const svt_hip_me_setup_params *params, int16_t *d_center, int16_t *d_area,
                           size_t ntasks, void *stream);
int svt_hip_me_fullpel_search_areas_batch(const uint8_t *d_src, uint32_t src_stride, const uint32_t *d_src_offsets,
                                          const uint8_t *d_ref, uint32_t ref_stride, const uint32_t *d_ref_offsets,
                                          const int16_t *d_areas, int max_search_w, int max_search_h, int flavour, int nsq,
                                          uint32_t *d_best_sad, uint32_t *d_best_mv, uint32_t pu_pitch, size_t nblocks,
                                          void *stream);
typedef struct svt_hip_me_result {                /* MeCuResults_t (EbMotionEstimationLcuResults.h:62-77) without me_nsq */
    int16_t x_mv_l0, y_mv_l0, x_mv_l1, y_mv_l1;
    uint32_t distortion[3];
    uint8_t direction[3];                         /* UNI_PRED_LIST_0 0, UNI_PRED_LIST_1 1, BI_PRED 2 */
    uint8_t total_me_candidate_index;
} svt_hip_me_result;
int svt_hip_me_bipred_batch(const uint8_t *d_src_pic, uint32_t src_stride, const uint8_t *d_ref0_pic, uint32_t ref0_stride,
                            const uint8_t *d_ref1_pic, uint32_t ref1_stride, const int16_t *d_sb_origin,
                            const uint32_t *d_best_sad0, const uint32_t *d_best_mv0, const uint32_t *d_best_sad1,
                            const uint32_t *d_best_mv1, uint32_t pu_pitch, int npus, int bipred_all_pus, int sub_sad,
                            uint32_t *d_bipred_sad, svt_hip_me_result *d_results, size_t nsb, void *stream);
/* HOST helper: index in the SAD / vector rows (EbMeTierZeroPu order) of raster PU `pu_index` (0 .. 208), or -1; the
 * reference's tab8x8 / tab16x16 / tab32x16 ... tables (EbMotionEstimation.h:90-175), derived from the rectangles */
int svt_hip_me_pu_storage_index(int pu_index);

/* K7 coefficient-domain distortion (full_distortion_kernel32_bits_func_ptr_array /
 * full_distortion_kernel_cbf_zero32_bits_func_ptr_array, EbPictureOperators.h:268-280;
 * C: EbPictureOperators.c:283-346).  d_out: uint64[nblocks][2] =
 * {DIST_CALC_RESIDUAL, DIST_CALC_PREDICTION}.  cbf_zero: d_recon is ignored. */
int svt_hip_full_distortion32_batch(const int32_t *d_coeff, uint32_t coeff_stride,
                                    size_t coeff_block_pitch, const int32_t *d_recon,
                                    uint32_t recon_stride, size_t recon_block_pitch,
                                    uint32_t width, uint32_t height, int cbf_zero,
                                    uint64_t *d_out, size_t nblocks, void *stream);

/* picture_full_distortion32_bits (EbPictureOperators.c:349-457), luma leg, for nblocks transform blocks: a 64-sample
 * dimension covers 32 coefficients, both buffers are dense with the (clamped) width as row stride and lie *_block_pitch
 * int32 apart, d_count_non_zero_coeffs[b] == 0 selects the cbf_zero kernel for block b (NULL: never).  flavour
 * SVT_HIP_FLAVOUR_AVX2 reproduces full_distortion_kernel32_bits_avx2's residual sum, which adds the low and high halves
 * of its 64-bit lanes separately (_mm256_add_epi32, EbPictureOperators_Intrinsic_AVX2.c:1989): it differs from the C
 * kernel once a lane's low halves sum past 2^32 (tests/golden/pins.npz holds both). */
int svt_hip_picture_full_distortion32_batch(const int32_t *d_coeff, size_t coeff_block_pitch, const int32_t *d_recon,
                                            size_t recon_block_pitch, uint32_t bwidth, uint32_t bheight,
                                            const uint32_t *d_count_non_zero_coeffs, int flavour, uint64_t *d_out,
                                            size_t nblocks, void *stream);

/* Open-loop intra search: open_loop_intra_search_sb, EbMotionEstimation.c:8694-8850,
 * for all blocks of ONE size of a picture (or of many pictures' worth of blocks) in one call.
 * d_pic points at picture sample (0, 0) (buffer_y + origin_y * stride_y + origin_x), width x height are
 * the picture's; d_xy[i] = x | y << 16 is a block's origin (the reference's cu_origin_x / _y).  Per block:
 * neighbour arrays gathered from the SOURCE picture exactly as update_neighbor_samples_array_open_loop does
 * (EbIntraPrediction.c:4707-4773: 127 / 129 / 128 fill at picture borders), then for each of the ncand
 * candidates (modes[c]: AV1 PredictionMode 0..12, angle_deltas[c]: -2..2 for the directional modes; HOST
 * arrays - the list the reference's loop :8747-8846 enumerates for this block size) the prediction of
 * intra_prediction_open_loop (:4778-4808: dr_predictor without edge filter / upsampling, DC by
 * availability, the other predictors as is) and its SAD against the source block.
 * d_distortion: uint32 [nblocks][ncand] (ois_candidate_t.distortion); d_best_index: int8 [nblocks]
 * (ois_sb_results_t.best_distortion_index: first strict minimum below 64*64*255, else 0).
 * d_work: scratch of svt_hip_ois_work_bytes(bsize, ncand, nblocks) bytes (neighbour arrays + one dense
 * prediction batch per candidate: split very large batches).  bsize 8 / 16 / 32 / 64, 8-bit. */
size_t svt_hip_ois_work_bytes(uint32_t bsize, int ncand, size_t nblocks);
int svt_hip_ois_search_batch(const uint8_t *d_pic, uint32_t stride, uint32_t width, uint32_t height,
                             const uint32_t *d_xy, uint32_t bsize, const uint8_t *modes,
                             const int8_t *angle_deltas, int ncand, uint32_t *d_distortion,
                             int8_t *d_best_index, void *d_work, size_t work_bytes, size_t nblocks,
                             void *stream);

/* The open-loop intra search of a whole PICTURE in one call: one group per block size (what svt_hip_ois_search_batch takes),
 * run concurrently on the library's internal streams (forked from / joined into `stream`; the call only enqueues).  The four
 * sizes of open_loop_intra_search_sb (EbMotionEstimation.c:8694) are independent, so the picture costs its slowest size, not
 * their sum. */
typedef struct svt_hip_ois_group {
    const uint32_t *d_xy;                 /* block origins x | y << 16 */
    uint32_t bsize;                       /* 8, 16, 32, 64 */
    const uint8_t *modes;                 /* HOST candidate list (svt_hip_ois_candidates) */
    const int8_t *angle_deltas;
    int32_t ncand;
    uint32_t *d_distortion;               /* [nblocks][ncand] */
    int8_t *d_best_index;                 /* [nblocks] */
    void *d_work;  size_t work_bytes;     /* svt_hip_ois_work_bytes(bsize, ncand, nblocks) */
    size_t nblocks;
} svt_hip_ois_group;
int svt_hip_ois_search_frame(const uint8_t *d_pic, uint32_t stride, uint32_t width, uint32_t height,
                             const svt_hip_ois_group *groups, int ngroups, void *stream);

/* K11 chroma-from-luma helpers of the encode pass (Av1EncodeLoop, EbCodingLoop.c:736-846) and the
 * entropy stage's level map - the remaining pieces of SURVEY.md 8(f) n3.
 *
 * svt_hip_cfl_luma_subsampling_420_batch replaces cfl_luma_subsampling_420_{lbd,hbd}_c
 * (EbIntraPrediction.c:1303-1332; aom_dsp_rtcd.h has no dispatch slot for them): per block, a
 * width x height LUMA area (8-bit, or 16-bit when is_16bit) becomes width/2 x height/2 Q3 sums
 * `(a + b + c + d) << 1`.  Blocks are addressed by d_xy[i] = x | y << 16 (sample units) in the
 * plane d_luma, or, when d_xy is NULL, lie luma_block_pitch samples apart.  Output rows are
 * q3_line int16 apart (CFL_BUF_LINE = 32 in the reference), blocks q3_block_pitch int16 apart
 * (CFL_BUF_SQUARE = 1024 there).  subtract_average != 0 also applies subtract_average with the
 * encode pass's arguments (round_offset = w*h/2, num_pel_log2 = log2(w*h) of the chroma block) in
 * the same kernel.  Chroma sizes 4..32 in both dimensions. */
int svt_hip_cfl_luma_subsampling_420_batch(const void *d_luma, uint32_t luma_stride,
                                           size_t luma_block_pitch, const uint32_t *d_xy, int is_16bit,
                                           int16_t *d_q3, uint32_t q3_line, size_t q3_block_pitch,
                                           uint32_t width, uint32_t height, int subtract_average,
                                           size_t nblocks, void *stream);
/* subtract_average (aom_dsp_rtcd.h:138, C: EbIntraPrediction.c:1333-1359), in place on Q3 blocks. */
int svt_hip_subtract_average_batch(int16_t *d_q3, uint32_t q3_line, size_t q3_block_pitch,
                                   uint32_t width, uint32_t height, int32_t round_offset,
                                   int32_t num_pel_log2, size_t nblocks, void *stream);
/* cfl_predict_lbd / cfl_predict_hbd (aom_dsp_rtcd.h:142-146, C: EbIntraPrediction.c:1361-1402):
 * dst = clip(pred + ROUND_POWER_OF_TWO_SIGNED(alpha_q3 * ac_q3, 6), bit_depth), one alpha per block
 * (d_alpha_q3[nblocks]).  pred and dst are planes addressed by the same d_xy (the encode pass
 * predicts in place: d_dst == d_pred is allowed), or dense blocks (stride * height apart) when
 * d_xy is NULL. */
int svt_hip_cfl_predict_batch(const int16_t *d_ac_q3, uint32_t q3_line, size_t q3_block_pitch,
                              const void *d_pred, uint32_t pred_stride, void *d_dst,
                              uint32_t dst_stride, const uint32_t *d_xy, const int32_t *d_alpha_q3,
                              int bit_depth, uint32_t width, uint32_t height, int is_16bit,
                              size_t nblocks, void *stream);
/* av1_txb_init_levels (aom_dsp_rtcd.h:2374, C: EbRateDistortionCost.c:125-150).  d_levels_buf holds
 * one WHOLE padded buffer per block (the reference's levels_buf; its `levels` pointer is
 * buffer + TX_PAD_TOP * (width + TX_PAD_HOR)), levels_block_pitch >= (width + 4) * (height + 6) + 16
 * bytes, a multiple of 4.  Every byte of the (width + 4) * (height + 6) + 16 is written. */
int svt_hip_txb_init_levels_batch(const int32_t *d_coeff, size_t coeff_block_pitch,
                                  uint8_t *d_levels_buf, size_t levels_block_pitch, uint32_t width,
                                  uint32_t height, size_t nblocks, void *stream);

/* K9/K10 intra prediction.  Replaces the aom_{dc,...,paeth}_predictor_WxH /
 * aom_highbd_* slots (aom_dsp_rtcd.h:442-1262, 1626-2330; C: EbIntraPrediction.c:
 * 1838-2260) and av1_dr_prediction_z{1,2,3} / av1_highbd_dr_prediction_z*
 * (aom_dsp_rtcd.h:2340; C: EbIntraPrediction.c:370-477, 3394-3506).
 * Neighbours: d_above / d_left hold one row of `nb_pitch` samples per block; the
 * reference's above_row[p] / left_col[p] (p >= -2) is element SVT_HIP_NB_ORIGIN + p.
 * Samples are uint8 (is_16bit = 0, bd = 8) or uint16.  mode: SVT_INTRA_*; for the
 * directional modes dx, dy, upsample_above, upsample_left are the reference's
 * arguments (dr_intra_derivative based).  Destination addressing as in
 * svt_hip_inv_txfm2d_add_batch. */
enum { SVT_INTRA_DC, SVT_INTRA_V, SVT_INTRA_H, SVT_INTRA_SMOOTH, SVT_INTRA_SMOOTH_V, SVT_INTRA_SMOOTH_H,
       SVT_INTRA_PAETH, SVT_INTRA_DC_TOP, SVT_INTRA_DC_LEFT, SVT_INTRA_DC_128, SVT_INTRA_Z1, SVT_INTRA_Z2,
       SVT_INTRA_Z3, SVT_INTRA_MODES };
#define SVT_HIP_NB_ORIGIN 16
int svt_hip_intra_pred_batch(void *d_dst, int32_t dst_stride, size_t dst_block_pitch,
                             const uint32_t *d_dst_offsets, const void *d_above, const void *d_left,
                             int32_t nb_pitch, int mode, int bw, int bh, int upsample_above,
                             int upsample_left, int dx, int dy, int is_16bit, int bd,
                             size_t nblocks, void *stream);
/* build_intra_predictors / build_intra_predictors_high (EbIntraPrediction.c:3667-3855, 3857-4076): the
 * neighbour-availability glue of av1_predict_intra_block, fused into ONE launch for a batch of prediction blocks of one
 * transform size with per-block mode and availability.  Per block: which edges the mode needs, the constant fill when the
 * needed edge is missing, edge extension (last available sample replicated; base +- 1 defaults), the corner sample, for
 * directional modes the corner / edge smoothing filters and 2x up-sampling chosen from size, angle and filt_type, DC by
 * availability, then the prediction itself.
 *   d_top_neigh / d_left_neigh: the caller's topNeighArray / leftNeighArray (EbCodingLoop.c:2862-2893), one per block,
 *     `neigh_pitch` samples apart: element 0 = the above-left corner sample, element 1 + i = above[i] / left[i];
 *     neigh_pitch >= 1 + 2 * max(width, height).  Samples uint8 (is_16bit = 0, bd 8) or uint16.
 *   d_blocks: one descriptor per block (device memory); n_top_px / n_left_px above the block's width / height are clamped.
 *   Destination addressing as in svt_hip_inv_txfm2d_add_batch. */
typedef struct svt_hip_intra_blk {
    uint8_t mode;                 /* AV1 PredictionMode: DC 0, V 1, H 2, D45 3, D135 4, D113 5, D157 6, D203 7, D67 8,
                                     SMOOTH 9, SMOOTH_V 10, SMOOTH_H 11, PAETH 12 */
    int8_t angle_delta;           /* -3 .. 3 (directional modes; angle = mode_to_angle_map[mode] + 3 * delta) */
    uint8_t filt_type;            /* get_filt_type (:146): 1 when the above or left block is a smooth-mode block */
    uint8_t disable_edge_filter;
    uint8_t n_top_px, n_topright_px, n_left_px, n_bottomleft_px;   /* available samples (svt_hip_intra_neighbor_px) */
} svt_hip_intra_blk;
int svt_hip_build_intra_predictors_batch(void *d_dst, int32_t dst_stride, size_t dst_block_pitch,
                                         const uint32_t *d_dst_offsets, const void *d_top_neigh,
                                         const void *d_left_neigh, int32_t neigh_pitch,
                                         const svt_hip_intra_blk *d_blocks, int tx_size, int is_16bit, int bd,
                                         size_t nblocks, void *stream);
/* The same walked through an ORDER of the batch (d_order[i] = block index): svt_hip_intra_order_blocks_batch groups the block
 * indices by the predictor kind each descriptor resolves to (DC variants, V, H, SMOOTH*, PAETH, the three directional zones),
 * on the device, inside tiles of SVT_HIP_INTRA_ORDER_TILE consecutive blocks (entries [t * TILE, (t + 1) * TILE) of d_order are a
 * permutation of those block indices, kinds contiguous), so that the lanes of a wave run one kind's code (mixed kinds in a wave
 * are correct, just slower: a wave runs every kind it holds).  d_order: uint32[nblocks]; d_work: unused (may be NULL; the first
 * version's counters).  Results do not depend on the order. */
#define SVT_HIP_INTRA_ORDER_TILE 4096
int svt_hip_intra_order_blocks_batch(const svt_hip_intra_blk *d_blocks, int tx_size, size_t nblocks, uint32_t *d_order,
                                     uint32_t *d_work, void *stream);
int svt_hip_build_intra_predictors_ordered_batch(void *d_dst, int32_t dst_stride, size_t dst_block_pitch,
                                                 const uint32_t *d_dst_offsets, const void *d_top_neigh,
                                                 const void *d_left_neigh, int32_t neigh_pitch,
                                                 const svt_hip_intra_blk *d_blocks, const uint32_t *d_order, int tx_size,
                                                 int is_16bit, int bd, size_t nblocks, void *stream);

/* HOST helper, no device work: the first half of av1_predict_intra_block / av1_predict_intra_block_16bit
 * (EbIntraPrediction.c:4078-4333, 4336-4566) - up / left availability from the block's mode-info position,
 * has_top_right (:1567) and has_bottom_left (:1755), and the four sample counts handed to build_intra_predictors.
 * The reference reads "is that neighbour block already coded" from bit tables (has_tr_* / has_bl_*); here it is the
 * comparison of two coding-order keys (Morton order of the enclosing squares; PARTITION_VERT_A / VERT_B visit the last
 * split column first).  is_16bit selects the tile handling of the 16-bit function (the 8-bit one treats the picture as
 * one tile, :4124-4137).  Fills blk->n_*_px; returns SVT_HIP_OK or SVT_HIP_ERR_INVALID. */
typedef struct svt_hip_intra_pos {
    int32_t is_16bit;
    int32_t sb_size_mi;                       /* mi_size_high[sb_size]: 16 (64x64 superblocks) or 32 */
    int32_t mi_rows, mi_cols;                 /* Av1Common */
    int32_t tile_mi_row_start, tile_mi_row_end, tile_mi_col_start, tile_mi_col_end;   /* TileInfo */
    int32_t partition;                        /* AV1 PartitionType (from_shape_to_part[blk_geom->shape]) */
    int32_t bsize;                            /* AV1 block_size, BLOCK_4X4 = 0 .. BLOCK_64X16 = 21 */
    int32_t tx_size, plane;
    int32_t bl_org_x_pict, bl_org_y_pict;     /* luma sample position of the block */
    int32_t col_off, row_off;                 /* transform block offset inside the block, 4-sample units of the plane */
    int32_t wpx, hpx;                         /* block size in samples of the plane */
} svt_hip_intra_pos;
int svt_hip_intra_neighbor_px(const svt_hip_intra_pos *pos, svt_hip_intra_blk *blk);
int svt_hip_intra_has_top_right(int sb_size_mi, int bsize, int mi_row, int mi_col, int top_available, int right_available,
                                int partition, int tx_size, int row_off, int col_off, int ss_x, int ss_y);
int svt_hip_intra_has_bottom_left(int sb_size_mi, int bsize, int mi_row, int mi_col, int bottom_available,
                                  int left_available, int partition, int tx_size, int row_off, int col_off, int ss_x,
                                  int ss_y);

/* av1_filter_intra_edge{,_high} / av1_upsample_intra_edge{,_high} (aom_dsp_rtcd.h:152,
 * 431; C: EbIntraPrediction.c:3539-3660) on nblocks edges laid out like d_above. */
int svt_hip_filter_intra_edge_batch(void *d_edges, int32_t nb_pitch, int sz, int strength,
                                    int is_16bit, size_t nblocks, void *stream);
int svt_hip_upsample_intra_edge_batch(void *d_edges, int32_t nb_pitch, int sz, int is_16bit,
                                      int bd, size_t nblocks, void *stream);

/* ============================================================================
 * (A) drop-in entry points — reference signatures, HOST pointers, synchronous.
 * ==========================================================================*/

/* av1_fwd_txfm2d_WxH (aom_dsp_rtcd.h:160-255) */
#define SVT_HIP_DECL_FWD(W, H) \
    void svt_hip_av1_fwd_txfm2d_##W##x##H(int16_t *input, int32_t *output, uint32_t input_stride, \
                                          svt_tx_type_t transform_type, uint8_t bit_depth);
SVT_HIP_DECL_FWD(4, 4) SVT_HIP_DECL_FWD(8, 8) SVT_HIP_DECL_FWD(16, 16) SVT_HIP_DECL_FWD(32, 32)
SVT_HIP_DECL_FWD(64, 64) SVT_HIP_DECL_FWD(4, 8) SVT_HIP_DECL_FWD(8, 4) SVT_HIP_DECL_FWD(8, 16)
SVT_HIP_DECL_FWD(16, 8) SVT_HIP_DECL_FWD(16, 32) SVT_HIP_DECL_FWD(32, 16) SVT_HIP_DECL_FWD(32, 64)
SVT_HIP_DECL_FWD(64, 32) SVT_HIP_DECL_FWD(4, 16) SVT_HIP_DECL_FWD(16, 4) SVT_HIP_DECL_FWD(8, 32)
SVT_HIP_DECL_FWD(32, 8) SVT_HIP_DECL_FWD(16, 64) SVT_HIP_DECL_FWD(64, 16)
#undef SVT_HIP_DECL_FWD

/* av1_inv_txfm2d_add_WxH (aom_dsp_rtcd.h:350-417): the three signature classes */
#define SVT_HIP_DECL_INV_SQ(W, H) \
    void svt_hip_av1_inv_txfm2d_add_##W##x##H(const int32_t *input, uint16_t *output, int32_t stride, \
                                              svt_tx_type_t tx_type, int32_t bd);
#define SVT_HIP_DECL_INV_R1(W, H) \
    void svt_hip_av1_inv_txfm2d_add_##W##x##H(const int32_t *input, uint16_t *output, int32_t stride, \
                                              svt_tx_type_t tx_type, svt_tx_size_t tx_size, int32_t eob, int32_t bd);
#define SVT_HIP_DECL_INV_R2(W, H) \
    void svt_hip_av1_inv_txfm2d_add_##W##x##H(const int32_t *input, uint16_t *output, int32_t stride, \
                                              svt_tx_type_t tx_type, svt_tx_size_t tx_size, int32_t bd);
SVT_HIP_DECL_INV_SQ(4, 4) SVT_HIP_DECL_INV_SQ(8, 8) SVT_HIP_DECL_INV_SQ(16, 16) SVT_HIP_DECL_INV_SQ(32, 32)
SVT_HIP_DECL_INV_SQ(64, 64)
SVT_HIP_DECL_INV_R1(8, 16) SVT_HIP_DECL_INV_R1(16, 8) SVT_HIP_DECL_INV_R1(16, 32) SVT_HIP_DECL_INV_R1(32, 16)
SVT_HIP_DECL_INV_R1(32, 64) SVT_HIP_DECL_INV_R1(64, 32) SVT_HIP_DECL_INV_R1(8, 32) SVT_HIP_DECL_INV_R1(32, 8)
SVT_HIP_DECL_INV_R1(16, 64) SVT_HIP_DECL_INV_R1(64, 16)
SVT_HIP_DECL_INV_R2(4, 8) SVT_HIP_DECL_INV_R2(8, 4) SVT_HIP_DECL_INV_R2(4, 16) SVT_HIP_DECL_INV_R2(16, 4)
#undef SVT_HIP_DECL_INV_SQ
#undef SVT_HIP_DECL_INV_R1
#undef SVT_HIP_DECL_INV_R2
/* av1_inv_txfm_add (aom_dsp_rtcd.h:421-423): 8-bit recon */
void svt_hip_av1_inv_txfm_add(const svt_tran_low_t *dqcoeff, uint8_t *dst, int32_t stride,
                              const svt_txfm_param *txfm_param);

/* aom_highbd_quantize_b{,_32x32,_64x64} and the 8-bit aom_quantize_b* slots
 * (aom_dsp_rtcd.h:323-342) */
#define SVT_HIP_DECL_QUANT(name) \
    void name(const svt_tran_low_t *coeff_ptr, intptr_t n_coeffs, int32_t skip_block, \
              const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr, \
              const int16_t *quant_shift_ptr, svt_tran_low_t *qcoeff_ptr, svt_tran_low_t *dqcoeff_ptr, \
              const int16_t *dequant_ptr, uint16_t *eob_ptr, const int16_t *scan, const int16_t *iscan);
SVT_HIP_DECL_QUANT(svt_hip_aom_highbd_quantize_b)
SVT_HIP_DECL_QUANT(svt_hip_aom_highbd_quantize_b_32x32)
SVT_HIP_DECL_QUANT(svt_hip_aom_highbd_quantize_b_64x64)
SVT_HIP_DECL_QUANT(svt_hip_aom_quantize_b)
SVT_HIP_DECL_QUANT(svt_hip_aom_quantize_b_32x32)
SVT_HIP_DECL_QUANT(svt_hip_aom_quantize_b_64x64)
#undef SVT_HIP_DECL_QUANT

/* EB_SADKERNELNxM_TYPE (EbComputeSAD.h:25-31): one row of NxMSadKernel_funcPtrArray */
uint32_t svt_hip_nxm_sad_kernel(const uint8_t *src, uint32_t src_stride, const uint8_t *ref,
                                uint32_t ref_stride, uint32_t height, uint32_t width);
/* EB_SADLOOPKERNELNxM_TYPE (EbComputeSAD.h:35-47) */
void svt_hip_sad_loop_kernel(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride,
                             uint32_t height, uint32_t width, uint64_t *best_sad,
                             int16_t *x_search_center, int16_t *y_search_center,
                             uint32_t src_stride_raw, int16_t search_area_width,
                             int16_t search_area_height);
/* spatial_full_distortion_kernel_func_ptr_array entry (EbPictureOperators.h:470) */
uint64_t svt_hip_spatial_full_distortion_kernel(uint8_t *input, uint32_t input_stride, uint8_t *recon,
                                                uint32_t recon_stride, uint32_t area_width,
                                                uint32_t area_height);
/* ResidualKernel (aom_dsp_rtcd.h:2372) */
void svt_hip_residual_kernel(uint8_t *input, uint32_t input_stride, uint8_t *pred, uint32_t pred_stride,
                             int16_t *residual, uint32_t residual_stride, uint32_t area_width,
                             uint32_t area_height);

/* full_distortion_kernel32_bits / _cbf_zero32_bits (EbPictureOperators.h:268-280) */
void svt_hip_full_distortion_kernel32_bits(int32_t *coeff, uint32_t coeff_stride, int32_t *recon_coeff,
                                           uint32_t recon_coeff_stride, uint64_t distortion_result[2],
                                           uint32_t area_width, uint32_t area_height);
void svt_hip_full_distortion_kernel_cbf_zero32_bits(int32_t *coeff, uint32_t coeff_stride,
                                                    int32_t *recon_coeff, uint32_t recon_coeff_stride,
                                                    uint64_t distortion_result[2], uint32_t area_width,
                                                    uint32_t area_height);

/* intra_pred_fn / intra_high_pred_fn (EbIntraPrediction.h:36-41): one entry per mode;
 * the per-size RTCD slots aom_<mode>_predictor_WxH bind to these with W, H fixed
 * (INTEGRATION.md).  mode = SVT_INTRA_DC .. SVT_INTRA_DC_128. */
void svt_hip_intra_predictor(int mode, int bw, int bh, uint8_t *dst, ptrdiff_t stride,
                             const uint8_t *above, const uint8_t *left);
void svt_hip_highbd_intra_predictor(int mode, int bw, int bh, uint16_t *dst, ptrdiff_t stride,
                                    const uint16_t *above, const uint16_t *left, int32_t bd);
/* av1_dr_prediction_z1/z2/z3 and highbd twins (aom_dsp_rtcd.h:2340-2362) */
void svt_hip_av1_dr_prediction_z1(uint8_t *dst, ptrdiff_t stride, int32_t bw, int32_t bh, const uint8_t *above,
                                  const uint8_t *left, int32_t upsample_above, int32_t dx, int32_t dy);
void svt_hip_av1_dr_prediction_z2(uint8_t *dst, ptrdiff_t stride, int32_t bw, int32_t bh, const uint8_t *above,
                                  const uint8_t *left, int32_t upsample_above, int32_t upsample_left,
                                  int32_t dx, int32_t dy);
void svt_hip_av1_dr_prediction_z3(uint8_t *dst, ptrdiff_t stride, int32_t bw, int32_t bh, const uint8_t *above,
                                  const uint8_t *left, int32_t upsample_left, int32_t dx, int32_t dy);
void svt_hip_av1_highbd_dr_prediction_z1(uint16_t *dst, ptrdiff_t stride, int32_t bw, int32_t bh,
                                         const uint16_t *above, const uint16_t *left, int32_t upsample_above,
                                         int32_t dx, int32_t dy, int32_t bd);
void svt_hip_av1_highbd_dr_prediction_z2(uint16_t *dst, ptrdiff_t stride, int32_t bw, int32_t bh,
                                         const uint16_t *above, const uint16_t *left, int32_t upsample_above,
                                         int32_t upsample_left, int32_t dx, int32_t dy, int32_t bd);
void svt_hip_av1_highbd_dr_prediction_z3(uint16_t *dst, ptrdiff_t stride, int32_t bw, int32_t bh,
                                         const uint16_t *above, const uint16_t *left, int32_t upsample_left,
                                         int32_t dx, int32_t dy, int32_t bd);

/* ---- per-size RTCD slot entry points ------------------------------------------------------------------------------
 * X-macro lists of the reference's slot families (a host can use them too). */
#define SVT_HIP_BLOCK_SIZES_2(X, A, B) \
    X(A, B, 4, 4) X(A, B, 8, 8) X(A, B, 16, 16) X(A, B, 32, 32) X(A, B, 64, 64) X(A, B, 4, 8) X(A, B, 8, 4) X(A, B, 8, 16) \
    X(A, B, 16, 8) X(A, B, 16, 32) X(A, B, 32, 16) X(A, B, 32, 64) X(A, B, 64, 32) X(A, B, 4, 16) X(A, B, 16, 4) \
    X(A, B, 8, 32) X(A, B, 32, 8) X(A, B, 16, 64) X(A, B, 64, 16)
/* SIZES(X, mode, MODE) applied to every non-directional mode: aom_<mode>_predictor_WxH <-> SVT_INTRA_* */
#define SVT_HIP_INTRA_MODES(SIZES, X) \
    SIZES(X, dc, SVT_INTRA_DC) SIZES(X, dc_top, SVT_INTRA_DC_TOP) SIZES(X, dc_left, SVT_INTRA_DC_LEFT) \
    SIZES(X, dc_128, SVT_INTRA_DC_128) SIZES(X, v, SVT_INTRA_V) SIZES(X, h, SVT_INTRA_H) SIZES(X, smooth, SVT_INTRA_SMOOTH) \
    SIZES(X, smooth_v, SVT_INTRA_SMOOTH_V) SIZES(X, smooth_h, SVT_INTRA_SMOOTH_H) SIZES(X, paeth, SVT_INTRA_PAETH)
/* the 22 sizes of aom_sadMxN / aom_sadMxNx4d (aom_dsp_rtcd.h:1328-1500) */
#define SVT_HIP_SAD_SIZES(X) \
    X(128, 128) X(128, 64) X(64, 128) X(64, 64) X(64, 32) X(32, 64) X(32, 32) X(32, 16) X(16, 32) X(16, 16) X(16, 8) \
    X(8, 16) X(8, 8) X(8, 4) X(4, 8) X(4, 4) X(4, 16) X(16, 4) X(8, 32) X(32, 8) X(16, 64) X(64, 16)

/* aom_<mode>_predictor_WxH / aom_highbd_<mode>_predictor_WxH: exactly intra_pred_fn / intra_high_pred_fn
 * (EbIntraPrediction.h:36-41), one per mode and size, storable in the reference's slots and its pred[][] / dc_pred[][][]
 * tables (EbIntraPrediction.c:2842-3350).  10 modes x 19 sizes x {8-bit, high bit depth}. */
#define SVT_HIP_DECL_PRED(mode, MODE, W, H)                                                                             \
    void svt_hip_aom_##mode##_predictor_##W##x##H(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left); \
    void svt_hip_aom_highbd_##mode##_predictor_##W##x##H(uint16_t *dst, ptrdiff_t stride, const uint16_t *above,          \
                                                         const uint16_t *left, int bd);
SVT_HIP_INTRA_MODES(SVT_HIP_BLOCK_SIZES_2, SVT_HIP_DECL_PRED)
#undef SVT_HIP_DECL_PRED
/* eb_smooth_v_predictor / eb_smooth_h_predictor (aom_dsp_rtcd.h:259-264) */
void svt_hip_eb_smooth_v_predictor(uint8_t *dst, ptrdiff_t stride, int32_t bw, int32_t bh, const uint8_t *above,
                                   const uint8_t *left);
void svt_hip_eb_smooth_h_predictor(uint8_t *dst, ptrdiff_t stride, int32_t bw, int32_t bh, const uint8_t *above,
                                   const uint8_t *left);
/* aom_sadMxN, aom_sadMxNx4d (aom_dsp_rtcd.h:1328-1500) */
#define SVT_HIP_DECL_SAD(W, H)                                                                                          \
    unsigned int svt_hip_aom_sad##W##x##H(const uint8_t *src_ptr, int src_stride, const uint8_t *ref_ptr, int ref_stride); \
    void svt_hip_aom_sad##W##x##H##x4d(const uint8_t *src_ptr, int src_stride, const uint8_t *const ref_ptr[],            \
                                       int ref_stride, uint32_t *sad_array);
SVT_HIP_SAD_SIZES(SVT_HIP_DECL_SAD)
#undef SVT_HIP_DECL_SAD
/* EB_SADAVGKERNELNxM_TYPE (EbComputeSAD.h:49-57): a row of NxMSadAveragingKernel_funcPtrArray */
uint32_t svt_hip_combined_averaging_sad(uint8_t *src, uint32_t src_stride, uint8_t *ref1, uint32_t ref1_stride,
                                        uint8_t *ref2, uint32_t ref2_stride, uint32_t height, uint32_t width);
/* residual_kernel16bit (EbPictureOperators.c:134) */
void svt_hip_residual_kernel16bit(uint16_t *input, uint32_t input_stride, uint16_t *pred, uint32_t pred_stride,
                                  int16_t *residual, uint32_t residual_stride, uint32_t area_width,
                                  uint32_t area_height);
/* av1_filter_intra_edge{,_high}, av1_upsample_intra_edge{,_high} (aom_dsp_rtcd.h:152-156, 431-437) */
void svt_hip_av1_filter_intra_edge(uint8_t *p, int32_t sz, int32_t strength);
void svt_hip_av1_filter_intra_edge_high(uint16_t *p, int32_t sz, int32_t strength);
void svt_hip_av1_upsample_intra_edge(uint8_t *p, int32_t sz);
void svt_hip_av1_upsample_intra_edge_high(uint16_t *p, int32_t sz, int32_t bd);
/* subtract_average, cfl_predict_lbd / _hbd (aom_dsp_rtcd.h:140-148), av1_txb_init_levels (:2376) */
void svt_hip_subtract_average(int16_t *pred_buf_q3, int32_t width, int32_t height, int32_t round_offset,
                              int32_t num_pel_log2);
void svt_hip_cfl_predict_lbd(const int16_t *pred_buf_q3, uint8_t *pred, int32_t pred_stride, uint8_t *dst,
                             int32_t dst_stride, int32_t alpha_q3, int32_t bit_depth, int32_t width, int32_t height);
void svt_hip_cfl_predict_hbd(const int16_t *pred_buf_q3, uint16_t *pred, int32_t pred_stride, uint16_t *dst,
                             int32_t dst_stride, int32_t alpha_q3, int32_t bit_depth, int32_t width, int32_t height);
void svt_hip_av1_txb_init_levels(const svt_tran_low_t *const coeff, const int32_t width, const int32_t height,
                                 uint8_t *const levels);

/* ---- picture input ("the y4m -> plane upload path") --------------------------------------------------
 * Host: the reference application's y4m reader (Source/App/EncApp/EbAppInputy4m.c) - the same header tokens, the same
 * accepted / rejected files (including its 7-character bound on the 'C' and 'F' tokens), "FRAME\n" delimiters. */
typedef struct svt_hip_y4m_info {
    uint32_t width, height, fr_n, fr_d, bit_depth, interlaced;
    char chroma[8];                /* "420" "422" "444" "411" "400" */
    char scan_type;                /* 'p' 't' 'b' */
} svt_hip_y4m_info;
typedef struct svt_hip_y4m svt_hip_y4m;
/* read_y4m_header (:35-243) on the line that follows the "YUV4MPEG2" signature */
int svt_hip_y4m_parse_header(const char *line, svt_hip_y4m_info *out);
size_t svt_hip_y4m_frame_bytes(const svt_hip_y4m_info *info);
/* check_if_y4m (:269) + read_y4m_header; read_y4m_frame_delimiter (:247) + the planes: 1 = frame read, 0 = end of file */
int svt_hip_y4m_open(const char *path, svt_hip_y4m **out, svt_hip_y4m_info *info);
int svt_hip_y4m_read_frame(svt_hip_y4m *h, void *host_dst, size_t capacity);
void svt_hip_y4m_close(svt_hip_y4m *h);
/* Device: a frame as the file holds it (d_frame: Y, Cb, Cr back to back, u8 or u16 samples, already in HBM) laid out in the
 * encoder's padded plane buffers in ONE launch: the copy, pad_input_picture (EbMcp.c:273; right / bottom extension to the
 * minimum CU size, PadPictureToMultipleOfMinCuSizeDimensions, EbPictureAnalysisProcess.c:4818) and generate_padding{,16_bit}
 * (EbMcp.c:176-267; the borders of origin_x / origin_y samples).  d_y / d_cb / d_cr = first sample of each BUFFER (the picture
 * origin is at (origin_x >> ss, origin_y >> ss)); strides in samples; d_cb = d_cr = NULL for luma only. */
int svt_hip_picture_import(const void *d_frame, uint32_t width, uint32_t height, int ss_x, int ss_y, int is_16bit, void *d_y,
                           uint32_t stride_y, void *d_cb, uint32_t stride_cb, void *d_cr, uint32_t stride_cr,
                           uint32_t origin_x, uint32_t origin_y, uint32_t pad_right, uint32_t pad_bottom, void *stream);
/* generate_padding / generate_padding16_bit in place (the reference also pads reconstructed reference pictures with it,
 * EbEncDecProcess.c:1040-1100): d_buf = first sample of the buffer, picture at (pad_w, pad_h) */
int svt_hip_picture_pad(void *d_buf, uint32_t stride, uint32_t width, uint32_t height, uint32_t pad_w, uint32_t pad_h,
                        int is_16bit, void *stream);
/* The 8-bit plane of a deeper picture, v >> (bd - 8): what the reference keeps as buffer_y next to its bit-increment plane and
 * what its HME / ME / open-loop intra search read; un_pack8_bit_data (C_DEFAULT/EbPackUnPack_C.c:152-175, bd = 10).  Whole padded buffer:
 * cols x rows samples from d_in (16-bit) to d_out (8-bit). */
int svt_hip_picture_luma8(const uint16_t *d_in, uint32_t in_stride, uint8_t *d_out, uint32_t out_stride, uint32_t cols,
                          uint32_t rows, int bd, void *stream);
/* DecimateInputPicture (EbPictureAnalysisProcess.c:4907-4958): Decimation2D (:170) at step 2 (quarter) and 4 (sixteenth) of
 * the luma picture at d_luma (its ORIGIN sample), each followed by generate_padding of the decimated buffer; either output may
 * be NULL.  d_quarter / d_sixteenth = first sample of the buffer. */
int svt_hip_picture_decimate(const uint8_t *d_luma, uint32_t luma_stride, uint32_t width, uint32_t height, uint8_t *d_quarter,
                             uint32_t q_stride, uint32_t q_origin_x, uint32_t q_origin_y, uint8_t *d_sixteenth,
                             uint32_t s_stride, uint32_t s_origin_x, uint32_t s_origin_y, void *stream);

/* ---- dispatch registration ---------------------------------------------------------------------------------------
 * The library keeps a registry {reference slot name -> drop-in of the same signature} for every RTCD global of
 * aom_dsp_rtcd.h it implements (svt_hip_rtcd_slot_count() entries: the 19 av1_fwd_txfm2d_WxH, the 19
 * av1_inv_txfm2d_add_WxH, av1_inv_txfm_add, the six quantisers, ResidualKernel, 380 intra predictor slots,
 * eb_smooth_v/h_predictor, av1_dr_prediction_z1-3 and their highbd twins, the edge filters / upsamplers, subtract_average,
 * cfl_predict_lbd/hbd, av1_txb_init_levels, 22 aom_sadMxN and 22 aom_sadMxNx4d).  After the stock
 * setup_rtcd_internal(asm_type) (EbEncHandle.c:917) and BEFORE init_intra_predictors_internal() the host calls, per slot,
 *     svt_hip_rtcd_override_slot("aom_dc_predictor_16x16", (void **)&aom_dc_predictor_16x16);
 * (INTEGRATION.md has the macro that does it for every slot).  Unknown names return SVT_HIP_ERR_INVALID and leave the
 * slot alone.  GRACEFUL REFUSAL: when the device is unusable (no HIP runtime, no gfx950 part) both override calls return
 * SVT_HIP_ERR_NO_DEVICE and store nothing - the host's pointers keep what setup_rtcd_internal put there (the AVX2 kernels:
 * the encoder's own fallback; integration/asm_hip.patch also takes asm_type back to ASM_AVX2 then).  Once a slot has been
 * handed out, a HIP error in the middle of a run still abort()s inside the drop-in: its reference signature has no error
 * channel and there is deliberately no CPU path in this library.  The *_funcPtrArray[asm_type] tables are static per translation unit in the reference, so their ASM_HIP rows
 * are added at compile time (integration/asm_hip.patch). */
int svt_hip_rtcd_slot_count(void);
const char *svt_hip_rtcd_slot_name(int index);        /* NULL when out of range */
void *svt_hip_rtcd_slot_function(const char *name);   /* the drop-in registered under a reference slot name, or NULL */
int svt_hip_rtcd_override_slot(const char *name, void **slot);

/* Pointer table the host fills after the stock setup_rtcd_internal(asm_type)
 * (EbEncHandle.c:917) and BEFORE init_intra_predictors_internal(): each member is
 * the address of the reference's RTCD global of the same name (or NULL to leave
 * that slot alone).  svt_hip_rtcd_override stores the svt_hip_* drop-ins in them. */
typedef struct svt_hip_rtcd_table {
    void **av1_fwd_txfm2d[SVT_TX_SIZES_ALL];      /* &av1_fwd_txfm2d_4x4 ... by TxSize */
    void **av1_inv_txfm2d_add[SVT_TX_SIZES_ALL];  /* &av1_inv_txfm2d_add_4x4 ... */
    void **av1_inv_txfm_add;
    void **aom_quantize_b, **aom_quantize_b_32x32, **aom_quantize_b_64x64;
    void **aom_highbd_quantize_b, **aom_highbd_quantize_b_32x32, **aom_highbd_quantize_b_64x64;
    void **ResidualKernel;
} svt_hip_rtcd_table;
int svt_hip_rtcd_override(const svt_hip_rtcd_table *table);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_DSP_H */
