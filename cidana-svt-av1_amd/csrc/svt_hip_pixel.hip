// svt_hip_pixel.hip — entry points of the pixel / search family of libsvt_hip_dsp.so (include/svt_hip_dsp.h): SAD, SSE,
// residual, SAD search, ME multi-size search, coefficient-domain distortion and their drop-ins.
#include "host_common.h"
#include "kernel_me.h"
#include "kernel_pixel.h"

using namespace svtdev;
using namespace svthost;

// mode 0 SAD, 1 SSE, 2 averaging SAD (kernel_pixel.h); a_shift = 2: four references per source block (x4d)
static int sad_sse_common(int mode, const uint8_t* a, uint32_t as, size_t ap, const uint32_t* a_offs, int a_shift,
                          const uint8_t* b, uint32_t bs, size_t bp, const uint32_t* b_offs, const uint8_t* c, uint32_t cst,
                          size_t cp, uint32_t w, uint32_t h, void* out, size_t n, void* stream) {
    if (int rc = require_init()) return rc;
    if (n == 0) return SVT_HIP_OK;
    if (!a || !b || !out || (mode == 2 && !c)) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (w == 0 || h == 0 || w > 128 || h > 128) return set_err(SVT_HIP_ERR_INVALID, "block %ux%u", w, h);
    const uint32_t grid = (uint32_t)((n + 15) / 16);
#define SADL(M) hipLaunchKernelGGL((sad_sse_kernel<M>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, as, ap, a_offs, a_shift, b, bs, bp, \
                                   b_offs, c, cst, cp, w, h, out, (uint32_t)n)
    if (mode == 1) SADL(1); else if (mode == 2) SADL(2); else SADL(0);
#undef SADL
    return launch_status(mode == 1 ? "sse" : (mode == 2 ? "sad_avg" : "sad"));
}
extern "C" int svt_hip_sad_batch(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch,
                                 const uint8_t* d_ref, uint32_t ref_stride, size_t ref_block_pitch, uint32_t width,
                                 uint32_t height, uint32_t* d_out, size_t nblocks, void* stream) {
    return sad_sse_common(0, d_src, src_stride, src_block_pitch, nullptr, 0, d_ref, ref_stride, ref_block_pitch, nullptr, nullptr, 0, 0,
                          width, height, d_out, nblocks, stream);
}
extern "C" int svt_hip_sse_batch(const uint8_t* d_a, uint32_t a_stride, size_t a_block_pitch, const uint8_t* d_b,
                                 uint32_t b_stride, size_t b_block_pitch, uint32_t width, uint32_t height,
                                 uint64_t* d_out, size_t nblocks, void* stream) {
    return sad_sse_common(1, d_a, a_stride, a_block_pitch, nullptr, 0, d_b, b_stride, b_block_pitch, nullptr, nullptr, 0, 0, width, height,
                          d_out, nblocks, stream);
}
extern "C" int svt_hip_sad_planes_batch(const uint8_t* d_src_plane, uint32_t src_stride, const uint32_t* d_src_offsets,
                                        const uint8_t* d_ref_plane, uint32_t ref_stride, const uint32_t* d_ref_offsets,
                                        uint32_t width, uint32_t height, uint32_t* d_out, size_t nblocks, void* stream) {
    if (nblocks && (!d_src_offsets || !d_ref_offsets)) { if (int rc = require_init()) return rc; return set_err(SVT_HIP_ERR_INVALID, "NULL offset table"); }
    return sad_sse_common(0, d_src_plane, src_stride, 0, d_src_offsets, 0, d_ref_plane, ref_stride, 0, d_ref_offsets, nullptr, 0, 0, width,
                          height, d_out, nblocks, stream);
}
extern "C" int svt_hip_sad_x4d_batch(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch,
                                     const uint32_t* d_src_offsets, const uint8_t* d_ref_plane, uint32_t ref_stride,
                                     const uint32_t* d_ref_offsets, uint32_t width, uint32_t height, uint32_t* d_out,
                                     size_t nblocks, void* stream) {
    if (nblocks && !d_ref_offsets) { if (int rc = require_init()) return rc; return set_err(SVT_HIP_ERR_INVALID, "NULL offset table"); }
    if (nblocks > 0x3fffffffu) { if (int rc = require_init()) return rc; return set_err(SVT_HIP_ERR_INVALID, "too many blocks"); }
    return sad_sse_common(0, d_src, src_stride, src_block_pitch, d_src_offsets, 2, d_ref_plane, ref_stride, 0, d_ref_offsets, nullptr, 0, 0,
                          width, height, d_out, nblocks * 4, stream);
}
extern "C" int svt_hip_sad_avg_batch(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch, const uint8_t* d_ref1,
                                     uint32_t ref1_stride, size_t ref1_block_pitch, const uint8_t* d_ref2,
                                     uint32_t ref2_stride, size_t ref2_block_pitch, uint32_t width, uint32_t height,
                                     uint32_t* d_out, size_t nblocks, void* stream) {
    return sad_sse_common(2, d_src, src_stride, src_block_pitch, nullptr, 0, d_ref1, ref1_stride, ref1_block_pitch, nullptr, d_ref2,
                          ref2_stride, ref2_block_pitch, width, height, d_out, nblocks, stream);
}
template <typename PixT>
static int residual_impl(const PixT* d_src, uint32_t src_stride, size_t src_block_pitch, const PixT* d_pred, uint32_t pred_stride,
                         size_t pred_block_pitch, int16_t* d_res, uint32_t res_stride, size_t res_block_pitch, uint32_t width,
                         uint32_t height, size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_src || !d_pred || !d_res) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (width == 0 || height == 0) return set_err(SVT_HIP_ERR_INVALID, "empty block");
    // samples per lane: 16 bytes of input where the width allows it
    constexpr uint32_t MAXCS = 16 / sizeof(PixT);
    const uint32_t rcs = (width % MAXCS) == 0 ? MAXCS : ((width & 7) == 0 ? 8u : ((width & 3) == 0 ? 4u : 1u));
    const size_t total = (size_t)(width / rcs) * height * nblocks;
    const size_t grid = (total + 255) / 256;
    if (grid > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many blocks for one launch");
    const uint32_t cpr = width / rcs;
    const size_t per = (size_t)cpr * height;
    const bool pow2 = (cpr & (cpr - 1)) == 0 && (per & (per - 1)) == 0;
#define RESL(CS, P2) hipLaunchKernelGGL((residual_kernel<CS, P2, PixT>), dim3((uint32_t)grid), dim3(256), 0, (hipStream_t)stream, d_src, src_stride, \
                       src_block_pitch, d_pred, pred_stride, pred_block_pitch, d_res, res_stride, res_block_pitch,   \
                       width, height, (uint32_t)nblocks)
#define RESC(CS) if (pow2) RESL(CS, true); else RESL(CS, false)
    if (rcs == 16) { if constexpr (MAXCS == 16) { RESC(16); } } else if (rcs == 8) { RESC(8); } else if (rcs == 4) { RESC(4); } else { RESC(1); }
#undef RESC
#undef RESL
    return launch_status("residual");
}
extern "C" int svt_hip_residual_batch(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch,
                                      const uint8_t* d_pred, uint32_t pred_stride, size_t pred_block_pitch,
                                      int16_t* d_res, uint32_t res_stride, size_t res_block_pitch, uint32_t width,
                                      uint32_t height, size_t nblocks, void* stream) {
    return residual_impl<uint8_t>(d_src, src_stride, src_block_pitch, d_pred, pred_stride, pred_block_pitch, d_res, res_stride,
                                  res_block_pitch, width, height, nblocks, stream);
}
extern "C" int svt_hip_residual16_batch(const uint16_t* d_src, uint32_t src_stride, size_t src_block_pitch,
                                        const uint16_t* d_pred, uint32_t pred_stride, size_t pred_block_pitch,
                                        int16_t* d_res, uint32_t res_stride, size_t res_block_pitch, uint32_t width,
                                        uint32_t height, size_t nblocks, void* stream) {
    return residual_impl<uint16_t>(d_src, src_stride, src_block_pitch, d_pred, pred_stride, pred_block_pitch, d_res, res_stride,
                                   res_block_pitch, width, height, nblocks, stream);
}

static int sad_search_impl(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch, const uint32_t* d_src_offs,
                           const uint8_t* d_ref, uint32_t ref_stride, uint32_t ref_stride_raw,
                           size_t ref_block_pitch, const uint32_t* d_ref_offs, uint32_t width, uint32_t height,
                           int16_t search_area_width, int16_t search_area_height,
                           uint64_t* d_best_sad, int16_t* d_x, int16_t* d_y, size_t nblocks,
                           void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_src || !d_ref || !d_best_sad || !d_x || !d_y) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (width == 0 || height == 0 || width > 64 || height > 64) return set_err(SVT_HIP_ERR_INVALID, "block %ux%u", width, height);
    if (search_area_width <= 0 || search_area_height <= 0) return set_err(SVT_HIP_ERR_INVALID, "empty search area");
    if (nblocks == 0) return SVT_HIP_OK;
    const uint32_t win_w = width + search_area_width - 1;
    const bool plain = ref_stride == ref_stride_raw;
    const uint32_t nrows = plain ? (uint32_t)(search_area_height + height - 1) : (uint32_t)search_area_height * height;
    const bool q16_for_16x16 = !g_tune_no_q16 && width == 16 && search_area_width % 16 == 0;      // see the q16 routing below
    if (plain && !g_tune_no_qsad && !g_tune_no_q2 && ((width == 16 && height == 16 && !q16_for_16x16) || (width == 8 && height == 8))) {
        // small blocks: source block in registers, 4 x 2 candidates per lane
        const uint32_t wpitch = ((win_w + 15) & ~15u) + 16;
        const uint32_t src_bytes = (width * height + 15) & ~15u;
        const uint32_t groups = (uint32_t)((search_area_width + 3) / 4) * (uint32_t)((search_area_height + 1) / 2);
        uint32_t lpb = 1;
        while (lpb < groups && lpb < 64) lpb <<= 1;
        // window + one spare row + 16 spare bytes per lane, padded to 8 (mod 32) bytes (LDS bank spread, see the kernel)
        uint32_t ref_bytes = wpitch * (nrows + 1) + 16 * lpb;
        ref_bytes = ((ref_bytes + 31) & ~31u) + 8;
        const uint32_t per_blk = src_bytes + ref_bytes;
        if (per_blk <= 64 * 1024) {
            uint32_t threads = 256;
            while (threads > lpb && (size_t)(threads / lpb) * per_blk > 64 * 1024) threads >>= 1;
            const uint32_t slots = threads / lpb;
            const uint32_t grid = (uint32_t)((nblocks + slots - 1) / slots);
#define SSQ2(CW, CH, SU)                                                                                                \
    hipLaunchKernelGGL((sad_search_q2_kernel<CW, CH, SU>), dim3(grid), dim3(threads), (size_t)slots * per_blk, (hipStream_t)stream, \
                       d_src, src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, (int)search_area_width,         \
                       (int)search_area_height, (unsigned long long*)d_best_sad, d_x, d_y, ref_bytes, lpb, cpr_magic,           \
                       d_src_offs, d_ref_offs, (uint32_t)nblocks)
            // exact j / cpr for j < 2^16 chunks (cpr <= 8): floor(2^32 / cpr) + 1
            const uint32_t cpr_magic = (uint32_t)(0x100000000ull / ((win_w + 15) >> 4)) + 1u;
            // staging depth: all of a lane's chunks in ONE batch of loads when that takes at most 8 per lane (one memory
            // latency per block instead of two: the kernel is latency-bound at 3 waves per SIMD)
            const uint32_t nchunk = ((win_w + 15) >> 4) * nrows;
            const bool deep = !g_tune_q2_su4 && nchunk > 4 * lpb;
            // persistent, software-pipelined form (loads of the next set of blocks in flight during the search): whenever a
            // lane's share of one block's chunks fits 8 registers-of-16-B
            const uint32_t nsrc_chunks = width * height / (width % 16 == 0 ? 16u : 8u);
            if (!g_tune_no_q2p && nsrc_chunks + nchunk <= 8 * lpb && search_area_width <= 256 && search_area_height <= 256) {
                // LDS per block: source + 16 B, window rows at 16 * cpr + 8 bytes, block stride == 64 (mod 128) (bank spread, see the kernel)
                const uint32_t cpr = (win_w + 15) >> 4, wpitch_p = 16 * cpr + 8;
                uint32_t wstride = wpitch_p * nrows + 8;
                wstride = ((wstride + 63) & ~127u) + 64;
                if (wstride < wpitch_p * nrows + 8) wstride += 128;
                const uint32_t per_blk_p = src_bytes + 16 + wstride;
                uint32_t pthreads = 256;
                while (pthreads > 64 && (size_t)(pthreads / lpb) * per_blk_p > 64 * 1024) pthreads >>= 1;
                const uint32_t pslots = pthreads / lpb, pwaves = pthreads / 64;
                const size_t plds = (size_t)pslots * per_blk_p;
                if (lpb <= 64 && pslots >= pwaves && plds <= 64 * 1024) {
                    const uint32_t nsets = (uint32_t)((nblocks + 64 / lpb - 1) / (64 / lpb));
                    const uint32_t wg_per_cu = (uint32_t)(160 * 1024 / plds);
                    const uint32_t by_waves = 16 / pwaves;                       // 4 waves per SIMD (124 VGPRs)
                    uint32_t pgrid = (uint32_t)g_num_cu * (wg_per_cu < by_waves ? (wg_per_cu ? wg_per_cu : 1) : by_waves);
                    const uint32_t need = (nsets + pwaves - 1) / pwaves;
                    if (pgrid > need) pgrid = need;
#define SSQ2P(CW, CH)                                                                                                   \
    hipLaunchKernelGGL((sad_search_q2p_kernel<CW, CH>), dim3(pgrid), dim3(pthreads), plds, (hipStream_t)stream,                     \
                       d_src, src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, (int)search_area_width,         \
                       (int)search_area_height, (unsigned long long*)d_best_sad, d_x, d_y, wstride, wpitch_p, lpb, cpr_magic,    \
                       d_src_offs, d_ref_offs, (uint32_t)nblocks)
                    if (width == 16) SSQ2P(16, 16); else SSQ2P(8, 8);
#undef SSQ2P
                    return launch_status("sad_search_q2p");
                }
            }
            if (width == 16) { if (deep) SSQ2(16, 16, 8); else SSQ2(16, 16, 4); }
            else { if (deep) SSQ2(8, 8, 8); else SSQ2(8, 8, 4); }
#undef SSQ2
            return launch_status("sad_search_q2");
        }
    }
    // 16-wide blocks: 16x32 / 16x64 always (measured 2.1-2.3x over sad_search_q_kernel); 16x16 when no candidate
    // of a lane is masked (search width % 16 == 0: 10 % over q2, equal otherwise)
    if (plain && !g_tune_no_qsad && !g_tune_no_q16 && height % (256 / (width ? width : 1)) == 0 &&
        (width == 32 || width == 64 || (width == 16 && (height != 16 || search_area_width % 16 == 0)))) {
        // wide blocks: 16 candidates per lane on b128 LDS reads (sad_search_q16_kernel)
        uint32_t wpitch = (((uint32_t)search_area_width + 15) & ~15u) + width;
        if (((wpitch >> 4) & 1) == 0) wpitch += 16;            // odd multiple of 16 B: bank spread over search rows
        const uint32_t ref_bytes = wpitch * nrows;
        const uint32_t per_blk = width * height + ref_bytes;
        if (per_blk <= 64 * 1024) {
            const uint32_t tasks = (uint32_t)((search_area_width + 15) / 16) * (uint32_t)search_area_height;
            uint32_t tsh = 0;
            while ((1u << tsh) < tasks && tsh < 6) tsh++;
            const uint32_t row_groups = height / (256 / width);
            uint32_t lpb = 1u << tsh;
            while (lpb < 64 && (lpb >> tsh) * 2 <= row_groups) lpb <<= 1;
            uint32_t threads = 256;
            while (threads > lpb && (size_t)(threads / lpb) * per_blk > 64 * 1024) threads >>= 1;
            const uint32_t slots = threads / lpb;
            const uint32_t grid = (uint32_t)((nblocks + slots - 1) / slots);
            const uint32_t cpr_magic = (uint32_t)(0x100000000ull / ((win_w + 15) >> 4)) + 1u;
#define SSQ16(CW)                                                                                                       \
    hipLaunchKernelGGL((sad_search_q16_kernel<CW>), dim3(grid), dim3(threads), (size_t)slots * per_blk, (hipStream_t)stream, \
                       d_src, src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, height, (int)search_area_width,  \
                       (int)search_area_height, (unsigned long long*)d_best_sad, d_x, d_y, wpitch, ref_bytes, lpb, tsh,         \
                       cpr_magic, d_src_offs, d_ref_offs, (uint32_t)nblocks)
            if (width == 16) SSQ16(16); else if (width == 32) SSQ16(32); else SSQ16(64);
#undef SSQ16
            return launch_status("sad_search_q16");
        }
    }
    if ((width & 3) == 0 && !g_tune_no_qsad) {
        // quad-SAD kernel: 4 candidates per lane
        const uint32_t wpitch = ((win_w + 15) & ~15u) + 16;
        const uint32_t src_bytes = (width * height + 15) & ~15u;
        const uint32_t ref_bytes = (wpitch * nrows + 16 + 15) & ~15u;
        const uint32_t per_blk = src_bytes + ref_bytes;
        if (per_blk > 64 * 1024) return set_err(SVT_HIP_ERR_INVALID, "search window needs %u B of LDS per block (> 64 KiB)", per_blk);
        const uint32_t groups = (uint32_t)((search_area_width + 3) / 4) * (uint32_t)search_area_height;
        uint32_t lpb = 1;
        while (lpb < groups && lpb < 64) lpb <<= 1;
        uint32_t threads = 256;
        while (threads > lpb && (size_t)(threads / lpb) * per_blk > 64 * 1024) threads >>= 1;
        const uint32_t slots = threads / lpb;
        const uint32_t grid = (uint32_t)((nblocks + slots - 1) / slots);
#define SSQ(CW, CH)                                                                                                     \
    hipLaunchKernelGGL((sad_search_q_kernel<CW, CH>), dim3(grid), dim3(threads), (size_t)slots * per_blk, (hipStream_t)stream, \
                       d_src, src_stride, src_block_pitch, d_ref, ref_stride, ref_stride_raw, ref_block_pitch, width, height,  \
                       (int)search_area_width, (int)search_area_height, (unsigned long long*)d_best_sad, d_x, d_y, src_bytes, \
                       ref_bytes, lpb, d_src_offs, d_ref_offs, (uint32_t)nblocks)
        if (width == 16 && height == 16) SSQ(16, 16);
        else if (width == 8 && height == 8) SSQ(8, 8);
        else if (width == 32 && height == 32) SSQ(32, 32);
        else if (width == 64 && height == 64) SSQ(64, 64);
        else SSQ(0, 0);
#undef SSQ
        return launch_status("sad_search_q");
    }
    const uint32_t wpitch = (win_w + 3 + 8) & ~3u;
    const uint32_t spitch = (width + 3) & ~3u;
    const uint32_t src_bytes = (spitch * height + 15) & ~15u;
    const uint32_t ref_bytes = (wpitch * nrows + 16 + 15) & ~15u;
    const uint32_t per_wave = src_bytes + ref_bytes;
    if (per_wave > 64 * 1024) return set_err(SVT_HIP_ERR_INVALID, "search window needs %u B of LDS per block (> 64 KiB)", per_wave);
    uint32_t waves = (64 * 1024) / per_wave;
    if (waves > 4) waves = 4;
    const uint32_t grid = (uint32_t)((nblocks + waves - 1) / waves);
    hipLaunchKernelGGL(sad_search_kernel, dim3(grid), dim3(waves * 64), waves * per_wave, (hipStream_t)stream, d_src,
                       src_stride, src_block_pitch, d_ref, ref_stride, ref_stride_raw, ref_block_pitch, width, height,
                       (int)search_area_width, (int)search_area_height, (unsigned long long*)d_best_sad, d_x, d_y,
                       src_bytes, ref_bytes, d_src_offs, d_ref_offs, (uint32_t)nblocks);
    return launch_status("sad_search");
}

extern "C" int svt_hip_sad_search_batch(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch,
                                        const uint8_t* d_ref, uint32_t ref_stride, uint32_t ref_stride_raw,
                                        size_t ref_block_pitch, uint32_t width, uint32_t height,
                                        int16_t search_area_width, int16_t search_area_height,
                                        uint64_t* d_best_sad, int16_t* d_x, int16_t* d_y, size_t nblocks,
                                        void* stream) {
    return sad_search_impl(d_src, src_stride, src_block_pitch, nullptr, d_ref, ref_stride, ref_stride_raw, ref_block_pitch,
                           nullptr, width, height, search_area_width, search_area_height, d_best_sad, d_x, d_y, nblocks, stream);
}
extern "C" int svt_hip_sad_search_planes_batch(const uint8_t* d_src_plane, uint32_t src_stride, const uint32_t* d_src_offsets,
                                               const uint8_t* d_ref_plane, uint32_t ref_stride, uint32_t ref_stride_raw,
                                               const uint32_t* d_ref_offsets, uint32_t width, uint32_t height,
                                               int16_t search_area_width, int16_t search_area_height,
                                               uint64_t* d_best_sad, int16_t* d_x, int16_t* d_y, size_t nblocks,
                                               void* stream) {
    if (nblocks && (!d_src_offsets || !d_ref_offsets)) { if (int rc = require_init()) return rc; return set_err(SVT_HIP_ERR_INVALID, "NULL offset table"); }
    return sad_search_impl(d_src_plane, src_stride, 0, d_src_offsets, d_ref_plane, ref_stride, ref_stride_raw, 0, d_ref_offsets,
                           width, height, search_area_width, search_area_height, d_best_sad, d_x, d_y, nblocks, stream);
}

// Row pitch of the reference window the motion-search kernels stage in LDS: whole 16-byte chunks + one chunk of slack for the
// sliding reads, and == 64 (mod 128) so that the lanes of consecutive SEARCH ROWS fall on different halves of the 32 banks.  A
// wave's b128 reads are served eight lanes at a time - four 16-point groups of one search row and four of the next - and with
// the first pitch (144 B for a 64-wide area, == 16 mod 128) the second row's lanes landed on the first row's banks: 31 % of the
// kernel's LDS cycles were bank conflicts (profiles/r03_a_pmc_me_sb.json); the 4-points-per-lane kernel's dword reads (16 lanes
// per search row) collide the same way.
static uint32_t me_window_pitch(uint32_t win_w) {
    uint32_t p = ((win_w + 15) & ~15u) + 16;
    while ((p & 127) != 64) p += 16;
    return p;
}

static int me_sb_search_impl(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch, const uint32_t* d_src_offs,
                             const uint8_t* d_ref, uint32_t ref_stride, size_t ref_block_pitch, const uint32_t* d_ref_offs,
                             int search_w, int search_h, const int16_t* d_origins, int x_origin,
                             int y_origin, uint32_t* d_best_sad, uint32_t* d_best_mv, size_t nblocks,
                             void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_src || !d_ref || !d_best_sad || !d_best_mv) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (search_w <= 0 || search_h <= 0 || search_w * search_h > 4096)
        return set_err(SVT_HIP_ERR_INVALID, "search area %dx%d (1..4096 points)", search_w, search_h);
    const uint32_t win_w = 64 + search_w - 1, win_h = 64 + search_h - 1;
    const uint32_t wpitch = me_window_pitch(win_w);
    const size_t lds = 32 * 64 + (size_t)wpitch * win_h;
    if (lds > 60 * 1024) return set_err(SVT_HIP_ERR_INVALID, "search window needs %zu B of LDS (> 60 KiB)", lds);
    // 16 points per lane; widths that are not a multiple of 16 mask the tail of each row
    if ((search_w & 15) == 0)
        hipLaunchKernelGGL(me_sb_search16_kernel<false>, dim3((uint32_t)nblocks), dim3(ME_THREADS), lds, (hipStream_t)stream, d_src,
                           src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, search_w, search_h, d_origins,
                           x_origin, y_origin, d_best_sad, d_best_mv, wpitch, d_src_offs, d_ref_offs, (uint32_t)nblocks);
    else
        hipLaunchKernelGGL(me_sb_search16_kernel<true>, dim3((uint32_t)nblocks), dim3(ME_THREADS), lds, (hipStream_t)stream, d_src,
                           src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, search_w, search_h, d_origins,
                           x_origin, y_origin, d_best_sad, d_best_mv, wpitch, d_src_offs, d_ref_offs, (uint32_t)nblocks);
    return launch_status("me_sb_search16");
}

extern "C" int svt_hip_me_sb_search_batch(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch,
                                          const uint8_t* d_ref, uint32_t ref_stride, size_t ref_block_pitch,
                                          int search_w, int search_h, const int16_t* d_origins, int x_origin,
                                          int y_origin, uint32_t* d_best_sad, uint32_t* d_best_mv, size_t nblocks,
                                          void* stream) {
    return me_sb_search_impl(d_src, src_stride, src_block_pitch, nullptr, d_ref, ref_stride, ref_block_pitch, nullptr, search_w,
                             search_h, d_origins, x_origin, y_origin, d_best_sad, d_best_mv, nblocks, stream);
}
extern "C" int svt_hip_me_sb_search_planes_batch(const uint8_t* d_src_plane, uint32_t src_stride, const uint32_t* d_src_offsets,
                                                 const uint8_t* d_ref_plane, uint32_t ref_stride, const uint32_t* d_ref_offsets,
                                                 int search_w, int search_h, const int16_t* d_origins, int x_origin,
                                                 int y_origin, uint32_t* d_best_sad, uint32_t* d_best_mv, size_t nblocks,
                                                 void* stream) {
    if (nblocks && (!d_src_offsets || !d_ref_offsets)) { if (int rc = require_init()) return rc; return set_err(SVT_HIP_ERR_INVALID, "NULL offset table"); }
    return me_sb_search_impl(d_src_plane, src_stride, 0, d_src_offsets, d_ref_plane, ref_stride, 0, d_ref_offsets, search_w,
                             search_h, d_origins, x_origin, y_origin, d_best_sad, d_best_mv, nblocks, stream);
}

// one HME level for many SBs, search-area clipping on the device; 1 .. 4 search regions per launch (include/svt_hip_dsp.h)
extern "C" int svt_hip_hme_level_regions_batch(const uint8_t* d_src_pic, uint32_t src_stride, const uint8_t* d_ref_pic, uint32_t ref_stride,
                                               const int16_t* d_sb_origin, const uint16_t* d_sb_size, const int16_t* d_centers,
                                               int center_shift, const svt_hip_hme_params* params, int nregions, uint64_t* d_best_sad,
                                               int16_t* d_mv, size_t ntasks, void* stream) {
    if (int rc = require_init()) return rc;
    if (ntasks == 0) return SVT_HIP_OK;
    if (!d_src_pic || !d_ref_pic || !d_sb_origin || !d_sb_size || !params || !d_best_sad || !d_mv) return set_err(SVT_HIP_ERR_INVALID, "NULL argument");
    if (nregions < 1 || nregions > 4) return set_err(SVT_HIP_ERR_INVALID, "%d search regions (1..4)", nregions);
    if (center_shift < 0 || center_shift > 2) return set_err(SVT_HIP_ERR_INVALID, "centre shift %d", center_shift);
    if (ntasks > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many tasks");
    HmeParamSets hp;
    static_assert(sizeof(HmeParams) == sizeof(svt_hip_hme_params), "layout");
    memset(&hp, 0, sizeof(hp));
    size_t lds = 0;
    uint32_t wpitch = 0;
    for (int r = 0; r < nregions; r++) {
        const svt_hip_hme_params& P = params[r];
        if (P.search_area_width < 1 || P.search_area_height < 1 || (P.round_down != 8 && P.round_down != 16) || P.mv_shift < 0 || P.mv_shift > 2 ||
            P.ref_width < 1 || P.ref_height < 1)
            return set_err(SVT_HIP_ERR_INVALID, "HME parameters: area %dx%d, round_down %d, mv_shift %d", P.search_area_width, P.search_area_height,
                           P.round_down, P.mv_shift);
        // the clipped area never exceeds the nominal one; blocks are at most 64 x 64 (32 compared rows)
        const uint32_t win_w = 64 + (uint32_t)P.search_area_width - 1;
        const uint32_t wp = ((win_w + 3) & ~3u) + 8;
        if (wp > wpitch) wpitch = wp;
        memcpy(&hp.p[r], &P, sizeof(HmeParams));
    }
    for (int r = 0; r < nregions; r++) {
        const size_t need = 32 * 64 + (size_t)wpitch * ((uint32_t)params[r].search_area_height + 62);
        if (need > lds) lds = need;
    }
    if (lds > 60 * 1024) return set_err(SVT_HIP_ERR_INVALID, "HME search window needs %zu B of LDS (> 60 KiB)", lds);
    hipLaunchKernelGGL(hme_level_kernel, dim3((uint32_t)ntasks, (uint32_t)nregions), dim3(ME_THREADS), lds, (hipStream_t)stream, d_src_pic, src_stride,
                       d_ref_pic, ref_stride, d_sb_origin, d_sb_size, d_centers, center_shift, hp, (unsigned long long*)d_best_sad, d_mv, wpitch,
                       (uint32_t)ntasks);
    return launch_status("hme_level");
}
extern "C" int svt_hip_hme_level_batch(const uint8_t* d_src_pic, uint32_t src_stride, const uint8_t* d_ref_pic, uint32_t ref_stride,
                                       const int16_t* d_sb_origin, const uint16_t* d_sb_size, const int16_t* d_centers,
                                       int center_shift, const svt_hip_hme_params* params, uint64_t* d_best_sad, int16_t* d_mv,
                                       size_t ntasks, void* stream) {
    return svt_hip_hme_level_regions_batch(d_src_pic, src_stride, d_ref_pic, ref_stride, d_sb_origin, d_sb_size, d_centers, center_shift, params, 1,
                                           d_best_sad, d_mv, ntasks, stream);
}

// K6 in the reference's result layout, both result flavours, square or all 209 PUs (include/svt_hip_dsp.h)
extern "C" int svt_hip_me_fullpel_search_batch(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch,
                                               const uint32_t* d_src_offsets, const uint8_t* d_ref, uint32_t ref_stride,
                                               size_t ref_block_pitch, const uint32_t* d_ref_offsets, int search_w,
                                               int search_h, const int16_t* d_origins, int x_origin, int y_origin,
                                               int flavour, int nsq, uint32_t* d_best_sad, uint32_t* d_best_mv,
                                               uint32_t pu_pitch, size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_src || !d_ref || !d_best_sad || !d_best_mv) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (flavour != SVT_HIP_FLAVOUR_C && flavour != SVT_HIP_FLAVOUR_AVX2) return set_err(SVT_HIP_ERR_INVALID, "flavour %d", flavour);
    const uint32_t npus = nsq ? SVT_HIP_ME_PUS_ALL : SVT_HIP_ME_PUS;
    if (pu_pitch < npus) return set_err(SVT_HIP_ERR_INVALID, "pu_pitch %u < %u PUs", pu_pitch, npus);
    if (search_w <= 0 || search_h <= 0 || search_w * search_h > 4096)
        return set_err(SVT_HIP_ERR_INVALID, "search area %dx%d (1..4096 points)", search_w, search_h);
    if (!nsq && !g_tune_me_exact) {
        // square PUs: the 16-points-per-lane kernel, any width; the AVX2 flavour only re-labels the 32x32 keys
        const uint32_t win_w = 64 + search_w - 1, win_h = 64 + search_h - 1;
        const uint32_t wpitch = me_window_pitch(win_w);
        const size_t lds = 32 * 64 + (size_t)wpitch * win_h;
        if (lds <= 60 * 1024) {
            const int w8q = flavour == SVT_HIP_FLAVOUR_AVX2 ? (search_w & ~7) : 0;
            if ((search_w & 15) == 0)
                hipLaunchKernelGGL(me_sb_search16_kernel<false>, dim3((uint32_t)nblocks), dim3(ME_THREADS), lds, (hipStream_t)stream, d_src,
                                   src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, search_w, search_h, d_origins,
                                   x_origin, y_origin, d_best_sad, d_best_mv, wpitch, d_src_offsets, d_ref_offsets, (uint32_t)nblocks,
                                   w8q, 1, pu_pitch);
            else
                hipLaunchKernelGGL(me_sb_search16_kernel<true>, dim3((uint32_t)nblocks), dim3(ME_THREADS), lds, (hipStream_t)stream, d_src,
                                   src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, search_w, search_h, d_origins,
                                   x_origin, y_origin, d_best_sad, d_best_mv, wpitch, d_src_offsets, d_ref_offsets, (uint32_t)nblocks,
                                   w8q, 1, pu_pitch);
            return launch_status("me_sb_search16 (reference layout)");
        }
    }
    if (nsq && !g_tune_me_exact && (search_w & 7) == 0) {
        // all 209 PUs, widths where every point takes the reference's eight-point form: 4 points per lane, every shape folded
        // out of the packed 8x8 SADs (me_nsq4_kernel); other widths keep the exact kernel (single-point quirks)
        const uint32_t win_w = 64 + search_w - 1, win_h = 64 + search_h - 1;
        const uint32_t wpitch = me_window_pitch(win_w);
        const size_t lds = 32 * 64 + (size_t)wpitch * win_h;
        if (lds <= 58 * 1024) {
            hipLaunchKernelGGL(me_nsq4_kernel, dim3((uint32_t)nblocks), dim3(ME_THREADS), lds, (hipStream_t)stream, d_src, src_stride,
                               src_block_pitch, d_ref, ref_stride, ref_block_pitch, search_w, search_h, d_origins, x_origin, y_origin,
                               d_best_sad, d_best_mv, wpitch, d_src_offsets, d_ref_offsets, (uint32_t)nblocks, pu_pitch);
            return launch_status("me_nsq4");
        }
    }
    hipLaunchKernelGGL(me_fullpel_exact_kernel, dim3((uint32_t)nblocks), dim3(ME_THREADS), 0, (hipStream_t)stream, d_src, src_stride,
                       src_block_pitch, d_src_offsets, d_ref, ref_stride, ref_block_pitch, d_ref_offsets, search_w, search_h,
                       d_origins, x_origin, y_origin, flavour, nsq ? 1 : 0, d_best_sad, d_best_mv, pu_pitch, (uint32_t)nblocks);
    return launch_status("me_fullpel_exact");
}

// ---- MotionEstimateLcu's glue: set-up, per-SB areas, bi-prediction + result rows (include/svt_hip_dsp.h) --------------------
extern "C" int svt_hip_me_setup_batch(const uint8_t* d_src_pic, uint32_t src_stride, const uint8_t* d_ref_pic, uint32_t ref_stride,
                                      const int16_t* d_sb_origin, const uint16_t* d_sb_size, const uint64_t* d_hme_sad, const int16_t* d_hme_mv,
                                      const svt_hip_me_setup_params* params, int16_t* d_center, int16_t* d_area, size_t ntasks, void* stream) {
    if (int rc = require_init()) return rc;
    if (ntasks == 0) return SVT_HIP_OK;
    if (!d_src_pic || !d_ref_pic || !d_sb_origin || !d_sb_size || !params || !d_area) return set_err(SVT_HIP_ERR_INVALID, "NULL argument");
    if ((d_hme_sad == nullptr) != (d_hme_mv == nullptr)) return set_err(SVT_HIP_ERR_INVALID, "d_hme_sad and d_hme_mv go together");
    const svt_hip_me_setup_params& P = *params;
    if (P.regions_w < 0 || P.regions_w > 2 || P.regions_h < 0 || P.regions_h > 2 || (P.regions_w == 0) != (P.regions_h == 0))
        return set_err(SVT_HIP_ERR_INVALID, "%d x %d search regions (0 x 0, or 1 .. 2 each)", P.regions_w, P.regions_h);
    // the reference's sort walks its [width][height] arrays through [q / regions_w][q % regions_w]: outside a square grid that reads
    // entries no level has written (stack garbage in the reference) - refused rather than guessed
    if (P.second_best && P.regions_w != P.regions_h) return set_err(SVT_HIP_ERR_INVALID, "second_best needs regions_w == regions_h");
    if (P.search_area_width < 1 || P.search_area_height < 1 || P.search_area_width > 4096 || P.search_area_height > 4096 || P.picture_width < 1 ||
        P.picture_height < 1 || P.ref_width < 1 || P.ref_height < 1)
        return set_err(SVT_HIP_ERR_INVALID, "search area %d x %d, picture %d x %d", P.search_area_width, P.search_area_height, P.picture_width, P.picture_height);
    if (ntasks > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many tasks");
    static_assert(sizeof(MeSetupParams) == sizeof(svt_hip_me_setup_params), "layout");
    MeSetupParams mp;
    memcpy(&mp, params, sizeof(mp));
    hipLaunchKernelGGL(me_setup_kernel, dim3((uint32_t)((ntasks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_src_pic, src_stride, d_ref_pic,
                       ref_stride, d_sb_origin, d_sb_size, (const unsigned long long*)d_hme_sad, d_hme_mv, mp, d_center, d_area, (uint32_t)ntasks);
    return launch_status("me_setup");
}

extern "C" int svt_hip_me_fullpel_search_areas_batch(const uint8_t* d_src, uint32_t src_stride, const uint32_t* d_src_offsets, const uint8_t* d_ref,
                                                     uint32_t ref_stride, const uint32_t* d_ref_offsets, const int16_t* d_areas, int max_search_w,
                                                     int max_search_h, int flavour, int nsq, uint32_t* d_best_sad, uint32_t* d_best_mv,
                                                     uint32_t pu_pitch, size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_src || !d_ref || !d_src_offsets || !d_ref_offsets || !d_areas || !d_best_sad || !d_best_mv) return set_err(SVT_HIP_ERR_INVALID, "NULL argument");
    if (flavour != SVT_HIP_FLAVOUR_C && flavour != SVT_HIP_FLAVOUR_AVX2) return set_err(SVT_HIP_ERR_INVALID, "flavour %d", flavour);
    const uint32_t npus = nsq ? SVT_HIP_ME_PUS_ALL : SVT_HIP_ME_PUS;
    if (pu_pitch < npus) return set_err(SVT_HIP_ERR_INVALID, "pu_pitch %u < %u PUs", pu_pitch, npus);
    if (max_search_w <= 0 || max_search_h <= 0 || max_search_w * max_search_h > 4096)
        return set_err(SVT_HIP_ERR_INVALID, "search area bound %dx%d (1..4096 points)", max_search_w, max_search_h);
    if (nblocks > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many blocks");
    const uint32_t win_w = 64 + max_search_w - 1, win_h = 64 + max_search_h - 1;
    const uint32_t wpitch = me_window_pitch(win_w);
    size_t lds = 32 * 64 + (size_t)wpitch * win_h;
    uint32_t pair_off = 0;
    if (nsq) {                                            // (64x32_1, 32x16_5) pairs of the narrow single-point areas: <= 7 x max_h points
        pair_off = (uint32_t)((lds + 15) & ~(size_t)15);
        lds = pair_off + (size_t)8 * 7 * max_search_h;
    }
    if (lds > 58 * 1024) return set_err(SVT_HIP_ERR_INVALID, "search window of %dx%d needs %zu B of LDS (> 58 KiB)", max_search_w, max_search_h, lds);
    if (nsq)
        hipLaunchKernelGGL(me_fullpel_areas_kernel<true>, dim3((uint32_t)nblocks), dim3(ME_THREADS), lds, (hipStream_t)stream, d_src, src_stride, d_src_offsets,
                           d_ref, ref_stride, d_ref_offsets, d_areas, max_search_w, max_search_h, flavour, d_best_sad, d_best_mv, pu_pitch, wpitch,
                           pair_off, (uint32_t)nblocks);
    else
        hipLaunchKernelGGL(me_fullpel_areas_kernel<false>, dim3((uint32_t)nblocks), dim3(ME_THREADS), lds, (hipStream_t)stream, d_src, src_stride, d_src_offsets,
                           d_ref, ref_stride, d_ref_offsets, d_areas, max_search_w, max_search_h, flavour, d_best_sad, d_best_mv, pu_pitch, wpitch,
                           pair_off, (uint32_t)nblocks);
    return launch_status("me_fullpel_areas");
}

// raster PU order (partitionWidth / partitionHeight / puSearchIndexMap, EbMotionEstimation.h:178-315): 14 shape groups, each a
// raster walk of its grid; the storage index is the PU of the result rows with the same rectangle (me_pu_rect)
static const MePuMap& me_pu_map() {
    static const MePuMap m = [] {
        MePuMap t;
        memset(&t, 0, sizeof(t));
        static const int gw[14] = {8, 4, 2, 1, 8, 4, 2, 4, 2, 1, 4, 1, 8, 2}, gh[14] = {8, 4, 2, 1, 4, 2, 1, 8, 4, 2, 1, 4, 2, 8};
        int p = 0, rows = 0;
        for (int g = 0; g < 14; g++) {
            const int cols = 8 / gw[g], n = cols * (8 / gh[g]);
            for (int i = 0; i < n; i++, p++) {
                const int x = (i % cols) * gw[g], y = (i / cols) * gh[g];
                t.x8[p] = (uint8_t)x; t.y8[p] = (uint8_t)y; t.w8[p] = (uint8_t)gw[g]; t.h8[p] = (uint8_t)gh[g];
                t.row0[p] = (uint16_t)rows;
                rows += 4 * gh[g];
                int found = 255;
                for (int q = 0; q < ME_PUS_ALL; q++) {
                    int qx, qy, qw, qh;
                    me_pu_rect(q, qx, qy, qw, qh);
                    if (qx == x && qy == y && qw == gw[g] && qh == gh[g]) { found = q; break; }
                }
                t.storage[p] = (uint8_t)found;
            }
        }
        t.row0[ME_PUS_ALL] = (uint16_t)rows;
        return t;
    }();
    return m;
}
extern "C" int svt_hip_me_pu_storage_index(int pu_index) {
    if (pu_index < 0 || pu_index >= ME_PUS_ALL) return -1;
    return me_pu_map().storage[pu_index];
}

extern "C" int svt_hip_me_bipred_batch(const uint8_t* d_src_pic, uint32_t src_stride, const uint8_t* d_ref0_pic, uint32_t ref0_stride,
                                       const uint8_t* d_ref1_pic, uint32_t ref1_stride, const int16_t* d_sb_origin, const uint32_t* d_best_sad0,
                                       const uint32_t* d_best_mv0, const uint32_t* d_best_sad1, const uint32_t* d_best_mv1, uint32_t pu_pitch, int npus,
                                       int bipred_all_pus, int sub_sad, uint32_t* d_bipred_sad, svt_hip_me_result* d_results, size_t nsb, void* stream) {
    if (int rc = require_init()) return rc;
    if (nsb == 0) return SVT_HIP_OK;
    if (!d_src_pic || !d_sb_origin || !d_best_sad0 || !d_best_mv0 || !d_results) return set_err(SVT_HIP_ERR_INVALID, "NULL argument");
    if ((d_best_sad1 == nullptr) != (d_best_mv1 == nullptr)) return set_err(SVT_HIP_ERR_INVALID, "d_best_sad1 and d_best_mv1 go together");
    if (d_best_sad1 && (!d_ref0_pic || !d_ref1_pic)) return set_err(SVT_HIP_ERR_INVALID, "two lists need both reference planes");
    if (npus != SVT_HIP_ME_PUS && npus != SVT_HIP_ME_PUS_ALL) return set_err(SVT_HIP_ERR_INVALID, "npus %d (85 or 209)", npus);
    if (pu_pitch < (uint32_t)npus) return set_err(SVT_HIP_ERR_INVALID, "pu_pitch %u < %d PUs", pu_pitch, npus);
    if (nsb > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many SBs");
    static_assert(sizeof(MeResult) == sizeof(svt_hip_me_result), "layout");
    hipLaunchKernelGGL(me_bipred_kernel, dim3((uint32_t)nsb), dim3(ME_THREADS), 0, (hipStream_t)stream, d_src_pic, src_stride, d_ref0_pic, ref0_stride,
                       d_ref1_pic, ref1_stride, d_sb_origin, d_best_sad0, d_best_mv0, d_best_sad1, d_best_mv1, pu_pitch, npus, bipred_all_pus ? 1 : 0,
                       sub_sad ? 1 : 0, me_pu_map(), d_bipred_sad, reinterpret_cast<MeResult*>(d_results), (uint32_t)nsb);
    return launch_status("me_bipred");
}

static int full_distortion32_impl(const int32_t* d_coeff, uint32_t coeff_stride, size_t coeff_block_pitch, const int32_t* d_recon,
                                  uint32_t recon_stride, size_t recon_block_pitch, uint32_t width, uint32_t height,
                                  int cbf_zero, const uint32_t* d_nz, int flavour, uint64_t* d_out, size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_coeff || !d_out || (!cbf_zero && !d_recon)) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (width == 0 || height == 0 || width > 128 || height > 128) return set_err(SVT_HIP_ERR_INVALID, "area %ux%u", width, height);
    if (flavour != SVT_HIP_FLAVOUR_C && flavour != SVT_HIP_FLAVOUR_AVX2) return set_err(SVT_HIP_ERR_INVALID, "flavour %d", flavour);
    if (flavour == SVT_HIP_FLAVOUR_AVX2 && (width & 3)) return set_err(SVT_HIP_ERR_INVALID, "the AVX2 kernel is defined for widths that are a multiple of 4");
    const dim3 grid((uint32_t)((nblocks + 15) / 16));
    if (flavour == SVT_HIP_FLAVOUR_AVX2)
        hipLaunchKernelGGL(full_distortion32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, d_coeff, coeff_stride, coeff_block_pitch,
                           d_recon, recon_stride, recon_block_pitch, width, height, cbf_zero, d_nz, (unsigned long long*)d_out, (uint32_t)nblocks);
    else
        hipLaunchKernelGGL(full_distortion32_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, d_coeff, coeff_stride, coeff_block_pitch,
                           d_recon, recon_stride, recon_block_pitch, width, height, cbf_zero, d_nz, (unsigned long long*)d_out, (uint32_t)nblocks);
    return launch_status("full_distortion32");
}
extern "C" int svt_hip_full_distortion32_batch(const int32_t* d_coeff, uint32_t coeff_stride, size_t coeff_block_pitch,
                                               const int32_t* d_recon, uint32_t recon_stride, size_t recon_block_pitch,
                                               uint32_t width, uint32_t height, int cbf_zero, uint64_t* d_out,
                                               size_t nblocks, void* stream) {
    return full_distortion32_impl(d_coeff, coeff_stride, coeff_block_pitch, d_recon, recon_stride, recon_block_pitch, width, height,
                                  cbf_zero, nullptr, SVT_HIP_FLAVOUR_C, d_out, nblocks, stream);
}
extern "C" int svt_hip_picture_full_distortion32_batch(const int32_t* d_coeff, size_t coeff_block_pitch, const int32_t* d_recon,
                                                       size_t recon_block_pitch, uint32_t bwidth, uint32_t bheight,
                                                       const uint32_t* d_count_non_zero_coeffs, int flavour, uint64_t* d_out,
                                                       size_t nblocks, void* stream) {
    // picture_full_distortion32_bits (EbPictureOperators.c:349-457): a 64-sample dimension covers 32 coefficients, the row
    // stride of both buffers is the (clamped) width, count_non_zero_coeffs == 0 selects the cbf_zero kernel per block
    const uint32_t w = bwidth < 64 ? bwidth : 32, h = bheight < 64 ? bheight : 32;
    if (nblocks && !d_recon) { if (int rc = require_init()) return rc; return set_err(SVT_HIP_ERR_INVALID, "NULL buffer"); }
    return full_distortion32_impl(d_coeff, w, coeff_block_pitch, d_recon, w, recon_block_pitch, w, h, 0, d_count_non_zero_coeffs, flavour,
                                  d_out, nblocks, stream);
}

// copies a w x h u8 block with `stride` into dense device memory
static void h2d_block(void* d, const uint8_t* h, uint32_t stride, uint32_t w, uint32_t ht, const char* fn) {
    HIP_DIE(hipMemcpy2DAsync(d, w, h, stride, w, ht, hipMemcpyHostToDevice, t_ctx.stream), fn);
}

extern "C" uint32_t svt_hip_nxm_sad_kernel(const uint8_t* src, uint32_t src_stride, const uint8_t* ref,
                                           uint32_t ref_stride, uint32_t height, uint32_t width) {
    const char* fn = "svt_hip_nxm_sad_kernel";
    const size_t bb = align256((size_t)width * height);
    DROPIN_TRY(t_ctx.ensure(2 * bb + 256), fn);
    uint8_t* d_a = (uint8_t*)t_ctx.dbuf;
    uint8_t* d_b = d_a + bb;
    uint32_t* d_o = (uint32_t*)(d_a + 2 * bb);
    h2d_block(d_a, src, src_stride, width, height, fn);
    h2d_block(d_b, ref, ref_stride, width, height, fn);
    DROPIN_TRY(svt_hip_sad_batch(d_a, width, 0, d_b, width, 0, width, height, d_o, 1, t_ctx.stream), fn);
    uint32_t out = 0;
    HIP_DIE(hipMemcpyAsync(&out, d_o, 4, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
    return out;
}
extern "C" uint64_t svt_hip_spatial_full_distortion_kernel(uint8_t* input, uint32_t input_stride, uint8_t* recon,
                                                           uint32_t recon_stride, uint32_t area_width,
                                                           uint32_t area_height) {
    const char* fn = "svt_hip_spatial_full_distortion_kernel";
    const size_t bb = align256((size_t)area_width * area_height);
    DROPIN_TRY(t_ctx.ensure(2 * bb + 256), fn);
    uint8_t* d_a = (uint8_t*)t_ctx.dbuf;
    uint8_t* d_b = d_a + bb;
    uint64_t* d_o = (uint64_t*)(d_a + 2 * bb);
    h2d_block(d_a, input, input_stride, area_width, area_height, fn);
    h2d_block(d_b, recon, recon_stride, area_width, area_height, fn);
    DROPIN_TRY(svt_hip_sse_batch(d_a, area_width, 0, d_b, area_width, 0, area_width, area_height, d_o, 1, t_ctx.stream), fn);
    uint64_t out = 0;
    HIP_DIE(hipMemcpyAsync(&out, d_o, 8, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
    return out;
}
extern "C" void svt_hip_residual_kernel(uint8_t* input, uint32_t input_stride, uint8_t* pred, uint32_t pred_stride,
                                        int16_t* residual, uint32_t residual_stride, uint32_t area_width,
                                        uint32_t area_height) {
    const char* fn = "svt_hip_residual_kernel";
    const size_t bb = align256((size_t)area_width * area_height);
    DROPIN_TRY(t_ctx.ensure(4 * bb), fn);
    uint8_t* d_a = (uint8_t*)t_ctx.dbuf;
    uint8_t* d_b = d_a + bb;
    int16_t* d_r = (int16_t*)(d_a + 2 * bb);
    h2d_block(d_a, input, input_stride, area_width, area_height, fn);
    h2d_block(d_b, pred, pred_stride, area_width, area_height, fn);
    DROPIN_TRY(svt_hip_residual_batch(d_a, area_width, 0, d_b, area_width, 0, d_r, area_width, 0, area_width, area_height, 1, t_ctx.stream), fn);
    HIP_DIE(hipMemcpy2DAsync(residual, (size_t)residual_stride * 2, d_r, (size_t)area_width * 2, (size_t)area_width * 2,
                             area_height, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
extern "C" void svt_hip_sad_loop_kernel(uint8_t* src, uint32_t src_stride, uint8_t* ref, uint32_t ref_stride,
                                        uint32_t height, uint32_t width, uint64_t* best_sad, int16_t* x_search_center,
                                        int16_t* y_search_center, uint32_t src_stride_raw, int16_t search_area_width,
                                        int16_t search_area_height) {
    const char* fn = "svt_hip_sad_loop_kernel";
    // stage the touched source rows and the touched reference span as-is (strides kept)
    const size_t src_span = (size_t)(height - 1) * src_stride + width;
    const size_t ref_span = (size_t)(search_area_height - 1) * src_stride_raw + (size_t)(height - 1) * ref_stride +
                            width + search_area_width - 1;
    const size_t sb = align256(src_span), rb = align256(ref_span);
    DROPIN_TRY(t_ctx.ensure(sb + rb + 256), fn);
    uint8_t* d_s = (uint8_t*)t_ctx.dbuf;
    uint8_t* d_r = d_s + sb;
    uint64_t* d_best = (uint64_t*)(d_r + rb);
    int16_t* d_xy = (int16_t*)(d_best + 1);
    int16_t xy[2] = {*x_search_center, *y_search_center};
    HIP_DIE(hipMemcpyAsync(d_s, src, src_span, hipMemcpyHostToDevice, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(d_r, ref, ref_span, hipMemcpyHostToDevice, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(d_xy, xy, 4, hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_sad_search_batch(d_s, src_stride, 0, d_r, ref_stride, src_stride_raw, 0, width, height,
                                        search_area_width, search_area_height, d_best, d_xy, d_xy + 1, 1, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(best_sad, d_best, 8, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(xy, d_xy, 4, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
    *x_search_center = xy[0];
    *y_search_center = xy[1];
}

static void dropin_dist32(int cbf_zero, int32_t* coeff, uint32_t cs, int32_t* recon, uint32_t rs, uint64_t out[2],
                          uint32_t w, uint32_t h, const char* fn) {
    const size_t bb = align256((size_t)w * h * 4);
    DROPIN_TRY(t_ctx.ensure(2 * bb + 256), fn);
    int32_t* d_c = (int32_t*)t_ctx.dbuf;
    int32_t* d_r = (int32_t*)(t_ctx.dbuf + bb);
    uint64_t* d_o = (uint64_t*)(t_ctx.dbuf + 2 * bb);
    HIP_DIE(hipMemcpy2DAsync(d_c, (size_t)w * 4, coeff, (size_t)cs * 4, (size_t)w * 4, h, hipMemcpyHostToDevice, t_ctx.stream), fn);
    if (!cbf_zero)
        HIP_DIE(hipMemcpy2DAsync(d_r, (size_t)w * 4, recon, (size_t)rs * 4, (size_t)w * 4, h, hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_full_distortion32_batch(d_c, w, 0, d_r, w, 0, w, h, cbf_zero, d_o, 1, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(out, d_o, 16, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
extern "C" void svt_hip_full_distortion_kernel32_bits(int32_t* coeff, uint32_t coeff_stride, int32_t* recon_coeff,
                                                      uint32_t recon_coeff_stride, uint64_t distortion_result[2],
                                                      uint32_t area_width, uint32_t area_height) {
    dropin_dist32(0, coeff, coeff_stride, recon_coeff, recon_coeff_stride, distortion_result, area_width, area_height,
                  "svt_hip_full_distortion_kernel32_bits");
}
extern "C" void svt_hip_full_distortion_kernel_cbf_zero32_bits(int32_t* coeff, uint32_t coeff_stride, int32_t* recon_coeff,
                                                               uint32_t recon_coeff_stride, uint64_t distortion_result[2],
                                                               uint32_t area_width, uint32_t area_height) {
    dropin_dist32(1, coeff, coeff_stride, recon_coeff, recon_coeff_stride, distortion_result, area_width, area_height,
                  "svt_hip_full_distortion_kernel_cbf_zero32_bits");
}


// combined_averaging_sad (EB_SADAVGKERNELNxM_TYPE, EbComputeSAD.h:49-57; NxMSadAveragingKernel_funcPtrArray)
extern "C" uint32_t svt_hip_combined_averaging_sad(uint8_t* src, uint32_t src_stride, uint8_t* ref1, uint32_t ref1_stride,
                                                   uint8_t* ref2, uint32_t ref2_stride, uint32_t height, uint32_t width) {
    const char* fn = "svt_hip_combined_averaging_sad";
    const size_t bb = align256((size_t)width * height);
    DROPIN_TRY(t_ctx.ensure(3 * bb + 256), fn);
    uint8_t* d_a = (uint8_t*)t_ctx.dbuf;
    uint8_t* d_b = d_a + bb;
    uint8_t* d_c = d_a + 2 * bb;
    uint32_t* d_o = (uint32_t*)(d_a + 3 * bb);
    h2d_block(d_a, src, src_stride, width, height, fn);
    h2d_block(d_b, ref1, ref1_stride, width, height, fn);
    h2d_block(d_c, ref2, ref2_stride, width, height, fn);
    DROPIN_TRY(svt_hip_sad_avg_batch(d_a, width, 0, d_b, width, 0, d_c, width, 0, width, height, d_o, 1, t_ctx.stream), fn);
    uint32_t out = 0;
    HIP_DIE(hipMemcpyAsync(&out, d_o, 4, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
    return out;
}

// aom_sadMxN / aom_sadMxNx4d (aom_dsp_rtcd.h:1328-1500; C: C_DEFAULT/EbComputeSAD_C.c:140-233)
static void dropin_sad_x4d(int w, int h, const uint8_t* src, int src_stride, const uint8_t* const ref[4], int ref_stride,
                           uint32_t* sad_array, const char* fn) {
    const size_t bb = align256((size_t)w * h);
    DROPIN_TRY(t_ctx.ensure(5 * bb + 256), fn);
    uint8_t* d = (uint8_t*)t_ctx.dbuf;
    uint32_t* d_o = (uint32_t*)(d + 5 * bb);
    h2d_block(d, src, (uint32_t)src_stride, w, h, fn);
    for (int i = 0; i < 4; i++) h2d_block(d + (size_t)(1 + i) * bb, ref[i], (uint32_t)ref_stride, w, h, fn);
    // four dense reference blocks bb bytes apart against one source block (a_shift = 2 through the x4d entry point)
    const uint32_t offs[4] = {0, (uint32_t)bb, (uint32_t)(2 * bb), (uint32_t)(3 * bb)};
    uint32_t* d_offs = d_o + 4;
    HIP_DIE(hipMemcpyAsync(d_offs, offs, sizeof(offs), hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_sad_x4d_batch(d, w, 0, nullptr, d + bb, w, d_offs, w, h, d_o, 1, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(sad_array, d_o, 16, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
#define DEF_SAD(W, H)                                                                                                          \
    extern "C" unsigned int svt_hip_aom_sad##W##x##H(const uint8_t* src_ptr, int src_stride, const uint8_t* ref_ptr, int ref_stride) { \
        return svt_hip_nxm_sad_kernel(src_ptr, (uint32_t)src_stride, ref_ptr, (uint32_t)ref_stride, H, W);                     \
    }                                                                                                                          \
    extern "C" void svt_hip_aom_sad##W##x##H##x4d(const uint8_t* src_ptr, int src_stride, const uint8_t* const ref_ptr[],     \
                                                  int ref_stride, uint32_t* sad_array) {                                      \
        dropin_sad_x4d(W, H, src_ptr, src_stride, ref_ptr, ref_stride, sad_array, "svt_hip_aom_sad" #W "x" #H "x4d");          \
    }
SVT_HIP_SAD_SIZES(DEF_SAD)
#undef DEF_SAD

// residual_kernel16bit (EbPictureOperators.c:134-164)
extern "C" void svt_hip_residual_kernel16bit(uint16_t* input, uint32_t input_stride, uint16_t* pred, uint32_t pred_stride,
                                             int16_t* residual, uint32_t residual_stride, uint32_t area_width,
                                             uint32_t area_height) {
    const char* fn = "svt_hip_residual_kernel16bit";
    const size_t bb = align256((size_t)area_width * area_height * 2);
    DROPIN_TRY(t_ctx.ensure(3 * bb), fn);
    uint16_t* d_a = (uint16_t*)t_ctx.dbuf;
    uint16_t* d_b = (uint16_t*)(t_ctx.dbuf + bb);
    int16_t* d_r = (int16_t*)(t_ctx.dbuf + 2 * bb);
    const size_t rowb = (size_t)area_width * 2;
    HIP_DIE(hipMemcpy2DAsync(d_a, rowb, input, (size_t)input_stride * 2, rowb, area_height, hipMemcpyHostToDevice, t_ctx.stream), fn);
    HIP_DIE(hipMemcpy2DAsync(d_b, rowb, pred, (size_t)pred_stride * 2, rowb, area_height, hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_residual16_batch(d_a, area_width, 0, d_b, area_width, 0, d_r, area_width, 0, area_width, area_height, 1, t_ctx.stream), fn);
    HIP_DIE(hipMemcpy2DAsync(residual, (size_t)residual_stride * 2, d_r, rowb, rowb, area_height, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
