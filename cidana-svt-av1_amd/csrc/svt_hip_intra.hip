// svt_hip_intra.hip — entry points of the intra family of libsvt_hip_dsp.so (include/svt_hip_dsp.h): intra prediction, edge
// filters, chroma-from-luma helpers, txb_init_levels, the open-loop intra search and their drop-ins.
#include "host_common.h"
#include "kernel_cfl.h"
#include "kernel_intra.h"
#include "kernel_bip.h"
#include "kernel_ois.h"

using namespace svtdev;
using namespace svthost;

// ---- K11 chroma-from-luma helpers + av1_txb_init_levels ----
static bool cfl_dim_ok(uint32_t v) { return v == 4 || v == 8 || v == 16 || v == 32; }

static int cfl_ac_launch(int in_mode, const void* d_luma, uint32_t luma_stride, size_t luma_block_pitch, const uint32_t* d_xy,
                         int16_t* d_q3, uint32_t q3_line, size_t q3_block_pitch, uint32_t w, uint32_t h, int subtract,
                         int round_offset, int num_pel_log2, size_t nblocks, void* stream, const char* what) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_q3 || (in_mode != 2 && !d_luma)) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (!cfl_dim_ok(w) || !cfl_dim_ok(h)) return set_err(SVT_HIP_ERR_INVALID, "chroma block %ux%u", w, h);
    if (q3_line < w || q3_block_pitch < (size_t)q3_line * (h - 1) + w) return set_err(SVT_HIP_ERR_INVALID, "q3 layout: line %u, block pitch %zu", q3_line, q3_block_pitch);
    if (num_pel_log2 < 0 || num_pel_log2 > 31) return set_err(SVT_HIP_ERR_INVALID, "num_pel_log2 %d", num_pel_log2);
    const uint32_t nchunks = (w / (w < 8 ? 4 : 8)) * h;
    const uint32_t lpb = nchunks < 64 ? nchunks : 64;
    const size_t lanes = nblocks * lpb;
    const size_t grid = (lanes + 255) / 256;
    if (grid > 0x7fffffffu || nblocks > 0x7fffffffu / 64) return set_err(SVT_HIP_ERR_INVALID, "too many blocks for one launch");
#define CFL_AC(IN)                                                                                                          \
    hipLaunchKernelGGL((cfl_ac_kernel<IN>), dim3((uint32_t)grid), dim3(256), 0, (hipStream_t)stream, d_luma, luma_stride,   \
                       luma_block_pitch, d_xy, d_q3, q3_line, q3_block_pitch, w, h, lpb, subtract, round_offset, num_pel_log2, \
                       (uint32_t)nblocks)
    if (in_mode == 0) CFL_AC(0); else if (in_mode == 1) CFL_AC(1); else CFL_AC(2);
#undef CFL_AC
    return launch_status(what);
}

extern "C" int svt_hip_cfl_luma_subsampling_420_batch(const void* d_luma, uint32_t luma_stride, size_t luma_block_pitch,
                                                      const uint32_t* d_xy, int is_16bit, int16_t* d_q3, uint32_t q3_line,
                                                      size_t q3_block_pitch, uint32_t width, uint32_t height,
                                                      int subtract_average, size_t nblocks, void* stream) {
    if ((width & 1) || (height & 1)) return set_err(SVT_HIP_ERR_INVALID, "luma block %ux%u", width, height);
    const uint32_t w = width >> 1, h = height >> 1;
    int lg = 0;
    while ((1u << lg) < w * h) lg++;
    return cfl_ac_launch(is_16bit ? 1 : 0, d_luma, luma_stride, luma_block_pitch, d_xy, d_q3, q3_line, q3_block_pitch, w, h,
                         subtract_average ? 1 : 0, (int)(w * h / 2), lg, nblocks, stream, "cfl_luma_subsampling_420");
}

extern "C" int svt_hip_subtract_average_batch(int16_t* d_q3, uint32_t q3_line, size_t q3_block_pitch, uint32_t width,
                                              uint32_t height, int32_t round_offset, int32_t num_pel_log2, size_t nblocks,
                                              void* stream) {
    return cfl_ac_launch(2, nullptr, 0, 0, nullptr, d_q3, q3_line, q3_block_pitch, width, height, 1, round_offset,
                         num_pel_log2, nblocks, stream, "subtract_average");
}

extern "C" int svt_hip_cfl_predict_batch(const int16_t* d_ac_q3, uint32_t q3_line, size_t q3_block_pitch, const void* d_pred,
                                         uint32_t pred_stride, void* d_dst, uint32_t dst_stride, const uint32_t* d_xy,
                                         const int32_t* d_alpha_q3, int bit_depth, uint32_t width, uint32_t height,
                                         int is_16bit, size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_ac_q3 || !d_pred || !d_dst || !d_alpha_q3) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (!cfl_dim_ok(width) || !cfl_dim_ok(height)) return set_err(SVT_HIP_ERR_INVALID, "chroma block %ux%u", width, height);
    if ((is_16bit && bit_depth != 8 && bit_depth != 10 && bit_depth != 12) || (!is_16bit && bit_depth != 8))
        return set_err(SVT_HIP_ERR_INVALID, "bit depth %d", bit_depth);
    if (q3_line < width || pred_stride < width || dst_stride < width) return set_err(SVT_HIP_ERR_INVALID, "stride smaller than the block");
    const uint32_t nchunks = (width / (width < 8 ? 4 : 8)) * height;
    uint32_t sh = 0;
    while ((1u << sh) < nchunks) sh++;
    const size_t grid = ((nblocks << sh) + 255) / 256;
    if (grid > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many blocks for one launch");
    const int hi = (1 << bit_depth) - 1;
    if (is_16bit)
        hipLaunchKernelGGL((cfl_predict_kernel<uint16_t>), dim3((uint32_t)grid), dim3(256), 0, (hipStream_t)stream, d_ac_q3, q3_line,
                           q3_block_pitch, (const uint16_t*)d_pred, pred_stride, (uint16_t*)d_dst, dst_stride, d_xy, d_alpha_q3, hi,
                           width, height, sh, (uint32_t)nblocks);
    else
        hipLaunchKernelGGL((cfl_predict_kernel<uint8_t>), dim3((uint32_t)grid), dim3(256), 0, (hipStream_t)stream, d_ac_q3, q3_line,
                           q3_block_pitch, (const uint8_t*)d_pred, pred_stride, (uint8_t*)d_dst, dst_stride, d_xy, d_alpha_q3, hi,
                           width, height, sh, (uint32_t)nblocks);
    return launch_status("cfl_predict");
}

extern "C" int svt_hip_txb_init_levels_batch(const int32_t* d_coeff, size_t coeff_block_pitch, uint8_t* d_levels_buf,
                                             size_t levels_block_pitch, uint32_t width, uint32_t height, size_t nblocks,
                                             void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_coeff || !d_levels_buf) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    auto ok = [](uint32_t v) { return v == 4 || v == 8 || v == 16 || v == 32 || v == 64; };
    if (!ok(width) || !ok(height)) return set_err(SVT_HIP_ERR_INVALID, "block %ux%u", width, height);
    const uint32_t bytes = (width + 4) * (height + 6) + 16;      // (W + TX_PAD_HOR) * (H + TX_PAD_VER) + TX_PAD_END
    if (levels_block_pitch < bytes || (levels_block_pitch & 3) || ((uintptr_t)d_levels_buf & 3))
        return set_err(SVT_HIP_ERR_INVALID, "levels buffer: %zu B per block (need >= %u, multiple of 4, 4-byte aligned)", levels_block_pitch, bytes);
    if (coeff_block_pitch < (size_t)width * height) return set_err(SVT_HIP_ERR_INVALID, "coeff_block_pitch %zu", coeff_block_pitch);
    const uint32_t ndw = bytes >> 2, dpr = (width + 4) >> 2;
    const bool wide = (levels_block_pitch & 15) == 0 && ((uintptr_t)d_levels_buf & 15) == 0;
    const uint32_t items = wide ? (ndw + 3) / 4 : ndw;
    uint32_t lpb = 1;
    while (lpb < items && lpb < 256) lpb <<= 1;
    const uint32_t slots = 256 / lpb;
    const size_t grid = (nblocks + slots - 1) / slots;
    if (grid > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many blocks for one launch");
    const uint32_t row_magic = (uint32_t)(0x100000000ull / dpr) + 1u;
    if (wide)
        hipLaunchKernelGGL(txb_init_levels_kernel<true>, dim3((uint32_t)grid), dim3(256), 0, (hipStream_t)stream, d_coeff, coeff_block_pitch,
                           d_levels_buf, levels_block_pitch, width, height, lpb, ndw, row_magic, (uint32_t)nblocks);
    else
        hipLaunchKernelGGL(txb_init_levels_kernel<false>, dim3((uint32_t)grid), dim3(256), 0, (hipStream_t)stream, d_coeff, coeff_block_pitch,
                           d_levels_buf, levels_block_pitch, width, height, lpb, ndw, row_magic, (uint32_t)nblocks);
    return launch_status("txb_init_levels");
}

// ---- the frame call with its chroma-from-luma step and level maps (header: svt_hip_encode_recon_frame_ex) ----
static int frame_levels_launch(const svt_hip_frame_group* groups, const svt_hip_frame_levels* levels, int ngroups, hipStream_t s) {
    LevelsFrameDesc fd;
    memset(&fd, 0, sizeof(fd));
    uint32_t total = 0;
    auto flush = [&]() -> int {
        if (!fd.ngroups) return SVT_HIP_OK;
        hipLaunchKernelGGL(levels_frame_kernel, dim3(total), dim3(256), 0, s, fd);
        fd.ngroups = 0; total = 0;
        return launch_status("levels_frame");
    };
    for (int g = 0; g < ngroups; g++) {
        const svt_hip_frame_group& G = groups[g];
        const svt_hip_frame_levels& L = levels[g];
        if (!G.nblocks || !L.d_levels_buf) continue;
        if (fd.ngroups == LEVELS_MAX_GROUPS) if (int rc = flush()) return rc;
        LevelsGroupDev& D = fd.g[fd.ngroups];
        // get_txb_wide / get_txb_high: the packed coefficient block (a 64-sample side keeps its 32 low-frequency columns / rows)
        const uint32_t w = (uint32_t)(kTxW[G.tx_size] > 32 ? 32 : kTxW[G.tx_size]), h = (uint32_t)(kTxH[G.tx_size] > 32 ? 32 : kTxH[G.tx_size]);
        const uint32_t bytes = (w + 4) * (h + 6) + 16, ndw = bytes >> 2, dpr = (w + 4) >> 2;
        const bool wide = (L.levels_block_pitch & 15) == 0 && ((uintptr_t)L.d_levels_buf & 15) == 0;
        const uint32_t items = wide ? (ndw + 3) / 4 : ndw;
        uint32_t lpb = 1;
        while (lpb < items && lpb < 256) lpb <<= 1;
        D.coeff = G.d_qcoeff; D.levels = L.d_levels_buf; D.levels_pitch = (uint32_t)L.levels_block_pitch; D.w = w; D.h = h; D.lpb = lpb; D.ndw = ndw;
        D.row_magic = (uint32_t)(0x100000000ull / dpr) + 1u; D.nblocks = G.nblocks;
        const uint32_t slots = 256 / lpb;
        total += (G.nblocks + slots - 1) / slots;
        D.wg_end = total | (wide ? 0x80000000u : 0u);
        fd.ngroups++;
    }
    return flush();
}

extern "C" int svt_hip_encode_recon_frame_ex(const svt_hip_frame_group* groups, int ngroups, int first_chroma_group,
                                             const svt_hip_frame_cfl_group* cfl, int ncfl, const svt_hip_frame_levels* levels,
                                             int is_16bit, int bd, const int16_t* zbin, const int16_t* round, const int16_t* quant,
                                             const int16_t* quant_shift, const int16_t* dequant, void* stream) {
    if (int rc = require_init()) return rc;
    if (ngroups < 0 || ncfl < 0 || (ngroups && !groups) || (ncfl && !cfl)) return set_err(SVT_HIP_ERR_INVALID, "group lists");
    if (ngroups > 256) return set_err(SVT_HIP_ERR_INVALID, "more than 256 groups in one call");
    if (ncfl > CFL_MAX_GROUPS) return set_err(SVT_HIP_ERR_INVALID, "%d chroma-from-luma groups (at most %d: one per chroma transform size)", ncfl, CFL_MAX_GROUPS);
    if (ncfl && (first_chroma_group < 0 || first_chroma_group > ngroups)) return set_err(SVT_HIP_ERR_INVALID, "first_chroma_group %d of %d", first_chroma_group, ngroups);
    if ((is_16bit && bd != 8 && bd != 10 && bd != 12) || (!is_16bit && bd != 8)) return set_err(SVT_HIP_ERR_INVALID, "bit depth %d", bd);
    // everything is validated before anything is enqueued
    if (int rc = frame_groups_check(groups, ngroups)) return rc;
    CflFrameDesc cd;
    memset(&cd, 0, sizeof(cd));
    uint32_t cfl_total = 0;
    for (int g = 0; g < ncfl; g++) {
        const svt_hip_frame_cfl_group& C = cfl[g];
        if (C.nblocks == 0) continue;
        if (!C.d_luma_recon || !C.d_pred_cb || !C.d_pred_cr || !C.d_xy || !C.d_alpha_q3_cb || !C.d_alpha_q3_cr)
            return set_err(SVT_HIP_ERR_INVALID, "chroma-from-luma group %d: NULL member", g);
        if (!cfl_dim_ok(C.width) || !cfl_dim_ok(C.height)) return set_err(SVT_HIP_ERR_INVALID, "chroma-from-luma group %d: chroma block %ux%u", g, C.width, C.height);
        if (C.luma_stride < 2 * C.width || C.pred_stride_cb < C.width || C.pred_stride_cr < C.width)
            return set_err(SVT_HIP_ERR_INVALID, "chroma-from-luma group %d: stride smaller than the block", g);
        CflGroupDev& D = cd.g[cd.ngroups++];
        const uint32_t nchunks = (C.width / (C.width < 8 ? 4 : 8)) * C.height;
        D.lpb = nchunks < 64 ? nchunks : 64;
        D.luma = C.d_luma_recon; D.cb = C.d_pred_cb; D.cr = C.d_pred_cr; D.xy = C.d_xy; D.alpha_cb = C.d_alpha_q3_cb; D.alpha_cr = C.d_alpha_q3_cr;
        D.luma_stride = C.luma_stride; D.cb_stride = C.pred_stride_cb; D.cr_stride = C.pred_stride_cr; D.nblocks = C.nblocks; D.w = C.width; D.h = C.height;
        D.round_offset = (int32_t)(C.width * C.height / 2);
        D.num_pel_log2 = __builtin_ctz(C.width) + __builtin_ctz(C.height);
        const uint32_t slots = 256 / D.lpb;
        cfl_total += (C.nblocks + slots - 1) / slots;
        D.wg_end = cfl_total;
    }
    if (levels)
        for (int g = 0; g < ngroups; g++) {
            const svt_hip_frame_levels& L = levels[g];
            if (!groups[g].nblocks || !L.d_levels_buf) continue;
            const uint32_t w = (uint32_t)(kTxW[groups[g].tx_size] > 32 ? 32 : kTxW[groups[g].tx_size]), h = (uint32_t)(kTxH[groups[g].tx_size] > 32 ? 32 : kTxH[groups[g].tx_size]);
            const size_t bytes = (size_t)(w + 4) * (h + 6) + 16;
            if (L.levels_block_pitch < bytes || (L.levels_block_pitch & 3) || ((uintptr_t)L.d_levels_buf & 3) || L.levels_block_pitch > 0xffffffffu)
                return set_err(SVT_HIP_ERR_INVALID, "group %d: levels buffer %zu B per block (need >= %zu, multiple of 4, 4-byte aligned)", g, L.levels_block_pitch, bytes);
        }
    hipStream_t s = (hipStream_t)stream;
    // stream order carries the dependencies: luma reconstruction -> chroma-from-luma prediction -> chroma encode -> level maps
    const int n_first = cd.ngroups ? first_chroma_group : ngroups;
    if (int rc = svt_hip_encode_recon_frame(groups, n_first, is_16bit, bd, zbin, round, quant, quant_shift, dequant, stream)) return rc;
    if (cd.ngroups) {
        const int hi = (1 << bd) - 1;
        if (is_16bit) hipLaunchKernelGGL((cfl_frame_kernel<uint16_t>), dim3(cfl_total), dim3(256), 0, s, cd, hi);
        else hipLaunchKernelGGL((cfl_frame_kernel<uint8_t>), dim3(cfl_total), dim3(256), 0, s, cd, hi);
        if (int rc = launch_status("cfl_frame")) return rc;
        if (int rc = svt_hip_encode_recon_frame(groups + n_first, ngroups - n_first, is_16bit, bd, zbin, round, quant, quant_shift, dequant, stream)) return rc;
    }
    if (levels) return frame_levels_launch(groups, levels, ngroups, s);
    return SVT_HIP_OK;
}

static bool intra_size_ok(int bw, int bh) {
    auto ok1 = [](int v) { return v == 4 || v == 8 || v == 16 || v == 32 || v == 64; };
    if (!ok1(bw) || !ok1(bh)) return false;
    const int m = bw > bh ? bw : bh, mn = bw < bh ? bw : bh;
    return m <= 4 * mn;     // the 19 TX sizes
}

static int intra_pred_impl(void* d_dst, int32_t dst_stride, size_t dst_block_pitch,
                           const uint32_t* d_dst_offsets, const void* d_above, const void* d_left,
                           int32_t nb_pitch, int mode, int bw, int bh, int upsample_above,
                           int upsample_left, int dx, int dy, int is_16bit, int bd, size_t nblocks,
                           void* stream, const DirMulti* multi);

extern "C" int svt_hip_intra_pred_batch(void* d_dst, int32_t dst_stride, size_t dst_block_pitch,
                                        const uint32_t* d_dst_offsets, const void* d_above, const void* d_left,
                                        int32_t nb_pitch, int mode, int bw, int bh, int upsample_above,
                                        int upsample_left, int dx, int dy, int is_16bit, int bd, size_t nblocks,
                                        void* stream) {
    return intra_pred_impl(d_dst, dst_stride, dst_block_pitch, d_dst_offsets, d_above, d_left, nb_pitch, mode, bw, bh,
                           upsample_above, upsample_left, dx, dy, is_16bit, bd, nblocks, stream, nullptr);
}

// multi (directional modes only): several (dx, dy) of the same zone in one launch on edges staged once, see DirMulti
static int intra_pred_impl(void* d_dst, int32_t dst_stride, size_t dst_block_pitch,
                           const uint32_t* d_dst_offsets, const void* d_above, const void* d_left,
                           int32_t nb_pitch, int mode, int bw, int bh, int upsample_above,
                           int upsample_left, int dx, int dy, int is_16bit, int bd, size_t nblocks,
                           void* stream, const DirMulti* multi) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_dst || !d_above || !d_left) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (mode < 0 || mode >= SVT_INTRA_MODES) return set_err(SVT_HIP_ERR_INVALID, "intra mode %d", mode);
    if (!intra_size_ok(bw, bh)) return set_err(SVT_HIP_ERR_INVALID, "block %dx%d is not an AV1 transform size", bw, bh);
    if ((is_16bit && bd != 10 && bd != 12 && bd != 8) || (!is_16bit && bd != 8)) return set_err(SVT_HIP_ERR_INVALID, "bit depth %d", bd);
    if ((upsample_above | upsample_left) & ~1) return set_err(SVT_HIP_ERR_INVALID, "upsample flags");
    if (mode >= SVT_INTRA_Z1) {
        if (dx <= 0 || dy <= 0) return set_err(SVT_HIP_ERR_INVALID, "dx/dy must be positive");
        const int need = NB_ORIGIN + (((bw + bh) << 1) + 2);
        if (nb_pitch < need) return set_err(SVT_HIP_ERR_INVALID, "nb_pitch %d < %d", nb_pitch, need);
    } else if (nb_pitch < NB_ORIGIN + (bw > bh ? bw : bh)) {
        return set_err(SVT_HIP_ERR_INVALID, "nb_pitch %d too small", nb_pitch);
    }
    const int es = is_16bit ? 2 : 1;
    const int pxl = 16 / es, ppl = bw < pxl ? bw : pxl;
    const size_t per_block = (size_t)(bw / ppl) * bh;            // lanes per block: power of two, 4..512
    const size_t items = per_block * nblocks;
    size_t grid = (items + 255) / 256;
    if (grid > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many blocks for one launch");
    // grid * 256 must be a multiple of per_block so that a lane keeps its (row, column) across iterations
    if (per_block > 256) grid = (grid + 1) & ~(size_t)1;
    hipStream_t s = (hipStream_t)stream;
    if (mode >= SVT_INTRA_Z1) {
        // last edge sample the reference may read: index max_base = (bw + bh - 1) << upsample; the LDS copy
        // continues with copies of it for one lane-row (+2) so the pixel loop needs no bounds test
        const int lim_a = NB_ORIGIN + ((bw + bh - 1) << upsample_above), lim_l = NB_ORIGIN + ((bw + bh - 1) << upsample_left);
        const int up = upsample_above > upsample_left ? upsample_above : upsample_left;
        const int n_pad = (lim_a > lim_l ? lim_a : lim_l) + ((16 / es) << up) + 3;
        const size_t slots = per_block >= 256 ? 1 : 256 / per_block;
        const size_t shmem = slots * (size_t)dir_slot_stride((n_pad + 10) & ~7, (uint32_t)(per_block >= 256 ? 256 : per_block)) * 4 + 2 * 64 * 4;   // pair dwords + zone 2's column table (see the kernel)
        DirMulti dm;
        if (multi) dm = *multi; else dm.n = 0;
        dm.z2_tab = 0;
        if (dm.n > 0 && mode == SVT_INTRA_Z2 && bw == ppl && (upsample_above | upsample_left) == 0) {     // see DirMulti
            dm.z2_tab = 1;
            for (int a = 0; a < dm.n; a++)
                for (int k = 0; k < 16; k++) {
                    const int ys = -(int)dm.dy[a] * (k + 1);
                    const uint32_t sh = ((uint32_t)ys & 63u) >> 1;
                    dm.z2_w2[a][k] = (32u - sh) | (sh << 16);
                    dm.z2_ol[a][k] = 4 * (ys >> 6);
                }
        }
        // angles per workgroup: all of them when the batch alone gives every CU a few workgroups, else split over grid.y (each part re-stages the edges)
        uint32_t gy = 1;
        dm.chunk = dm.n > 0 ? dm.n : 1;
        if (dm.n > 1 && !g_tune_dir_no_split) {
            const size_t want = (size_t)g_tune_dir_split_target * (size_t)g_num_cu;
            size_t parts = grid >= want ? 1 : (want + grid - 1) / grid;
            if (parts > (size_t)dm.n) parts = (size_t)dm.n;
            dm.chunk = (int)((dm.n + parts - 1) / parts);
            gy = (uint32_t)((dm.n + dm.chunk - 1) / dm.chunk);
        }
#define IDL(T, M, P, TB)                                                                                                     \
    hipLaunchKernelGGL((intra_dir_kernel<T, M, P, TB>), dim3((uint32_t)grid, gy), dim3(256), shmem, s, (T*)d_dst, dst_stride, \
                       dst_block_pitch, d_dst_offsets, (const T*)d_above, (const T*)d_left, nb_pitch, bw, bh,               \
                       upsample_above, upsample_left, dx, dy, lim_a, lim_l, n_pad, bd, (uint32_t)nblocks, dm)
#define IDM(T, P)                                                                                                             \
    switch (mode) {                                                                                                           \
    case SVT_INTRA_Z1: IDL(T, IM_Z1, P, false); break; default: IDL(T, IM_Z3, P, false); break;                                \
    case SVT_INTRA_Z2: if (dm.z2_tab) IDL(T, IM_Z2, P, true); else IDL(T, IM_Z2, P, false); break;                              \
    }
        // samples per lane (ppl) is a template parameter: 4 / 8 / 16 for bytes, 4 / 8 for 16-bit samples
        if (is_16bit) { if (ppl == 8) { IDM(uint16_t, 8) } else { IDM(uint16_t, 4) } }
        else if (ppl == 16) { IDM(uint8_t, 16) } else if (ppl == 8) { IDM(uint8_t, 8) } else { IDM(uint8_t, 4) }
#undef IDM
#undef IDL
        return launch_status("intra_dir");
    }
    const int cnt = mode == SVT_INTRA_DC ? bw + bh : (mode == SVT_INTRA_DC_TOP ? bw : bh);
    const uint32_t dc_magic = (uint32_t)(0x100000000ull / (uint64_t)cnt) + 1u;
    // blocks per lane in the wide kernels: 4 for the modes with per-pixel arithmetic (SMOOTH*, PAETH: + 9 % on 2^21 32x32 blocks, A/B),
    // 2 for the copy-like ones (DC*, V, H: 2 - 4 % SLOWER at 4)
    const bool iu4 = mode == SVT_INTRA_SMOOTH || mode == SVT_INTRA_SMOOTH_V || mode == SVT_INTRA_SMOOTH_H || mode == SVT_INTRA_PAETH;
    const size_t grid_w = (grid + (iu4 ? 4 : 2) - 1) / (iu4 ? 4 : 2);
#define IPL(T, M)                                                                                                     \
    if (bw >= 16 / (int)sizeof(T))                                                                                    \
        hipLaunchKernelGGL((intra_pred_kernel<T, M, true, (M == IM_SMOOTH || M == IM_SMOOTH_V || M == IM_SMOOTH_H || M == IM_PAETH) ? 4 : 2>), dim3((uint32_t)(per_block > 256 ? (grid_w + 1) & ~(size_t)1 : grid_w)), dim3(256), 0, s, (T*)d_dst, dst_stride, \
                           dst_block_pitch, d_dst_offsets, (const T*)d_above, (const T*)d_left, nb_pitch, bw, bh, bd,    \
                           dc_magic, (uint32_t)nblocks);                                                              \
    else                                                                                                              \
        hipLaunchKernelGGL((intra_pred_kernel<T, M, false, 1>), dim3((uint32_t)grid), dim3(256), 0, s, (T*)d_dst, dst_stride, \
                           dst_block_pitch, d_dst_offsets, (const T*)d_above, (const T*)d_left, nb_pitch, bw, bh, bd,    \
                           dc_magic, (uint32_t)nblocks)
#define IPM(T)                                                                                                        \
    switch (mode) {                                                                                                   \
    case SVT_INTRA_DC: IPL(T, IM_DC); break; case SVT_INTRA_V: IPL(T, IM_V); break; case SVT_INTRA_H: IPL(T, IM_H); break; \
    case SVT_INTRA_SMOOTH: IPL(T, IM_SMOOTH); break; case SVT_INTRA_SMOOTH_V: IPL(T, IM_SMOOTH_V); break;             \
    case SVT_INTRA_SMOOTH_H: IPL(T, IM_SMOOTH_H); break; case SVT_INTRA_PAETH: IPL(T, IM_PAETH); break;               \
    case SVT_INTRA_DC_TOP: IPL(T, IM_DC_TOP); break; case SVT_INTRA_DC_LEFT: IPL(T, IM_DC_LEFT); break;               \
    default: IPL(T, IM_DC_128); break;                                                                                \
    }
    if (is_16bit) { IPM(uint16_t) } else { IPM(uint8_t) }
#undef IPM
#undef IPL
    return launch_status("intra_pred");
}

extern "C" int svt_hip_filter_intra_edge_batch(void* d_edges, int32_t nb_pitch, int sz, int strength, int is_16bit,
                                               size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0 || strength == 0) return SVT_HIP_OK;
    if (!d_edges) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (sz < 1 || sz > 129 || strength < 0 || strength > 3 || nb_pitch < NB_ORIGIN + sz)
        return set_err(SVT_HIP_ERR_INVALID, "edge sz %d strength %d pitch %d", sz, strength, nb_pitch);
    hipStream_t s = (hipStream_t)stream;
    if (is_16bit)
        hipLaunchKernelGGL((filter_edge_kernel<uint16_t>), dim3((uint32_t)nblocks), dim3(256), 0, s, (uint16_t*)d_edges, nb_pitch, NB_ORIGIN, sz, strength, (uint32_t)nblocks);
    else
        hipLaunchKernelGGL((filter_edge_kernel<uint8_t>), dim3((uint32_t)nblocks), dim3(256), 0, s, (uint8_t*)d_edges, nb_pitch, NB_ORIGIN, sz, strength, (uint32_t)nblocks);
    return launch_status("filter_intra_edge");
}
extern "C" int svt_hip_upsample_intra_edge_batch(void* d_edges, int32_t nb_pitch, int sz, int is_16bit, int bd,
                                                 size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_edges) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (sz < 1 || sz > 16 || nb_pitch < NB_ORIGIN + 2 * sz) return set_err(SVT_HIP_ERR_INVALID, "upsample sz %d pitch %d", sz, nb_pitch);
    hipStream_t s = (hipStream_t)stream;
    if (is_16bit)
        hipLaunchKernelGGL((upsample_edge_kernel<uint16_t>), dim3((uint32_t)nblocks), dim3(64), 0, s, (uint16_t*)d_edges, nb_pitch, NB_ORIGIN, sz, bd, (uint32_t)nblocks);
    else
        hipLaunchKernelGGL((upsample_edge_kernel<uint8_t>), dim3((uint32_t)nblocks), dim3(64), 0, s, (uint8_t*)d_edges, nb_pitch, NB_ORIGIN, sz, 8, (uint32_t)nblocks);
    return launch_status("upsample_intra_edge");
}

// ===========================================================================
// (A) drop-in entry points: host pointers, one block, synchronous
// ---- build_intra_predictors{,_high} for a batch (EbIntraPrediction.c:3667-4076), see kernel_bip.h ----
static_assert(sizeof(svt_hip_intra_blk) == sizeof(BipBlk), "svt_hip_intra_blk layout");
template <int W, int H>
static int bip_launch(void* d_dst, int32_t dst_stride, size_t dst_block_pitch, const uint32_t* d_dst_offsets, const void* d_top_neigh,
                      const void* d_left_neigh, int32_t neigh_pitch, const svt_hip_intra_blk* d_blocks, const uint32_t* d_order, int is_16bit,
                      int bd, size_t nblocks, uint32_t grid, hipStream_t s) {
    if (is_16bit)
        hipLaunchKernelGGL((bip_kernel<uint16_t, W, H>), dim3(grid), dim3(64 * BIP_WAVES), 0, s, (uint16_t*)d_dst, dst_stride, dst_block_pitch, d_dst_offsets,
                           (const uint16_t*)d_top_neigh, (const uint16_t*)d_left_neigh, neigh_pitch, (const BipBlk*)d_blocks, d_order, bd, (uint32_t)nblocks);
    else
        hipLaunchKernelGGL((bip_kernel<uint8_t, W, H>), dim3(grid), dim3(64 * BIP_WAVES), 0, s, (uint8_t*)d_dst, dst_stride, dst_block_pitch, d_dst_offsets,
                           (const uint8_t*)d_top_neigh, (const uint8_t*)d_left_neigh, neigh_pitch, (const BipBlk*)d_blocks, d_order, bd, (uint32_t)nblocks);
    return launch_status("build_intra_predictors");
}
static int bip_impl(void* d_dst, int32_t dst_stride, size_t dst_block_pitch, const uint32_t* d_dst_offsets, const void* d_top_neigh,
                    const void* d_left_neigh, int32_t neigh_pitch, const svt_hip_intra_blk* d_blocks, const uint32_t* d_order, int tx_size,
                    int is_16bit, int bd, size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_dst || !d_top_neigh || !d_left_neigh || !d_blocks) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (tx_size < 0 || tx_size >= SVT_TX_SIZES_ALL) return set_err(SVT_HIP_ERR_INVALID, "tx_size %d", tx_size);
    if ((is_16bit && bd != 10 && bd != 12 && bd != 8) || (!is_16bit && bd != 8)) return set_err(SVT_HIP_ERR_INVALID, "bit depth %d", bd);
    const int w = kTxW[tx_size], h = kTxH[tx_size];
    if (neigh_pitch < 1 + 2 * (w > h ? w : h)) return set_err(SVT_HIP_ERR_INVALID, "neigh_pitch %d < %d", neigh_pitch, 1 + 2 * (w > h ? w : h));
    if (dst_stride < w) return set_err(SVT_HIP_ERR_INVALID, "dst_stride %d < width %d", dst_stride, w);
    if (!d_dst_offsets && dst_block_pitch == 0) return set_err(SVT_HIP_ERR_INVALID, "dst_block_pitch 0 without offsets");
    const size_t per_wg = (size_t)BIP_WAVES * (64 / bip_lanes_per_block(w, h));       // blocks per workgroup
    const size_t grid = (nblocks + per_wg - 1) / per_wg;
    if (grid > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many blocks for one launch");
    hipStream_t s = (hipStream_t)stream;
#define BIP_LAUNCH(W, H) bip_launch<W, H>(d_dst, dst_stride, dst_block_pitch, d_dst_offsets, d_top_neigh, d_left_neigh, neigh_pitch, d_blocks, d_order, is_16bit, bd, \
                                          nblocks, (uint32_t)grid, s)
    TX_SWITCH(tx_size, BIP_LAUNCH)
#undef BIP_LAUNCH
}
extern "C" int svt_hip_build_intra_predictors_batch(void* d_dst, int32_t dst_stride, size_t dst_block_pitch,
                                                    const uint32_t* d_dst_offsets, const void* d_top_neigh,
                                                    const void* d_left_neigh, int32_t neigh_pitch,
                                                    const svt_hip_intra_blk* d_blocks, int tx_size, int is_16bit, int bd,
                                                    size_t nblocks, void* stream) {
    return bip_impl(d_dst, dst_stride, dst_block_pitch, d_dst_offsets, d_top_neigh, d_left_neigh, neigh_pitch, d_blocks, nullptr, tx_size, is_16bit, bd,
                    nblocks, stream);
}
extern "C" int svt_hip_build_intra_predictors_ordered_batch(void* d_dst, int32_t dst_stride, size_t dst_block_pitch,
                                                            const uint32_t* d_dst_offsets, const void* d_top_neigh,
                                                            const void* d_left_neigh, int32_t neigh_pitch,
                                                            const svt_hip_intra_blk* d_blocks, const uint32_t* d_order, int tx_size,
                                                            int is_16bit, int bd, size_t nblocks, void* stream) {
    if (nblocks && !d_order) { if (int rc = require_init()) return rc; return set_err(SVT_HIP_ERR_INVALID, "NULL order"); }
    return bip_impl(d_dst, dst_stride, dst_block_pitch, d_dst_offsets, d_top_neigh, d_left_neigh, neigh_pitch, d_blocks, d_order, tx_size, is_16bit, bd,
                    nblocks, stream);
}
extern "C" int svt_hip_intra_order_blocks_batch(const svt_hip_intra_blk* d_blocks, int tx_size, size_t nblocks, uint32_t* d_order,
                                                uint32_t* d_work, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    static_assert(BIP_ORDER_TILE == SVT_HIP_INTRA_ORDER_TILE, "header constant");
    (void)d_work;                                             // (the first version's global counters; unused)
    if (!d_blocks || !d_order) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (tx_size < 0 || tx_size >= SVT_TX_SIZES_ALL) return set_err(SVT_HIP_ERR_INVALID, "tx_size %d", tx_size);
    if (nblocks > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many blocks");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((uint32_t)((nblocks + BIP_ORDER_TILE - 1) / BIP_ORDER_TILE));
    hipLaunchKernelGGL(bip_order_tile_kernel, grid, dim3(BIP_ORDER_THREADS), 0, s, (const BipBlk*)d_blocks, kTxW[tx_size], kTxH[tx_size], d_order, (uint32_t)nblocks);
    return launch_status("intra_order_blocks");
}

// ---- open-loop intra search ----
static size_t ois_nb_pitch(uint32_t bsize) { return (size_t)NB_ORIGIN + 4 * bsize + 16; }     // multiple of 16
static size_t ois_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t svt_hip_ois_work_bytes(uint32_t bsize, int ncand, size_t nblocks) {
    if ((bsize != 8 && bsize != 16 && bsize != 32 && bsize != 64) || ncand <= 0 || ncand > 61) return 0;
    return 2 * ois_align(nblocks * ois_nb_pitch(bsize)) + ois_align(nblocks) + (size_t)ncand * ois_align(nblocks * (size_t)bsize * bsize);
}

// dr_intra_derivative (AV1 spec 7.11.2.4; reference EbIntraPrediction.c:299), non-zero entries
static int ois_dr_derivative(int angle) {
    static const uint16_t at[][2] = {{3, 1023}, {6, 547}, {9, 372}, {14, 273}, {17, 215}, {20, 178}, {23, 151}, {26, 132},
                                     {29, 116}, {32, 102}, {36, 90}, {39, 80}, {42, 71}, {45, 64}, {48, 57}, {51, 51},
                                     {54, 45}, {58, 40}, {61, 35}, {64, 31}, {67, 27}, {70, 23}, {73, 19}, {76, 15},
                                     {81, 11}, {84, 7}, {87, 3}};
    for (const auto& e : at)
        if (e[0] == angle) return e[1];
    return 0;
}

// the directional candidates of the three zones in ONE launch (ois_dir3_kernel): 8x8 / 16x16 SAD mode
static int ois_dir3_launch(const uint8_t* d_above, const uint8_t* d_left, size_t pitch, uint32_t bsize, size_t nblocks, const DirMulti zone[3],
                           hipStream_t s) {
    const size_t per_block = bsize;                              // one lane per row
    const size_t grid = (per_block * nblocks + 255) / 256;
    const int lim = NB_ORIGIN + (int)(2 * bsize - 1), n_pad = lim + 16 + 3;
    const size_t slots = 256 / per_block;
    const size_t shmem = slots * (size_t)dir_slot_stride((n_pad + 10) & ~7, (uint32_t)per_block) * 4 + 2 * 64 * 4;
    DirOis3 m;
    memset(&m, 0, sizeof(m));
    auto lite = [](DirMultiLite& d, const DirMulti& z) {
        d.n = z.n; memcpy(d.dx, z.dx, sizeof(d.dx)); memcpy(d.dy, z.dy, sizeof(d.dy)); memcpy(d.slot, z.slot, sizeof(d.slot));
        d.batch_pitch = 0; d.sad_pic = z.sad_pic; d.sad_stride = z.sad_stride; d.sad_xy = z.sad_xy; d.sad_dist = z.sad_dist; d.sad_ncand = z.sad_ncand; d.z2_tab = 0;
    };
    lite(m.z1, zone[0]); lite(m.z3, zone[2]);
    m.z2 = zone[1];
    m.z2.z2_tab = 1;
    for (int a = 0; a < m.z2.n; a++)
        for (int k = 0; k < 16; k++) {
            const int ys = -(int)m.z2.dy[a] * (k + 1);
            const uint32_t sh = ((uint32_t)ys & 63u) >> 1;
            m.z2.z2_w2[a][k] = (32u - sh) | (sh << 16);
            m.z2.z2_ol[a][k] = 4 * (ys >> 6);
        }
    // angles per workgroup, per zone: as separate launches each zone wants dir_split_target (4) workgroups per CU before it stops
    // spreading its angles over grid.y; with the three zones in one launch the sum counts - swept 1 .. 8 on the 1080p search
    // (gpurun_out r03_f): 1 is best for 8x8 (0.0526 against 0.0567 ms at 4) and 16x16 (0.0443 against 0.0471)
    const int nz[3] = {m.z1.n, m.z2.n, m.z3.n};
    int chunk[3];
    uint32_t gy[3];
    const size_t want = (size_t)g_num_cu;
    for (int z = 0; z < 3; z++) {
        size_t parts = (g_tune_dir_no_split || grid >= want) ? 1 : (want + grid - 1) / grid;
        if (parts > (size_t)(nz[z] ? nz[z] : 1)) parts = (size_t)(nz[z] ? nz[z] : 1);
        chunk[z] = nz[z] ? (int)((nz[z] + parts - 1) / parts) : 1;
        gy[z] = nz[z] ? (uint32_t)((nz[z] + chunk[z] - 1) / chunk[z]) : 0;
    }
    m.z1.chunk = chunk[0]; m.z2.chunk = chunk[1]; m.z3.chunk = chunk[2];
    m.y_end[0] = gy[0]; m.y_end[1] = gy[0] + gy[1];
    const uint32_t gyt = gy[0] + gy[1] + gy[2];
    if (bsize == 8)
        hipLaunchKernelGGL(ois_dir3_kernel<8>, dim3((uint32_t)grid, gyt), dim3(256), shmem, s, d_above, d_left, (int32_t)pitch, (int)bsize, lim, n_pad, (uint32_t)nblocks, m);
    else
        hipLaunchKernelGGL(ois_dir3_kernel<16>, dim3((uint32_t)grid, gyt), dim3(256), shmem, s, d_above, d_left, (int32_t)pitch, (int)bsize, lim, n_pad, (uint32_t)nblocks, m);
    return launch_status("ois_dir3");
}

// `defer` (svt_hip_ois_search_frame): the non-directional launch of the group is not enqueued but described in *defer (nblocks != 0),
// so that the caller can run every group's in ONE launch (ois_nd_multi_kernel) after the chains it depends on
static int ois_search_impl(const uint8_t* d_pic, uint32_t stride, uint32_t width, uint32_t height,
                           const uint32_t* d_xy, uint32_t bsize, const uint8_t* modes, const int8_t* angle_deltas,
                           int ncand, uint32_t* d_distortion, int8_t* d_best_index, void* d_work,
                           size_t work_bytes, size_t nblocks, void* stream, OisNdGroup* defer, int phase = 0, OisGatherGroup* gather = nullptr) {
    // phase 0: everything; 1: enqueue nothing, describe the group's neighbour gather in *gather (nblocks != 0) if it takes the fused
    // path; 2: the gather has been done by the caller (fused path), enqueue the rest
    if (defer) defer->nblocks = 0;
    if (gather) gather->nblocks = 0;
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_pic || !d_xy || !d_distortion || !d_best_index || !d_work || !modes || !angle_deltas) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (bsize != 8 && bsize != 16 && bsize != 32 && bsize != 64) return set_err(SVT_HIP_ERR_INVALID, "block size %u", bsize);
    if (ncand <= 0 || ncand > 61) return set_err(SVT_HIP_ERR_INVALID, "%d candidates (1..61, MAX_OIS_CANDIDATES)", ncand);
    if (width == 0 || height == 0 || width > 0xffffu || height > 0xffffu || stride < width) return set_err(SVT_HIP_ERR_INVALID, "picture %ux%u stride %u", width, height, stride);
    if (work_bytes < svt_hip_ois_work_bytes(bsize, ncand, nblocks)) return set_err(SVT_HIP_ERR_INVALID, "work buffer: %zu B, need %zu", work_bytes, svt_hip_ois_work_bytes(bsize, ncand, nblocks));
    if (nblocks > 0x7fffffffu / 256) return set_err(SVT_HIP_ERR_INVALID, "too many blocks for one launch");
    static const int mode_angle[13] = {0, 90, 180, 45, 135, 113, 157, 203, 67, 0, 0, 0, 0};      // mode_to_angle_map, EbCodingUnit.h:129
    for (int c = 0; c < ncand; c++) {
        if (modes[c] > 12) return set_err(SVT_HIP_ERR_INVALID, "candidate %d: prediction mode %u", c, modes[c]);
        if (modes[c] >= 1 && modes[c] <= 8) {
            const int a = mode_angle[modes[c]] + 3 * angle_deltas[c];
            if (a <= 0 || a >= 270) return set_err(SVT_HIP_ERR_INVALID, "candidate %d: angle %d", c, a);
            if (a != 90 && a != 180) {
                const int d1 = a < 90 ? a : (a < 180 ? 180 - a : 270 - a), d2 = a < 180 && a > 90 ? a - 90 : d1;
                if (!ois_dr_derivative(d1) || !ois_dr_derivative(d2)) return set_err(SVT_HIP_ERR_INVALID, "candidate %d: angle %d has no derivative", c, a);
            }
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t pitch = ois_nb_pitch(bsize);
    char* w = (char*)d_work;
    // ---- fused form: the directional candidates (angles other than 90 / 180) sum their SADs inside intra_dir_kernel (blocks of
    // at most 64 lanes: 8x8, 16x16), everything else - DC, V, H, SMOOTH*, PAETH - is predicted and summed inside ois_nd_kernel
    // straight from the picture; no prediction ever goes to memory.  Lists without directional candidates (all 32x32 / 64x64
    // lists) are ONE launch.
    {
        OisKinds kinds;
        memset(&kinds, 0, sizeof(kinds));
        bool any_dir = false, nd_fits = true;             // nd_fits: the kernel's own list (7 distinct kinds; a list may repeat them)
        for (int c = 0; c < ncand; c++) {
            const int m = modes[c];
            int k;
            if (m == 0) k = OIS_K_DC;
            else if (m <= 8) {
                const int a = mode_angle[m] + 3 * angle_deltas[c];
                k = a == 90 ? OIS_K_V : (a == 180 ? OIS_K_H : OIS_K_FOLDED);
                any_dir = any_dir || k == OIS_K_FOLDED;
            } else k = m == 9 ? OIS_K_SMOOTH : (m == 10 ? OIS_K_SMOOTH_V : (m == 11 ? OIS_K_SMOOTH_H : OIS_K_PAETH));
            kinds.k[c] = (uint8_t)k;
            if (k != OIS_K_FOLDED) {
                if (kinds.n_nd >= sizeof(kinds.nd_c)) nd_fits = false;
                else { kinds.nd_c[kinds.n_nd] = (uint8_t)c; kinds.nd_kind[kinds.n_nd] = (uint8_t)k; kinds.n_nd++; }
            }
        }
        kinds.k[OIS_MAX_CAND + 2] = any_dir ? 1 : 0;              // ois_nd_kernel: rows of dist hold folded sums to pick up
        const bool can_fold = bsize <= 16 && !g_tune_ois_no_fold;
        if (!g_tune_ois_no_nd && nd_fits && (!any_dir || can_fold)) {
            if (any_dir) {
                uint8_t* d_above = (uint8_t*)w;
                uint8_t* d_left = d_above + ois_align(nblocks * pitch);
                uint8_t* d_dc = d_left + ois_align(nblocks * pitch);
                // (no clearing of the neighbour arrays: the directional kernels stage positions [-2, 2 * bsize) only, all written by the gather)
                const uint32_t slots = 256 / (2 * bsize);
                if (phase == 1) {
                    gather->xy = d_xy; gather->above = d_above; gather->left = d_left; gather->dc = d_dc; gather->bsize = bsize;
                    gather->nb_pitch = (uint32_t)pitch; gather->nblocks = (uint32_t)nblocks; gather->wg_end = (uint32_t)((nblocks + slots - 1) / slots);
                    return SVT_HIP_OK;
                }
                if (phase == 0) {
                    hipLaunchKernelGGL(ois_gather_kernel, dim3((uint32_t)((nblocks + slots - 1) / slots)), dim3(256), 0, st, d_pic, stride, width,
                                       height, d_xy, bsize, d_above, d_left, (uint32_t)pitch, d_dc, (uint32_t)nblocks);
                    if (int rc = launch_status("ois_gather")) return rc;
                }
                DirMulti zone[3];
                for (auto& z : zone) {
                    z.n = 0; z.batch_pitch = 0;
                    z.sad_pic = d_pic; z.sad_stride = stride; z.sad_xy = d_xy; z.sad_dist = d_distortion; z.sad_ncand = (uint32_t)ncand;
                }
                bool flushed = false;                             // a zone with more than 20 angles went out on its own
                auto flush = [&](int zi) -> int {
                    DirMulti& z = zone[zi];
                    if (!z.n) return SVT_HIP_OK;
                    flushed = true;
                    const int rc = intra_pred_impl(d_distortion /* unused in SAD mode */, (int32_t)bsize, (size_t)bsize * bsize, nullptr, d_above, d_left,
                                                   (int32_t)pitch, SVT_INTRA_Z1 + zi, (int)bsize, (int)bsize, 0, 0, 1, 1, 0, 8, nblocks, stream, &z);
                    z.n = 0;
                    return rc;
                };
                for (int c = 0; c < ncand; c++) {
                    if (kinds.k[c] != OIS_K_FOLDED) continue;
                    const int a = mode_angle[modes[c]] + 3 * angle_deltas[c];
                    const int zi = a < 90 ? 0 : (a < 180 ? 1 : 2);
                    DirMulti& z = zone[zi];
                    if (z.n == 20) if (int rc = flush(zi)) return rc;
                    z.dx[z.n] = (int16_t)(zi == 0 ? ois_dr_derivative(a) : (zi == 1 ? ois_dr_derivative(180 - a) : 1));
                    z.dy[z.n] = (int16_t)(zi == 0 ? 1 : (zi == 1 ? ois_dr_derivative(a - 90) : ois_dr_derivative(270 - a)));
                    z.slot[z.n] = (uint8_t)c;
                    z.n++;
                }
                if (!flushed && !g_tune_ois_no_dir3 && (zone[0].n > 0) + (zone[1].n > 0) + (zone[2].n > 0) >= 2) {
                    if (int rc = ois_dir3_launch(d_above, d_left, pitch, bsize, nblocks, zone, st)) return rc;
                } else
                    for (int zi = 0; zi < 3; zi++) if (int rc = flush(zi)) return rc;
            }
            if (phase == 1) return SVT_HIP_OK;
            const uint32_t cs = bsize < 16 ? 8 : 16, lpb = bsize * bsize / cs;
            const uint32_t nd_slots = 256 / lpb;
            const uint32_t nd_grid = (uint32_t)((nblocks + nd_slots - 1) / nd_slots);
            const size_t shmem = (((size_t)nd_slots + (lpb > 64 ? 4 : 0)) * (size_t)ncand + 4) * sizeof(uint32_t);
            if (defer) {
                defer->xy = d_xy; defer->dist = d_distortion; defer->best_index = d_best_index; defer->bsize = bsize; defer->ncand = (uint32_t)ncand;
                defer->nblocks = (uint32_t)nblocks; defer->wg_end = nd_grid /* the caller turns it into a running end */; defer->kinds = kinds;
                return SVT_HIP_OK;
            }
            if (cs == 8)
                hipLaunchKernelGGL(ois_nd_kernel<8>, dim3(nd_grid), dim3(256), shmem, st, d_pic, stride, width, height, d_xy, bsize, kinds, d_distortion,
                                   d_best_index, (uint32_t)ncand, (uint32_t)nblocks);
            else
                hipLaunchKernelGGL(ois_nd_kernel<16>, dim3(nd_grid), dim3(256), shmem, st, d_pic, stride, width, height, d_xy, bsize, kinds, d_distortion,
                                   d_best_index, (uint32_t)ncand, (uint32_t)nblocks);
            return launch_status("ois_nd");
        }
    }
    if (phase == 1) return SVT_HIP_OK;                    // (the general path below gathers for itself, in phase 2)
    uint8_t* d_above = (uint8_t*)w;
    uint8_t* d_left = d_above + ois_align(nblocks * pitch);
    uint8_t* d_dc = d_left + ois_align(nblocks * pitch);
    uint8_t* d_pred = d_dc + ois_align(nblocks);
    if (hipMemsetAsync(d_above, 0, 2 * ois_align(nblocks * pitch), st) != hipSuccess) return set_err(SVT_HIP_ERR_RUNTIME, "hipMemsetAsync");
    {
        const uint32_t slots = 256 / (2 * bsize);
        hipLaunchKernelGGL(ois_gather_kernel, dim3((uint32_t)((nblocks + slots - 1) / slots)), dim3(256), 0, st, d_pic, stride, width,
                           height, d_xy, bsize, d_above, d_left, (uint32_t)pitch, d_dc, (uint32_t)nblocks);
        if (int rc = launch_status("ois_gather")) return rc;
    }
    // every candidate's prediction into its own dense batch, then ONE SAD launch over (block, candidate).  The
    // directional candidates of one zone (up to 19) share one launch (DirMulti): 9 prediction launches for the
    // reference's 45-candidate list instead of 44.
    const size_t cand_pitch = ois_align(nblocks * (size_t)bsize * bsize);
    unsigned long long const_mask = 0;
    DirMulti zone[3];
    // 8x8 / 16x16 (a block's lanes share a wave): the directional kernels compare each angle's prediction with the source
    // block themselves (DirMulti SAD mode) - no prediction scratch round trip for 38 of the 45 candidates
    const bool fold = bsize <= 16 && !g_tune_ois_no_fold;
    unsigned long long fold_mask = 0;
    for (auto& z : zone) {
        z.n = 0; z.batch_pitch = cand_pitch;
        z.sad_pic = d_pic; z.sad_stride = stride; z.sad_xy = d_xy; z.sad_dist = fold ? d_distortion : nullptr; z.sad_ncand = (uint32_t)ncand;
    }
    for (int c = 0; c < ncand; c++) {
        const int m = modes[c];
        if (m == 0) { const_mask |= 1ull << c; continue; }     // DC_PRED under the availability rule: constant prediction
        int mode = -1;
        if (m >= 1 && m <= 8) {                                               // dr_predictor, EbIntraPrediction.c:3352-3383
            const int a = mode_angle[m] + 3 * angle_deltas[c];
            if (a == 90) mode = SVT_INTRA_V;
            else if (a == 180) mode = SVT_INTRA_H;
            else {
                const int zi = a < 90 ? 0 : (a < 180 ? 1 : 2);
                DirMulti& z = zone[zi];
                if (z.n == 20) {                                              // flush a full group (longer candidate lists)
                    if (int rc = intra_pred_impl(d_pred, (int32_t)bsize, (size_t)bsize * bsize, nullptr, d_above, d_left, (int32_t)pitch,
                                                 SVT_INTRA_Z1 + zi, (int)bsize, (int)bsize, 0, 0, 1, 1, 0, 8, nblocks, stream, &z))
                        return rc;
                    z.n = 0;
                }
                z.dx[z.n] = (int16_t)(zi == 0 ? ois_dr_derivative(a) : (zi == 1 ? ois_dr_derivative(180 - a) : 1));
                z.dy[z.n] = (int16_t)(zi == 0 ? 1 : (zi == 1 ? ois_dr_derivative(a - 90) : ois_dr_derivative(270 - a)));
                z.slot[z.n] = (uint8_t)c;
                z.n++;
                if (fold) fold_mask |= 1ull << c;
                continue;
            }
        } else {
            mode = m == 9 ? SVT_INTRA_SMOOTH : m == 10 ? SVT_INTRA_SMOOTH_V : m == 11 ? SVT_INTRA_SMOOTH_H : SVT_INTRA_PAETH;
        }
        if (int rc = svt_hip_intra_pred_batch(d_pred + (size_t)c * cand_pitch, (int32_t)bsize, (size_t)bsize * bsize, nullptr, d_above,
                                              d_left, (int32_t)pitch, mode, (int)bsize, (int)bsize, 0, 0, 1, 1, 0, 8, nblocks, stream))
            return rc;
    }
    for (int zi = 0; zi < 3; zi++)
        if (zone[zi].n)
            if (int rc = intra_pred_impl(d_pred, (int32_t)bsize, (size_t)bsize * bsize, nullptr, d_above, d_left, (int32_t)pitch,
                                         SVT_INTRA_Z1 + zi, (int)bsize, (int)bsize, 0, 0, 1, 1, 0, 8, nblocks, stream, &zone[zi]))
                return rc;
    {
        const uint32_t lpb = bsize * bsize / (bsize < 16 ? 8 : 16);
        const uint32_t sad_slots = 256 / lpb;
        const uint32_t sad_grid = (uint32_t)((nblocks + sad_slots - 1) / sad_slots);
        const size_t shmem = ((size_t)sad_slots + (lpb > 64 ? 4 : 0)) * (size_t)ncand * sizeof(uint32_t);
        hipLaunchKernelGGL(ois_sad_kernel, dim3(sad_grid), dim3(256), shmem, st, d_pic, stride, d_xy, bsize, d_pred, cand_pitch,
                           d_dc, const_mask, fold_mask, d_distortion, d_best_index, (uint32_t)ncand, (uint32_t)nblocks);
    }
    return launch_status("ois_sad");
}

extern "C" int svt_hip_ois_search_batch(const uint8_t* d_pic, uint32_t stride, uint32_t width, uint32_t height,
                                        const uint32_t* d_xy, uint32_t bsize, const uint8_t* modes, const int8_t* angle_deltas,
                                        int ncand, uint32_t* d_distortion, int8_t* d_best_index, void* d_work,
                                        size_t work_bytes, size_t nblocks, void* stream) {
    return ois_search_impl(d_pic, stride, width, height, d_xy, bsize, modes, angle_deltas, ncand, d_distortion, d_best_index, d_work, work_bytes,
                           nblocks, stream, nullptr);
}

extern "C" int svt_hip_ois_search_frame(const uint8_t* d_pic, uint32_t stride, uint32_t width, uint32_t height,
                                        const svt_hip_ois_group* groups, int ngroups, void* stream) {
    if (int rc = require_init()) return rc;
    if (ngroups == 0) return SVT_HIP_OK;
    if (!groups || ngroups < 0 || ngroups > 64) return set_err(SVT_HIP_ERR_INVALID, "group list");
    // Every group's chain (neighbour gather, the three directional zones in one launch) goes out on the caller's stream, one after
    // the other - each fills the GPU - and the non-directional launch of EVERY group, which also takes each block's best index,
    // follows as ONE launch (ois_nd_multi_kernel).  Round 2 ran the single-launch groups (every 32x32 / 64x64 list) on a side stream;
    // they did not hide behind the chains (the call cost 0.134 ms per 1080p picture against 0.140 for the four groups in sequence:
    // the chains leave no free units), so the side stream, its fork / join events and two ~ 20 us latency-bound launches are gone.
    // (svt_hip_tune("ois_no_nd_multi", 1): one non-directional launch per group, on the caller's stream.)
    hipStream_t s = (hipStream_t)stream;
    OisNdMulti m;
    memset(&m, 0, sizeof(m));
    uint32_t total = 0;
    size_t shmem = 0;
    auto flush = [&]() -> int {
        if (!m.ngroups) return SVT_HIP_OK;
        // largest blocks first: their workgroups are the longest latency chains (64x64: one block per workgroup, two barriers)
        for (int i = 1; i < m.ngroups; i++) {
            const OisNdGroup v = m.g[i];
            int j = i - 1;
            while (j >= 0 && m.g[j].bsize < v.bsize) { m.g[j + 1] = m.g[j]; j--; }
            m.g[j + 1] = v;
        }
        total = 0;
        for (int i = 0; i < m.ngroups; i++) { total += m.g[i].wg_end; m.g[i].wg_end = total; }
        hipLaunchKernelGGL(ois_nd_multi_kernel, dim3(total), dim3(256), shmem, s, d_pic, stride, width, height, m);
        m.ngroups = 0; total = 0; shmem = 0;
        return launch_status("ois_nd_multi");
    };
    // the neighbour gathers of the groups that have directional candidates: one launch, first (arguments are validated here, before
    // anything is enqueued)
    const bool multi = !g_tune_ois_no_nd_multi;
    {
        OisGatherMulti gm;
        memset(&gm, 0, sizeof(gm));
        uint32_t gtotal = 0;
        auto gflush = [&]() -> int {
            if (!gm.ngroups) return SVT_HIP_OK;
            hipLaunchKernelGGL(ois_gather_multi_kernel, dim3(gtotal), dim3(256), 0, s, d_pic, stride, width, height, gm);
            gm.ngroups = 0; gtotal = 0;
            return launch_status("ois_gather_multi");
        };
        for (int g = 0; g < ngroups && multi; g++) {
            const svt_hip_ois_group& G = groups[g];
            if (G.nblocks == 0) continue;
            OisGatherGroup gd;
            if (int rc = ois_search_impl(d_pic, stride, width, height, G.d_xy, G.bsize, G.modes, G.angle_deltas, G.ncand, G.d_distortion, G.d_best_index,
                                         G.d_work, G.work_bytes, G.nblocks, stream, nullptr, 1, &gd)) {
                return gm.ngroups ? (gflush(), rc) : rc;
            }
            if (gd.nblocks == 0) continue;
            if (gm.ngroups == OIS_GATHER_MAX_GROUPS) if (int rc = gflush()) return rc;
            gtotal += gd.wg_end;
            gd.wg_end = gtotal;
            gm.g[gm.ngroups++] = gd;
        }
        if (int rc = gflush()) return rc;
    }
    for (int g = 0; g < ngroups; g++) {
        const svt_hip_ois_group& G = groups[g];
        if (G.nblocks == 0) continue;
        OisNdGroup d;
        if (int rc = ois_search_impl(d_pic, stride, width, height, G.d_xy, G.bsize, G.modes, G.angle_deltas, G.ncand, G.d_distortion, G.d_best_index,
                                     G.d_work, G.work_bytes, G.nblocks, stream, multi ? &d : nullptr, multi ? 2 : 0)) {
            (void)flush();                      // what was enqueued for the earlier groups stays complete
            return rc;
        }
        if (g_tune_ois_no_nd_multi || d.nblocks == 0) continue;
        if (m.ngroups == OIS_ND_MAX_GROUPS) if (int rc = flush()) return rc;
        const uint32_t cs = d.bsize < 16 ? 8 : 16, lpb = d.bsize * d.bsize / cs, nd_slots = 256 / lpb;
        const size_t sh = (((size_t)nd_slots + (lpb > 64 ? 4 : 0)) * (size_t)d.ncand + 4) * sizeof(uint32_t);
        shmem = sh > shmem ? sh : shmem;
        m.g[m.ngroups++] = d;                   // (wg_end = the group's own workgroup count until flush() orders the groups)
    }
    return flush();
}

// one intra block: stage [lo, hi) of above / left around the origin, predict, copy the block back
static void dropin_intra(int mode, int bw, int bh, void* dst, ptrdiff_t stride, const void* above, const void* left,
                         int a_lo, int a_hi, int l_lo, int l_hi, int ua, int ul, int dx, int dy, int is16, int bd,
                         const char* fn) {
    const size_t es = is16 ? 2 : 1;
    const int pitch = NB_ORIGIN + 2 * (bw + bh) + 16;
    const size_t nb_b = align256((size_t)pitch * es), px_b = (size_t)bw * bh * es;
    DROPIN_TRY(t_ctx.ensure(2 * nb_b + px_b), fn);
    char* d_a = t_ctx.dbuf;
    char* d_l = t_ctx.dbuf + nb_b;
    char* d_px = t_ctx.dbuf + 2 * nb_b;
    HIP_DIE(hipMemsetAsync(d_a, 0, 2 * nb_b, t_ctx.stream), fn);
    if (a_hi > a_lo)
        HIP_DIE(hipMemcpyAsync(d_a + (size_t)(NB_ORIGIN + a_lo) * es, (const char*)above + (ptrdiff_t)a_lo * (ptrdiff_t)es,
                               (size_t)(a_hi - a_lo) * es, hipMemcpyHostToDevice, t_ctx.stream), fn);
    if (l_hi > l_lo)
        HIP_DIE(hipMemcpyAsync(d_l + (size_t)(NB_ORIGIN + l_lo) * es, (const char*)left + (ptrdiff_t)l_lo * (ptrdiff_t)es,
                               (size_t)(l_hi - l_lo) * es, hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_intra_pred_batch(d_px, bw, (size_t)bw * bh, nullptr, d_a, d_l, pitch, mode, bw, bh, ua, ul, dx, dy,
                                        is16, bd, 1, t_ctx.stream), fn);
    HIP_DIE(hipMemcpy2DAsync(dst, (size_t)stride * es, d_px, (size_t)bw * es, (size_t)bw * es, bh, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
extern "C" void svt_hip_intra_predictor(int mode, int bw, int bh, uint8_t* dst, ptrdiff_t stride, const uint8_t* above,
                                        const uint8_t* left) {
    dropin_intra(mode, bw, bh, dst, stride, above, left, -1, bw, 0, bh, 0, 0, 1, 1, 0, 8, "svt_hip_intra_predictor");
}
extern "C" void svt_hip_highbd_intra_predictor(int mode, int bw, int bh, uint16_t* dst, ptrdiff_t stride,
                                               const uint16_t* above, const uint16_t* left, int32_t bd) {
    dropin_intra(mode, bw, bh, dst, stride, above, left, -1, bw, 0, bh, 0, 0, 1, 1, 1, bd, "svt_hip_highbd_intra_predictor");
}
#define DR_RANGES_Z1 0, (((bw + bh - 1) << upsample_above) + 2), 0, 0
#define DR_RANGES_Z3 0, 0, 0, (((bw + bh - 1) << upsample_left) + 2)
#define DR_RANGES_Z2 -(1 << upsample_above), (((bw - 1) << upsample_above) + 2), -(1 << upsample_left), (((bh - 1) << upsample_left) + 2)
extern "C" void svt_hip_av1_dr_prediction_z1(uint8_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh, const uint8_t* above,
                                             const uint8_t* left, int32_t upsample_above, int32_t dx, int32_t dy) {
    dropin_intra(SVT_INTRA_Z1, bw, bh, dst, stride, above, left, DR_RANGES_Z1, upsample_above, 0, dx, dy, 0, 8, "svt_hip_av1_dr_prediction_z1");
}
extern "C" void svt_hip_av1_dr_prediction_z2(uint8_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh, const uint8_t* above,
                                             const uint8_t* left, int32_t upsample_above, int32_t upsample_left,
                                             int32_t dx, int32_t dy) {
    dropin_intra(SVT_INTRA_Z2, bw, bh, dst, stride, above, left, DR_RANGES_Z2, upsample_above, upsample_left, dx, dy, 0, 8, "svt_hip_av1_dr_prediction_z2");
}
extern "C" void svt_hip_av1_dr_prediction_z3(uint8_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh, const uint8_t* above,
                                             const uint8_t* left, int32_t upsample_left, int32_t dx, int32_t dy) {
    dropin_intra(SVT_INTRA_Z3, bw, bh, dst, stride, above, left, DR_RANGES_Z3, 0, upsample_left, dx, dy, 0, 8, "svt_hip_av1_dr_prediction_z3");
}
extern "C" void svt_hip_av1_highbd_dr_prediction_z1(uint16_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh,
                                                    const uint16_t* above, const uint16_t* left, int32_t upsample_above,
                                                    int32_t dx, int32_t dy, int32_t bd) {
    dropin_intra(SVT_INTRA_Z1, bw, bh, dst, stride, above, left, DR_RANGES_Z1, upsample_above, 0, dx, dy, 1, bd, "svt_hip_av1_highbd_dr_prediction_z1");
}
extern "C" void svt_hip_av1_highbd_dr_prediction_z2(uint16_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh,
                                                    const uint16_t* above, const uint16_t* left, int32_t upsample_above,
                                                    int32_t upsample_left, int32_t dx, int32_t dy, int32_t bd) {
    dropin_intra(SVT_INTRA_Z2, bw, bh, dst, stride, above, left, DR_RANGES_Z2, upsample_above, upsample_left, dx, dy, 1, bd, "svt_hip_av1_highbd_dr_prediction_z2");
}
extern "C" void svt_hip_av1_highbd_dr_prediction_z3(uint16_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh,
                                                    const uint16_t* above, const uint16_t* left, int32_t upsample_left,
                                                    int32_t dx, int32_t dy, int32_t bd) {
    dropin_intra(SVT_INTRA_Z3, bw, bh, dst, stride, above, left, DR_RANGES_Z3, 0, upsample_left, dx, dy, 1, bd, "svt_hip_av1_highbd_dr_prediction_z3");
}


// ---- per-size RTCD slot entry points (intra_pred_fn / intra_high_pred_fn, EbIntraPrediction.h:36-41): what
// init_intra_predictors_internal stores in pred[][] / dc_pred[][][] (EbIntraPrediction.c:2842-3350) ----
#define DEF_PRED(mode, MODE, W, H)                                                                                          \
    extern "C" void svt_hip_aom_##mode##_predictor_##W##x##H(uint8_t* dst, ptrdiff_t stride, const uint8_t* above,        \
                                                             const uint8_t* left) {                                       \
        dropin_intra(MODE, W, H, dst, stride, above, left, -1, W, 0, H, 0, 0, 1, 1, 0, 8, "svt_hip_aom_" #mode "_predictor_" #W "x" #H); \
    }                                                                                                                     \
    extern "C" void svt_hip_aom_highbd_##mode##_predictor_##W##x##H(uint16_t* dst, ptrdiff_t stride, const uint16_t* above, \
                                                                    const uint16_t* left, int bd) {                       \
        dropin_intra(MODE, W, H, dst, stride, above, left, -1, W, 0, H, 0, 0, 1, 1, 1, bd, "svt_hip_aom_highbd_" #mode "_predictor_" #W "x" #H); \
    }
SVT_HIP_INTRA_MODES(SVT_HIP_BLOCK_SIZES_2, DEF_PRED)
#undef DEF_PRED
extern "C" void svt_hip_eb_smooth_v_predictor(uint8_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh, const uint8_t* above,
                                              const uint8_t* left) {
    dropin_intra(SVT_INTRA_SMOOTH_V, bw, bh, dst, stride, above, left, -1, bw, 0, bh, 0, 0, 1, 1, 0, 8, "svt_hip_eb_smooth_v_predictor");
}
extern "C" void svt_hip_eb_smooth_h_predictor(uint8_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh, const uint8_t* above,
                                              const uint8_t* left) {
    dropin_intra(SVT_INTRA_SMOOTH_H, bw, bh, dst, stride, above, left, -1, bw, 0, bh, 0, 0, 1, 1, 0, 8, "svt_hip_eb_smooth_h_predictor");
}

// ---- av1_filter_intra_edge{,_high}, av1_upsample_intra_edge{,_high} (aom_dsp_rtcd.h:152-156, 431-437) ----
static void dropin_edge(int upsample, void* p, int sz, int strength_or_bd, int is16, const char* fn) {
    const size_t es = is16 ? 2 : 1;
    const int pitch = NB_ORIGIN + 2 * sz + 16;
    DROPIN_TRY(t_ctx.ensure((size_t)pitch * es), fn);
    char* d = t_ctx.dbuf;
    // filter: p[0 .. sz-1] in and out; upsample: p[-1 .. sz-1] in, p[-2 .. 2*sz-2] out (EbIntraPrediction.c:3597-3660)
    const int in_lo = upsample ? -1 : 0, in_n = upsample ? sz + 1 : sz, out_lo = upsample ? -2 : 0, out_n = upsample ? 2 * sz + 1 : sz;
    HIP_DIE(hipMemcpyAsync(d + (size_t)(NB_ORIGIN + in_lo) * es, (const char*)p + (ptrdiff_t)in_lo * (ptrdiff_t)es, (size_t)in_n * es,
                           hipMemcpyHostToDevice, t_ctx.stream), fn);
    if (upsample) DROPIN_TRY(svt_hip_upsample_intra_edge_batch(d, pitch, sz, is16, strength_or_bd, 1, t_ctx.stream), fn);
    else DROPIN_TRY(svt_hip_filter_intra_edge_batch(d, pitch, sz, strength_or_bd, is16, 1, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync((char*)p + (ptrdiff_t)out_lo * (ptrdiff_t)es, d + (size_t)(NB_ORIGIN + out_lo) * es, (size_t)out_n * es,
                           hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
extern "C" void svt_hip_av1_filter_intra_edge(uint8_t* p, int32_t sz, int32_t strength) {
    if (strength) dropin_edge(0, p, sz, strength, 0, "svt_hip_av1_filter_intra_edge");
}
extern "C" void svt_hip_av1_filter_intra_edge_high(uint16_t* p, int32_t sz, int32_t strength) {
    if (strength) dropin_edge(0, p, sz, strength, 1, "svt_hip_av1_filter_intra_edge_high");
}
extern "C" void svt_hip_av1_upsample_intra_edge(uint8_t* p, int32_t sz) { dropin_edge(1, p, sz, 8, 0, "svt_hip_av1_upsample_intra_edge"); }
extern "C" void svt_hip_av1_upsample_intra_edge_high(uint16_t* p, int32_t sz, int32_t bd) {
    dropin_edge(1, p, sz, bd, 1, "svt_hip_av1_upsample_intra_edge_high");
}

// ---- subtract_average, cfl_predict_lbd / _hbd, av1_txb_init_levels (aom_dsp_rtcd.h:140-148, 2376) ----
extern "C" void svt_hip_subtract_average(int16_t* pred_buf_q3, int32_t width, int32_t height, int32_t round_offset,
                                         int32_t num_pel_log2) {
    const char* fn = "svt_hip_subtract_average";
    const size_t bytes = (size_t)32 * height * 2;                     // CFL_BUF_LINE = 32 int16 per row
    DROPIN_TRY(t_ctx.ensure(bytes), fn);
    HIP_DIE(hipMemcpyAsync(t_ctx.dbuf, pred_buf_q3, bytes, hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_subtract_average_batch((int16_t*)t_ctx.dbuf, 32, (size_t)32 * height, (uint32_t)width, (uint32_t)height, round_offset, num_pel_log2, 1,
                                              t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(pred_buf_q3, t_ctx.dbuf, bytes, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
static void dropin_cfl_predict(const int16_t* q3, const void* pred, int32_t pred_stride, void* dst, int32_t dst_stride,
                               int32_t alpha_q3, int32_t bit_depth, int32_t width, int32_t height, int is16, const char* fn) {
    const size_t es = is16 ? 2 : 1;
    const size_t qb = align256((size_t)32 * height * 2), pb = align256((size_t)width * height * es);
    DROPIN_TRY(t_ctx.ensure(qb + 2 * pb + 256), fn);
    char* d_q = t_ctx.dbuf;
    char* d_p = d_q + qb;
    char* d_d = d_p + pb;
    int32_t* d_alpha = (int32_t*)(d_d + pb);
    HIP_DIE(hipMemcpyAsync(d_q, q3, (size_t)32 * height * 2, hipMemcpyHostToDevice, t_ctx.stream), fn);
    HIP_DIE(hipMemcpy2DAsync(d_p, (size_t)width * es, pred, (size_t)pred_stride * es, (size_t)width * es, height, hipMemcpyHostToDevice, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(d_alpha, &alpha_q3, 4, hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_cfl_predict_batch((const int16_t*)d_q, 32, (size_t)32 * height, d_p, (uint32_t)width, d_d, (uint32_t)width, nullptr, d_alpha, bit_depth,
                                         (uint32_t)width, (uint32_t)height, is16, 1, t_ctx.stream), fn);
    HIP_DIE(hipMemcpy2DAsync(dst, (size_t)dst_stride * es, d_d, (size_t)width * es, (size_t)width * es, height, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
extern "C" void svt_hip_cfl_predict_lbd(const int16_t* pred_buf_q3, uint8_t* pred, int32_t pred_stride, uint8_t* dst,
                                        int32_t dst_stride, int32_t alpha_q3, int32_t bit_depth, int32_t width, int32_t height) {
    dropin_cfl_predict(pred_buf_q3, pred, pred_stride, dst, dst_stride, alpha_q3, bit_depth, width, height, 0, "svt_hip_cfl_predict_lbd");
}
extern "C" void svt_hip_cfl_predict_hbd(const int16_t* pred_buf_q3, uint16_t* pred, int32_t pred_stride, uint16_t* dst,
                                        int32_t dst_stride, int32_t alpha_q3, int32_t bit_depth, int32_t width, int32_t height) {
    dropin_cfl_predict(pred_buf_q3, pred, pred_stride, dst, dst_stride, alpha_q3, bit_depth, width, height, 1, "svt_hip_cfl_predict_hbd");
}
extern "C" void svt_hip_av1_txb_init_levels(const svt_tran_low_t* const coeff, const int32_t width, const int32_t height,
                                            uint8_t* const levels) {
    const char* fn = "svt_hip_av1_txb_init_levels";
    // `levels` points TX_PAD_TOP rows into the padded buffer (EbRateDistortionCost.c:125-150): whole buffer = (w+4)*(h+6)+16
    const int stride = width + 4;
    const size_t cb = align256((size_t)width * height * 4), lb = (size_t)stride * (height + 6) + 16;
    const size_t lpitch = (lb + 3) & ~(size_t)3;
    DROPIN_TRY(t_ctx.ensure(cb + lpitch), fn);
    HIP_DIE(hipMemcpyAsync(t_ctx.dbuf, coeff, (size_t)width * height * 4, hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_txb_init_levels_batch((const int32_t*)t_ctx.dbuf, (size_t)width * height, (uint8_t*)t_ctx.dbuf + cb, lpitch, (uint32_t)width, (uint32_t)height, 1,
                                             t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(levels - 2 * stride, t_ctx.dbuf + cb, lb, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
