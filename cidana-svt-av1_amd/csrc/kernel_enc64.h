// kernel_enc64.h — the encode-pass chain (ResidualKernel -> FwdTxfm2d -> quantise / dequantise -> InvTxfm2d + add, see
// enc32_kernel) for 64x64 DCT_DCT blocks, TWO BLOCKS PER WAVE.
//
// Only 32x32 coefficients of a 64x64 transform are coded (EbTransforms.c:4351-4408, :8226-8240), so the forward row pass and
// the inverse row pass have 32 rows of work per block: in the generic staged kernel (one block per wave) half the lanes of a
// wave idle there while the wave still issues every instruction.  Here a wave owns two blocks; the column passes run once per
// block (64 lanes = 64 columns), the two row passes run ONCE for both blocks (lanes 0-31: rows of block A, 32-63: block B):
// 6 pass executions per two blocks instead of 8.  Further: the forward networks compute only the 32 outputs that are kept
// (svt_fdct64_out32), the inverse ones know inputs 32..63 are zero (svt_idct64_low32) and drop their stage clamps under the
// L1 bound (idct64_pass below, as idct32_pass), the residual is formed once while staging (packed int16, SAD with v_sad_u8 on
// the raw words) instead of per element in the column pass, and the prediction is re-read (L2) for the reconstruction
// instead of being held in 32 VGPRs across the transforms.
//
// LDS per wave: 2 x 8 320 B.  Block b's region [b * 8320, +8320) holds in turn: its residual (64 rows x 128 B), its column-
// pass output (32 rows x 64 words at a pitch of 65), its 32x32 coefficients (rows of 128 B, 16-B slots XOR-swizzled by row),
// their dequantised values, the inverse row-pass output (32 x 65 words) and the reconstructed residual (64 x 128 B).
// Every hand-over is "all lanes read into registers, fence, write".
#pragma once
#include "kernel_txfm_staged.h"

namespace svtdev {

constexpr int E64_WAVES = 2;
constexpr int E64_TILE = 32 * 65 * 4;            // 8320
constexpr int E64_WAVE_LDS = 2 * E64_TILE;

// one 64-point inverse pass with 32 live inputs (raw as loaded): input clamp + idct64_low32, clamp-free under the L1 bound
template <int IN_BITS, int STAGE_BITS>
__device__ __forceinline__ void idct64_pass(int (&x)[64]) {
    constexpr int in_max = (1 << (IN_BITS - 1)) - 1, st_max = (1 << (STAGE_BITS - 1)) - 1;
    constexpr int lim_net = svtgen::svt_clamp_free_l1(svtgen::svt_idct64_low32_gain_q10, svtgen::svt_idct64_low32_slack, st_max);
    constexpr int lim = lim_net < in_max ? lim_net : in_max;
    const int l1 = svtgen::svt_l1<64>(x, 32);
    if (__builtin_amdgcn_ballot_w64(l1 > lim) == 0) {
        svtgen::svt_idct64_low32<12, false, false>(x, 0, 0);
        return;
    }
    const int in_hi = svtgen::svt_vgpr(in_max), in_lo = ~in_hi;
    const int st_hi = STAGE_BITS == IN_BITS ? in_hi : svtgen::svt_vgpr(st_max), st_lo = ~st_hi;
#pragma unroll
    for (int i = 0; i < 32; i++) x[i] = svtgen::svt_clamp(x[i], in_lo, in_hi);
    svtgen::svt_idct64_low32<12>(x, st_lo, st_hi);
}

// MODE: E64_FULL the whole chain; E64_FWD stops after the quantiser (svt_hip_fwd_quant_*: coeff / qcoeff / dqcoeff / eob / sad).
// (An inverse-only mode was measured against inv_staged_kernel<64, 64> with its 16-bit tile at 4 waves / SIMD: equal, 0.93-0.96 ms
// per 2^18 blocks, so the standalone inverse keeps that kernel.)
enum { E64_FULL = 0, E64_FWD = 1 };
template <typename PixT, int BD, bool KEEP, int MODE = E64_FULL>
__device__ __forceinline__ void enc64_body(
    const PixT* __restrict__ src, const PixT* __restrict__ pred, PixT* __restrict__ recon,
    int32_t* __restrict__ coeff, int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff,
    uint16_t* __restrict__ eob, uint32_t* __restrict__ sad, const int16_t* __restrict__ iscan, const QParams& qp,
    uint32_t nblocks, const uint32_t* __restrict__ xy, uint32_t src_stride, uint32_t pred_stride, uint32_t recon_stride,
    uint32_t bid, char* lds) {
    constexpr int ES = (int)sizeof(PixT), PPC = 16 / ES, CPR = 64 / PPC, CPB = 64 * CPR, NIT = 2 * CPB / 64;
    constexpr int in_bits = BD + 8, row_bits = BD == 8 ? 16 : (BD == 10 ? 18 : 20);      // av1_gen_inv_stage_range (:5404-5456)
    constexpr int cin_bits = BD + 6 > 16 ? BD + 6 : 16, col_bits = BD == 12 ? 18 : 16, maxpix = (1 << BD) - 1;
    constexpr int CBC = fwd_cos_col(64, 64), CBR = fwd_cos_row(64, 64);
    static_assert(fwd_shift(64, 64, 0) == 0 && fwd_shift(64, 64, 1) == -2 && fwd_shift(64, 64, 2) == -2 && inv_shift0(64, 64) == -2, "64x64 shifts");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* wl = lds + wave * E64_WAVE_LDS;
    if (wave >= E64_WAVES) return;                         // (run inside a larger workgroup: the spare waves have nothing to do)
    const uint32_t first = (bid * E64_WAVES + wave) * 2;
    if (first >= nblocks) return;                          // wave-uniform
    const bool two = first + 1 < nblocks;                  // the last wave of an odd batch: its second block repeats the first, unstored
    size_t sb[2], pb[2], rb[2];
    uint32_t ss = 64, ps = 64, rs = 64;
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const uint32_t blk = (b == 1 && !two) ? first : first + b;
        if (xy) {
            const uint32_t o = xy[blk];
            const size_t y = o >> 16, x = o & 0xffffu;
            sb[b] = y * src_stride + x; pb[b] = y * pred_stride + x; rb[b] = y * recon_stride + x;
        } else {
            sb[b] = pb[b] = rb[b] = (size_t)blk * 4096;
        }
    }
    if (xy) { ss = src_stride; ps = pred_stride; rs = recon_stride; }

    const int hb = lane >> 5, hk = lane & 31;
    char* myrow = wl + hb * E64_TILE + hk * 128;           // this lane's coefficient row (32 x int32, slots swizzled by hk & 7)
    unsigned sad_acc[2] = {0, 0};
    constexpr int BATCH = 8;
    {
    // ---- residual -> LDS (packed int16), SAD on the raw words -------------------------------------------------------------
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += BATCH) {
        uint4 sv[BATCH], pv[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; k++) {
            const int it = it0 + k, b = it / (CPB / 64), w = (it % (CPB / 64)) * 64 + lane;
            const int row = w / CPR, col = (w % CPR) * PPC;
            // (memcpy: plane origins need not be 16-B aligned; gfx950 takes the unaligned dwordx4)
            __builtin_memcpy(&sv[k], src + sb[b] + (size_t)row * ss + col, 16);
            __builtin_memcpy(&pv[k], pred + pb[b] + (size_t)row * ps + col, 16);
        }
#pragma unroll
        for (int k = 0; k < BATCH; k++) {
            const int it = it0 + k, b = it / (CPB / 64), w = (it % (CPB / 64)) * 64 + lane;
            const int row = w / CPR, col = (w % CPR) * PPC;
            const uint32_t a[4] = {sv[k].x, sv[k].y, sv[k].z, sv[k].w}, p[4] = {pv[k].x, pv[k].y, pv[k].z, pv[k].w};
            char* d = wl + b * E64_TILE + row * 128 + col * 2;
            if constexpr (ES == 1) {
                uint32_t r[8];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    sad_acc[b] = __builtin_amdgcn_sad_u8(a[j], p[j], sad_acc[b]);
                    const uint32_t a01 = __builtin_amdgcn_perm(0u, a[j], 0x0c010c00u), a23 = __builtin_amdgcn_perm(0u, a[j], 0x0c030c02u);
                    const uint32_t p01 = __builtin_amdgcn_perm(0u, p[j], 0x0c010c00u), p23 = __builtin_amdgcn_perm(0u, p[j], 0x0c030c02u);
                    r[2 * j] = pk_sub_i16(a01, p01); r[2 * j + 1] = pk_sub_i16(a23, p23);
                }
                *reinterpret_cast<uint4*>(d) = make_uint4(r[0], r[1], r[2], r[3]);
                *reinterpret_cast<uint4*>(d + 16) = make_uint4(r[4], r[5], r[6], r[7]);
            } else {
                uint32_t r[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    sad_acc[b] = __builtin_amdgcn_sad_u16(a[j], p[j], sad_acc[b]);
                    r[j] = pk_sub_i16(a[j], p[j]);
                }
                *reinterpret_cast<uint4*>(d) = make_uint4(r[0], r[1], r[2], r[3]);
            }
        }
    }
    wave_lds_fence();
    // ---- forward column pass, once per block: lane = column -------------------------------------------------------------
#pragma unroll 1
    for (int b = 0; b < 2; b++) {
        int x[64];
        const char* rb0 = wl + b * E64_TILE + lane * 2;
#pragma unroll
        for (int r = 0; r < 64; r++) x[r] = (int)*reinterpret_cast<const short*>(rb0 + r * 128);
        svtgen::svt_fdct64_out32<CBC>(x);
        wave_lds_fence();                                  // the block's residual is dead: its tile may overwrite it
        int32_t* tile = reinterpret_cast<int32_t*>(wl + b * E64_TILE);
#pragma unroll
        for (int k = 0; k < 32; k++) tile[k * 65 + lane] = (x[k] + 2) >> 2;        // shift[1] = -2
    }
    wave_lds_fence();
    // ---- forward row pass, both blocks at once: lanes 0-31 rows of block A, 32-63 rows of block B ------------------------
    {
        int y[64];
        const int32_t* trow = reinterpret_cast<const int32_t*>(wl + hb * E64_TILE) + hk * 65;
#pragma unroll
        for (int c = 0; c < 64; c++) y[c] = trow[c];
        svtgen::svt_fdct64_out32<CBR>(y);
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < 8; s++)
            *reinterpret_cast<int4*>(myrow + ((s ^ (hk & 7)) << 4)) =
                make_int4((y[4 * s] + 2) >> 2, (y[4 * s + 1] + 2) >> 2, (y[4 * s + 2] + 2) >> 2, (y[4 * s + 3] + 2) >> 2);   // shift[2] = -2
    }
    wave_lds_fence();
    }
    // ---- quantise in linear chunk order (coalesced stores); dequantised chunks stay in registers --------------------------
    {
        int4 dvs[8];
        int eob_acc[2] = {0, 0};
        {
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int b = it >> 2, w = (it & 3) * 64 + lane;       // chunk w of block b: row w / 8, slot w % 8
            const int row = w >> 3, slot = w & 7;
            const int4 c = *reinterpret_cast<const int4*>(wl + b * E64_TILE + row * 128 + ((slot ^ (row & 7)) << 4));
            int4 qv, dv;
            quant_one<2>(c.x, w == 0 ? 0 : 1, qp, qv.x, dv.x);
            quant_one<2>(c.y, 1, qp, qv.y, dv.y);
            quant_one<2>(c.z, 1, qp, qv.z, dv.z);
            quant_one<2>(c.w, 1, qp, qv.w, dv.w);
            dvs[it] = dv;
            const uint2 is = *reinterpret_cast<const uint2*>(iscan + w * 4);
            const int e = max(max(qv.x ? (int)(is.x & 0xffffu) + 1 : 0, qv.y ? (int)(is.x >> 16) + 1 : 0),
                              max(qv.z ? (int)(is.y & 0xffffu) + 1 : 0, qv.w ? (int)(is.y >> 16) + 1 : 0));
            eob_acc[b] = max(eob_acc[b], e);
            if (b == 0 || two) {
                const size_t o = (size_t)(first + b) * 1024 + (size_t)w * 4;
                *reinterpret_cast<int4*>(qcoeff + o) = qv;
                if (KEEP) { *reinterpret_cast<int4*>(coeff + o) = c; *reinterpret_cast<int4*>(dqcoeff + o) = dv; }
            }
        }
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int m = group_max<64>(eob_acc[b]);
            const unsigned sd = group_sum<64>(sad_acc[b]);
            if (lane == 0 && (b == 0 || two)) {
                eob[first + b] = (uint16_t)m;
                if (sad) sad[first + b] = sd;
            }
        }
        }
        if constexpr (MODE == E64_FWD) return;
        wave_lds_fence();                                  // every chunk is read: the dequantised ones take their places
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int b = it >> 2, w = (it & 3) * 64 + lane;
            const int row = w >> 3, slot = w & 7;
            *reinterpret_cast<int4*>(wl + b * E64_TILE + row * 128 + ((slot ^ (row & 7)) << 4)) = dvs[it];
        }
    }
    wave_lds_fence();
    // ---- inverse row pass, both blocks at once ------------------------------------------------------------------------------
    {
        int x[64];
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int4 v = *reinterpret_cast<const int4*>(myrow + ((s ^ (hk & 7)) << 4));
            x[4 * s] = v.x; x[4 * s + 1] = v.y; x[4 * s + 2] = v.z; x[4 * s + 3] = v.w;
        }
#pragma unroll
        for (int c = 32; c < 64; c++) x[c] = 0;
        idct64_pass<in_bits, row_bits>(x);
        wave_lds_fence();
        int32_t* trow = reinterpret_cast<int32_t*>(wl + hb * E64_TILE) + hk * 65;
#pragma unroll
        for (int c = 0; c < 64; c++) trow[c] = (x[c] + 2) >> 2;                    // inv shift[0] = -2
    }
    wave_lds_fence();
    // ---- inverse column pass, once per block: lane = column; residual back as int16 rows -----------------------------------
#pragma unroll 1
    for (int b = 0; b < 2; b++) {
        int y[64];
        const int32_t* tile = reinterpret_cast<const int32_t*>(wl + b * E64_TILE);
#pragma unroll
        for (int r = 0; r < 32; r++) y[r] = tile[r * 65 + lane];
#pragma unroll
        for (int r = 32; r < 64; r++) y[r] = 0;
        idct64_pass<cin_bits, col_bits>(y);
        wave_lds_fence();
        short* res = reinterpret_cast<short*>(wl + b * E64_TILE);
#pragma unroll
        for (int r = 0; r < 64; r++) res[r * 64 + lane] = (short)((y[r] + 8) >> 4);   // inv shift[1] = -4
    }
    wave_lds_fence();
    // ---- reconstruction = prediction (re-read: L2) + residual ---------------------------------------------------------------
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += BATCH) {
        uint4 pv[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; k++) {
            const int it = it0 + k, b = it / (CPB / 64), w = (it % (CPB / 64)) * 64 + lane;
            const int row = w / CPR, col = (w % CPR) * PPC;
            __builtin_memcpy(&pv[k], pred + pb[b] + (size_t)row * ps + col, 16);
        }
#pragma unroll
        for (int k = 0; k < BATCH; k++) {
            const int it = it0 + k, b = it / (CPB / 64), w = (it % (CPB / 64)) * 64 + lane;
            const int row = w / CPR, col = (w % CPR) * PPC;
            const uint4* rs4 = reinterpret_cast<const uint4*>(wl + b * E64_TILE + row * 128 + col * 2);
            const uint32_t pw[4] = {pv[k].x, pv[k].y, pv[k].z, pv[k].w};
            uint32_t ow[4];
            if constexpr (ES == 1) {
                const uint4 ra = rs4[0], rb4 = rs4[1];
                const uint32_t rw[8] = {ra.x, ra.y, ra.z, ra.w, rb4.x, rb4.y, rb4.z, rb4.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t p01 = __builtin_amdgcn_perm(0u, pw[j], 0x0c010c00u), p23 = __builtin_amdgcn_perm(0u, pw[j], 0x0c030c02u);
                    const uint32_t u01 = sat_pk_u8_i16(pk_add_i16(p01, rw[2 * j])), u23 = sat_pk_u8_i16(pk_add_i16(p23, rw[2 * j + 1]));
                    ow[j] = (u23 << 16) | (u01 & 0xffffu);
                }
            } else {
                const uint4 ra = rs4[0];
                const uint32_t rw[4] = {ra.x, ra.y, ra.z, ra.w};
#pragma unroll
                for (int j = 0; j < 4; j++) ow[j] = pk_clamp_i16(pk_add_i16(pw[j], rw[j]), maxpix);
            }
            if (b == 0 || two) __builtin_memcpy(recon + rb[b] + (size_t)row * rs + col, ow, 16);
        }
    }
}

// SAD = false: the caller takes no SAD (the frame-level calls): without that path the 10-bit form needs 192 registers instead of
// 256 + 30 AGPRs - two waves per SIMD (what the LDS allows) instead of one
template <typename PixT, int BD, bool KEEP, bool SAD = true>
__global__ __launch_bounds__(E64_WAVES * 64) void enc64_kernel(
    const PixT* __restrict__ src, const PixT* __restrict__ pred, PixT* __restrict__ recon,
    int32_t* __restrict__ coeff, int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff,
    uint16_t* __restrict__ eob, uint32_t* __restrict__ sad, const int16_t* __restrict__ iscan, QParams qp,
    uint32_t nblocks, const uint32_t* __restrict__ xy, uint32_t src_stride, uint32_t pred_stride, uint32_t recon_stride) {
    __shared__ __attribute__((aligned(16))) char lds[E64_WAVES * E64_WAVE_LDS];
    enc64_body<PixT, BD, KEEP>(src, pred, recon, coeff, qcoeff, dqcoeff, eob, SAD ? sad : nullptr, iscan, qp, nblocks, xy, src_stride, pred_stride,
                               recon_stride, blockIdx.x, lds);
}

// svt_hip_fwd_quant_*: the forward half (coeff, qcoeff, dqcoeff, eob, sad out; no three_quad_energy - the pruned networks never
// form the discarded coefficients, so the dispatcher keeps the one-block kernel when the caller asks for the energy)
template <typename PixT, int BD>
__global__ __launch_bounds__(E64_WAVES * 64) void fq64_kernel(
    const PixT* __restrict__ src, const PixT* __restrict__ pred, int32_t* __restrict__ coeff, int32_t* __restrict__ qcoeff,
    int32_t* __restrict__ dqcoeff, uint16_t* __restrict__ eob, uint32_t* __restrict__ sad, const int16_t* __restrict__ iscan, QParams qp,
    uint32_t nblocks, const uint32_t* __restrict__ xy, uint32_t src_stride, uint32_t pred_stride) {
    __shared__ __attribute__((aligned(16))) char lds[E64_WAVES * E64_WAVE_LDS];
    enc64_body<PixT, BD, true, E64_FWD>(src, pred, nullptr, coeff, qcoeff, dqcoeff, eob, sad, iscan, qp, nblocks, xy, src_stride, pred_stride, 0,
                                        blockIdx.x, lds);
}
}  // namespace svtdev
