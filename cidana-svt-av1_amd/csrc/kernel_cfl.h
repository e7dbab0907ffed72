// kernel_cfl.h — chroma-from-luma helpers of the encode pass (K11: cfl_luma_subsampling_420_{lbd,hbd},
// subtract_average, cfl_predict_{lbd,hbd}; reference EbIntraPrediction.c:1303-1402, called from
// Av1EncodeLoop, EbCodingLoop.c:736-846) and the entropy stage's level map (av1_txb_init_levels,
// EbRateDistortionCost.c:125-150).  All four are HBM-bound byte / int16 work: one 16-byte item per lane,
// a grid as large as the job, no LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_common.h"

namespace svtdev {

// one (possibly unaligned) 16- / 8-byte global access: a plain memcpy from a 1- or 2-byte-aligned pointer is split
// into dwordx2 / dword pieces by the compiler
typedef unsigned cfl_v4u __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned cfl_v2u __attribute__((ext_vector_type(2), aligned(1)));
template <int BYTES, typename T>
__device__ __forceinline__ void cfl_ld(T* dst, const void* src) {
    static_assert(BYTES == 32 || BYTES == 16 || BYTES == 8, "chunk");
    if constexpr (BYTES == 32) {
        const cfl_v4u a = reinterpret_cast<const cfl_v4u*>(src)[0], b = reinterpret_cast<const cfl_v4u*>(src)[1];
        __builtin_memcpy(dst, &a, 16); __builtin_memcpy(reinterpret_cast<char*>(dst) + 16, &b, 16);
    } else if constexpr (BYTES == 16) { const cfl_v4u a = *reinterpret_cast<const cfl_v4u*>(src); __builtin_memcpy(dst, &a, 16); }
    else { const cfl_v2u a = *reinterpret_cast<const cfl_v2u*>(src); __builtin_memcpy(dst, &a, 8); }
}
template <int BYTES, typename T>
__device__ __forceinline__ void cfl_st(void* dst, const T* src) {
    static_assert(BYTES == 16 || BYTES == 8 || BYTES == 4, "chunk");
    if constexpr (BYTES == 16) { cfl_v4u a; __builtin_memcpy(&a, src, 16); *reinterpret_cast<cfl_v4u*>(dst) = a; }
    else if constexpr (BYTES == 8) { cfl_v2u a; __builtin_memcpy(&a, src, 8); *reinterpret_cast<cfl_v2u*>(dst) = a; }
    else { unsigned a; __builtin_memcpy(&a, src, 4); __builtin_memcpy(dst, &a, 4); }
}

// ---------------------------------------------------------------------------
// cfl_ac_kernel<IN> — per chroma block: 2x2 luma sums * 2 (Q3), optionally minus the block average.
//   IN = 0: 8-bit luma, 1: 16-bit luma, 2: the Q3 buffer itself (subtract_average alone, in place).
// A lane owns one chunk of CS = min(W, 8) chroma samples of one row; a block's chunks sit in lpb <= 64
// consecutive lanes (two passes when a 32x32 block has 128 chunks), so the block sum is a __shfl_xor
// reduction.  Blocks are addressed by xy = x | y << 16 in a plane (d_xy) or densely (block pitch).
// ---------------------------------------------------------------------------
template <int IN>
__global__ __launch_bounds__(256) void cfl_ac_kernel(const void* __restrict__ luma, uint32_t luma_stride, size_t luma_block_pitch,
                                                     const uint32_t* __restrict__ xy, int16_t* __restrict__ q3, uint32_t q3_line,
                                                     size_t q3_block_pitch, uint32_t w, uint32_t h, uint32_t lpb, int subtract,
                                                     int round_offset, int num_pel_log2, uint32_t nblocks) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lsh = __builtin_ctz(lpb);
    const uint32_t blk = tid >> lsh, l = tid & (lpb - 1);
    const bool valid = blk < nblocks;
    const uint32_t cs = w < 8 ? 4u : 8u;                    // chroma samples per chunk
    const uint32_t cpr_sh = w == 32 ? 2u : (w == 16 ? 1u : 0u);
    const uint32_t nchunks = (w / cs) * h;
    int v[2][8];
    int sum = 0;
    int16_t* qb = q3 + (valid ? (size_t)blk * q3_block_pitch : 0);
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const uint32_t c = l + p * lpb;
        const bool on = valid && c < nchunks;
        const uint32_t cc = on ? c : 0;
        const uint32_t row = cc >> cpr_sh, col = (cc & ((1u << cpr_sh) - 1)) * cs;
#pragma unroll
        for (int i = 0; i < 8; i++) v[p][i] = 0;
        if (on) {
            if (IN == 2) {
                const int16_t* s = qb + (size_t)row * q3_line + col;
                if (cs == 8) { short vv[8]; cfl_ld<16>(vv, s);
#pragma unroll
                    for (int i = 0; i < 8; i++) v[p][i] = vv[i];
                } else { short vv[4]; cfl_ld<8>(vv, s);
#pragma unroll
                    for (int i = 0; i < 4; i++) v[p][i] = vv[i];
                }
            } else {
                size_t base;
                if (xy) { const uint32_t q = xy[blk]; base = (size_t)(q >> 16) * luma_stride + (q & 0xffffu); }
                else base = (size_t)blk * luma_block_pitch;
                base += (size_t)(2 * row) * luma_stride + 2 * col;
                if (IN == 0) {
                    const uint8_t* s = reinterpret_cast<const uint8_t*>(luma) + base;
                    uint8_t r0[16], r1[16];
                    if (cs == 8) { cfl_ld<16>(r0, s); cfl_ld<16>(r1, s + luma_stride); }
                    else { cfl_ld<8>(r0, s); cfl_ld<8>(r1, s + luma_stride); }
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (i < (int)cs) v[p][i] = ((int)r0[2 * i] + r0[2 * i + 1] + r1[2 * i] + r1[2 * i + 1]) << 1;
                } else {
                    const uint16_t* s = reinterpret_cast<const uint16_t*>(luma) + base;
                    uint16_t r0[16], r1[16];
                    if (cs == 8) { cfl_ld<32>(r0, s); cfl_ld<32>(r1, s + luma_stride); }
                    else { cfl_ld<16>(r0, s); cfl_ld<16>(r1, s + luma_stride); }
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (i < (int)cs) v[p][i] = (int)(int16_t)(uint16_t)(((int)r0[2 * i] + r0[2 * i + 1] + r1[2 * i] + r1[2 * i + 1]) << 1);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) sum += v[p][i];
        }
    }
    int avg = 0;
    if (subtract) {
        for (uint32_t m = lpb >> 1; m >= 1; m >>= 1) sum += __shfl_xor(sum, (int)m, 64);
        avg = (int)(int16_t)((sum + round_offset) >> num_pel_log2);
    }
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const uint32_t c = l + p * lpb;
        if (valid && c < nchunks) {
            const uint32_t row = c >> cpr_sh, col = (c & ((1u << cpr_sh) - 1)) * cs;
            short o[8];
#pragma unroll
            for (int i = 0; i < 8; i++) o[i] = (short)(v[p][i] - avg);
            int16_t* d = qb + (size_t)row * q3_line + col;
            if (cs == 8) cfl_st<16>(d, o); else cfl_st<8>(d, o);
        }
    }
}

// ---------------------------------------------------------------------------
// cfl_predict_kernel<PixT> — dst = clip(pred + round_signed(alpha_q3 * ac_q3, 6)); one chunk of min(W, 8)
// samples per lane.  pred and dst may be the same plane (the encode pass predicts in place).
// ---------------------------------------------------------------------------
template <typename PixT>
__global__ __launch_bounds__(256) void cfl_predict_kernel(const int16_t* __restrict__ ac, uint32_t q3_line, size_t q3_block_pitch,
                                                          const PixT* pred, uint32_t pred_stride, PixT* dst, uint32_t dst_stride,
                                                          const uint32_t* __restrict__ xy, const int32_t* __restrict__ alpha_q3,
                                                          int hi, uint32_t w, uint32_t h, uint32_t per_block_sh, uint32_t nblocks) {
    const size_t tid = (size_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t blk = (uint32_t)(tid >> per_block_sh);
    if (blk >= nblocks) return;
    const uint32_t c = (uint32_t)tid & ((1u << per_block_sh) - 1);
    const uint32_t cs = w < 8 ? 4u : 8u;
    const uint32_t cpr_sh = w == 32 ? 2u : (w == 16 ? 1u : 0u);
    const uint32_t row = c >> cpr_sh, col = (c & ((1u << cpr_sh) - 1)) * cs;
    size_t pb, db;
    if (xy) { const uint32_t q = xy[blk]; pb = (size_t)(q >> 16) * pred_stride + (q & 0xffffu); db = (size_t)(q >> 16) * dst_stride + (q & 0xffffu); }
    else { pb = (size_t)blk * pred_stride * h; db = (size_t)blk * dst_stride * h; }
    const int a = alpha_q3[blk];
    short q[8];
    PixT pv[8], ov[8];
    const int16_t* s = ac + (size_t)blk * q3_block_pitch + (size_t)row * q3_line + col;
    const PixT* pp = pred + pb + (size_t)row * pred_stride + col;
    if (cs == 8) { cfl_ld<16>(q, s); cfl_ld<8 * sizeof(PixT)>(pv, pp); }
    else {
        cfl_ld<8>(q, s);
        if constexpr (sizeof(PixT) == 2) cfl_ld<8>(pv, pp); else __builtin_memcpy(pv, pp, 4);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i < (int)cs) {
            const int q6 = a * q[i];
            const int mag = ((q6 < 0 ? -q6 : q6) + 32) >> 6;
            int v = (int)(int16_t)pv[i] + (q6 < 0 ? -mag : mag);
            v = v < 0 ? 0 : (v > hi ? hi : v);
            ov[i] = (PixT)v;
        }
    }
    PixT* dp = dst + db + (size_t)row * dst_stride + col;
    if (cs == 8) cfl_st<8 * sizeof(PixT)>(dp, ov); else cfl_st<4 * sizeof(PixT)>(dp, ov);
}

// ---------------------------------------------------------------------------
// txb_init_levels_kernel — levels buffer of one block = (W + 4) x (H + 6) bytes + 16: two zero rows, H rows of
// min(|coeff|, 127) with 4 zero bytes of right padding, four zero rows + 16 zero bytes.  A lane writes one
// dword of the buffer (W + 4 is a multiple of 4, so a dword never straddles a row) from one 16-byte load.
// ---------------------------------------------------------------------------
template <bool WIDE>
__device__ __forceinline__ void txb_levels_body(const int32_t* __restrict__ cb, uint32_t* __restrict__ ob, uint32_t w, uint32_t h, uint32_t l,
                                                uint32_t lpb, uint32_t ndw, uint32_t row_magic) {
    const uint32_t dpr = (w + 4) >> 2;                       // dwords per row
    auto dword = [&](uint32_t d) -> uint32_t {
        const uint32_t r = __umulhi(d, row_magic), cq = d - r * dpr;          // d / dpr, d % dpr
        uint32_t out = 0;
        const uint32_t y = r - 2;                            // TX_PAD_TOP rows above (wraps to huge for r < 2)
        if (y < h && cq < (w >> 2)) {
            int4 c4;
            __builtin_memcpy(&c4, cb + (size_t)y * w + 4 * cq, 16);
            const int cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // |INT32_MIN| is undefined in the reference (abs); every other value is exact
                const uint32_t a = (uint32_t)(cv[i] < 0 ? -(long long)cv[i] : (long long)cv[i]);
                out |= (a > 127u ? 127u : a) << (8 * i);
            }
        }
        return out;
    };
    if (WIDE) {        // 16 output bytes per lane (block buffers 16-byte aligned): up to four 16-byte loads, one 16-byte store
        for (uint32_t d = 4 * l; d < ndw; d += 4 * lpb) {
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = dword(min(d + k, ndw - 1));
            if (d + 4 <= ndw) *reinterpret_cast<uint4*>(ob + d) = make_uint4(o[0], o[1], o[2], o[3]);
            else {
#pragma unroll
                for (int k = 0; k < 3; k++) if (d + k < ndw) ob[d + k] = o[k];
            }
        }
    } else {
        for (uint32_t d = l; d < ndw; d += lpb) ob[d] = dword(d);
    }
}

template <bool WIDE>
__global__ __launch_bounds__(256) void txb_init_levels_kernel(const int32_t* __restrict__ coeff, size_t coeff_block_pitch,
                                                              uint8_t* __restrict__ levels_buf, size_t levels_block_pitch,
                                                              uint32_t w, uint32_t h, uint32_t lpb, uint32_t ndw, uint32_t row_magic,
                                                              uint32_t nblocks) {
    const uint32_t lsh = __builtin_ctz(lpb);
    const uint32_t slots = 256u >> lsh;
    const uint32_t blk = blockIdx.x * slots + (threadIdx.x >> lsh);
    if (blk >= nblocks) return;
    txb_levels_body<WIDE>(coeff + (size_t)blk * coeff_block_pitch, reinterpret_cast<uint32_t*>(levels_buf + (size_t)blk * levels_block_pitch), w, h,
                          threadIdx.x & (lpb - 1), lpb, ndw, row_magic);
}

// ---------------------------------------------------------------------------
// The two pieces of the encode pass that sit beside the transform chain, at picture scale (svt_hip_encode_recon_frame_ex):
// one launch covers every group of a call, the group table rides in the kernel arguments (as enc_frame_kernel's).
//
// levels_frame_kernel  av1_txb_init_levels of every block's quantised coefficients, straight from the dense qcoeff output of the
//                      encode launch that precedes it in the stream.
// cfl_frame_kernel     the chroma-from-luma step between a picture's luma and chroma encode passes (Av1EncodeLoop,
//                      EbCodingLoop.c:736-846): cfl_luma_subsampling_420 of the luma RECONSTRUCTION under the chroma block,
//                      subtract_average, then cfl_predict on the Cb and the Cr prediction in place.  The Q3 values never leave
//                      the registers (the reference keeps them in pred_buf_q3 between its four calls).
// ---------------------------------------------------------------------------
constexpr int LEVELS_MAX_GROUPS = 48;
struct LevelsGroupDev {
    const int32_t* coeff; uint8_t* levels;
    uint32_t levels_pitch, w, h, lpb, ndw, row_magic, nblocks;
    uint32_t wg_end;                 // bit 31: 16-byte stores (buffer and pitch 16-byte aligned)
};
struct LevelsFrameDesc { int32_t ngroups; LevelsGroupDev g[LEVELS_MAX_GROUPS]; };
static_assert(sizeof(LevelsFrameDesc) <= 4000, "kernel arguments");

__global__ __launch_bounds__(256) void levels_frame_kernel(const LevelsFrameDesc fd) {
    int gi = 0;
    uint32_t start = 0;
#pragma unroll 1
    for (int i = 0; i < fd.ngroups; i++) {
        const uint32_t e = fd.g[i].wg_end & 0x7fffffffu;
        if (blockIdx.x >= e) { gi = i + 1; start = e; }
    }
    if (gi >= fd.ngroups) return;
    const LevelsGroupDev& G = fd.g[gi];
    const uint32_t lsh = __builtin_ctz(G.lpb);
    const uint32_t blk = (blockIdx.x - start) * (256u >> lsh) + (threadIdx.x >> lsh);
    if (blk >= G.nblocks) return;
    const int32_t* cb = G.coeff + (size_t)blk * (G.w * G.h);
    uint32_t* ob = reinterpret_cast<uint32_t*>(G.levels + (size_t)blk * G.levels_pitch);
    if (G.wg_end >> 31) txb_levels_body<true>(cb, ob, G.w, G.h, threadIdx.x & (G.lpb - 1), G.lpb, G.ndw, G.row_magic);
    else txb_levels_body<false>(cb, ob, G.w, G.h, threadIdx.x & (G.lpb - 1), G.lpb, G.ndw, G.row_magic);
}

constexpr int CFL_MAX_GROUPS = 16;      // the chroma transform sizes 4 .. 32 in both dimensions
struct CflGroupDev {
    const void* luma; void* cb; void* cr;
    const uint32_t* xy; const int32_t* alpha_cb; const int32_t* alpha_cr;
    uint32_t luma_stride, cb_stride, cr_stride, nblocks, w, h, lpb, wg_end;
    int32_t round_offset, num_pel_log2;
};
struct CflFrameDesc { int32_t ngroups; CflGroupDev g[CFL_MAX_GROUPS]; };

template <typename PixT>
__global__ __launch_bounds__(256) void cfl_frame_kernel(const CflFrameDesc fd, int hi) {
    int gi = 0;
    uint32_t start = 0;
#pragma unroll 1
    for (int i = 0; i < fd.ngroups; i++) {
        if (blockIdx.x >= fd.g[i].wg_end) { gi = i + 1; start = fd.g[i].wg_end; }
    }
    if (gi >= fd.ngroups) return;
    const CflGroupDev& G = fd.g[gi];
    const uint32_t lpb = G.lpb, w = G.w, h = G.h;
    const uint32_t lsh = __builtin_ctz(lpb);
    const uint32_t blk = (blockIdx.x - start) * (256u >> lsh) + (threadIdx.x >> lsh), l = threadIdx.x & (lpb - 1);
    const bool valid = blk < G.nblocks;
    const uint32_t cs = w < 8 ? 4u : 8u;                    // chroma samples per chunk
    const uint32_t cpr_sh = w == 32 ? 2u : (w == 16 ? 1u : 0u);
    const uint32_t nchunks = (w / cs) * h;
    const uint32_t q = valid ? G.xy[blk] : 0u;
    const uint32_t bx = q & 0xffffu, by = q >> 16;
    int v[2][8];
    int sum = 0;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const uint32_t c = l + p * lpb;
        const bool on = valid && c < nchunks;
        const uint32_t row = on ? c >> cpr_sh : 0u, col = on ? (c & ((1u << cpr_sh) - 1)) * cs : 0u;
#pragma unroll
        for (int i = 0; i < 8; i++) v[p][i] = 0;
        if (on) {
            const PixT* s = reinterpret_cast<const PixT*>(G.luma) + (size_t)(2 * (by + row)) * G.luma_stride + 2 * (bx + col);
            PixT r0[16], r1[16];
            if (cs == 8) { cfl_ld<16 * sizeof(PixT)>(r0, s); cfl_ld<16 * sizeof(PixT)>(r1, s + G.luma_stride); }
            else { cfl_ld<8 * sizeof(PixT)>(r0, s); cfl_ld<8 * sizeof(PixT)>(r1, s + G.luma_stride); }
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (i < (int)cs) v[p][i] = (int)(int16_t)(uint16_t)(((int)r0[2 * i] + r0[2 * i + 1] + r1[2 * i] + r1[2 * i + 1]) << 1);
#pragma unroll
            for (int i = 0; i < 8; i++) sum += v[p][i];
        }
    }
    for (uint32_t m = lpb >> 1; m >= 1; m >>= 1) sum += __shfl_xor(sum, (int)m, 64);
    const int avg = (int)(int16_t)((sum + G.round_offset) >> G.num_pel_log2);
    if (!valid) return;
    const int a_cb = G.alpha_cb[blk], a_cr = G.alpha_cr[blk];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const uint32_t c = l + p * lpb;
        if (c >= nchunks) continue;
        const uint32_t row = c >> cpr_sh, col = (c & ((1u << cpr_sh) - 1)) * cs;
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            PixT* pp = reinterpret_cast<PixT*>(pl ? G.cr : G.cb) + (size_t)(by + row) * (pl ? G.cr_stride : G.cb_stride) + bx + col;
            const int a = pl ? a_cr : a_cb;
            PixT pv[8], ov[8];
            if (cs == 8) cfl_ld<8 * sizeof(PixT)>(pv, pp);
            else if constexpr (sizeof(PixT) == 2) cfl_ld<8>(pv, pp); else __builtin_memcpy(pv, pp, 4);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (i < (int)cs) {
                    const int q6 = a * (int)(int16_t)(v[p][i] - avg);
                    const int mag = ((q6 < 0 ? -q6 : q6) + 32) >> 6;
                    int o = (int)(int16_t)pv[i] + (q6 < 0 ? -mag : mag);
                    o = o < 0 ? 0 : (o > hi ? hi : o);
                    ov[i] = (PixT)o;
                }
            }
            if (cs == 8) cfl_st<8 * sizeof(PixT)>(pp, ov); else cfl_st<4 * sizeof(PixT)>(pp, ov);
        }
    }
}

}  // namespace svtdev
