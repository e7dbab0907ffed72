// kernel_intra.h — batched intra prediction (K9 non-directional, K10
// directional + edge filter / upsample) and coefficient-domain distortion.
//
// Reference: EbIntraPrediction.c:1838-2260 (dc/v/h/smooth/paeth, highbd twins),
// :370-477 and :3394-3506 (av1_dr_prediction_z1/z2/z3), :3539-3660 (edge filter,
// upsample); EbPictureOperators.c:283-346 (full_distortion_kernel32_bits).
//
// Write-bound: 1 (u8) or 2 (u16) bytes out per pixel, ~(bw+bh) neighbour samples
// in per block.  One lane produces 16 bytes of one output row (16 u8 / 8 u16
// pixels; narrower blocks: the whole row) so stores are as wide as the block
// allows; neighbour rows are tiny and served from L1/L2.
#pragma once
#include "dev_common.h"

namespace svtdev {

enum { IM_DC = 0, IM_V, IM_H, IM_SMOOTH, IM_SMOOTH_V, IM_SMOOTH_H, IM_PAETH, IM_DC_TOP, IM_DC_LEFT, IM_DC_128,
       IM_Z1, IM_Z2, IM_Z3, IM_MODES };

// AV1 smooth weights (sm_weight_arrays, ASM_AVX2/EbIntraPrediction_AVX2.h:19-38), index [bs + i]
__device__ constexpr uint8_t kSmWeights[128] = {
    0, 0, 255, 128, 255, 149, 85, 64, 255, 197, 146, 105, 73, 50, 37, 32,
    255, 225, 196, 170, 145, 123, 102, 84, 68, 54, 43, 33, 26, 20, 17, 16,
    255, 240, 225, 210, 196, 182, 169, 157, 145, 133, 122, 111, 101, 92, 83, 74,
    66, 59, 52, 45, 39, 34, 29, 25, 21, 17, 14, 12, 10, 9, 8, 8,
    255, 248, 240, 233, 225, 218, 210, 203, 196, 189, 182, 176, 169, 163, 156, 150,
    144, 138, 133, 127, 121, 116, 111, 106, 101, 96, 91, 86, 82, 77, 73, 69,
    65, 61, 57, 54, 50, 47, 44, 41, 38, 35, 32, 29, 27, 25, 22, 20,
    18, 16, 15, 13, 12, 10, 9, 8, 7, 6, 6, 5, 5, 4, 4, 4};

// Neighbour layout per block: `nb_pitch` samples; position p of the reference's
// above_row / left_col lives at index NB_ORIGIN + p (p >= -2).
constexpr int NB_ORIGIN = 16;

template <typename PixT, int MODE>
__device__ __forceinline__ void intra_row(PixT (&out)[16 / sizeof(PixT)], const PixT* __restrict__ above,
                                          const PixT* __restrict__ left, int r, int c0, int ppl, int bw, int bh, int wh,
                                          int dc, int up_above, int up_left, int dx, int dy, int maxv) {
    constexpr int PXL = 16 / (int)sizeof(PixT);
    PixT ab[PXL];                                          // above[c0 .. c0+ppl-1] in one wide load
    if (MODE == IM_V || MODE == IM_SMOOTH || MODE == IM_SMOOTH_V || MODE == IM_PAETH) {
        if (ppl == PXL) __builtin_memcpy(ab, above + c0, 16);
        else {
#pragma unroll
            for (int k = 0; k < PXL; k++) ab[k] = k < ppl ? above[c0 + k] : (PixT)0;
        }
    }
    const int lft = (MODE == IM_H || MODE == IM_SMOOTH || MODE == IM_SMOOTH_H || MODE == IM_PAETH) ? (int)left[r] : 0;
    const int bl = (MODE == IM_SMOOTH || MODE == IM_SMOOTH_V) ? (int)left[bh - 1] : 0;
    const int tr = (MODE == IM_SMOOTH || MODE == IM_SMOOTH_H) ? (int)above[bw - 1] : 0;
    const int tl = MODE == IM_PAETH ? (int)above[-1] : 0;
#pragma unroll
    for (int k = 0; k < PXL; k++) {
        const int c = c0 + (k < ppl ? k : 0);          // lanes of narrow blocks recompute pixel 0 (discarded)
        int v;
        if (MODE == IM_V) v = ab[k];
        else if (MODE == IM_H) v = lft;
        else if (MODE == IM_SMOOTH) {
            const int ww = kSmWeights[bw + c];
            v = (wh * ab[k] + (256 - wh) * bl + ww * lft + (256 - ww) * tr + 256) >> 9;
        } else if (MODE == IM_SMOOTH_V) v = (wh * ab[k] + (256 - wh) * bl + 128) >> 8;
        else if (MODE == IM_SMOOTH_H) {
            const int ww = kSmWeights[bw + c];
            v = (ww * lft + (256 - ww) * tr + 128) >> 8;
        } else if (MODE == IM_PAETH) {
            const int t = ab[k];
            const int base = t + lft - tl;
            const int pl = abs(base - lft), pt = abs(base - t), ptl = abs(base - tl);
            v = (pl <= pt && pl <= ptl) ? lft : (pt <= ptl ? t : tl);
        } else if (MODE == IM_Z1) {
            const int x = dx * (r + 1);
            const int base = (x >> (6 - up_above)) + (c << up_above);
            const int sh = ((x << up_above) & 0x3f) >> 1;
            const int max_base = (bw + bh - 1) << up_above;
            if (base < max_base) v = min(max((above[base] * (32 - sh) + above[base + 1] * sh + 16) >> 5, 0), maxv);
            else v = above[max_base];
        } else if (MODE == IM_Z3) {
            const int y = dy * (c + 1);
            const int base = (y >> (6 - up_left)) + (r << up_left);
            const int sh = ((y << up_left) & 0x3f) >> 1;
            const int max_base = (bw + bh - 1) << up_left;
            if (base < max_base) v = min(max((left[base] * (32 - sh) + left[base + 1] * sh + 16) >> 5, 0), maxv);
            else v = left[max_base];
        } else if (MODE == IM_Z2) {
            const int x = -dx * (r + 1);
            const int base1 = (x >> (6 - up_above)) + (c << up_above);
            if (base1 >= -(1 << up_above)) {
                const int s1 = ((x * (1 << up_above)) & 0x3f) >> 1;
                v = (above[base1] * (32 - s1) + above[base1 + 1] * s1 + 16) >> 5;
            } else {
                const int y = (r << 6) - dy * (c + 1);
                const int base2 = y >> (6 - up_left);
                const int s2 = ((y * (1 << up_left)) & 0x3f) >> 1;
                v = (left[base2] * (32 - s2) + left[base2 + 1] * s2 + 16) >> 5;
            }
            v = min(max(v, 0), maxv);
        } else v = dc;
        out[k] = (PixT)v;
    }
}

// Work split: a block is covered by LB = bw*bh/ppl consecutive lanes (ppl = pixels per lane =
// min(bw, 16 B / sizeof(PixT))), i.e. one lane writes 16 B (or a whole narrow row) of one output
// row.  DC sums are formed cooperatively by the lanes of a block (strided partial sums + a
// cross-lane reduction), and each lane fetches its `above` segment with one wide unaligned load.
template <typename PixT, int MODE>
__global__ __launch_bounds__(256) void intra_pred_kernel(
    PixT* __restrict__ dst, int32_t dst_stride, size_t dst_block_pitch, const uint32_t* __restrict__ dst_offsets,
    const PixT* __restrict__ above_all, const PixT* __restrict__ left_all, int32_t nb_pitch, int mode_rt, int bw,
    int bh, int up_above, int up_left, int dx, int dy, int bd, uint32_t nblocks) {
    constexpr int PXL = 16 / (int)sizeof(PixT);            // pixels per lane when the block is wide enough
    const int ppl = bw < PXL ? bw : PXL;                   // pixels per lane
    const int lanes_per_row = bw / ppl;
    const uint32_t per_block = (uint32_t)(lanes_per_row * bh);      // LB: power of two, 4 .. 512
    const size_t total = (size_t)per_block * nblocks;
    const int maxv = (1 << bd) - 1;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    // the grid is a multiple of per_block (host), so every lane keeps its (row, column) and only
    // the block index advances: no per-iteration division
    const size_t item0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int pb_shift = __builtin_ctz(per_block);                  // per_block is a power of two
    const uint32_t j = (uint32_t)(item0 & (per_block - 1));
    const int r = (int)(j / lanes_per_row), c0 = (int)(j % lanes_per_row) * ppl;
    const int wh = kSmWeights[bh + r];
    constexpr int mode = MODE;
    (void)mode_rt;
    for (size_t item = item0; item < total; item += nthreads) {
        const uint32_t blk = (uint32_t)(item >> pb_shift);
        const PixT* above = above_all + (size_t)blk * nb_pitch + NB_ORIGIN;
        const PixT* left = left_all + (size_t)blk * nb_pitch + NB_ORIGIN;
        int dc = 0;
        if (mode == IM_DC || mode == IM_DC_TOP || mode == IM_DC_LEFT) {
            // cooperative sum: lane j of the block adds samples j, j+LB, ... then the block's lanes
            // reduce (LB <= 64: shuffles inside the wave; LB > 64: every wave sums everything itself)
            const int na = mode != IM_DC_LEFT ? bw : 0, nl = mode != IM_DC_TOP ? bh : 0;
            const int grp = per_block < 64 ? (int)per_block : 64;
            const int lj = (int)(j % (uint32_t)grp);
            int sum = 0;
            for (int i = lj; i < na; i += grp) sum += above[i];
            for (int i = lj; i < nl; i += grp) sum += left[i];
            for (int m = grp >> 1; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
            const int cnt = na + nl;
            dc = (sum + (cnt >> 1)) / cnt;                 // exact division (EbIntraPrediction.c:1880-1896)
        } else if (mode == IM_DC_128) {
            dc = 128 << (bd - 8);
        }
        PixT out[PXL];
        // MODE is a template parameter: one kernel per mode keeps the register footprint of the
        // gather-heavy directional bodies away from the simple ones
        intra_row<PixT, MODE>(out, above, left, r, c0, ppl, bw, bh, wh, dc, up_above, up_left, dx, dy, maxv);
        const size_t base_off = dst_offsets ? (size_t)dst_offsets[blk] : (size_t)blk * dst_block_pitch;
        PixT* d = dst + base_off + (size_t)r * dst_stride + c0;
        if (ppl == PXL && ((reinterpret_cast<uintptr_t>(d) & 15) == 0)) {
            *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(out);
        } else {
#pragma unroll
            for (int k = 0; k < PXL; k++) if (k < ppl) d[k] = out[k];
        }
    }
}

// av1_filter_intra_edge(_high) (:3539) — out-of-place on the device: every output
// sample depends only on the unfiltered edge, so all samples go in parallel.
// edge block b: p = edges + b*pitch + origin; p[1..sz-1] filtered, p[0] kept.
template <typename PixT>
__global__ __launch_bounds__(256) void filter_edge_kernel(PixT* __restrict__ edges, int32_t pitch, int32_t origin, int sz,
                                                          int strength, uint32_t nblocks) {
    __shared__ PixT sh[256];
    // one workgroup handles 256/128 = up to 1 edge of <=129 samples per 129-thread slice: keep it simple,
    // one edge per workgroup (edges are tiny; this kernel is never the bottleneck)
    const uint32_t blk = blockIdx.x;
    if (blk >= nblocks || strength == 0) return;
    PixT* p = edges + (size_t)blk * pitch + origin;
    const int i = threadIdx.x;
    if (i < sz) sh[i] = p[i];
    __syncthreads();
    if (i >= 1 && i < sz) {
        const int k0 = strength == 3 ? 2 : 0, k1 = strength == 1 ? 4 : (strength == 2 ? 5 : 4),
                  k2 = strength == 1 ? 8 : (strength == 2 ? 6 : 4);
        auto at = [&](int q) { q = q < 0 ? 0 : (q > sz - 1 ? sz - 1 : q); return (int)sh[q]; };
        const int s = k0 * at(i - 2) + k1 * at(i - 1) + k2 * at(i) + k1 * at(i + 1) + k0 * at(i + 2);
        p[i] = (PixT)((s + 8) >> 4);
    }
}

// av1_upsample_intra_edge(_high) (:3597): p[-2 .. 2*sz-2] from p[-1 .. sz-1]
template <typename PixT>
__global__ __launch_bounds__(64) void upsample_edge_kernel(PixT* __restrict__ edges, int32_t pitch, int32_t origin, int sz,
                                                           int bd, uint32_t nblocks) {
    __shared__ int in[16 + 3];
    const uint32_t blk = blockIdx.x;
    if (blk >= nblocks) return;
    PixT* p = edges + (size_t)blk * pitch + origin;
    const int i = threadIdx.x;
    if (i < sz + 3) in[i] = (i < 2) ? p[-1] : (i < sz + 2 ? p[i - 2] : p[sz - 1]);
    __syncthreads();
    const int maxv = (1 << bd) - 1;
    if (i == 0) p[-2] = (PixT)in[0];
    if (i < sz) {
        int s = -in[i] + 9 * in[i + 1] + 9 * in[i + 2] - in[i + 3];
        s = min(max((s + 8) >> 4, 0), maxv);
        p[2 * i - 1] = (PixT)s;
        p[2 * i] = (PixT)in[i + 2];
    }
}

// full_distortion_kernel32_bits / _cbf_zero32_bits (EbPictureOperators.c:283-346):
// out[blk][0] = sum (c - r)^2 (or sum c^2 when cbf_zero), out[blk][1] = sum c^2.
// 16 lanes per block, 4 blocks per wave.
__global__ __launch_bounds__(256) void full_distortion32_kernel(
    const int32_t* __restrict__ coeff, uint32_t coeff_stride, size_t coeff_block_pitch,
    const int32_t* __restrict__ recon, uint32_t recon_stride, size_t recon_block_pitch, uint32_t width,
    uint32_t height, int cbf_zero, unsigned long long* __restrict__ out, uint32_t nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4, l = lane & 15;
    const uint32_t blk = (blockIdx.x * 4 + wave) * 4 + sub;
    const bool valid = blk < nblocks;
    unsigned long long resid = 0, pred = 0;
    if (valid) {
        const int32_t* pc = coeff + (size_t)blk * coeff_block_pitch;
        const int32_t* pr = cbf_zero ? nullptr : recon + (size_t)blk * recon_block_pitch;
        const uint32_t total = width * height;
        for (uint32_t i = l; i < total; i += 16) {
            const uint32_t y = i / width, x = i - y * width;
            const long long c = pc[(size_t)y * coeff_stride + x];
            pred += (unsigned long long)(c * c);
            if (!cbf_zero) {
                const long long d = c - (long long)pr[(size_t)y * recon_stride + x];
                resid += (unsigned long long)(d * d);
            }
        }
    }
    resid = group_sum64<16>(resid);
    pred = group_sum64<16>(pred);
    if (valid && l == 0) {
        out[(size_t)blk * 2 + 0] = cbf_zero ? pred : resid;
        out[(size_t)blk * 2 + 1] = pred;
    }
}

}  // namespace svtdev
