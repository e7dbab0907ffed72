// kernel_me.h — K6: ME multi-size SAD of 64x64 superblocks over a full-pel search
// area, all 85 PUs (64 8x8 + 16 16x16 + 4 32x32 + 1 64x64) at once.
//
// Reference: FullPelSearch_LCU (EbMotionEstimation.c:3199-3247) ->
// GetSearchPointResults (:2932-3057) -> ext_sad_calculation_8x8_16x16 (:208-262,
// 8x8 SADs on every other row, doubled) + ext_sad_calculation_32x32_64x64
// (:267-311); per-PU running best with strict '<' in raster search order and the
// packed MV ((uint16)y << 18) | (uint16)(x << 2).
//
// Mapping: one workgroup (4 waves) per superblock.  The even source rows (32 x 64 B)
// and the whole reference window live in LDS; lane t evaluates search points
// t, t+256, ...  For one search point a lane walks the 8 bands of 8 rows: 4 even
// rows x 16 dwords of v_sad_u8 against reference dwords rebuilt with v_alignbyte
// into 8 accumulators (= the band's eight 8x8 SADs), folding them on the fly into
// 16x16 / 32x32 / 64x64 sums.  Every PU keeps one packed 32-bit key
// (sad << 12 | search-point index) per lane, so "first strict minimum" is a plain
// unsigned min; the 85 keys are min-reduced over the workgroup at the end.
// Limits: search_w * search_h <= 4096 (12-bit index), window must fit 64 KiB LDS.
#pragma once
#include "dev_common.h"

namespace svtdev {

constexpr int ME_THREADS = 256;
constexpr int ME_PUS = 85;

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, m, 64));
    return v;
}

__global__ __launch_bounds__(ME_THREADS) void me_sb_search_kernel(
    const uint8_t* __restrict__ src, uint32_t src_stride, size_t src_block_pitch,
    const uint8_t* __restrict__ ref, uint32_t ref_stride, size_t ref_block_pitch, int search_w, int search_h,
    const int16_t* __restrict__ origins /* [n][2] x,y or NULL */, int x_origin, int y_origin,
    uint32_t* __restrict__ best_sad, uint32_t* __restrict__ best_mv, uint32_t wpitch, uint32_t nblocks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* s_src = reinterpret_cast<uint32_t*>(smem);            // [32 even rows][16 dwords]
    uint8_t* s_ref = smem + 32 * 64;                                // [(64+sh-1)][wpitch]
    __shared__ unsigned s_red[4][ME_PUS];
    const uint32_t blk = blockIdx.x;
    if (blk >= nblocks) return;
    const int tid = threadIdx.x;
    const uint8_t* gs = src + (size_t)blk * src_block_pitch;
    const uint8_t* gr = ref + (size_t)blk * ref_block_pitch;
    // stage even source rows (dword loads when aligned, else bytes)
    for (int i = tid; i < 32 * 64; i += ME_THREADS) {
        const int r = i >> 6, c = i & 63;
        reinterpret_cast<uint8_t*>(s_src)[i] = gs[(size_t)(2 * r) * src_stride + c];
    }
    const uint32_t win_w = 64 + search_w - 1, win_h = 64 + search_h - 1;
    for (uint32_t i = tid; i < wpitch * win_h; i += ME_THREADS) {
        const uint32_t y = i / wpitch, x = i - y * wpitch;
        s_ref[i] = x < win_w ? gr[(size_t)y * ref_stride + x] : 0;
    }
    __syncthreads();

    unsigned best[ME_PUS];
#pragma unroll
    for (int i = 0; i < ME_PUS; i++) best[i] = 0xffffffffu;

    const int ncand = search_w * search_h;
    for (int cand = tid; cand < ncand; cand += ME_THREADS) {
        const int ys = cand / search_w, xs = cand - ys * search_w;
        const unsigned sh = (unsigned)(xs & 3);
        const uint8_t* rbase = s_ref + (size_t)ys * wpitch + (xs & ~3);
        unsigned s32acc[4] = {0, 0, 0, 0};
        unsigned s16acc[4] = {0, 0, 0, 0};   // the four 16x16 of the current 16-row band
#pragma unroll
        for (int band = 0; band < 8; band++) {          // 8-row band = one row of 8x8 blocks
            unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int r = band * 8 + rr * 2;
                const uint32_t* rrow = reinterpret_cast<const uint32_t*>(rbase + (size_t)r * wpitch);
                const uint32_t* srow = s_src + (r >> 1) * 16;
                uint32_t lo = rrow[0];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const uint32_t hi = rrow[q + 1];
                    const uint32_t rv = __builtin_amdgcn_alignbyte(hi, lo, sh);
                    acc[q >> 1] = __builtin_amdgcn_sad_u8(srow[q], rv, acc[q >> 1]);
                    lo = hi;
                }
            }
            const int by16 = band >> 1, kr = band & 1;
#pragma unroll
            for (int bx = 0; bx < 8; bx++) {
                const unsigned s = acc[bx] << 1;
                const int bx16 = bx >> 1;
                const int z = ((by16 >> 1) * 2 + (bx16 >> 1)) * 4 + (by16 & 1) * 2 + (bx16 & 1);
                const int idx = 4 * z + kr * 2 + (bx & 1);
                best[idx] = min(best[idx], (s << 12) | (unsigned)cand);
                s16acc[bx16] += s;
            }
            if (kr == 1) {
#pragma unroll
                for (int bx16 = 0; bx16 < 4; bx16++) {
                    const int z = ((by16 >> 1) * 2 + (bx16 >> 1)) * 4 + (by16 & 1) * 2 + (bx16 & 1);
                    best[64 + z] = min(best[64 + z], (s16acc[bx16] << 12) | (unsigned)cand);
                    s32acc[(by16 >> 1) * 2 + (bx16 >> 1)] += s16acc[bx16];
                    s16acc[bx16] = 0;
                }
            }
        }
        unsigned s64 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            best[80 + q] = min(best[80 + q], (s32acc[q] << 12) | (unsigned)cand);
            s64 += s32acc[q];
        }
        best[84] = min(best[84], (s64 << 12) | (unsigned)cand);
    }
    // workgroup min-reduction of the 85 keys
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < ME_PUS; i++) {
        const unsigned v = wave_min_u32(best[i]);
        if (lane == 0) s_red[wave][i] = v;
    }
    __syncthreads();
    if (tid < ME_PUS && ncand > 0) {
        const unsigned key = min(min(s_red[0][tid], s_red[1][tid]), min(s_red[2][tid], s_red[3][tid]));
        const unsigned sad = key >> 12, cand = key & 0xfffu;
        const int ys = (int)cand / search_w, xs = (int)cand - ys * search_w;
        const int ox = origins ? origins[2 * blk] : x_origin, oy = origins ? origins[2 * blk + 1] : y_origin;
        uint32_t* bs = best_sad + (size_t)blk * ME_PUS;
        uint32_t* bm = best_mv + (size_t)blk * ME_PUS;
        if (sad < bs[tid]) {
            bs[tid] = sad;
            bm[tid] = (((uint32_t)(uint16_t)(ys + oy)) << 18) | (uint32_t)(uint16_t)((xs + ox) << 2);
        }
    }
}

}  // namespace svtdev
