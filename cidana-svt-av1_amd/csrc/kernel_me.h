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
// and the whole reference window live in LDS; a lane evaluates groups of FOUR
// horizontally adjacent search points with v_qsad_pk_u16_u8 (aligned reference dword
// pairs, 4 points x 4 pixels per instruction), walking the 8 bands of 8 rows: 4 even
// rows x 16 dwords into 8 packed accumulators (= the band's eight 8x8 SADs of the 4
// points), folded on the fly into 16x16 (still packed u16) and 32x32 / 64x64 (u32)
// sums.  Every PU keeps one packed 32-bit key (sad << 12 | search-point index) per
// lane, so "first strict minimum" is a plain unsigned min; the 85 keys are
// min-reduced over the workgroup at the end.
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

__device__ __forceinline__ unsigned wave_min_u32_to_lane63(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, true));   // row_half_mirror
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, true));   // row_mirror
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast15 -> rows 1, 3
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast31 -> rows 2, 3
    return v;
}

template <bool MASKED>
__device__ __forceinline__ unsigned me_key16_min(const unsigned long long (&a)[4], unsigned idb, unsigned nvalid) {
    // a[g] = packed SADs of points 4g .. 4g+3; returns min over the lane's first `nvalid` points (all 16 unless
    // MASKED: search widths that are not a multiple of 16) of (sad << 16 | idb + point)
    unsigned best = 0xffffffffu;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const unsigned lo = (unsigned)a[g], hi = (unsigned)(a[g] >> 32);
        unsigned k0 = ((lo << 16) | idb) + (4 * g + 0), k1 = ((lo & 0xffff0000u) | idb) + (4 * g + 1);
        unsigned k2 = ((hi << 16) | idb) + (4 * g + 2), k3 = ((hi & 0xffff0000u) | idb) + (4 * g + 3);
        if (MASKED) {
            k0 = 4 * g + 0 < nvalid ? k0 : 0xffffffffu; k1 = 4 * g + 1 < nvalid ? k1 : 0xffffffffu;
            k2 = 4 * g + 2 < nvalid ? k2 : 0xffffffffu; k3 = 4 * g + 3 < nvalid ? k3 : 0xffffffffu;
        }
        best = min(best, min(k0, k1));
        best = min(best, min(k2, k3));
    }
    return best;
}
__device__ __forceinline__ unsigned long long me_pk_add(unsigned long long a, unsigned long long b) {
    // lane-wise u16 add of two packed words whose lane sums stay below 2^16: two independent 32-bit adds
    const unsigned lo = (unsigned)a + (unsigned)b, hi = (unsigned)(a >> 32) + (unsigned)(b >> 32);
    return ((unsigned long long)hi << 32) | lo;
}

// MASKED: the search width is not a multiple of 16; the last 16-point group of a row is partly outside the area.
// The search of one SB by one workgroup (body shared by me_sb_search16_kernel and me_fullpel_areas_kernel): gs / gr = the SB's
// source block and the top-left sample of its search window, bs / bm = its result rows, (ox, oy) = the search area's origin.
template <bool MASKED>
__device__ __forceinline__ void me_sb_search16_body(
    uint8_t* smem, const uint8_t* __restrict__ gs, uint32_t src_stride, const uint8_t* __restrict__ gr, uint32_t ref_stride,
    int search_w, int search_h, int ox, int oy, uint32_t* __restrict__ bs, uint32_t* __restrict__ bm, uint32_t wpitch,
    // w8q > 0: SVT_HIP_FLAVOUR_AVX2 - inside the full groups of eight search points of a row (xs < w8q = search_w & ~7) the four
    // 32x32 PUs rank and report point p of the group as p ^ 4 (see me_fullpel_exact_kernel).  ref_layout: results in the
    // reference's EbMeTierZeroPu order instead of 8x8 | 16x16 | 32x32 | 64x64 back to back.
    int w8q, int ref_layout) {
    uint32_t* s_src = reinterpret_cast<uint32_t*>(smem);            // [32 even rows][16 dwords]
    uint8_t* s_ref = smem + 32 * 64;                                // [(64+sh-1)][wpitch], wpitch % 16 == 0
    __shared__ unsigned s_red[4][ME_PUS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- stage the even source rows and the reference window (16-B unaligned loads, 4 in flight) ----
    if (tid < 128) {
        const int r = tid >> 2, c = tid & 3;
        uint4 v;
        __builtin_memcpy(&v, gs + (size_t)(2 * r) * src_stride + c * 16, 16);
        reinterpret_cast<uint4*>(s_src)[tid] = v;
    }
    for (int i = tid; i < 4 * ME_PUS; i += ME_THREADS) (&s_red[0][0])[i] = 0xffffffffu;
    const uint32_t win_w = 64 + search_w - 1, win_h = 64 + search_h - 1;
    {
        const uint32_t cpr = (win_w + 15) >> 4;
        const size_t span = (size_t)(win_h - 1) * ref_stride + win_w;
        for (uint32_t c = tid & 15; c < cpr; c += 16)
            for (uint32_t y0 = tid >> 4; y0 < win_h; y0 += 64) {
                uint4 v[4];
                uint32_t back[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t y = y0 + 16 * k;
                    v[k] = make_uint4(0, 0, 0, 0);
                    back[k] = 0;
                    if (y < win_h) {
                        const size_t off = (size_t)y * ref_stride + c * 16;
                        if (off + 16 <= span) __builtin_memcpy(&v[k], gr + off, 16);
                        else if (off < span) {           // footprint tail: the last 16 bytes of the window's footprint, stored `back` bytes earlier
                            back[k] = (uint32_t)(off - (span - 16));
                            __builtin_memcpy(&v[k], gr + (span - 16), 16);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t y = y0 + 16 * k;
                    if (y < win_h) {
                        if (back[k] == 0) *reinterpret_cast<uint4*>(s_ref + (size_t)y * wpitch + c * 16) = v[k];
                        else {
                            const uint8_t* vb = reinterpret_cast<const uint8_t*>(&v[k]);
                            uint8_t* d = s_ref + (size_t)y * wpitch + c * 16 - back[k];
#pragma unroll
                            for (int b = 0; b < 16; b++) d[b] = vb[b];
                        }
                    }
                }
            }
    }
    __syncthreads();

    const int xqn = (search_w + 15) >> 4;
    const int ntasks = xqn * search_h;
    for (int t0 = 0; t0 < ntasks; t0 += ME_THREADS) {
        const int t = t0 + tid;
        const bool act = t < ntasks;
        const int tc = act ? t : 0;
        const int ys = tc / xqn, xs0 = (tc - ys * xqn) * 16;
        const unsigned idb = (unsigned)(ys * search_w + xs0);         // point index of the lane's first point (< 4096)
        const unsigned nvalid = MASKED ? (unsigned)min(16, search_w - xs0) : 16u;
        const unsigned dead = act ? 0u : 0xffffffffu;
        const unsigned q4[2] = {xs0 < w8q ? 4u : 0u, xs0 + 8 < w8q ? 4u : 0u};      // per group of eight of the lane's 16 points
        const uint8_t* rbase = s_ref + (size_t)ys * wpitch + xs0;
        unsigned s64[16];
#pragma unroll
        for (int i = 0; i < 16; i++) s64[i] = 0;
#pragma unroll 1
        for (int h32 = 0; h32 < 2; h32++) {                           // top / bottom 32 rows of the SB
            unsigned long long PA[2][4];                              // 32x16 sums of the upper 16 rows: 32x32 column c32, point group
#pragma unroll
            for (int h16 = 0; h16 < 2; h16++) {
                // A 16-row band is walked as two 32-column halves (hq = the 32x32 column), each as two 8-row bands: 4 x 4 packed
                // accumulators, 4 x 2 16x16 sums and 8 + 12 operand dwords are live at a time (the first version kept 4 x 8, 4 x 4 and
                // 16 + 20 and spilled 69 dwords per lane to scratch at its 256-register budget).  The third reference chunk of the left
                // half is read again by the right half (6 instead of 5 b128 reads per row: the LDS pipe has the room, the kernel is
                // bound by v_qsad issue).
#pragma unroll
                for (int hq = 0; hq < 2; hq++) {
                    unsigned long long s16[4][2];                     // [point group][16x16 column inside the half]
#pragma unroll
                    for (int kb = 0; kb < 2; kb++) {
                        const int band_in = h16 * 2 + kb;             // band inside the 32-row half (compile time)
                        unsigned long long acc[4][4];
#pragma unroll
                        for (int g = 0; g < 4; g++)
#pragma unroll
                            for (int bx = 0; bx < 4; bx++) acc[g][bx] = 0;
#pragma unroll 2
                        for (int rr = 0; rr < 4; rr++) {
                            const int row = band_in * 8 + rr * 2;     // SB row inside the half
                            const uint4* sp = reinterpret_cast<const uint4*>(s_src + (h32 * 16 + (row >> 1)) * 16) + 2 * hq;
                            const uint4* rp = reinterpret_cast<const uint4*>(rbase + (size_t)(h32 * 32 + row) * wpitch) + 2 * hq;
                            uint32_t sw[8], rw[12];
#pragma unroll
                            for (int i = 0; i < 2; i++) { const uint4 a = sp[i]; sw[4 * i] = a.x; sw[4 * i + 1] = a.y; sw[4 * i + 2] = a.z; sw[4 * i + 3] = a.w; }
#pragma unroll
                            for (int i = 0; i < 3; i++) { const uint4 a = rp[i]; rw[4 * i] = a.x; rw[4 * i + 1] = a.y; rw[4 * i + 2] = a.z; rw[4 * i + 3] = a.w; }
                            unsigned long long pr[11];                // dword pairs (d, d+1): odd d costs one register copy, shared by 4 qsads
#pragma unroll
                            for (int d = 0; d < 11; d++) pr[d] = ((unsigned long long)rw[d + 1] << 32) | rw[d];
#pragma unroll
                            for (int g = 0; g < 4; g++)
#pragma unroll
                                for (int q = 0; q < 8; q++)
                                    acc[g][q >> 1] = __builtin_amdgcn_qsad_pk_u16_u8(pr[g + q], sw[q], acc[g][q >> 1]);
                        }
                        // ---- 8x8 PUs of this half band: 16 points -> one key per PU -> wave -> LDS table ----
#pragma unroll
                        for (int bxl = 0; bxl < 4; bxl++) {
                            const int bx = 4 * hq + bxl;
                            const unsigned long long a4[4] = {acc[0][bxl], acc[1][bxl], acc[2][bxl], acc[3][bxl]};
                            const unsigned k = wave_min_u32_to_lane63(me_key16_min<MASKED>(a4, idb, nvalid) | dead);
                            const int bx16 = bx >> 1;
                            const int zc = (bx16 >> 1) * 4 + h16 * 2 + (bx16 & 1);           // z-order inside the half
                            const int idx = 32 * h32 + 4 * zc + kb * 2 + (bx & 1);
                            if (lane == 63) s_red[wave][idx] = min(s_red[wave][idx], k);
                        }
#pragma unroll
                        for (int g = 0; g < 4; g++)
#pragma unroll
                            for (int cl = 0; cl < 2; cl++) {
                                const unsigned long long v = me_pk_add(acc[g][2 * cl], acc[g][2 * cl + 1]);
                                s16[g][cl] = kb == 0 ? v : me_pk_add(s16[g][cl], v);
                            }
                    }
                    // ---- the two 16x16 PUs of this half, and its 32x16 sum ----
#pragma unroll
                    for (int cl = 0; cl < 2; cl++) {
                        const int c16 = 2 * hq + cl;
                        const unsigned long long a4[4] = {s16[0][cl], s16[1][cl], s16[2][cl], s16[3][cl]};
                        const unsigned k = wave_min_u32_to_lane63(me_key16_min<MASKED>(a4, idb, nvalid) | dead);
                        const int zc = (c16 >> 1) * 4 + h16 * 2 + (c16 & 1);
                        const int idx = 64 + 8 * h32 + zc;
                        if (lane == 63) s_red[wave][idx] = min(s_red[wave][idx], k);
                    }
                    if (h16 == 0) {
#pragma unroll
                        for (int g = 0; g < 4; g++) PA[hq][g] = me_pk_add(s16[g][0], s16[g][1]);      // 32x16, <= 65 280 per lane
                    } else {
                        // ---- the 32x32 PU of this half (c32 = hq): widen, double (SADs are on every other row), key = sad << 12 | point ----
                        unsigned best = 0xffffffffu;
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            const unsigned long long pb = me_pk_add(s16[g][0], s16[g][1]);
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) {
                                const unsigned a = (unsigned)((PA[hq][g] >> (16 * jj)) & 0xffffu), b = (unsigned)((pb >> (16 * jj)) & 0xffffu);
                                const unsigned sd = (a + b) << 1;
                                s64[4 * g + jj] += sd;
                                const unsigned key = (sd << 12) | (idb + ((unsigned)(4 * g + jj) ^ q4[g >> 1]));
                                best = min(best, (MASKED && (unsigned)(4 * g + jj) >= nvalid) ? 0xffffffffu : key);
                            }
                        }
                        const unsigned k = wave_min_u32_to_lane63(best | dead);
                        const int idx = 80 + 2 * h32 + hq;
                        if (lane == 63) s_red[wave][idx] = min(s_red[wave][idx], k);
                    }
                }
            }
        }
        {
            unsigned best = 0xffffffffu;
#pragma unroll
            for (int i = 0; i < 16; i++) best = min(best, (MASKED && (unsigned)i >= nvalid) ? 0xffffffffu : (s64[i] << 12) | (idb + i));
            const unsigned k = wave_min_u32_to_lane63(best | dead);
            if (lane == 63) s_red[wave][84] = min(s_red[wave][84], k);
        }
    }
    __syncthreads();
    if (tid < ME_PUS) {
        const unsigned key = min(min(s_red[0][tid], s_red[1][tid]), min(s_red[2][tid], s_red[3][tid]));
        // PUs 0..79 (8x8, 16x16): sad16 << 16 | point, SAD still to be doubled; 80..84: (2*sad) << 12 | point
        const unsigned sad = tid < 80 ? (key >> 16) << 1 : key >> 12;
        const unsigned cand = key & 0xfffu;
        const int ys = (int)cand / search_w, xs = (int)cand - ys * search_w;
        // legacy order 8x8 [0..63] | 16x16 [64..79] | 32x32 [80..83] | 64x64 [84] -> EbMeTierZeroPu order
        const int o = !ref_layout ? tid : (tid < 64 ? 21 + tid : (tid < 80 ? 5 + (tid - 64) : (tid < 84 ? 1 + (tid - 80) : 0)));
        if (sad < bs[o]) {
            bs[o] = sad;
            bm[o] = (((uint32_t)(uint16_t)(ys + oy)) << 18) | (uint32_t)(uint16_t)((xs + ox) << 2);
        }
    }
}

template <bool MASKED>
__global__ __launch_bounds__(ME_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void me_sb_search16_kernel(
    const uint8_t* __restrict__ src, uint32_t src_stride, size_t src_block_pitch,
    const uint8_t* __restrict__ ref, uint32_t ref_stride, size_t ref_block_pitch, int search_w, int search_h,
    const int16_t* __restrict__ origins /* [n][2] x,y or NULL */, int x_origin, int y_origin,
    uint32_t* __restrict__ best_sad, uint32_t* __restrict__ best_mv, uint32_t wpitch,
    const uint32_t* __restrict__ src_offs, const uint32_t* __restrict__ ref_offs, uint32_t nblocks,
    int w8q = 0, int ref_layout = 0, uint32_t pu_pitch = ME_PUS) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t blk = blockIdx.x;
    if (blk >= nblocks) return;
    const uint8_t* gs = src + (src_offs ? (size_t)src_offs[blk] : (size_t)blk * src_block_pitch);
    const uint8_t* gr = ref + (ref_offs ? (size_t)ref_offs[blk] : (size_t)blk * ref_block_pitch);
    const int ox = origins ? origins[2 * blk] : x_origin, oy = origins ? origins[2 * blk + 1] : y_origin;
    me_sb_search16_body<MASKED>(smem, gs, src_stride, gr, ref_stride, search_w, search_h, ox, oy, best_sad + (size_t)blk * pu_pitch,
                                best_mv + (size_t)blk * pu_pitch, wpitch, w8q, ref_layout);
}

// ---------------------------------------------------------------------------
// K6 in the reference's own result layout: 209 PUs in EbMeTierZeroPu order (EbMotionEstimationContext.h:47-270),
// i.e. the encoder's p_sb_best_sad / p_sb_best_mv rows: 64x64 [0], 32x32 [1..4], 16x16 [5..20], 8x8 [21..84], and the
// non-square shapes of open_loop_me_fullpel_search_sblock (EbMotionEstimation.c:3251): 64x32 [85..86], 32x16 [87..94],
// 16x8 [95..126], 32x64 [127..128], 16x32 [129..136], 8x16 [137..168], 32x8 [169..184], 8x32 [185..200], 64x16 [201..204],
// 16x64 [205..208].  me_pu_rect gives a PU's rectangle in units of 8 pixels, derived from how ext_eigth_sad_calculation_nsq_c
// (:1455-2490) / ExtSadCalculation (:655-1440) build the sums out of the z-ordered 8x8 / 16x16 / 32x32 SADs.
// ---------------------------------------------------------------------------
constexpr int ME_PUS_ALL = 209;

__host__ __device__ inline void me_pu_rect(int pu, int& x, int& y, int& w, int& h) {
    // z-order index of a 16x16 -> its position in 16-pixel units
    auto z16 = [](int z, int& bx16, int& by16) { bx16 = ((z >> 2) & 1) * 2 + (z & 1); by16 = (z >> 3) * 2 + ((z >> 1) & 1); };
    int bx, by;
    if (pu == 0) { x = 0; y = 0; w = 8; h = 8; }
    else if (pu < 5) { const int q = pu - 1; x = (q & 1) * 4; y = (q >> 1) * 4; w = 4; h = 4; }
    else if (pu < 21) { z16(pu - 5, bx, by); x = bx * 2; y = by * 2; w = 2; h = 2; }
    else if (pu < 85) { const int i = pu - 21; z16(i >> 2, bx, by); x = bx * 2 + (i & 1); y = by * 2 + ((i >> 1) & 1); w = 1; h = 1; }
    else if (pu < 87) { x = 0; y = (pu - 85) * 4; w = 8; h = 4; }                                         // 64x32
    else if (pu < 95) { const int i = pu - 87, q = i >> 1; x = (q & 1) * 4; y = (q >> 1) * 4 + (i & 1) * 2; w = 4; h = 2; }   // 32x16
    else if (pu < 127) { const int i = pu - 95; z16(i >> 1, bx, by); x = bx * 2; y = by * 2 + (i & 1); w = 2; h = 1; }          // 16x8
    else if (pu < 129) { x = (pu - 127) * 4; y = 0; w = 4; h = 8; }                                       // 32x64
    else if (pu < 137) { const int i = pu - 129, q = i >> 1; x = (q & 1) * 4 + (i & 1) * 2; y = (q >> 1) * 4; w = 2; h = 4; }  // 16x32
    else if (pu < 169) { const int i = pu - 137; z16(i >> 1, bx, by); x = bx * 2 + (i & 1); y = by * 2; w = 1; h = 2; }         // 8x16
    else if (pu < 185) { const int i = pu - 169, m = i >> 1, q = m >> 1; x = (q & 1) * 4; y = (q >> 1) * 4 + (m & 1) * 2 + (i & 1); w = 4; h = 1; }  // 32x8
    else if (pu < 201) { const int i = pu - 185, q = i >> 2; x = (q & 1) * 4 + (i & 3); y = (q >> 1) * 4; w = 1; h = 4; }       // 8x32
    else if (pu < 205) { x = 0; y = (pu - 201) * 2; w = 8; h = 2; }                                       // 64x16
    else { x = (pu - 205) * 2; y = 0; w = 2; h = 8; }                                                     // 16x64
}

// ---------------------------------------------------------------------------
// me_nsq4_kernel — all 209 PUs (open_loop_me_fullpel_search_sblock, EbMotionEstimation.c:3251) for search widths that are a
// multiple of 8, where every search point goes through the eight-point form and a PU's result is simply the first strict
// minimum of its SAD over the search points in raster order - both result flavours agree there (the quirks
// me_fullpel_exact_kernel restates live in the single-point form and in the square-PU AVX2 path).
//
// Staging as me_sb_search16_kernel (even source rows + reference window in LDS).  A lane owns FOUR adjacent search points
// (one packed u16 x 4 accumulator per SAD), walks the SB in 8-row bands and folds each band's eight 8x8 SADs into every
// shape as soon as its parts exist:   band: 8x8, 16x8, 32x8 | two bands: 8x16, 16x16, 32x16, 64x16 | 32-row half: 8x32,
// 16x32, 32x32, 64x32 | SB: 16x64, 32x64, 64x64.  Sums that fit 16 bits on every other row (up to 32x16 / 16x32: 65 280) stay
// packed; larger ones are widened per point.  Each PU's four keys (sad << 16 | point, or 2 * sad << 12 | point for the wide
// ones) are min-reduced over the wave sixteen PUs at a time (me_wave_min16 below) into the wave's table; the last step maps the
// table back to the reference's EbMeTierZeroPu order (kNsqLoc).  With four points per lane the reference window costs one dword
// read per v_qsad.
// ---------------------------------------------------------------------------
// ---- batched wave minimum -------------------------------------------------------------------------------------------------
// A PU's result is the minimum of its key over every lane.  One DPP reduction per PU (six dependent v_min_u32_dpp + a lane-63
// table update, 209 times per pass) cost more than the v_qsad work itself.  Sixteen keys are reduced TOGETHER instead, halving
// the number of live registers at every step (a transposing reduction): gfx950's v_permlane32_swap / v_permlane16_swap exchange
// half-waves / odd-even rows between two registers, so  min(swap(a, b))  leaves a's result in one half of the lanes and b's in
// the other; the last two register-halving steps select between two mirrored-DPP minima.  35 VALU instructions per 16 keys
// instead of 96 + 16 table updates; afterwards lane L holds the wave minimum of key (L >> 2) & 15 and lanes L % 4 == 0 fold
// the batch into the wave's table with ONE ds_min_u32.
__device__ __forceinline__ unsigned me_min_swap32(unsigned a, unsigned b) {     // lanes 0..31: min over halves of a; 32..63: of b
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return min((unsigned)r[0], (unsigned)r[1]);
}
__device__ __forceinline__ unsigned me_min_swap16(unsigned a, unsigned b) {     // even rows of 16 lanes: a; odd rows: b
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    return min((unsigned)r[0], (unsigned)r[1]);
}
__device__ __forceinline__ unsigned me_wave_min16(unsigned (&k)[16], bool bit3, bool bit2) {
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = me_min_swap32(k[i], k[i + 8]);           // lane bit 5 picks k[i] / k[i + 8]
#pragma unroll
    for (int i = 0; i < 4; i++) k[i] = me_min_swap16(k[i], k[i + 4]);           // lane bit 4 picks k[i] / k[i + 4]
#pragma unroll
    for (int i = 0; i < 2; i++) {                                                // lane bit 3 picks k[i] / k[i + 2] (row_mirror: i <-> 15 - i)
        const unsigned a = min(k[i], (unsigned)__builtin_amdgcn_update_dpp((int)k[i], (int)k[i], 0x140, 0xF, 0xF, true));
        const unsigned b = min(k[i + 2], (unsigned)__builtin_amdgcn_update_dpp((int)k[i + 2], (int)k[i + 2], 0x140, 0xF, 0xF, true));
        k[i] = bit3 ? b : a;
    }
    {                                                                            // lane bit 2 picks k[0] / k[1] (row_half_mirror)
        const unsigned a = min(k[0], (unsigned)__builtin_amdgcn_update_dpp((int)k[0], (int)k[0], 0x141, 0xF, 0xF, true));
        const unsigned b = min(k[1], (unsigned)__builtin_amdgcn_update_dpp((int)k[1], (int)k[1], 0x141, 0xF, 0xF, true));
        k[0] = bit2 ? b : a;
    }
    unsigned v = k[0];
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    return v;
}

// Batches of me_nsq4_kernel (15 per pass x 16 slots; -1 = unused slot) and where each PU's minimum lands in the wave's table.
// z16: z-order index of the 16x16 at column c16 of 16-row band h16 of half h32 (EbMeTierZeroPu numbering).
constexpr int nsq_z16(int h32, int h16, int c16) { return 8 * h32 + 4 * (c16 >> 1) + 2 * h16 + (c16 & 1); }
constexpr int nsq_band_pu(int h32, int h16, int kb, int s) {          // 8-row band: 8x8 [0..7], 16x8 [8..11], 32x8 [12..13]
    if (s < 8) return 21 + 4 * nsq_z16(h32, h16, s >> 1) + 2 * kb + (s & 1);
    if (s < 12) return 95 + 2 * nsq_z16(h32, h16, s - 8) + kb;
    if (s < 14) return 169 + 4 * (2 * h32 + (s - 12)) + 2 * h16 + kb;
    return -1;
}
constexpr int nsq_row16_pu(int h32, int h16, int s) {                  // 16-row band: 8x16 [0..7], 16x16 [8..11], 32x16 [12..13], 64x16 [14]
    if (s < 8) return 137 + 2 * nsq_z16(h32, h16, s >> 1) + (s & 1);
    if (s < 12) return 5 + nsq_z16(h32, h16, s - 8);
    if (s < 14) return 87 + 2 * (2 * h32 + (s - 12)) + h16;
    if (s == 14) return 201 + 2 * h32 + h16;
    return -1;
}
constexpr int nsq_half_pu(int h32, int s) {                            // 32-row half: 8x32 [0..7], 16x32 [8..11], 32x32 [12..13], 64x32 [14]
    if (s < 8) return 185 + 4 * (2 * h32 + (s >> 2)) + (s & 3);
    if (s < 12) return 129 + 2 * (2 * h32 + ((s - 8) >> 1)) + ((s - 8) & 1);
    if (s < 14) return 1 + 2 * h32 + (s - 12);
    if (s == 14) return 85 + h32;
    return -1;
}
constexpr int nsq_sb_pu(int s) { return s < 4 ? 205 + s : (s < 6 ? 127 + (s - 4) : (s == 6 ? 0 : -1)); }   // 16x64, 32x64, 64x64
constexpr int NSQ_BATCHES = 15;                                        // per half: band (h16, kb) -> 3 * h16 + kb, row16 -> 3 * h16 + 2, half -> 6; SB -> 14
struct NsqLoc { uint8_t v[ME_PUS_ALL]; int filled; };
constexpr NsqLoc make_nsq_loc() {
    NsqLoc t{};
    for (int h32 = 0; h32 < 2; h32++)
        for (int s = 0; s < 16; s++) {
            for (int h16 = 0; h16 < 2; h16++) {
                for (int kb = 0; kb < 2; kb++)
                    if (const int pu = nsq_band_pu(h32, h16, kb, s); pu >= 0) { t.v[pu] = (uint8_t)((7 * h32 + 3 * h16 + kb) * 16 + s); t.filled++; }
                if (const int pu = nsq_row16_pu(h32, h16, s); pu >= 0) { t.v[pu] = (uint8_t)((7 * h32 + 3 * h16 + 2) * 16 + s); t.filled++; }
            }
            if (const int pu = nsq_half_pu(h32, s); pu >= 0) { t.v[pu] = (uint8_t)((7 * h32 + 6) * 16 + s); t.filled++; }
        }
    for (int s = 0; s < 16; s++)
        if (const int pu = nsq_sb_pu(s); pu >= 0) { t.v[pu] = (uint8_t)(14 * 16 + s); t.filled++; }
    return t;
}
constexpr bool nsq_loc_is_a_bijection() {
    const NsqLoc t = make_nsq_loc();
    if (t.filled != ME_PUS_ALL) return false;
    bool seen[NSQ_BATCHES * 16] = {};
    for (int p = 0; p < ME_PUS_ALL; p++) { if (seen[t.v[p]]) return false; seen[t.v[p]] = true; }
    return true;
}
static_assert(nsq_loc_is_a_bijection(), "every PU must own exactly one (batch, slot)");
__device__ constexpr NsqLoc kNsqLoc = make_nsq_loc();

// keys of a lane's four points: sad << 16 | point (packed u16 sums, SAD still to be doubled) or (2 * sad) << 12 | point (wide
// sums); c[j] = point index of the lane's j-th point, or 0xffffffff in a lane without work (the key is then all ones)
__device__ __forceinline__ unsigned me_key4_min(unsigned long long a, const unsigned (&c)[4]) {
    const unsigned lo = (unsigned)a, hi = (unsigned)(a >> 32);
    const unsigned k0 = (lo << 16) | c[0], k1 = (lo & 0xffff0000u) | c[1], k2 = (hi << 16) | c[2], k3 = (hi & 0xffff0000u) | c[3];
    return min(min(k0, k1), min(k2, k3));
}
__device__ __forceinline__ void me_unpack4(unsigned long long a, unsigned (&o)[4]) {
    o[0] = (unsigned)a & 0xffffu; o[1] = ((unsigned)a) >> 16; o[2] = (unsigned)(a >> 32) & 0xffffu; o[3] = (unsigned)(a >> 48);
}
__device__ __forceinline__ unsigned me_bigkey4_min(const unsigned (&s2)[4], const unsigned (&c)[4]) {      // s2 = doubled SADs (< 2^20)
    return min(min((s2[0] << 12) | c[0], (s2[1] << 12) | c[1]), min((s2[2] << 12) | c[2], (s2[3] << 12) | c[3]));
}

// One SB by one workgroup (shared by me_nsq4_kernel and me_fullpel_areas_kernel).  Widths: a multiple of 8 (every point in the
// eight-point form), or BELOW 8 with s_pair != NULL - the C flavour's single-search-point form, which production reaches when
// the clipped search area of a picture-edge SB is narrower than 8 (EbMotionEstimation.c:8016-8021 rounds every other width down
// to a multiple of 8).  There the SADs are the same and every PU still keeps its first strict minimum, except 32x16_5 (index 92):
// ExtSadCalculation (:732-736) tests the stale `sad` of 64x32_1 against its best and then stores sad_32x16[5] - a sequential
// rule, not a minimum.  The lanes leave (64x32_1 SAD, 32x16_5 SAD) of every point in s_pair [search_w * search_h] and one lane
// replays the rule in search order at the end (<= 7 x search_h steps); lanes past the row's end carry all-ones keys.
__device__ __forceinline__ void me_nsq4_body(
    uint8_t* smem, const uint8_t* __restrict__ gs, uint32_t src_stride, const uint8_t* __restrict__ gr, uint32_t ref_stride,
    int search_w, int search_h, int ox, int oy, uint32_t* __restrict__ bs, uint32_t* __restrict__ bm, uint32_t wpitch, uint2* s_pair) {
    uint32_t* s_src = reinterpret_cast<uint32_t*>(smem);            // [32 even rows][16 dwords]
    uint8_t* s_ref = smem + 32 * 64;                                // [(64+sh-1)][wpitch], wpitch % 16 == 0
    __shared__ unsigned s_red[4][NSQ_BATCHES * 16];                 // [wave][batch][slot]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool sp = (search_w & 7) != 0;                            // single-search-point form (search_w < 8)
    if (tid < 128) {
        const int r = tid >> 2, c = tid & 3;
        uint4 v;
        __builtin_memcpy(&v, gs + (size_t)(2 * r) * src_stride + c * 16, 16);
        reinterpret_cast<uint4*>(s_src)[tid] = v;
    }
    for (int i = tid; i < 4 * NSQ_BATCHES * 16; i += ME_THREADS) (&s_red[0][0])[i] = 0xffffffffu;
    const uint32_t win_w = 64 + search_w - 1, win_h = 64 + search_h - 1;
    {
        const uint32_t cpr = (win_w + 15) >> 4;
        const size_t span = (size_t)(win_h - 1) * ref_stride + win_w;
        for (uint32_t c = tid & 15; c < cpr; c += 16)
            for (uint32_t y = tid >> 4; y < win_h; y += 16) {
                const size_t off = (size_t)y * ref_stride + c * 16;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (off + 16 <= span) {
                    __builtin_memcpy(&v, gr + off, 16);
                    *reinterpret_cast<uint4*>(s_ref + (size_t)y * wpitch + c * 16) = v;
                } else if (off < span) {                   // footprint tail: the last 16 bytes, stored earlier 
                    const uint32_t back = (uint32_t)(off - (span - 16));
                    __builtin_memcpy(&v, gr + (span - 16), 16);
                    struct __attribute__((packed, aligned(1))) U4 { uint32_t a, b, c, d; };
                    *reinterpret_cast<U4*>(s_ref + (size_t)y * wpitch + c * 16 - back) = U4{v.x, v.y, v.z, v.w};
                }
            }
    }
    __syncthreads();

    const bool bit3 = (lane & 8) != 0, bit2 = (lane & 4) != 0, writer = (lane & 3) == 0;
    unsigned* my_red = &s_red[wave][(lane >> 2) & 15];
    auto put16 = [&](int batch, unsigned (&k)[16]) {       // wave minima of 16 keys -> this wave's row of the table
        const unsigned v = me_wave_min16(k, bit3, bit2);
        if (writer) __hip_atomic_fetch_min(my_red + batch * 16, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    };
    const int xq = (search_w + 3) >> 2;
    const int ntasks = xq * search_h;
    for (int t0 = 0; t0 < ntasks; t0 += ME_THREADS) {
        const int t = t0 + tid;
        const bool act = t < ntasks;
        const int tc = act ? t : 0;
        const int ys = tc / xq, xs0 = (tc - ys * xq) * 4;
        const unsigned idb = (unsigned)(ys * search_w + xs0);
        const int nv = act ? search_w - xs0 : 0;                       // valid points of the lane (>= 4 except at a narrow row's end)
        const unsigned cj[4] = {nv > 0 ? idb : 0xffffffffu, nv > 1 ? idb + 1u : 0xffffffffu, nv > 2 ? idb + 2u : 0xffffffffu, nv > 3 ? idb + 3u : 0xffffffffu};
        const uint8_t* rbase = s_ref + (size_t)ys * wpitch + xs0;
        unsigned s64[4] = {0, 0, 0, 0};
        unsigned s32top[2][4];
        unsigned long long v16x32top[4];
#pragma unroll 1
        for (int h32 = 0; h32 < 2; h32++) {
            unsigned long long s16h0[4], v8x16h0[8], PA[2], PB[2], v16x32[4];
            unsigned khalf[16];
#pragma unroll
            for (int h16 = 0; h16 < 2; h16++) {
                unsigned long long acc0[8], s16[4], v8x16[8];
                unsigned krow[16];
#pragma unroll
                for (int kb = 0; kb < 2; kb++) {
                    unsigned long long acc[8];
#pragma unroll
                    for (int bx = 0; bx < 8; bx++) acc[bx] = 0;
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        const int row = (h16 * 2 + kb) * 8 + rr * 2;                  // SB row inside the 32-row half
                        const uint4* sp = reinterpret_cast<const uint4*>(s_src + (h32 * 16 + (row >> 1)) * 16);
                        const uint32_t* rp = reinterpret_cast<const uint32_t*>(rbase + (size_t)(h32 * 32 + row) * wpitch);
                        uint32_t sw[16], rw[17];
#pragma unroll
                        for (int i = 0; i < 4; i++) { const uint4 a = sp[i]; sw[4 * i] = a.x; sw[4 * i + 1] = a.y; sw[4 * i + 2] = a.z; sw[4 * i + 3] = a.w; }
#pragma unroll
                        for (int i = 0; i < 17; i++) rw[i] = rp[i];
#pragma unroll
                        for (int q = 0; q < 16; q++)
                            acc[q >> 1] = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)rw[q + 1] << 32) | rw[q], sw[q], acc[q >> 1]);
                    }
                    // ---- this 8-row band: 8x8, 16x8, 32x8 (slots as nsq_band_pu) ----
                    unsigned long long p16x8[4];
                    unsigned kband[16];
#pragma unroll
                    for (int bx = 0; bx < 8; bx++) kband[bx] = me_key4_min(acc[bx], cj);
#pragma unroll
                    for (int c16 = 0; c16 < 4; c16++) {
                        p16x8[c16] = me_pk_add(acc[2 * c16], acc[2 * c16 + 1]);
                        kband[8 + c16] = me_key4_min(p16x8[c16], cj);
                    }
#pragma unroll
                    for (int c32 = 0; c32 < 2; c32++) kband[12 + c32] = me_key4_min(me_pk_add(p16x8[2 * c32], p16x8[2 * c32 + 1]), cj);
                    kband[14] = kband[15] = 0xffffffffu;
                    put16(7 * h32 + 3 * h16 + kb, kband);
                    if (kb == 0) {
#pragma unroll
                        for (int bx = 0; bx < 8; bx++) acc0[bx] = acc[bx];
#pragma unroll
                        for (int c16 = 0; c16 < 4; c16++) s16[c16] = p16x8[c16];
                    } else {
#pragma unroll
                        for (int bx = 0; bx < 8; bx++) {                              // 8x16
                            v8x16[bx] = me_pk_add(acc0[bx], acc[bx]);
                            krow[bx] = me_key4_min(v8x16[bx], cj);
                        }
#pragma unroll
                        for (int c16 = 0; c16 < 4; c16++) s16[c16] = me_pk_add(s16[c16], p16x8[c16]);
                    }
                }
                // ---- this 16-row band: 8x16 (above), 16x16, 32x16, 64x16 (slots as nsq_row16_pu) ----
                unsigned long long p32x16[2];
#pragma unroll
                for (int c16 = 0; c16 < 4; c16++) krow[8 + c16] = me_key4_min(s16[c16], cj);
#pragma unroll
                for (int c32 = 0; c32 < 2; c32++) {
                    p32x16[c32] = me_pk_add(s16[2 * c32], s16[2 * c32 + 1]);         // <= 65 280 per lane
                    krow[12 + c32] = me_key4_min(p32x16[c32], cj);
                }
                {
                    unsigned a[4], b[4], s2[4];
                    me_unpack4(p32x16[0], a); me_unpack4(p32x16[1], b);
#pragma unroll
                    for (int j = 0; j < 4; j++) s2[j] = (a[j] + b[j]) << 1;
                    krow[14] = me_bigkey4_min(s2, cj);
                }
                krow[15] = 0xffffffffu;
                if (h16 == 1 && sp && h32 == 1) krow[12] = 0xffffffffu;               // 32x16_5 in the single-point form: replayed below
                put16(7 * h32 + 3 * h16 + 2, krow);
                if (h16 == 0) {
#pragma unroll
                    for (int c16 = 0; c16 < 4; c16++) s16h0[c16] = s16[c16];
#pragma unroll
                    for (int bx = 0; bx < 8; bx++) v8x16h0[bx] = v8x16[bx];
                    PA[0] = p32x16[0]; PA[1] = p32x16[1];
                } else {
#pragma unroll
                    for (int bx = 0; bx < 8; bx++) khalf[bx] = me_key4_min(me_pk_add(v8x16h0[bx], v8x16[bx]), cj);   // 8x32
#pragma unroll
                    for (int c16 = 0; c16 < 4; c16++) {                               // 16x32 (<= 65 280)
                        v16x32[c16] = me_pk_add(s16h0[c16], s16[c16]);
                        khalf[8 + c16] = me_key4_min(v16x32[c16], cj);
                    }
                    PB[0] = p32x16[0]; PB[1] = p32x16[1];
                }
            }
            // ---- this 32-row half: 8x32, 16x32 (above), 32x32, 64x32 (slots as nsq_half_pu); parts of 32x64, 16x64, 64x64 ----
            unsigned s32[2][4];
#pragma unroll
            for (int c32 = 0; c32 < 2; c32++) {
                unsigned a[4], b[4];
                me_unpack4(PA[c32], a); me_unpack4(PB[c32], b);
#pragma unroll
                for (int j = 0; j < 4; j++) { s32[c32][j] = (a[j] + b[j]) << 1; s64[j] += s32[c32][j]; }
                khalf[12 + c32] = me_bigkey4_min(s32[c32], cj);
            }
            {
                unsigned w[4];
#pragma unroll
                for (int j = 0; j < 4; j++) w[j] = s32[0][j] + s32[1][j];
                khalf[14] = me_bigkey4_min(w, cj);
                if (sp && h32 == 1) {                                                 // (64x32_1, 32x16_5) of the lane's points
                    unsigned b5[4];
                    me_unpack4(PB[0], b5);
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (cj[j] != 0xffffffffu) s_pair[cj[j]] = make_uint2(w[j], b5[j] << 1);
                }
            }
            khalf[15] = 0xffffffffu;
            put16(7 * h32 + 6, khalf);
            if (h32 == 0) {
#pragma unroll
                for (int c32 = 0; c32 < 2; c32++)
#pragma unroll
                    for (int j = 0; j < 4; j++) s32top[c32][j] = s32[c32][j];
#pragma unroll
                for (int c16 = 0; c16 < 4; c16++) v16x32top[c16] = v16x32[c16];
            } else {
                unsigned ksb[16];                                                     // slots as nsq_sb_pu
#pragma unroll
                for (int c16 = 0; c16 < 4; c16++) {                                   // 16x64
                    unsigned a[4], b[4], w[4];
                    me_unpack4(v16x32top[c16], a); me_unpack4(v16x32[c16], b);
#pragma unroll
                    for (int j = 0; j < 4; j++) w[j] = (a[j] + b[j]) << 1;
                    ksb[c16] = me_bigkey4_min(w, cj);
                }
#pragma unroll
                for (int c32 = 0; c32 < 2; c32++) {                                   // 32x64
                    unsigned w[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) w[j] = s32top[c32][j] + s32[c32][j];
                    ksb[4 + c32] = me_bigkey4_min(w, cj);
                }
                ksb[6] = me_bigkey4_min(s64, cj);                                     // 64x64
#pragma unroll
                for (int i = 7; i < 16; i++) ksb[i] = 0xffffffffu;
                put16(14, ksb);
            }
        }
    }
    __syncthreads();
    if (tid < ME_PUS_ALL) {
        const int loc = kNsqLoc.v[tid];
        const unsigned key = min(min(s_red[0][loc], s_red[1][loc]), min(s_red[2][loc], s_red[3][loc]));
        // wide PUs carry (2 * sad) << 12 | point; the others sad16 << 16 | point with the SAD still to be doubled
        const bool wide = tid < 5 || tid == 85 || tid == 86 || tid == 127 || tid == 128 || tid >= 201;
        const unsigned sad = wide ? key >> 12 : (key >> 16) << 1;
        const unsigned cand = key & 0xfffu;
        const int ys = (int)cand / search_w, xs = (int)cand - ys * search_w;
        if (sad < bs[tid] && !(sp && tid == 92)) {         // (32x16_5 in the single-point form has no key: it is replayed below)
            bs[tid] = sad;
            bm[tid] = (((uint32_t)(uint16_t)(ys + oy)) << 18) | (uint32_t)(uint16_t)((xs + ox) << 2);
        }
    }
    if (sp && tid == ME_THREADS - 1) {                     // 32x16_5, single-point form: the reference's rule in search order
        unsigned best = bs[92], mv = bm[92];
        const int np = search_w * search_h;
        for (int p = 0, ys = 0, xs = 0; p < np; p++) {
            const uint2 v = s_pair[p];
            if (v.x < best) { best = v.y; mv = (((uint32_t)(uint16_t)(ys + oy)) << 18) | (uint32_t)(uint16_t)((xs + ox) << 2); }
            if (++xs == search_w) { xs = 0; ys++; }
        }
        bs[92] = best; bm[92] = mv;
    }
}

__global__ __launch_bounds__(ME_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void me_nsq4_kernel(
    const uint8_t* __restrict__ src, uint32_t src_stride, size_t src_block_pitch,
    const uint8_t* __restrict__ ref, uint32_t ref_stride, size_t ref_block_pitch, int search_w, int search_h,
    const int16_t* __restrict__ origins /* [n][2] x,y or NULL */, int x_origin, int y_origin,
    uint32_t* __restrict__ best_sad, uint32_t* __restrict__ best_mv, uint32_t wpitch,
    const uint32_t* __restrict__ src_offs, const uint32_t* __restrict__ ref_offs, uint32_t nblocks, uint32_t pu_pitch) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t blk = blockIdx.x;
    if (blk >= nblocks) return;
    const uint8_t* gs = src + (src_offs ? (size_t)src_offs[blk] : (size_t)blk * src_block_pitch);
    const uint8_t* gr = ref + (ref_offs ? (size_t)ref_offs[blk] : (size_t)blk * ref_block_pitch);
    const int ox = origins ? origins[2 * blk] : x_origin, oy = origins ? origins[2 * blk + 1] : y_origin;
    me_nsq4_body(smem, gs, src_stride, gr, ref_stride, search_w, search_h, ox, oy, best_sad + (size_t)blk * pu_pitch,
                 best_mv + (size_t)blk * pu_pitch, wpitch, nullptr);
}

// me_fullpel_exact_kernel — the reference's search, search point by search point, in its own order: every width, both
// result flavours (0 = its C / SSE4.1 kernels, 1 = what its AVX2 build compiled by GCC / clang computes), square PUs only
// or all 209.  This is the general path behind svt_hip_me_fullpel_search_batch; the fast kernels take the shapes they
// cover (me_sb_search16_kernel: square PUs, any width; me_nsq16_kernel: all PUs, widths that are a multiple of 8).
// One workgroup per SB; a step = the (up to) eight search points of one group of FullPelSearch_LCU's row loop (:3210-3243):
//   phase 1  256 lanes compute the 8 x 64 8x8 SADs of the step (every other row, doubled: Compute8x4SAD_Kernel with
//            2x strides) from the even source rows in LDS and the reference window in global memory / L2;
//   phase 2  lane p < npus owns PU p: its SAD per point is the sum over its rectangle of 8x8 SADs, its running best is
//            updated in the reference's order with strict '<'.
// Restated quirks (oracle/pixel.c holds the same, pinned to the reference by tests/golden/me.npz):
//   * flavour 1, square-PU search, full group of 8: the four 32x32 PUs rank and report the group's point p as p ^ 4
//     (EbComputeSAD_Intrinsic_AVX2.c:3989-4001, the `#ifdef __GNUC__` lane swap);
//   * NSQ search, points outside a full group (single-search-point form): 32x16_5 is replaced when the 64x32_1 SAD (not its
//     own) beats its best (ExtSadCalculation's stale `sad`, EbMotionEstimation.c:732-736) - both flavours;
//   * the same points, flavour 1: the lower 8x8 pair of every 16x16 reads its first reference row 8 SOURCE strides below
//     the 16x16's reference origin (ext_sad_calculation_8x8_16x16_avx2_intrin, EbComputeSAD_Intrinsic_AVX2.c:50-52).
// body: s_src = 2 KiB (even source rows), s_s8 = 2 KiB ([point of the step][8x8 block, raster]) of the caller's LDS
__device__ __forceinline__ void me_exact_body(
    uint32_t* s_src, uint32_t (*s_s8)[64], const uint8_t* __restrict__ gs, uint32_t src_stride, const uint8_t* __restrict__ gr,
    uint32_t ref_stride, int search_w, int search_h, int ox, int oy, int flavour, int nsq, uint32_t* __restrict__ bs_row,
    uint32_t* __restrict__ bm_row) {
    const int tid = threadIdx.x;
    if (tid < 128) {
        const int r = tid >> 2, c = tid & 3;
        uint4 v;
        __builtin_memcpy(&v, gs + (size_t)(2 * r) * src_stride + c * 16, 16);
        reinterpret_cast<uint4*>(s_src)[tid] = v;
    }
    const int npus = nsq ? ME_PUS_ALL : ME_PUS;
    int px = 0, py = 0, pw = 0, ph = 0;
    uint32_t bsad = 0, bmv = 0;
    if (tid < npus) {
        me_pu_rect(tid, px, py, pw, ph);
        bsad = bs_row[tid];
        bmv = bm_row[tid];
    }
    const int w8 = search_w & ~7;
    __syncthreads();
    for (int ys = 0; ys < search_h; ys++) {
        const uint32_t mvy = ((uint32_t)(uint16_t)(ys + oy)) << 18;
        for (int xg = 0; xg < search_w; xg += 8) {
            const int np = min(8, search_w - xg);
            const bool single = xg >= w8;                 // GetSearchPointResults / open_loop_me_get_search_point_results_block
            const bool row_bug = flavour == 1 && nsq && single;
            for (int item = tid; item < np * 64; item += ME_THREADS) {
                const int p = item >> 6, b8 = item & 63, by = b8 >> 3, bx = b8 & 7;
                const uint8_t* r0 = gr + (size_t)ys * ref_stride + (xg + p);
                unsigned sad = 0;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const uint8_t* rrow = r0 + (size_t)(by * 8 + 2 * rr) * ref_stride;
                    if (row_bug && (by & 1) && rr == 0) rrow = r0 + (size_t)((by >> 1) * 16) * ref_stride + (size_t)8 * src_stride;
                    uint2 rv;
                    __builtin_memcpy(&rv, rrow + bx * 8, 8);
                    const uint32_t* sp = s_src + (by * 4 + rr) * 16 + bx * 2;
                    sad = __builtin_amdgcn_sad_u8(sp[0], rv.x, sad);
                    sad = __builtin_amdgcn_sad_u8(sp[1], rv.y, sad);
                }
                s_s8[p][b8] = sad << 1;
            }
            __syncthreads();
            if (tid < npus) {
                const bool swap32 = flavour == 1 && !nsq && !single && tid >= 1 && tid <= 4;
                for (int pa = 0; pa < np; pa++) {
                    const int p = swap32 ? (pa ^ 4) : pa;        // the point whose SAD is attributed to position pa
                    unsigned sum = 0;
                    for (int yy = 0; yy < ph; yy++)
                        for (int xx = 0; xx < pw; xx++) sum += s_s8[p][(py + yy) * 8 + px + xx];
                    unsigned test = sum;
                    if (nsq && single && tid == 87 + 5) {       // the stale `sad` of the 64x32_1 sum
                        test = 0;
                        for (int i = 32; i < 64; i++) test += s_s8[p][i];
                    }
                    if (test < bsad) {
                        bsad = sum;
                        bmv = mvy | (uint32_t)(uint16_t)((xg + pa + ox) << 2);
                    }
                }
            }
            __syncthreads();
        }
    }
    if (tid < npus) {
        bs_row[tid] = bsad;
        bm_row[tid] = bmv;
    }
}

__global__ __launch_bounds__(ME_THREADS) void me_fullpel_exact_kernel(
    const uint8_t* __restrict__ src, uint32_t src_stride, size_t src_block_pitch, const uint32_t* __restrict__ src_offs,
    const uint8_t* __restrict__ ref, uint32_t ref_stride, size_t ref_block_pitch, const uint32_t* __restrict__ ref_offs,
    int search_w, int search_h, const int16_t* __restrict__ origins, int x_origin, int y_origin, int flavour, int nsq,
    uint32_t* __restrict__ best_sad, uint32_t* __restrict__ best_mv, uint32_t pu_pitch, uint32_t nblocks) {
    __shared__ __attribute__((aligned(16))) uint32_t s_src[32 * 16];      // even source rows
    __shared__ uint32_t s_s8[8][64];                                       // [point of the step][8x8 block, raster]
    const uint32_t blk = blockIdx.x;
    if (blk >= nblocks) return;
    const uint8_t* gs = src + (src_offs ? (size_t)src_offs[blk] : (size_t)blk * src_block_pitch);
    const uint8_t* gr = ref + (ref_offs ? (size_t)ref_offs[blk] : (size_t)blk * ref_block_pitch);
    const int ox = origins ? origins[2 * blk] : x_origin, oy = origins ? origins[2 * blk + 1] : y_origin;
    me_exact_body(s_src, s_s8, gs, src_stride, gr, ref_stride, search_w, search_h, ox, oy, flavour, nsq, best_sad + (size_t)blk * pu_pitch,
                  best_mv + (size_t)blk * pu_pitch);
}

// ---------------------------------------------------------------------------
// me_fullpel_areas_kernel — the full-pel search of a batch whose SBs each have their OWN search area (x_origin, y_origin,
// width, height), as svt_hip_me_setup_batch derives them on the device (MotionEstimateLcu clips every SB's area against the
// picture, EbMotionEstimation.c:7955-8040): one launch for a picture's interior and edge SBs alike.  A workgroup reads its
// area, points at its window (the SB's co-located position in the reference plane + the area's origin) and takes the body
// that covers its shape: square search -> me_sb_search16_body (any width, both flavours); all 209 PUs -> me_nsq4_body when the
// width is a multiple of 8, or below 8 in the C flavour; everything else (widths >= 9 that are not a multiple of 8, which the
// reference's rounding never produces, and narrow areas in the AVX2 flavour with its source-stride row fetch) ->
// me_exact_body, whose cost is proportional to the number of search points.  An area with a non-positive size or more than
// max_w x max_h points (what the launch's LDS was sized for) is skipped: the SB's rows keep their incoming values.
// ---------------------------------------------------------------------------
template <bool NSQ>
__global__ __launch_bounds__(ME_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void me_fullpel_areas_kernel(
    const uint8_t* __restrict__ src, uint32_t src_stride, const uint32_t* __restrict__ src_offs, const uint8_t* __restrict__ ref,
    uint32_t ref_stride, const uint32_t* __restrict__ ref_offs, const int16_t* __restrict__ areas /* [n][4] */, int max_w, int max_h,
    int flavour, uint32_t* __restrict__ best_sad, uint32_t* __restrict__ best_mv, uint32_t pu_pitch, uint32_t wpitch, uint32_t pair_off,
    uint32_t nblocks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t blk = blockIdx.x;
    if (blk >= nblocks) return;
    const int ox = areas[4 * blk], oy = areas[4 * blk + 1], sw = areas[4 * blk + 2], sh = areas[4 * blk + 3];
    if (sw < 1 || sh < 1 || sw > max_w || sh > max_h) return;
    const uint8_t* gs = src + (size_t)src_offs[blk];
    const uint8_t* gr = ref + (size_t)ref_offs[blk] + (ptrdiff_t)oy * (ptrdiff_t)ref_stride + ox;
    uint32_t* bs = best_sad + (size_t)blk * pu_pitch;
    uint32_t* bm = best_mv + (size_t)blk * pu_pitch;
    if (!NSQ) {
        me_sb_search16_body<true>(smem, gs, src_stride, gr, ref_stride, sw, sh, ox, oy, bs, bm, wpitch, flavour == 1 ? (sw & ~7) : 0, 1);
    } else if ((sw & 7) == 0 || (sw < 8 && flavour == 0 && pair_off)) {
        me_nsq4_body(smem, gs, src_stride, gr, ref_stride, sw, sh, ox, oy, bs, bm, wpitch, reinterpret_cast<uint2*>(smem + pair_off));
    } else {
        me_exact_body(reinterpret_cast<uint32_t*>(smem), reinterpret_cast<uint32_t(*)[64]>(smem + 2048), gs, src_stride, gr, ref_stride, sw, sh, ox,
                      oy, flavour, 1, bs, bm);
    }
}

// ---------------------------------------------------------------------------
// me_setup_kernel — what MotionEstimateLcu does per SB x reference between the HME levels and the full-pel search
// (EbMotionEstimation.c:7849-8040), for a whole batch: (1) the search centre = the first strict minimum of the last HME level's
// SADs over its search regions in the order the reference visits them, or, for list 1 of a picture whose two references are the
// same picture, the second entry after its sort of the regions (:7906-7936); no HME result is used for SBs that are not 64 rows
// high ("no HME in boundaries", :7678); (2) CheckZeroZeroCenter (:6844-6930): the centre is clipped into the reference picture
// and kept only if the SB's SAD there (every other row, doubled) is strictly below its SAD at (0, 0); (3) the search area: width
// rounded up to 8, centred, clipped left / right / top / bottom in the reference's statement order (its "shrink" statements test
// the corrected origin and never fire), width rounded down to 8 unless below 8.  One wave per task: lanes 0..31 take the rows of
// the (0, 0) SAD, lanes 32..63 those of the HME-centre SAD.
// ---------------------------------------------------------------------------
struct MeSetupParams {      // == svt_hip_me_setup_params (include/svt_hip_dsp.h)
    int32_t picture_width, picture_height, ref_width, ref_height, search_area_width, search_area_height, regions_w, regions_h,
        second_best, zz_check;
};

__global__ __launch_bounds__(256) void me_setup_kernel(
    const uint8_t* __restrict__ src_pic, uint32_t src_stride, const uint8_t* __restrict__ ref_pic, uint32_t ref_stride,
    const int16_t* __restrict__ sb_origin /* [n][2] */, const uint16_t* __restrict__ sb_size /* [n][2] */,
    const unsigned long long* __restrict__ hme_sad /* [regions][n] or NULL */, const int16_t* __restrict__ hme_mv /* [regions][n][2] */,
    const MeSetupParams p, int16_t* __restrict__ center /* [n][2] or NULL */, int16_t* __restrict__ area /* [n][4] */, uint32_t ntasks) {
    const uint32_t task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (task >= ntasks) return;                             // wave-uniform; nothing below synchronises across waves
    const int ox = sb_origin[2 * task], oy = sb_origin[2 * task + 1];
    const int sbw = sb_size[2 * task], sbh = sb_size[2 * task + 1];
    const int nreg = p.regions_w * p.regions_h;
    int xc = 0, yc = 0;
    if (hme_sad && nreg > 0 && sbh == 64) {
        // region r = rh * regions_w + rw: r = 0, 1, 2, 3 is the reference's visiting order [w][h] = [0][0], [1][0], [0][1], [1][1]
        unsigned long long s[4];
        int cx[4], cy[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const bool on = r < nreg;
            s[r] = on ? hme_sad[(size_t)r * ntasks + task] : ~0ull;
            cx[r] = on ? hme_mv[((size_t)r * ntasks + task) * 2] : 0;
            cy[r] = on ? hme_mv[((size_t)r * ntasks + task) * 2 + 1] : 0;
        }
        unsigned long long best = s[0];
        xc = cx[0]; yc = cy[0];
#pragma unroll
        for (int r = 1; r < 4; r++)
            if (r < nreg && s[r] < best) { best = s[r]; xc = cx[r]; yc = cy[r]; }
        if (p.second_best && nreg > 1) {
            // the reference sorts its [width][height] arrays through the index [q / regions_w][q % regions_w] (:7912-7930): entry q
            // is region (rw, rh) = (q / W, q % W), i.e. r = (q % W) * W + q / W (square region grids only; the host checks)
            const int W = p.regions_w;
            int rq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) rq[q] = q < nreg ? (q % W) * W + q / W : 0;
            unsigned long long t[4];
            int tx[4], ty[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { t[q] = s[rq[q]]; tx[q] = cx[rq[q]]; ty[q] = cy[rq[q]]; }
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int n = q + 1; n < 4; n++)
                    if (n < nreg && t[q] > t[n]) {
                        const unsigned long long ts = t[q]; t[q] = t[n]; t[n] = ts;
                        const int a = tx[q]; tx[q] = tx[n]; tx[n] = a;
                        const int b = ty[q]; ty[q] = ty[n]; ty[n] = b;
                    }
            xc = tx[1]; yc = ty[1];
        }
    }
    const int pad = 63;                                     // BLOCK_SIZE_64 - 1
    if ((xc != 0 || yc != 0) && p.zz_check) {
        int hx = xc, hy = yc;
        if (ox + hx < -pad) hx = -pad - ox;
        if (ox + hx > p.ref_width - 1) hx -= (ox + hx) - (p.ref_width - 1);
        if (oy + hy < -pad) hy = -pad - oy;
        if (oy + hy > p.ref_height - 1) hy -= (oy + hy) - (p.ref_height - 1);
        const bool second = lane >= 32;
        const int row = lane & 31;
        unsigned sad = 0;
        if (row < (sbh >> 1)) {
            const uint8_t* a = src_pic + (ptrdiff_t)(oy + 2 * row) * (ptrdiff_t)src_stride + ox;
            const uint8_t* b = ref_pic + (ptrdiff_t)(oy + (second ? hy : 0) + 2 * row) * (ptrdiff_t)ref_stride + ox + (second ? hx : 0);
            if (sbw == 64) {
                // the full-width SB (all but the last column of a picture): the row's eight 16-byte loads in flight together - the general
                // loop below is a chain of sixteen dependent 4-byte round trips, 23 us per picture for a kernel with one wave per SB
                typedef unsigned me_v4u __attribute__((ext_vector_type(4), aligned(1)));
                me_v4u va[4], vb[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { va[q] = reinterpret_cast<const me_v4u*>(a)[q]; vb[q] = reinterpret_cast<const me_v4u*>(b)[q]; }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    sad = __builtin_amdgcn_sad_u8(va[q].x, vb[q].x, sad); sad = __builtin_amdgcn_sad_u8(va[q].y, vb[q].y, sad);
                    sad = __builtin_amdgcn_sad_u8(va[q].z, vb[q].z, sad); sad = __builtin_amdgcn_sad_u8(va[q].w, vb[q].w, sad);
                }
            } else
            for (int c = 0; c < sbw; c += 4) {
                uint32_t va = 0, vb = 0;
                if (c + 4 <= sbw) { __builtin_memcpy(&va, a + c, 4); __builtin_memcpy(&vb, b + c, 4); }
                else for (int k = 0; c + k < sbw; k++) { va |= (uint32_t)a[c + k] << (8 * k); vb |= (uint32_t)b[c + k] << (8 * k); }
                sad = __builtin_amdgcn_sad_u8(va, vb, sad);
            }
        }
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) sad += (unsigned)__shfl_xor((int)sad, m, 64);
        const unsigned zero = (unsigned)__shfl((int)sad, 0, 64) << 1, hme = (unsigned)__shfl((int)sad, 32, 64) << 1;
        // MIN(zero cost, hme cost) == zero cost: hmeMvdRate is 0 and MD_OFFSET >> MD_SHIFT is 0, the costs are the SADs << 8
        if (zero <= hme) { xc = 0; yc = 0; } else { xc = hx; yc = hy; }
    }
    int saw = (p.search_area_width + 7) & ~7, sah = p.search_area_height;
    int xo = xc - (saw >> 1), yo = yc - (sah >> 1);
    const int W = p.picture_width, H = p.picture_height;
    if (ox + xo < -pad) xo = -pad - ox;
    if (ox + xo < -pad) saw -= -pad - (ox + xo);            // never true after the line above (as in the reference)
    if (ox + xo > W - 1) xo -= (ox + xo) - (W - 1);
    if (ox + xo + saw > W) saw = max(1, saw - ((ox + xo + saw) - W));
    if (saw >= 8) saw &= ~7;
    if (oy + yo < -pad) yo = -pad - oy;
    if (oy + yo < -pad) sah -= -pad - (oy + yo);
    if (oy + yo > H - 1) yo -= (oy + yo) - (H - 1);
    if (oy + yo + sah > H) sah = max(1, sah - ((oy + yo + sah) - H));
    if (lane == 0) {
        if (center) { center[2 * task] = (int16_t)xc; center[2 * task + 1] = (int16_t)yc; }
        area[4 * task] = (int16_t)xo; area[4 * task + 1] = (int16_t)yo; area[4 * task + 2] = (int16_t)saw; area[4 * task + 3] = (int16_t)sah;
    }
}

// ---------------------------------------------------------------------------
// me_bipred_kernel — BiPredictionSearch (EbMotionEstimation.c:6639 -> BiPredictionCompensation :6457 -> BiPredAverging :6317,
// integer vectors) and the candidate ordering MotionEstimateLcu writes into me_results (:8308-8440) for every SB of a batch.
// One workgroup per SB.  Phase 1: a work item is one compared row of one PU (src row against the rounded average of the two
// lists' blocks at that PU's best vectors; every other row and doubled when sub_sad); the items of a PU add into its LDS cell.
// Phase 2: lane p < npus takes RASTER PU p (the order of me_results, partitionWidth / puSearchIndexMap), whose vectors and SADs
// sit at storage index n = map.storage[p] of the result rows (EbMeTierZeroPu order, 16x16 / 8x8 derived shapes in z-order), and
// orders {list 0, list 1, bi} as Sort3Elements (:6809) / the two-candidate rule do.
// ---------------------------------------------------------------------------
struct MePuMap {
    uint8_t storage[ME_PUS_ALL];        // raster PU index -> storage index
    uint8_t x8[ME_PUS_ALL], y8[ME_PUS_ALL], w8[ME_PUS_ALL], h8[ME_PUS_ALL];   // rectangle of RASTER PU p in units of 8 samples
    uint16_t row0[ME_PUS_ALL + 1];      // number of every-other-row rows (h / 2 per PU) of the raster PUs before p: the work-item numbering
};
struct MeResult {                       // == svt_hip_me_result (include/svt_hip_dsp.h)
    int16_t x_mv_l0, y_mv_l0, x_mv_l1, y_mv_l1;
    uint32_t distortion[3];
    uint8_t direction[3], total_me_candidate_index;
};

__global__ __launch_bounds__(ME_THREADS) void me_bipred_kernel(
    const uint8_t* __restrict__ src_pic, uint32_t src_stride, const uint8_t* __restrict__ ref0_pic, uint32_t ref0_stride,
    const uint8_t* __restrict__ ref1_pic, uint32_t ref1_stride, const int16_t* __restrict__ sb_origin,
    const uint32_t* __restrict__ best_sad0, const uint32_t* __restrict__ best_mv0, const uint32_t* __restrict__ best_sad1,
    const uint32_t* __restrict__ best_mv1, uint32_t pu_pitch, int npus, int bipred_all_pus, int sub_sad, const MePuMap map,
    uint32_t* __restrict__ bipred_sad /* [n][pu_pitch] storage order, or NULL */, MeResult* __restrict__ results /* [n][npus] raster order */,
    uint32_t nsb) {
    __shared__ unsigned s_bi[ME_PUS_ALL];
    // the PU map in LDS: the item loop below looks a PU up by binary search and reads five of its fields - through the kernel arguments
    // (per-lane indices: vector loads) that was a dozen dependent round trips per work item
    __shared__ MePuMap s_map;
    static_assert(sizeof(MePuMap) % 2 == 0 && alignof(MePuMap) >= 2, "copied as 16-bit words");
    const uint32_t sb = blockIdx.x;
    if (sb >= nsb) return;
    const int tid = threadIdx.x;
    const int ox = sb_origin[2 * sb], oy = sb_origin[2 * sb + 1];
    const bool two = best_sad1 != nullptr;
    for (int i = tid; i < ME_PUS_ALL; i += ME_THREADS) s_bi[i] = 0;
    for (int i = tid; i < (int)(sizeof(MePuMap) / 2); i += ME_THREADS) reinterpret_cast<uint16_t*>(&s_map)[i] = reinterpret_cast<const uint16_t*>(&map)[i];
    __syncthreads();
    if (two) {
        const int nbi = bipred_all_pus ? npus : min(npus, 21);
        const int nitems = sub_sad ? s_map.row0[nbi] : 2 * s_map.row0[nbi];
        for (int it = tid; it < nitems; it += ME_THREADS) {
            const int key = sub_sad ? it : (it >> 1);        // position in the sub-sampled row numbering
            int lo = 0, hi = nbi - 1;                        // PU whose item range holds `key`
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_map.row0[mid] <= key) lo = mid; else hi = mid - 1; }
            const int p = lo, n = s_map.storage[p];
            const int row = sub_sad ? 2 * (key - s_map.row0[p]) : 2 * (key - s_map.row0[p]) + (it & 1);
            const uint32_t m0 = best_mv0[(size_t)sb * pu_pitch + n], m1 = best_mv1[(size_t)sb * pu_pitch + n];
            const int x0 = (int16_t)(m0 & 0xffffu) >> 2, y0 = (int16_t)(m0 >> 16) >> 2, x1 = (int16_t)(m1 & 0xffffu) >> 2, y1 = (int16_t)(m1 >> 16) >> 2;
            const int px = ox + 8 * s_map.x8[p], py = oy + 8 * s_map.y8[p] + row, w = 8 * s_map.w8[p];
            const uint8_t* a = src_pic + (ptrdiff_t)py * (ptrdiff_t)src_stride + px;
            const uint8_t* b = ref0_pic + (ptrdiff_t)(py + y0) * (ptrdiff_t)ref0_stride + px + x0;
            const uint8_t* c = ref1_pic + (ptrdiff_t)(py + y1) * (ptrdiff_t)ref1_stride + px + x1;
            unsigned sad = 0;
            // rounded average of four byte pairs: (x | y) - (((x ^ y) >> 1) & 0x7f7f7f7f).  Two 8-byte chunks per round (w is 8 or a
            // multiple of 16): six loads in flight instead of three
            auto chunk = [&](const uint2 va, const uint2 vb, const uint2 vc) {
                const uint32_t ax = (vb.x | vc.x) - (((vb.x ^ vc.x) >> 1) & 0x7f7f7f7fu), ay = (vb.y | vc.y) - (((vb.y ^ vc.y) >> 1) & 0x7f7f7f7fu);
                sad = __builtin_amdgcn_sad_u8(va.x, ax, sad);
                sad = __builtin_amdgcn_sad_u8(va.y, ay, sad);
            };
            if (w == 8) {
                uint2 va, vb, vc;
                __builtin_memcpy(&va, a, 8); __builtin_memcpy(&vb, b, 8); __builtin_memcpy(&vc, c, 8);
                chunk(va, vb, vc);
            } else
            for (int k = 0; k < w; k += 16) {
                uint2 va[2], vb[2], vc[2];
                __builtin_memcpy(va, a + k, 16); __builtin_memcpy(vb, b + k, 16); __builtin_memcpy(vc, c + k, 16);
                chunk(va[0], vb[0], vc[0]); chunk(va[1], vb[1], vc[1]);
            }
            atomicAdd(&s_bi[n], sub_sad ? sad << 1 : sad);
        }
    }
    __syncthreads();
    if (tid < npus) {
        const int p = tid, n = s_map.storage[p];
        const uint32_t l0 = best_sad0[(size_t)sb * pu_pitch + n], m0 = best_mv0[(size_t)sb * pu_pitch + n];
        const uint32_t l1 = two ? best_sad1[(size_t)sb * pu_pitch + n] : 0, m1 = two ? best_mv1[(size_t)sb * pu_pitch + n] : 0;
        const bool has_bi = two && (bipred_all_pus || p < 21);
        const uint32_t bi = s_bi[n];
        if (bipred_sad && has_bi) bipred_sad[(size_t)sb * pu_pitch + n] = bi;
        MeResult r;
        r.x_mv_l0 = (int16_t)(m0 & 0xffffu); r.y_mv_l0 = (int16_t)(m0 >> 16); r.x_mv_l1 = (int16_t)(m1 & 0xffffu); r.y_mv_l1 = (int16_t)(m1 >> 16);
        const uint32_t d[3] = {l0, l1, bi};
        int o0 = 0, o1 = 1, o2 = 2, total = two ? 2 : 1;
        if (has_bi) {
            total = 3;
            if (l0 <= l1 && l0 <= bi) { o0 = 0; o1 = l1 <= bi ? 1 : 2; o2 = l1 <= bi ? 2 : 1; }
            else if (l1 <= l0 && l1 <= bi) { o0 = 1; o1 = l0 <= bi ? 0 : 2; o2 = l0 <= bi ? 2 : 0; }
            else if (l0 <= l1) { o0 = 2; o1 = 0; o2 = 1; }
            else { o0 = 2; o1 = 1; o2 = 0; }
        } else if (two) {
            o0 = l0 <= l1 ? 0 : 1; o1 = 1 - o0;
        }
        r.distortion[0] = d[o0]; r.direction[0] = (uint8_t)o0;
        r.distortion[1] = total > 1 ? d[o1] : 0; r.direction[1] = total > 1 ? (uint8_t)o1 : 0;
        r.distortion[2] = total > 2 ? d[o2] : 0; r.direction[2] = total > 2 ? (uint8_t)o2 : 0;
        r.total_me_candidate_index = (uint8_t)total;
        results[(size_t)sb * npus + p] = r;
    }
}

// ---------------------------------------------------------------------------
// hme_level_kernel — one level of the hierarchical motion estimation for every SB of a picture (and every reference /
// search region handed in as separate tasks): HmeLevel0 / HmeLevel1 / HmeLevel2 (EbMotionEstimation.c:5689-6150) INCLUDING the
// per-SB search-area placement and clipping the reference does on the host (:5729-5798): origin = offset + search centre,
// clipped against the padded reference picture in the reference's own statement order (its left / top "shrink" statements
// test the already corrected origin and never fire: restated as written), width rounded down to a multiple of 16 / 8.
// The search itself is sad_loop_kernel on EVERY OTHER ROW of the block (the 1/16 SB buffer holds every other row,
// EbMotionEstimationProcess.c:548-556; levels 1 / 2 double both strides), first strict minimum in raster order.
// One workgroup per task: the block's even rows and the clipped window live in LDS, the candidates are strided over the
// 256 lanes (v_sad_u8 on dwords rebuilt with v_alignbyte), argmin key = sad << 32 | candidate.
// Results as the reference leaves them: SAD x 2, (x + origin) << mv_shift.
// ---------------------------------------------------------------------------
struct HmeParams {          // == svt_hip_hme_params (include/svt_hip_dsp.h)
    int32_t search_area_width, search_area_height, x_origin_offset, y_origin_offset, pad_width, pad_height, ref_width,
        ref_height, round_down, mv_shift;
};

// Several search regions of one level in ONE launch (the reference splits a level's area into up to 2 x 2 regions and carries
// each region's vector through the next levels): blockIdx.y = region, its parameter set from ps, its centres / results in
// planes of ntasks entries ([region][task]).
struct HmeParamSets { HmeParams p[4]; };

__global__ __launch_bounds__(ME_THREADS) void hme_level_kernel(
    const uint8_t* __restrict__ src_pic, uint32_t src_stride, const uint8_t* __restrict__ ref_pic, uint32_t ref_stride,
    const int16_t* __restrict__ sb_origin /* [n][2] */, const uint16_t* __restrict__ sb_size /* [n][2] */,
    const int16_t* __restrict__ centers /* [regions][n][2] or NULL */, int center_shift, const HmeParamSets ps,
    unsigned long long* __restrict__ best_sad /* [regions][n] */, int16_t* __restrict__ mv /* [regions][n][2] */, uint32_t wpitch, uint32_t ntasks) {
    const HmeParams& p = ps.p[blockIdx.y];
    if (centers) centers += (size_t)blockIdx.y * ntasks * 2;
    best_sad += (size_t)blockIdx.y * ntasks;
    mv += (size_t)blockIdx.y * ntasks * 2;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* s_src = reinterpret_cast<uint32_t*>(smem);            // [32 rows][16 dwords]: the block's even rows, zero-padded
    uint8_t* s_ref = smem + 32 * 64;
    __shared__ unsigned long long s_red[4];
    const uint32_t task = blockIdx.x;
    if (task >= ntasks) return;
    const int tid = threadIdx.x;
    const int ox = sb_origin[2 * task], oy = sb_origin[2 * task + 1];
    const int sbw = sb_size[2 * task], sbh = sb_size[2 * task + 1];
    // The size table lives in device memory, the host cannot check it; s_src holds 32 rows of 64 bytes and the window is sized for
    // blocks of at most 64x64 (include/svt_hip_dsp.h).  An entry outside 1..64 x 2..64 is answered, not searched.
    if (sbw < 1 || sbh < 2 || sbw > 64 || sbh > 64) {
        if (tid == 0) { best_sad[task] = ~0ull; mv[2 * task] = 0; mv[2 * task + 1] = 0; }
        return;
    }
    const int xc = centers ? (centers[2 * task] >> center_shift) : 0, yc = centers ? (centers[2 * task + 1] >> center_shift) : 0;
    int saw = p.search_area_width, sah = p.search_area_height;
    int xo = p.x_origin_offset + xc, yo = p.y_origin_offset + yc;
    const int W = p.ref_width, H = p.ref_height;
    xo = (ox + xo < -p.pad_width) ? -p.pad_width - ox : xo;
    saw = (ox + xo < -p.pad_width) ? saw - (-p.pad_width - (ox + xo)) : saw;        // never true after the line above (as in the reference)
    xo = (ox + xo > W - 1) ? xo - ((ox + xo) - (W - 1)) : xo;
    if (ox + xo + saw > W) saw = max(1, saw - ((ox + xo + saw) - W));
    if (saw >= p.round_down) saw &= ~(p.round_down - 1);
    yo = (oy + yo < -p.pad_height) ? -p.pad_height - oy : yo;
    sah = (oy + yo < -p.pad_height) ? sah - (-p.pad_height - (oy + yo)) : sah;
    yo = (oy + yo > H - 1) ? yo - ((oy + yo) - (H - 1)) : yo;
    if (oy + yo + sah > H) sah = max(1, sah - ((oy + yo + sah) - H));
    const int hh = sbh >> 1;                                          // rows compared
    const int win_w = sbw + saw - 1, win_h = sah + 2 * hh - 2;
    const uint8_t* gs = src_pic + (ptrdiff_t)oy * (ptrdiff_t)src_stride + ox;
    const uint8_t* gr = ref_pic + (ptrdiff_t)(oy + yo) * (ptrdiff_t)ref_stride + (ox + xo);
    // ---- stage: even block rows (zero-padded to 64 B) and the window, dword by dword (bytes at a row's ragged end) ----
    for (int i = tid; i < 32 * 16; i += ME_THREADS) {
        const int r = i >> 4, q = i & 15;
        uint32_t v = 0;
        if (r < hh) {
            const uint8_t* g = gs + (size_t)(2 * r) * src_stride + 4 * q;
            if (4 * q + 4 <= sbw) __builtin_memcpy(&v, g, 4);
            else for (int b = 0; 4 * q + b < sbw; b++) v |= (uint32_t)g[b] << (8 * b);
        }
        s_src[i] = v;
    }
    const int wq = (win_w + 3) >> 2;                                  // dwords per window row
    for (int i = tid; i < win_h * wq; i += ME_THREADS) {
        const int r = i / wq, q = i - r * wq;
        const uint8_t* g = gr + (ptrdiff_t)r * (ptrdiff_t)ref_stride + 4 * q;
        uint32_t v = 0;
        if (4 * q + 4 <= win_w) __builtin_memcpy(&v, g, 4);
        else for (int b = 0; 4 * q + b < win_w; b++) v |= (uint32_t)g[b] << (8 * b);
        *reinterpret_cast<uint32_t*>(s_ref + (size_t)r * wpitch + 4 * q) = v;
    }
    __syncthreads();
    // ---- candidates ----
    unsigned long long best = ~0ull;
    const int ncand = saw * sah;
    const int bq = (sbw + 3) >> 2;                                    // dwords per block row
    // whole SBs of the three levels (16x16 / 32x32 / 64x64: the block size is the same for every lane of the workgroup) take an
    // unrolled row body - the general loop's trip counts are run-time values, each LDS read waits for the one before it
    auto search = [&](auto bq_c) {
        constexpr int BQ = decltype(bq_c)::value;                     // dwords per block row; 0 = run-time width (ragged allowed)
        for (int cand = tid; cand < ncand; cand += ME_THREADS) {
            const int ys = cand / saw, xs = cand - ys * saw;
            const unsigned sh = (unsigned)(xs & 3);
            unsigned acc = 0;
            for (int r = 0; r < hh; r++) {
                const uint32_t* rrow = reinterpret_cast<const uint32_t*>(s_ref + (size_t)(ys + 2 * r) * wpitch) + (xs >> 2);
                const uint32_t* srow = s_src + r * 16;
                if constexpr (BQ > 0) {
                    uint32_t rw[BQ + 1];
#pragma unroll
                    for (int q = 0; q <= BQ; q++) rw[q] = rrow[q];    // wpitch leaves 8 spare bytes per row
#pragma unroll
                    for (int q = 0; q < BQ; q++) acc = __builtin_amdgcn_sad_u8(srow[q], __builtin_amdgcn_alignbyte(rw[q + 1], rw[q], sh), acc);
                } else {
                    uint32_t lo = rrow[0];
                    for (int q = 0; q < bq; q++) {
                        const uint32_t hi = rrow[q + 1];
                        uint32_t rv = __builtin_amdgcn_alignbyte(hi, lo, sh);
                        const int rem = sbw - 4 * q;
                        if (rem < 4) rv &= (1u << (8 * rem)) - 1;     // ragged last dword: the source side is zero-padded
                        acc = __builtin_amdgcn_sad_u8(srow[q], rv, acc);
                        lo = hi;
                    }
                }
            }
            const unsigned long long key = ((unsigned long long)acc << 32) | (unsigned)cand;
            best = key < best ? key : best;
        }
    };
    if (sbw == 16) search(std::integral_constant<int, 4>{});
    else if (sbw == 32) search(std::integral_constant<int, 8>{});
    else if (sbw == 64) search(std::integral_constant<int, 16>{});
    else search(std::integral_constant<int, 0>{});
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = __shfl_xor(best, m, 64);
        best = o < best ? o : best;
    }
    if ((tid & 63) == 0) s_red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long b = s_red[0];
        for (int i = 1; i < 4; i++) b = s_red[i] < b ? s_red[i] : b;
        const int cand = (int)(unsigned)b;
        const int ys = cand / saw, xs = cand - ys * saw;
        best_sad[task] = (b >> 32) * 2ull;
        mv[2 * task] = (int16_t)((xs + xo) << p.mv_shift);
        mv[2 * task + 1] = (int16_t)((ys + yo) << p.mv_shift);
    }
}

}  // namespace svtdev
