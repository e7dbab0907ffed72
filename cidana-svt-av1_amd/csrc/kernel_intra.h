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
#include <type_traits>

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

// ---- non-directional modes -------------------------------------------------------------
// Work split: a block is covered by LB = bw*bh/ppl consecutive lanes (ppl = pixels per lane =
// min(bw, 16 B / sizeof(PixT))), i.e. one lane writes 16 B (or a whole narrow row) of one output
// row, ONE item per lane and a grid as large as the job.  tools/probe/store_probe.hip: for a
// write-only stream "one 16-B store per lane, huge grid" reaches 6.8 TB/s, 4 stores per lane 6.0,
// grid-stride loops 4.1-5.7 (on gfx9 loads and stores share vmcnt, so a loop of load -> store makes
// each iteration wait for the previous iteration's stores); 4 blocks per lane measured slower as well.

// One lane's output: `ppl` pixels (16 B when the block is wide enough).  The value is packed into
// four dwords first and stored with the widest instruction the address allows; keeping the three
// paths structurally different stops the compiler from "hoisting common stores" out of them, which
// had turned the aligned 16-B store into byte + short + dwordx3 + byte stores (half the bandwidth).
template <typename PixT>
__device__ __forceinline__ void intra_store(PixT* d, const uint4 v, int ppl) {
    const int nbytes = ppl * (int)sizeof(PixT);
    const uintptr_t a = reinterpret_cast<uintptr_t>(d);
    if (nbytes == 16 && (a & 15) == 0) {
        *reinterpret_cast<uint4*>(d) = v;
    } else if ((a & 3) == 0) {                       // nbytes is 4, 8 or 16
        // (static register indexing + predicates: a runtime-indexed copy of `v` would live in scratch memory,
        // i.e. one extra 16-B scratch store per lane even when this path is not taken - measured as 1.6x WRITE_SIZE)
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t* d4 = reinterpret_cast<uint32_t*>(d);
#pragma unroll
        for (int q = 0; q < 4; q++) if (q < (nbytes >> 2)) d4[q] = w[q];
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint8_t* d1 = reinterpret_cast<uint8_t*>(d);
#pragma unroll
        for (int q = 0; q < 16; q++) if (q < nbytes) d1[q] = (uint8_t)(w[q >> 2] >> (8 * (q & 3)));
    }
}

__device__ const uint32_t kZeroWord[1] = {0};     // stand-in offset table (keeps the offset fetch branch-free)

template <typename PixT>
struct IntraNb {                       // what one lane needs from a block's neighbours
    uint4 abv;                         // above[c0 .. c0+ppl-1], packed as loaded (kept as registers: a byte array here
                                       // makes the compiler split the 16-B load into eight 2-B loads)
    uint4 wwv;                         // smooth weights (bytes) of those columns
    int lft, bl, tr, tl, sum;
};

// WIDE = the block is at least 16 B wide (ppl == PXL): every access below is then a single wide
// instruction and the whole kernel is straight-line code up to the store, so all of a lane's loads
// are in flight together (with the runtime narrow/wide branches the compiler had put an
// s_waitcnt vmcnt(0) after every load).
template <typename PixT, int MODE, bool WIDE>
__device__ __forceinline__ void intra_load(IntraNb<PixT>& nb, const PixT* __restrict__ above, const PixT* __restrict__ left,
                                           int r, int c0, int ppl, int bw, int bh, int lj, int grp) {
    constexpr int PXL = 16 / (int)sizeof(PixT);
    typedef unsigned svt_v4u_unaligned __attribute__((ext_vector_type(4), aligned(1)));
    nb.abv = make_uint4(0, 0, 0, 0); nb.wwv = nb.abv;
    if (MODE == IM_V || MODE == IM_SMOOTH || MODE == IM_SMOOTH_V || MODE == IM_PAETH) {
        if (WIDE) {
            const svt_v4u_unaligned t = *reinterpret_cast<const svt_v4u_unaligned*>(above + c0);     // one unaligned 16-B load
            nb.abv = make_uint4(t.x, t.y, t.z, t.w);
        } else {
            PixT tmp[PXL];
#pragma unroll
            for (int k = 0; k < PXL; k++) tmp[k] = k < ppl ? above[c0 + k] : (PixT)0;
            __builtin_memcpy(&nb.abv, tmp, 16);
        }
    }
    if (MODE == IM_SMOOTH || MODE == IM_SMOOTH_H) {
        uint8_t tmp[16] = {0};
        if (WIDE) __builtin_memcpy(tmp, kSmWeights + bw + c0, PXL);
        else {
#pragma unroll
            for (int k = 0; k < PXL; k++) tmp[k] = kSmWeights[bw + c0 + (k < ppl ? k : 0)];
        }
        __builtin_memcpy(&nb.wwv, tmp, 16);
    }
    nb.lft = (MODE == IM_H || MODE == IM_SMOOTH || MODE == IM_SMOOTH_H || MODE == IM_PAETH) ? (int)left[r] : 0;
    nb.bl = (MODE == IM_SMOOTH || MODE == IM_SMOOTH_V) ? (int)left[bh - 1] : 0;
    nb.tr = (MODE == IM_SMOOTH || MODE == IM_SMOOTH_H) ? (int)above[bw - 1] : 0;
    nb.tl = MODE == IM_PAETH ? (int)above[-1] : 0;
    nb.sum = 0;
    if (MODE == IM_DC || MODE == IM_DC_TOP || MODE == IM_DC_LEFT) {
        // cooperative sum over 4-sample chunks (bw, bh are multiples of 4): lane lj of the block's
        // reduction group takes chunks lj, lj+grp, ...; the group is reduced after all loads are out
        const int na = MODE != IM_DC_LEFT ? bw : 0, nl = MODE != IM_DC_TOP ? bh : 0;
        const int nch = (na + nl) >> 2;
        if (WIDE && nch <= grp) {
            // one predicated chunk per lane, no loop (every wide block except 16x4)
            const int o = lj << 2;
            const PixT* q = o < na ? above + o : left + (o - na);
            PixT t[4] = {0, 0, 0, 0};
            if (lj < nch) __builtin_memcpy(t, q, 4 * sizeof(PixT));
            nb.sum = (int)t[0] + (int)t[1] + (int)t[2] + (int)t[3];
        } else {
            for (int ch = lj; ch < nch; ch += grp) {
                const int o = ch << 2;
                const PixT* q = o < na ? above + o : left + (o - na);
                PixT t[4];
                __builtin_memcpy(t, q, 4 * sizeof(PixT));
                nb.sum += (int)t[0] + (int)t[1] + (int)t[2] + (int)t[3];
            }
        }
    }
}

template <typename PixT, int MODE, bool WIDE>
__device__ __forceinline__ uint4 intra_row(const IntraNb<PixT>& nb, int wh, int dc) {
    constexpr int PXL = 16 / (int)sizeof(PixT), ES = (int)sizeof(PixT), BITS = 8 * ES;
    constexpr uint32_t MASK = ES == 1 ? 0xffu : 0xffffu;
    if (MODE == IM_V) return nb.abv;                                   // the above segment as loaded
    if (MODE == IM_H || MODE == IM_DC || MODE == IM_DC_TOP || MODE == IM_DC_LEFT || MODE == IM_DC_128) {
        const uint32_t v = (uint32_t)(MODE == IM_H ? nb.lft : dc);
        const uint32_t w = ES == 1 ? v * 0x01010101u : v * 0x00010001u;
        return make_uint4(w, w, w, w);
    }
    // terms that do not depend on the column
    const int sm_c = (256 - wh) * nb.bl + 256 * nb.tr + 256;          // SMOOTH: + wh*ab + ww*(lft - tr)
    const int smv_c = (256 - wh) * nb.bl + 128;                       // SMOOTH_V
    const int smh_c = 256 * nb.tr + 128;                              // SMOOTH_H: + ww*(lft - tr)
    const int dlt = nb.lft - nb.tr;
    const int p_t = abs(nb.lft - nb.tl);                              // PAETH: |base - top|  = |left - topleft|
    const int p_k = nb.lft - 2 * nb.tl;                               //        |base - tl|   = |top + left - 2 topleft|
    const uint32_t aw[4] = {nb.abv.x, nb.abv.y, nb.abv.z, nb.abv.w}, ww4[4] = {nb.wwv.x, nb.wwv.y, nb.wwv.z, nb.wwv.w};
    uint32_t ow[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < PXL; k++) {
        const int t = (int)((aw[(k * ES) >> 2] >> (BITS * (k & (4 / ES - 1)))) & MASK);      // above[c0 + k]
        const int ww = (int)((ww4[k >> 2] >> (8 * (k & 3))) & 0xffu);                          // smooth weight of that column
        int v;
        if (MODE == IM_SMOOTH) v = (wh * t + ww * dlt + sm_c) >> 9;
        else if (MODE == IM_SMOOTH_V) v = (wh * t + smv_c) >> 8;
        else if (MODE == IM_SMOOTH_H) v = (ww * dlt + smh_c) >> 8;
        else {   // IM_PAETH
            const int pl = abs(t - nb.tl), ptl = abs(t + p_k);        // |base - left| = |top - topleft|
            v = (pl <= p_t && pl <= ptl) ? nb.lft : (p_t <= ptl ? t : nb.tl);
        }
        ow[(k * ES) >> 2] |= ((uint32_t)v & MASK) << (BITS * (k & (4 / ES - 1)));
    }
    return make_uint4(ow[0], ow[1], ow[2], ow[3]);
}

template <typename PixT, int MODE, bool WIDE, int IU>
__global__ __launch_bounds__(256) void intra_pred_kernel(
    PixT* __restrict__ dst, int32_t dst_stride, size_t dst_block_pitch, const uint32_t* __restrict__ dst_offsets,
    const PixT* __restrict__ above_all, const PixT* __restrict__ left_all, int32_t nb_pitch, int bw, int bh, int bd,
    uint32_t dc_magic, uint32_t nblocks) {
    constexpr int PXL = 16 / (int)sizeof(PixT);            // pixels per lane when the block is wide enough
    const int ppl = WIDE ? PXL : bw;                       // pixels per lane (host picks WIDE = bw >= PXL)
    const int lanes_per_row = WIDE ? bw >> __builtin_ctz((uint32_t)PXL) : 1;
    const uint32_t per_block = (uint32_t)(lanes_per_row * bh);      // LB: power of two, 4 .. 512
    const size_t total = (size_t)per_block * nblocks;
    const size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int pb_shift = __builtin_ctz(per_block);                  // per_block is a power of two
    const int lr_shift = __builtin_ctz((uint32_t)lanes_per_row);
    const uint32_t j = (uint32_t)(item & (per_block - 1));
    const int r = (int)(j >> lr_shift), c0 = (int)(j & (uint32_t)(lanes_per_row - 1)) * ppl;
    const int wh = (MODE == IM_SMOOTH || MODE == IM_SMOOTH_V) ? kSmWeights[bh + r] : 0;
    const int grp = per_block < 64 ? (int)per_block : 64;          // DC reduction group (inside one wave)
    const int lj = (int)(j & (uint32_t)(grp - 1));
    constexpr bool IS_DC = MODE == IM_DC || MODE == IM_DC_TOP || MODE == IM_DC_LEFT;
    const int cnt = MODE == IM_DC ? bw + bh : (MODE == IM_DC_TOP ? bw : bh);
    // a lane owns IU items, `nthreads` apart (nthreads is a multiple of per_block: same row / column)
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    bool live[IU];
    uint32_t blk[IU];
    IntraNb<PixT> nb[IU];
    size_t boff[IU];
    const bool has_offs = dst_offsets != nullptr;
    // ---- every load of this lane, back to back -----------------------------------------------
#pragma unroll
    for (int u = 0; u < IU; u++) {
        const size_t it = item + (size_t)u * nthreads;
        live[u] = it < total;
        blk[u] = live[u] ? (uint32_t)(it >> pb_shift) : 0u;         // dead slots re-read block 0 (never stored)
        const PixT* above = above_all + (size_t)blk[u] * nb_pitch + NB_ORIGIN;
        const PixT* left = left_all + (size_t)blk[u] * nb_pitch + NB_ORIGIN;
        intra_load<PixT, MODE, WIDE>(nb[u], above, left, r, c0, ppl, bw, bh, lj, grp);
        const uint32_t off_word = (has_offs ? dst_offsets : kZeroWord)[has_offs ? blk[u] : 0u];
        boff[u] = has_offs ? (size_t)off_word : (size_t)blk[u] * dst_block_pitch;
    }
    // ---- reduce, predict, store ---------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < IU; u++) {
        int dc = 0;
        if (IS_DC) {
            int sum = nb[u].sum;
            for (int m = grp >> 1; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
            // (sum + cnt/2) / cnt, exact (EbIntraPrediction.c:1880-1896): cnt <= 128, numerator < 2^20,
            // dc_magic = floor(2^32 / cnt) + 1 (host) makes the high product the exact quotient
            dc = (int)__umulhi((uint32_t)(sum + (cnt >> 1)), dc_magic);
        } else if (MODE == IM_DC_128) {
            dc = 128 << (bd - 8);
        }
        const uint4 out = intra_row<PixT, MODE, WIDE>(nb[u], wh, dc);
        if (live[u]) intra_store<PixT>(dst + boff[u] + (size_t)r * dst_stride + c0, out, ppl);
    }
}

// ---- directional modes (av1_dr_prediction_z1/z2/z3, :370-477 / :3394-3506) ------------------
// Same lane -> (row, 16-B column segment) mapping, one step per workgroup.  The two edge arrays of
// the blocks a workgroup works on are staged in LDS in PAIR form: dword i = edge[i] | edge[i+1] << 16,
// padded with copies of the last valid sample edge[max_base] (the reference's "base >= max_base ->
// edge[max_base]" case then falls out of the same interpolation: (32*e + 16) >> 5 == e, so the pixel
// loop has no bounds test).  Staging takes the samples in groups of four pairs per lane (one 8-byte load, four v_perm_b32, two
// ds_write_b64).  A pixel is then ONE aligned ds_read_b32 and ONE v_dot2_u32_u16:
//   (a*(32-sh) + b*sh + 16) >> 5  =  dot2((a, b), (32-sh, sh), 16) >> 5.
// (The first LDS version kept samples as bytes: the compiler merged the per-sample reads into unaligned
// ds_read_b128/b64 at ~23 LDS cycles each and the LDS was 86 % busy; one ds_read_u8 per sample was slower
// still.)  8- and 16-bit samples share the pair format.
__device__ __forceinline__ uint32_t dir_lerp2(uint32_t pair, uint32_t w) {          // w = (32 - sh) | sh << 16
    uint32_t r;
    asm("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(pair), "v"(w), "v"(16u));
    return r >> 5;
}

// Several angles of one zone in one launch (the open-loop intra search runs up to 19 per zone on the same
// neighbours): the block's edges are staged in LDS ONCE and the lanes loop over the angles; angle k goes to the dense
// output batch `slot[k] * batch_pitch` samples into dst.
struct DirMulti {
    int n;                       // 0: single-angle launch (dx, dy arguments)
    int chunk;                   // angles per workgroup: blockIdx.y takes angles [y * chunk, y * chunk + chunk).  A picture-sized
                                 // batch leaves the GPU a few workgroups per CU each walking all (up to 19) angles one after the
                                 // other - a latency chain; spreading the angles over grid.y shortens it (edges are re-staged per y)
    int16_t dx[20], dy[20];
    uint8_t slot[20];
    size_t batch_pitch;
    // SAD mode (8-bit, blocks of at most 64 lanes: 8x8, 16x16): no prediction is stored; angle k's prediction is
    // compared with the source block at xy[blk] in sad_pic and its SAD goes to sad_dist[blk * sad_ncand + slot[k]]
    const uint8_t* sad_pic;
    uint32_t sad_stride;
    const uint32_t* sad_xy;
    uint32_t* sad_dist;
    uint32_t sad_ncand;
    // zone 2, one lane per row (bw == samples per lane: the open-loop search's 8x8 and 16x16), no up-sampling: the LEFT-edge terms
    // of pixel k depend on (dy, k) only, so the host tabulates them per angle - the kernel fetches a row of each with one scalar
    // load and the per-pixel work drops to an add, two selects, the dot and the shift.
    //   z2_w2[a][k] = (32 - s) | s << 16,  s = ((-dy (k + 1)) & 63) >> 1;   z2_ol[a][k] = 4 * ((-dy (k + 1)) >> 6)
    int z2_tab;                  // tables valid (host: zone 2, lanes per row == 1, no up-sampling)
    uint32_t z2_w2[20][16];
    int32_t z2_ol[20][16];
};
// the same without zone 2's tables (zones 1 and 3 of ois_dir3_kernel: the kernel-argument block is 4 KiB)
struct DirMultiLite {
    int n, chunk;
    int16_t dx[20], dy[20];
    uint8_t slot[20];
    size_t batch_pitch;
    const uint8_t* sad_pic;
    uint32_t sad_stride;
    const uint32_t* sad_xy;
    uint32_t* sad_dist;
    uint32_t sad_ncand;
    int z2_tab;
};

// NPX = samples per lane = min(bw, 16 B worth): a compile-time constant so that narrow blocks (bw 4 / 8) do not compute
// a full 16-B segment per lane and throw most of it away (the open-loop search's 8x8 pass spent half its VALU there).
// TAB: zone 2 with the host's per-angle tables (DirMulti::z2_w2 / z2_ol); a separate instantiation so that the dense-output kernels
// keep their own code (with both paths in one kernel the 32x32 single-angle form ran 15 % slower).
// LDS pair dwords between the staged edges of consecutive blocks of a workgroup.  Lanes of `lpb` rows read a window of about lpb
// consecutive dwords of their block's edge, and a half wave (32 lanes, one LDS pass) holds 32 / lpb blocks: a stride that is
// lpb modulo 32 puts those windows on disjoint banks (2 * estride alone is 0 or 16 modulo 32: the open-loop search's 8x8 pass,
// 8 lanes per block, measured 50 % of its LDS cycles in bank conflicts).  Even, so that the 8-byte staging stores stay aligned.
__host__ __device__ inline int dir_slot_stride(int estride, uint32_t lpb) {
    const int base = 2 * estride;
    return lpb < 32u ? base + (int)(((uint32_t)lpb - (uint32_t)base) & 31u) : base;
}

// (the body is a device function so that ois_dir3_kernel below can run the three zones of one candidate list in ONE launch; MT is
// DirMulti or DirMultiLite, ypart the angle chunk this workgroup takes)
template <typename PixT, int MODE, int NPX, bool TAB, typename MT>
__device__ __forceinline__ void intra_dir_body(
    PixT* __restrict__ dst, int32_t dst_stride, size_t dst_block_pitch, const uint32_t* __restrict__ dst_offsets,
    const PixT* __restrict__ above_all, const PixT* __restrict__ left_all, int32_t nb_pitch, int bw, int bh,
    int up_above, int up_left, int dx, int dy, int lim_a, int lim_l, int n_pad, int bd, uint32_t nblocks, const MT& multi, const uint32_t ypart) {
    // interpolated values of in-range samples are in range; 16-bit input may carry out-of-range
    // samples, which clip_pixel_highbd (EbIntraPrediction.c:3394-3506) would clip: keep that
    const uint32_t maxv = (1u << bd) - 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t dir_smem[];
    uint32_t* sm = reinterpret_cast<uint32_t*>(dir_smem);
    constexpr int PXL = NPX;
    constexpr int PPW = 4 / (int)sizeof(PixT);                             // pixels per output dword
    constexpr int NW = NPX / PPW;                                          // output dwords per lane
    static_assert(NPX * (int)sizeof(PixT) <= 16 && NW >= 1, "a lane writes 4 .. 16 bytes");
    constexpr int ppl = NPX;                                               // host: NPX == min(bw, 16 / sizeof(PixT))
    const int lanes_per_row = bw >> __builtin_ctz((uint32_t)ppl);
    const uint32_t per_block = (uint32_t)(lanes_per_row * bh);             // power of two, 4 .. 512
    const int pb_shift = __builtin_ctz(per_block);
    const int lr_shift = __builtin_ctz((uint32_t)lanes_per_row);
    const uint32_t lpb = per_block >= 256 ? 256u : per_block;              // this workgroup's lanes on one block
    const uint32_t slot = threadIdx.x >> __builtin_ctz(lpb);               // which of the workgroup's blocks
    const uint32_t jl = threadIdx.x & (lpb - 1);                           // lane inside that block's group
    const size_t total = (size_t)per_block * nblocks;
    const size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = (uint32_t)(item & (per_block - 1));
    const int r = (int)(j >> lr_shift), c0 = (int)(j & (uint32_t)(lanes_per_row - 1)) * ppl;
    const int estride = (n_pad + 10) & ~7;                                 // pair dwords per staged edge (+ 3: the last group of four may pass n_pad)
    uint32_t* sa = sm + (size_t)slot * dir_slot_stride(estride, lpb);                      // above edge of this lane's block
    uint32_t* sl = sa + estride;
    const bool live = item < total;
    const uint32_t blk = live ? (uint32_t)(item >> pb_shift) : 0u;
    const bool has_offs = dst_offsets != nullptr;
    const uint32_t off_word = (has_offs ? dst_offsets : kZeroWord)[has_offs ? blk : 0u];
    // ---- stage the block's two edges as pairs: array positions [NB_ORIGIN-2, n_pad) - sample i from memory up to lim (= NB_ORIGIN +
    // max_base), edge[lim] after it.  A lane takes GROUPS of four consecutive pair dwords: one (unaligned) 8- or 12-byte load
    // gives the five samples they need, four v_perm / v_alignbit build the pairs, two ds_write_b64 store them - against two byte
    // loads and ~7 VALU instructions per pair before (the staging cost as much as the prediction of a 32x32 z1 block).
    {
        const PixT* ga = above_all + (size_t)blk * nb_pitch;
        const PixT* gl = left_all + (size_t)blk * nb_pitch;
        const int ng = (n_pad - (NB_ORIGIN - 2) + 3) >> 2;                 // groups per edge (estride is a multiple of 8: room for the last one)
        for (int g = (int)jl; g < 2 * ng; g += (int)lpb) {
            const bool left_edge = g >= ng;
            const int gi = left_edge ? g - ng : g;
            const PixT* ge = left_edge ? gl : ga;
            const int lim = left_edge ? lim_l : lim_a;
            const int idx = NB_ORIGIN - 2 + 4 * gi;
            const uint32_t tail = (uint32_t)ge[lim] * 0x10001u;
            uint32_t pr[4];
            if (sizeof(PixT) == 1) {
                // the 8-byte window never reaches past sample lim (the row of a block ends soon after it): groups near the tail read the
                // window that ENDS at lim and shift their byte selectors by o; pairs at or past lim are replaced below
                const int wb = min(idx, lim - 7), o = idx - wb;
                uint32_t d[2];
                __builtin_memcpy(d, ge + wb, 8);
                const uint32_t so = (uint32_t)o * 0x00010001u;
#pragma unroll
                for (int k = 0; k < 4; k++) pr[k] = __builtin_amdgcn_perm(d[1], d[0], (0x0c000c00u | ((uint32_t)(k + 1) << 16) | (uint32_t)k) + so);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) __builtin_memcpy(&pr[k], ge + min(idx + k, lim - 1), 4);      // samples (i, i + 1): one 4-byte load
            }
#pragma unroll
            for (int k = 0; k < 4; k++) pr[k] = idx + k < lim ? pr[k] : tail;
            uint2* o = reinterpret_cast<uint2*>((left_edge ? sl : sa) + idx);      // 8-byte aligned (NB_ORIGIN - 2 is even)
            o[0] = make_uint2(pr[0], pr[1]); o[1] = make_uint2(pr[2], pr[3]);
        }
    }
    // zone 2, one angle per launch, no up-sampling: the left-edge terms of column c - weight pair and 4 * ((-dy (c + 1)) >> 6) - are
    // the same in every row; 64 lanes tabulate them once per workgroup behind the staged edges (DirMulti documents the terms)
    const bool z2_lds_tab = MODE == IM_Z2 && !TAB && multi.n == 0 && (up_above | up_left) == 0;
    uint32_t* tabw = sm + (size_t)(256u >> __builtin_ctz(lpb)) * dir_slot_stride(estride, lpb);
    if (z2_lds_tab && threadIdx.x < 64) {
        const int ys = -dy * ((int)threadIdx.x + 1);
        const uint32_t sh = ((uint32_t)ys & 63u) >> 1;
        tabw[threadIdx.x] = (32u - sh) | (sh << 16);
        tabw[64 + threadIdx.x] = (uint32_t)(4 * (ys >> 6));
    }
    __syncthreads();
    const uint32_t* A = sa + NB_ORIGIN;
    const uint32_t* L = sl + NB_ORIGIN;
    // SAD mode: this lane's chunk of the source block, loaded once for all angles
    uint32_t srcw[4] = {0, 0, 0, 0};
    const bool sad_mode = sizeof(PixT) == 1 && multi.n != 0 && multi.sad_dist != nullptr;
    if (sad_mode && live) {
        const uint32_t q = multi.sad_xy[blk];
        const uint8_t* sp = multi.sad_pic + (size_t)((q >> 16) + (uint32_t)r) * multi.sad_stride + (q & 0xffffu) + (uint32_t)c0;
        if (ppl == 16) __builtin_memcpy(srcw, sp, 16); else if (ppl == 8) __builtin_memcpy(srcw, sp, 8); else __builtin_memcpy(srcw, sp, 4);
    }
    auto emit = [&](const int dx, const int dy, PixT* __restrict__ dst, const int kslot, const int kslot_tab) {
        uint32_t px[PXL];
        if (MODE == IM_Z1) {
            const int x = dx * (r + 1);
            const uint32_t sh = (uint32_t)(((x << up_above) & 0x3f) >> 1), w = (32u - sh) | (sh << 16);
            // a start at or past max_base reads only padding (= above[max_base]), as the reference's tail fill does
            const uint32_t* q = A + min((x >> (6 - up_above)) + (c0 << up_above), lim_a - NB_ORIGIN);
            if (up_above == 0) {
#pragma unroll
                for (int k = 0; k < PXL; k++) px[k] = dir_lerp2(q[k], w);
            } else {
#pragma unroll
                for (int k = 0; k < PXL; k++) px[k] = dir_lerp2(q[2 * k], w);
            }
        } else if (MODE == IM_Z3) {
            int y = dy * (c0 + 1);
            if (lanes_per_row == 1 && up_left == 0) {
                // one lane per row: y, its fraction and the weight pair of pixel k are the same in every lane (SGPRs)
#pragma unroll
                for (int k = 0; k < PXL; k++) {
                    const int ys = dy * (k + 1);
                    const uint32_t sh = (uint32_t)((ys & 0x3f) >> 1);
                    px[k] = dir_lerp2(L[min((ys >> 6) + r, lim_l - NB_ORIGIN)], (32u - sh) | (sh << 16));
                }
            } else
#pragma unroll
            for (int k = 0; k < PXL; k++) {
                const uint32_t sh = (uint32_t)(((y << up_left) & 0x3f) >> 1);
                px[k] = dir_lerp2(L[min((y >> (6 - up_left)) + (r << up_left), lim_l - NB_ORIGIN)], (32u - sh) | (sh << 16));
                y += dy;
            }
        } else {   // IM_Z2
            const int x = -dx * (r + 1);
            const uint32_t s1 = (uint32_t)(((x * (1 << up_above)) & 0x3f) >> 1);
            const int lim = -(1 << up_above);
            const int aoff = (int)(A - sm) + (x >> (6 - up_above)) + (c0 << up_above), loff = (int)(L - sm);
            int y = (r << 6) - dy * (c0 + 1);
            if ((up_above | up_left) == 0) {
                // no upsampling (every block larger than 8x8, and the whole open-loop search): constant shifts, LDS byte
                // addresses and the weight pair as one multiply-add - the kernel is VALU-bound on this per-pixel selection
                const uint32_t w1 = (32u - s1) | (s1 << 16);
                const int aoffB = aoff * 4, loffB = loff * 4;
                const char* smb = reinterpret_cast<const char*>(sm);
                auto row_px = [&](const int c0v) {
                    const int b0 = (x >> 6) + c0v;                             // base1 of pixel k = b0 + k; above edge when >= -1
                    int y = (r << 6) - dy * (c0v + 1);
#pragma unroll
                    for (int k = 0; k < PXL; k++) {
                        const bool ab = b0 >= -1 - k;
                        const uint32_t s2 = ((uint32_t)y >> 1) & 31u;          // (y & 0x3f) >> 1
                        const uint32_t w2 = s2 * 0xffffu + 32u;                // (32 - s2) | s2 << 16
                        const int offL = loffB + ((y >> 4) & ~3);              // 4 * (loff + (y >> 6))
                        const int off = ab ? aoffB + 4 * k : offL;
                        px[k] = dir_lerp2(*reinterpret_cast<const uint32_t*>(smb + off), ab ? w1 : w2);
                        y -= dy;
                    }
                };
                if constexpr (TAB) {
                    // LDS byte addresses as integers (the pointer form costs an add of the LDS base per read)
                    typedef const __attribute__((address_space(3))) uint32_t* LdsWord;
                    const uint32_t ldsb = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)smb);
                    const uint32_t aB = ldsb + (uint32_t)aoffB, lB = ldsb + (uint32_t)(loffB + (r << 2));
                    const int xb = x >> 6;
                    // the angle's two table rows, fetched unconditionally into SGPRs (left to itself the compiler sinks each
                    // element's scalar load into a branch on `ab`: two exec-mask branches and two waits per pixel)
                    uint32_t tw[PXL], tol[PXL];
#pragma unroll
                    for (int k = 0; k < PXL; k++) { tw[k] = multi.z2_w2[kslot_tab][k]; tol[k] = (uint32_t)multi.z2_ol[kslot_tab][k]; }
#pragma unroll
                    for (int k = 0; k < PXL; k++) asm volatile("" : "+s"(tw[k]), "+s"(tol[k]));
                    uint32_t ev[PXL];
#pragma unroll
                    for (int k = 0; k < PXL; k++) ev[k] = *(LdsWord)(uintptr_t)(xb >= -1 - k ? aB + 4u * (uint32_t)k : lB + tol[k]);
#pragma unroll
                    for (int k = 0; k < PXL; k++) px[k] = dir_lerp2(ev[k], xb >= -1 - k ? w1 : tw[k]);
                } else if (z2_lds_tab) {
                    typedef const __attribute__((address_space(3))) uint32_t* LdsWord;
                    const uint32_t ldsb = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)smb);
                    const uint32_t aB = ldsb + (uint32_t)aoffB, lB = ldsb + (uint32_t)(loffB + (r << 2));
                    const int b0 = (x >> 6) + c0;
                    uint32_t tw[PXL], tol[PXL];
#pragma unroll
                    for (int q = 0; q < PXL / 4; q++) {
                        const uint4 a = *reinterpret_cast<const uint4*>(tabw + c0 + 4 * q), b = *reinterpret_cast<const uint4*>(tabw + 64 + c0 + 4 * q);
                        tw[4 * q] = a.x; tw[4 * q + 1] = a.y; tw[4 * q + 2] = a.z; tw[4 * q + 3] = a.w;
                        tol[4 * q] = b.x; tol[4 * q + 1] = b.y; tol[4 * q + 2] = b.z; tol[4 * q + 3] = b.w;
                    }
                    uint32_t ev[PXL];
#pragma unroll
                    for (int k = 0; k < PXL; k++) ev[k] = *(LdsWord)(uintptr_t)(b0 >= -1 - k ? aB + 4u * (uint32_t)k : lB + tol[k]);
#pragma unroll
                    for (int k = 0; k < PXL; k++) px[k] = dir_lerp2(ev[k], b0 >= -1 - k ? w1 : tw[k]);
                } else row_px(c0);
            } else
#pragma unroll
            for (int k = 0; k < PXL; k++) {
                const int base1 = (x >> (6 - up_above)) + ((c0 + k) << up_above);
                const bool ab = base1 >= lim;
                const uint32_t s2 = (uint32_t)(((y * (1 << up_left)) & 0x3f) >> 1);
                const int idx = ab ? aoff + (k << up_above) : loff + (y >> (6 - up_left));
                const uint32_t sh = ab ? s1 : s2;
                px[k] = dir_lerp2(sm[idx], (32u - sh) | (sh << 16));
                y -= dy;
            }
        }
        if (live) {
            if (sizeof(PixT) == 2) {
#pragma unroll
                for (int k = 0; k < PXL; k++) px[k] = min(px[k], maxv);
            }
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < NW; q++) {
                uint32_t v = px[q * PPW];
#pragma unroll
                for (int t = 1; t < PPW; t++) v |= px[q * PPW + t] << (8 * (int)sizeof(PixT) * t);
                w[q] = v;
            }
            if (!sad_mode) {
                const size_t base_off = has_offs ? (size_t)off_word : (size_t)blk * dst_block_pitch;
                intra_store<PixT>(dst + base_off + (size_t)r * dst_stride + c0, make_uint4(w[0], w[1], w[2], w[3]), ppl);
            } else {
                // lanes of a block are per_block (4 .. 64) consecutive lanes of one wave: group sum (DPP inside a row), lane 0 writes
                uint32_t sad = __builtin_amdgcn_sad_u8(srcw[0], w[0], 0u);
                if (ppl >= 8) sad = __builtin_amdgcn_sad_u8(srcw[1], w[1], sad);
                if (ppl == 16) { sad = __builtin_amdgcn_sad_u8(srcw[2], w[2], sad); sad = __builtin_amdgcn_sad_u8(srcw[3], w[3], sad); }
                sad = group_sum_rt(sad, per_block);
                if (j == 0) multi.sad_dist[(size_t)blk * multi.sad_ncand + (uint32_t)kslot] = sad;
            }
        }
    };
    if (multi.n == 0) emit(dx, dy, dst, 0, 0);
    else {
        const int k0 = (int)ypart * multi.chunk, k1 = min(multi.n, k0 + multi.chunk);
        for (int k = k0; k < k1; k++) emit(multi.dx[k], multi.dy[k], dst + (size_t)multi.slot[k] * multi.batch_pitch, (int)multi.slot[k], k);
    }
}

template <typename PixT, int MODE, int NPX, bool TAB = false>
__global__ __launch_bounds__(256) void intra_dir_kernel(
    PixT* __restrict__ dst, int32_t dst_stride, size_t dst_block_pitch, const uint32_t* __restrict__ dst_offsets,
    const PixT* __restrict__ above_all, const PixT* __restrict__ left_all, int32_t nb_pitch, int bw, int bh,
    int up_above, int up_left, int dx, int dy, int lim_a, int lim_l, int n_pad, int bd, uint32_t nblocks, const DirMulti multi) {
    intra_dir_body<PixT, MODE, NPX, TAB>(dst, dst_stride, dst_block_pitch, dst_offsets, above_all, left_all, nb_pitch, bw, bh, up_above, up_left, dx, dy,
                                         lim_a, lim_l, n_pad, bd, nblocks, multi, blockIdx.y);
}

// The open-loop intra search's directional candidates of ALL THREE zones in one launch (8-bit SAD mode, square blocks of one lane per
// row: 8x8, 16x16): blockIdx.y walks zone 1's angle chunks, then zone 2's, then zone 3's.  As three launches each zone was one
// residency round of latency-bound waves (12 / 21 / 11 us per 1080p picture at 8x8, the VALU busy for 6 / 14 / 6 of them,
// profiles/r03_ois8_pmc_after.json): one launch lets the zones fill each other's stalls and drops two launch boundaries.
struct DirOis3 {
    DirMultiLite z1, z3;
    DirMulti z2;
    uint32_t y_end[2];           // zone 1 takes blockIdx.y in [0, y_end[0]), zone 2 [y_end[0], y_end[1]), zone 3 the rest
};
static_assert(sizeof(DirOis3) <= 3900, "kernel arguments");
template <int NPX>
__global__ __launch_bounds__(256) void ois_dir3_kernel(const uint8_t* __restrict__ above_all, const uint8_t* __restrict__ left_all, int32_t nb_pitch,
                                                       int bsize, int lim, int n_pad, uint32_t nblocks, const DirOis3 m) {
    const uint32_t y = blockIdx.y;
    if (y < m.y_end[0])
        intra_dir_body<uint8_t, IM_Z1, NPX, false>(nullptr, bsize, 0, nullptr, above_all, left_all, nb_pitch, bsize, bsize, 0, 0, 1, 1, lim, lim, n_pad, 8,
                                                   nblocks, m.z1, y);
    else if (y < m.y_end[1])
        intra_dir_body<uint8_t, IM_Z2, NPX, true>(nullptr, bsize, 0, nullptr, above_all, left_all, nb_pitch, bsize, bsize, 0, 0, 1, 1, lim, lim, n_pad, 8,
                                                  nblocks, m.z2, y - m.y_end[0]);
    else
        intra_dir_body<uint8_t, IM_Z3, NPX, false>(nullptr, bsize, 0, nullptr, above_all, left_all, nb_pitch, bsize, bsize, 0, 0, 1, 1, lim, lim, n_pad, 8,
                                                   nblocks, m.z3, y - m.y_end[1]);
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


}  // namespace svtdev
