// kernel_bip.h — build_intra_predictors / build_intra_predictors_high (EbIntraPrediction.c:3667-3855, 3857-4076) for a
// batch of prediction blocks of one transform size, fused into one launch: edge selection and extension, corner, the
// directional modes' smoothing filters and 2x up-sampling, DC by availability, and the prediction.
//
// A group of LPB = 4 .. 64 lanes per block (the smallest power of two that covers a block's w + h + 1 edge samples in three
// rounds), 64 / LPB blocks per wave, four waves per workgroup: a wave's life is a chain of dependent loads (descriptor -> edge
// samples -> LDS stages -> stores) whatever the block size, so small blocks share it (one 16x16 block per wave: 1.3 G blocks/s;
// four: see DESIGN 4.15).  Both edges of a block live in LDS as 16-bit samples and go through the reference's stages in place;
// every stage is a read phase into registers, a wave-level LDS fence (a wave's edges are its own: no workgroup barrier) and a
// write phase, so the result is what the reference's sequential loops produce.  The stage sequence (and so the number of
// fences) is the same for every block whatever its mode; stages a block does not need are predicated off.  Output: 4 samples
// per lane per step (one 4- or 8-byte store when the address allows).
//
// Round 3.  PMC on the first version (profiles/r02_pmc_c_bip.json): 1 400 VALU instructions per wave of four 16x16 blocks, 0.07 of
// HBM - every wave ran every stage (a stage is predicated per lane, but the wave issues it if ONE of its blocks needs it) and the
// per-pixel `switch` over the predictor kind was executed once per kind present in the wave.  Now:
//   * the block size is a template parameter (loops unrolled, no run-time divisions);
//   * svt_hip_intra_order_blocks_batch (bip_order_*_kernel below) sorts the batch's block indices by predictor kind on the
//     device (count, scan of 13 bins, scatter) and the kernel walks the batch through that order: a wave's blocks then share
//     a kind except at the 12 class boundaries;
//   * stages 2 - 4 are skipped by the whole wave when none of its blocks is angled / up-sampled (a ballot), and a wave whose
//     blocks share a kind runs that kind's own pixel loop - the `switch` is taken once per wave, on a scalar.
// It stays the encode-pass glue, not the search loop; the dense per-mode kernels (kernel_intra.h) are the specialised form.
#pragma once
#include <type_traits>
#include "kernel_intra.h"

namespace svtdev {

struct BipBlk {                 // == svt_hip_intra_blk
    uint8_t mode;
    int8_t angle_delta;
    uint8_t filt_type, disable_edge_filter;
    uint8_t n_top_px, n_topright_px, n_left_px, n_bottomleft_px;
};
static_assert(sizeof(BipBlk) == 8, "descriptor layout");

// dr_intra_derivative (EbIntraPrediction.c:299; AV1 spec 7.11.2.4): defined at the 27 angles the prediction can take (1 elsewhere, as
// nothing reads it there).  A table in constant memory, one load per lane: as a chain of 27 compare-and-select pairs it cost the
// wave 54 VALU instructions per call, two calls for zone 2 - a quarter of an angled wave's instructions for a per-BLOCK value.  The
// load is issued when the descriptor arrives and is first needed in the pixel stage.
__device__ __constant__ uint16_t kBipDrDerivative[92] = {
    1, 1, 1, 1023, 1, 1, 547, 1, 1, 372, 1, 1, 1, 1, 273, 1, 1, 215, 1, 1, 178, 1, 1,
    151, 1, 1, 132, 1, 1, 116, 1, 1, 102, 1, 1, 1, 90, 1, 1, 80, 1, 1, 71, 1, 1, 64,
    1, 1, 57, 1, 1, 51, 1, 1, 45, 1, 1, 1, 40, 1, 1, 35, 1, 1, 31, 1, 1, 27, 1,
    1, 23, 1, 1, 19, 1, 1, 15, 1, 1, 1, 1, 11, 1, 1, 7, 1, 1, 3, 1, 1, 1, 1,
};
__device__ __forceinline__ int bip_dr_derivative(int angle) { return (int)kBipDrDerivative[min(max(angle, 0), 91)]; }
// intra_edge_filter_strength (:225-268)
__device__ __forceinline__ int bip_filter_strength(int bs0, int bs1, int delta, int type) {
    const int d = abs(delta), wh = bs0 + bs1;
    int s = 0;
    if (type == 0) {
        if (wh <= 8) s = d >= 56;
        else if (wh <= 16) s = d >= 40;
        else if (wh <= 24) s = d >= 32 ? 3 : (d >= 16 ? 2 : (d >= 8 ? 1 : 0));
        else if (wh <= 32) s = d >= 32 ? 3 : (d >= 4 ? 2 : (d >= 1 ? 1 : 0));
        else s = d >= 1 ? 3 : 0;
    } else {
        if (wh <= 8) s = d >= 64 ? 2 : (d >= 40 ? 1 : 0);
        else if (wh <= 16) s = d >= 48 ? 2 : (d >= 20 ? 1 : 0);
        else if (wh <= 24) s = d >= 4 ? 3 : 0;
        else s = d >= 1 ? 3 : 0;
    }
    return s;
}
// use_intra_edge_upsample (:167-172)
__device__ __forceinline__ int bip_use_upsample(int bs0, int bs1, int delta, int type) {
    const int d = abs(delta), wh = bs0 + bs1;
    if (d <= 0 || d >= 40) return 0;
    return type ? (wh <= 8) : (wh <= 16);
}

constexpr int BIP_WAVES = 4;
// staged samples per edge: positions [-16, n): n covers w + h samples, twice that where the edge can be up-sampled (w + h <= 16)
__host__ __device__ constexpr int bip_edge_len(int w, int h) { return (16 + (w + h <= 16 ? 2 * (w + h) : w + h) + 8 + 7) & ~7; }
// lanes per block: w + h + 1 edge samples in at most three rounds (the smoothing stage keeps three values per lane)
__host__ __device__ constexpr int bip_lanes_per_block(int w, int h) {
    const int need = (w + h + 1 + 2) / 3;
    int l = 4;
    while (l < need) l <<= 1;
    return l > 64 ? 64 : l;
}

// the predictor kind a block resolves to (stage 5 of bip_kernel; also the sort key of the ordering pass)
__device__ __forceinline__ int bip_kind_of(const BipBlk& d, int w, int h, int& p_angle_out) {
    const int mode = d.mode > 12 ? 12 : d.mode;
    const int n_top = min((int)d.n_top_px, w), n_left = min((int)d.n_left_px, h);
    const bool is_dr = mode >= 1 && mode <= 8;
    int p_angle = 0;
    bool need_left = true, need_above = true;
    if (is_dr) {
        const int delta = max(-3, min(3, (int)d.angle_delta));
        const int ang[9] = {0, 90, 180, 45, 135, 113, 157, 203, 67};      // mode_to_angle_map (EbCodingUnit.h:129)
        int a0 = 0;
#pragma unroll
        for (int i = 1; i < 9; i++) a0 = i == mode ? ang[i] : a0;
        p_angle = a0 + 3 * delta;
        need_above = p_angle < 180; need_left = p_angle > 90;
    }
    p_angle_out = p_angle;
    const bool const_fill = (!need_above && n_left == 0) || (!need_left && n_top == 0);
    if (const_fill) return IM_DC_128;
    if (is_dr) return p_angle == 90 ? IM_V : (p_angle == 180 ? IM_H : (p_angle < 90 ? IM_Z1 : (p_angle < 180 ? IM_Z2 : IM_Z3)));
    if (mode == 0) return n_left > 0 ? (n_top > 0 ? IM_DC : IM_DC_LEFT) : (n_top > 0 ? IM_DC_TOP : IM_DC_128);       // dc_pred[left][top], :3851
    return mode == 9 ? IM_SMOOTH : (mode == 10 ? IM_SMOOTH_V : (mode == 11 ? IM_SMOOTH_H : IM_PAETH));
}

// ---- ordering pass: block indices grouped by kind inside TILES of BIP_ORDER_TILE consecutive blocks (a counting sort over the
// IM_MODES = 13 bins in LDS; the order inside a bin is whatever the atomics give - the prediction of a block does not depend on its
// place in the order).  One kernel, no global counters: the first version sorted the whole batch (count kernel + scatter kernel, 17 +
// 20 us per 2^20 blocks, each a single residency round of 256 workgroups waiting on 13 contended global atomics); a wave only needs
// ITS four blocks to share a kind, and a tile of 4 096 blocks holds ~ 315 of each, so the waves of a tile are uniform except at its
// (at most 12) kind boundaries - 1.2 % of the waves; a mixed wave runs every kind it holds, several times a uniform wave's cost
// (tiles of 1 024 blocks measured no faster than the global sort for that reason).  1 024 threads per workgroup: 256 workgroups of
// 16 waves keep enough loads in flight.
constexpr int BIP_ORDER_ITEMS = 4;
constexpr int BIP_ORDER_THREADS = 1024;
constexpr int BIP_ORDER_TILE = BIP_ORDER_THREADS * BIP_ORDER_ITEMS;
__global__ __launch_bounds__(BIP_ORDER_THREADS) void bip_order_tile_kernel(const BipBlk* __restrict__ blks, int w, int h, uint32_t* __restrict__ order, uint32_t nblocks) {
    __shared__ uint32_t s_cnt[16], s_base[16];
    if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)BIP_ORDER_TILE;
    BipBlk d[BIP_ORDER_ITEMS];
#pragma unroll
    for (int k = 0; k < BIP_ORDER_ITEMS; k++) {                 // the tile's descriptors, coalesced, all loads in flight together
        const uint32_t i = base + k * (uint32_t)BIP_ORDER_THREADS + threadIdx.x;
        d[k] = blks[i < nblocks ? i : 0u];
    }
    uint32_t kind[BIP_ORDER_ITEMS], local[BIP_ORDER_ITEMS];
#pragma unroll
    for (int k = 0; k < BIP_ORDER_ITEMS; k++) {
        const uint32_t i = base + k * (uint32_t)BIP_ORDER_THREADS + threadIdx.x;
        int pa;
        kind[k] = (uint32_t)bip_kind_of(d[k], w, h, pa);
        local[k] = i < nblocks ? atomicAdd(&s_cnt[kind[k]], 1u) : 0u;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        uint32_t start = 0;                                     // bin start = the counts of the bins before it
        for (int k = 0; k < (int)threadIdx.x; k++) start += s_cnt[k];
        s_base[threadIdx.x] = start;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BIP_ORDER_ITEMS; k++) {
        const uint32_t i = base + k * (uint32_t)BIP_ORDER_THREADS + threadIdx.x;
        if (i < nblocks) order[base + s_base[kind[k]] + local[k]] = i;
    }
}

template <typename PixT, int W, int H>
__global__ __launch_bounds__(64 * BIP_WAVES) void bip_kernel(
    PixT* __restrict__ dst, int32_t dst_stride, size_t dst_block_pitch, const uint32_t* __restrict__ dst_offsets,
    const PixT* __restrict__ top_all, const PixT* __restrict__ left_all, int32_t neigh_pitch, const BipBlk* __restrict__ blks,
    const uint32_t* __restrict__ order, int bd, uint32_t nblocks) {
    constexpr int w = W, h = H;
    constexpr int LPB = bip_lanes_per_block(W, H), bpw = 64 / LPB, EL = bip_edge_len(W, H);
    constexpr int lsh = LPB == 4 ? 2 : (LPB == 8 ? 3 : (LPB == 16 ? 4 : (LPB == 32 ? 5 : 6)));
    __shared__ __attribute__((aligned(16))) uint16_t s_edge[BIP_WAVES * bpw * 2 * EL];          // [wave][block of the wave][above | left][EL]
    // the directional pixel loops read the edges in PAIR form (dword i = edge[i] | edge[i + 1] << 16, kernel_intra.h): built once per
    // block after the edge stages, clamped at the last valid sample - a pixel is then one ds_read_b32 + one v_dot2_u32_u16 and the
    // reference's "base >= max_base -> edge[max_base]" falls out of the same interpolation ((32 e + 16) >> 5 == e)
    __shared__ __attribute__((aligned(16))) uint32_t s_pair[BIP_WAVES * bpw * 2 * EL];
    const int wv = threadIdx.x >> 6, sub = (threadIdx.x & 63) >> lsh;
    const int lane = threadIdx.x & (LPB - 1);                 // lane inside the block's group
    const uint32_t blk_id = (blockIdx.x * BIP_WAVES + (uint32_t)wv) * (uint32_t)bpw + (uint32_t)sub;
    const bool live = blk_id < nblocks;
    const uint32_t b = live ? (order ? order[blk_id] : blk_id) : (order ? order[0] : 0u);      // a spare group replays a block without storing
    const PixT* __restrict__ top = top_all + (size_t)b * neigh_pitch + 1;       // element 0 is the corner: top[-1]
    const PixT* __restrict__ left = left_all + (size_t)b * neigh_pitch + 1;
    // Every global load of the block is issued here, side by side: the descriptor AND the raw w + h samples of both edges (the row
    // of a block holds them whatever is "available": neigh_pitch >= 1 + 2 max(w, h)).  Which of them count, and what replaces the
    // others, is applied in LDS below - the first version loaded the descriptor, then the samples it selected (top[min(i, avail - 1)]):
    // two dependent round trips to HBM in a kernel whose waves do little else than wait for them.
    constexpr int NR = (W + H + LPB - 1) / LPB;               // rounds of LPB lanes over w + h samples (<= 3)
    int raw_t[NR], raw_l[NR];
#pragma unroll
    for (int t = 0; t < NR; t++) {
        const int i = lane + t * LPB;
        raw_t[t] = i < W + H ? (int)top[i] : 0;
        raw_l[t] = i < W + H ? (int)left[i] : 0;
    }
    const int top_m1 = (int)top[-1], top_0 = (int)top[0], left_0 = (int)left[0];
    const BipBlk d = blks[b];
    uint16_t* A = s_edge + (size_t)((wv * bpw + sub) * 2) * EL + 16;
    uint16_t* L = A + EL;
    uint32_t* PA = s_pair + (size_t)((wv * bpw + sub) * 2) * EL + 16;
    uint32_t* PL = PA + EL;
    const int maxv = (1 << bd) - 1, base = 128 << (bd - 8);

    const int mode = d.mode > 12 ? 12 : d.mode;
    const int n_top = min((int)d.n_top_px, w), n_left = min((int)d.n_left_px, h);
    // (the edge loops below never read past need - 1 <= w + h - 1, so the two corner counts need no clamp)
    const int n_tr = d.n_topright_px, n_bl = d.n_bottomleft_px;
    const int ft = d.filt_type ? 1 : 0;
    const bool is_dr = mode >= 1 && mode <= 8;
    // extend_modes (:1410-1424) and the directional angle classes (:3702-3722)
    bool need_left, need_above, need_above_left, need_right, need_bottom;
    int p_angle = 0;
    if (is_dr) {
        const int delta = max(-3, min(3, (int)d.angle_delta));
        const int ang[9] = {0, 90, 180, 45, 135, 113, 157, 203, 67};      // mode_to_angle_map (EbCodingUnit.h:129)
        int a0 = 0;
#pragma unroll
        for (int i = 1; i < 9; i++) a0 = i == mode ? ang[i] : a0;
        p_angle = a0 + 3 * delta;
        need_above = p_angle < 180; need_left = p_angle > 90; need_above_left = true;
        need_right = p_angle < 90; need_bottom = p_angle > 180;
    } else {
        need_left = true; need_above = true;                  // DC, SMOOTH*, PAETH (V / H are directional here)
        need_above_left = mode == 12;
        need_right = false; need_bottom = false;
    }
    // dx / dy of the angled kinds (one table load each, in flight while the edge stages run): zone 1 dx(p), zone 2 dx(180 - p) and
    // dy(p - 90), zone 3 dy(270 - p); every other argument reads 1
    const int dx = bip_dr_derivative(p_angle < 90 ? p_angle : 180 - p_angle), dy = bip_dr_derivative(p_angle < 180 ? p_angle - 90 : 270 - p_angle);
    const bool const_fill = (!need_above && n_left == 0) || (!need_left && n_top == 0);
    const int const_val = need_left ? (n_top > 0 ? top_0 : base + 1) : (n_left > 0 ? left_0 : base - 1);

    // ---- stage 1: edge extension (:3747-3800) -----------------------------------------------------------------
    // raw samples to LDS, then: a missing edge takes its default, samples past the available run the last available one (that
    // position is never written in the second phase, so it needs no buffer of its own)
#pragma unroll
    for (int t = 0; t < NR; t++) {
        const int i = lane + t * LPB;
        if (i < W + H) { A[i] = (uint16_t)raw_t[t]; L[i] = (uint16_t)raw_l[t]; }
    }
    wave_lds_fence();
    {
        const int need_l = need_left ? h + (need_bottom ? w : 0) : 0;
        const int avail_l = (need_bottom && n_bl > 0) ? h + n_bl : n_left;
        const int def_l = n_top > 0 ? top_0 : base + 1;
        const int need_a = need_above ? w + (need_right ? h : 0) : 0;
        const int avail_a = (need_right && n_tr > 0) ? n_top + n_tr : n_top;
        const int def_a = n_left > 0 ? left_0 : base - 1;
#pragma unroll
        for (int t = 0; t < NR; t++) {
            const int i = lane + t * LPB;
            if (i < need_l) { if (n_left == 0) L[i] = (uint16_t)def_l; else if (i >= avail_l) L[i] = L[avail_l - 1]; }
            if (i < need_a) { if (n_top == 0) A[i] = (uint16_t)def_a; else if (i >= avail_a) A[i] = A[avail_a - 1]; }
        }
        if (lane == 0 && need_above_left) {
            const int c = (n_top > 0 && n_left > 0) ? top_m1 : (n_top > 0 ? top_0 : (n_left > 0 ? left_0 : base));
            A[-1] = (uint16_t)c; L[-1] = (uint16_t)c;
        }
    }
    wave_lds_fence();
    const bool filt = is_dr && !d.disable_edge_filter && !const_fill;
    const bool angled = filt && p_angle != 90 && p_angle != 180;
    const bool wave_angled = __builtin_amdgcn_ballot_w64(angled) != 0;      // stages 2 - 4 are skipped by a wave none of whose blocks needs them
    // ---- stage 2: corner filter (filter_intra_edge_corner, :3383) ----------------------------------------------------
    if (wave_angled) {
    if (angled && need_above && need_left && w + h >= 24 && lane == 0) {
        const int s = ((int)L[0] * 5 + (int)A[-1] * 6 + (int)A[0] * 5 + 8) >> 4;
        A[-1] = (uint16_t)s; L[-1] = (uint16_t)s;
    }
    wave_lds_fence();
    }
    // ---- stage 3: edge smoothing (av1_filter_intra_edge, :3539): sample 0 of the run is kept ---------------------------
    if (wave_angled) {
        const int ab_le = need_above_left ? 1 : 0;
        const int sa = (angled && need_above && n_top > 0) ? bip_filter_strength(w, h, p_angle - 90, ft) : 0;
        const int sl = (angled && need_left && n_left > 0) ? bip_filter_strength(h, w, p_angle - 180, ft) : 0;
        const int na = n_top + ab_le + (need_right ? h : 0), nl = n_left + ab_le + (need_bottom ? w : 0);      // <= 129
        int va[3], vl[3];
        auto taps = [](const uint16_t* p, int i, int sz, int st) {
            const int k0 = st == 3 ? 2 : 0, k1 = st == 2 ? 5 : 4, k2 = st == 1 ? 8 : (st == 2 ? 6 : 4);
            auto at = [&](int q) { return (int)p[min(max(q, 0), sz - 1)]; };
            return (k0 * at(i - 2) + k1 * at(i - 1) + k2 * at(i) + k1 * at(i + 1) + k0 * at(i + 2) + 8) >> 4;
        };
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int i = lane + LPB * t;
            va[t] = (sa && i >= 1 && i < na) ? taps(A - ab_le, i, na, sa) : -1;
            vl[t] = (sl && i >= 1 && i < nl) ? taps(L - ab_le, i, nl, sl) : -1;
        }
        wave_lds_fence();
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int i = lane + LPB * t;
            if (va[t] >= 0) A[i - ab_le] = (uint16_t)va[t];
            if (vl[t] >= 0) L[i - ab_le] = (uint16_t)vl[t];
        }
    }
    wave_lds_fence();
    // ---- stage 4: 2x up-sampling (av1_upsample_intra_edge, :3597): p[-2 .. 2 sz - 2] from p[-1 .. sz - 1] ---------------
    // (use_intra_edge_upsample is 0 for every block with w + h > 16: there the shifts below are compile-time constants)
    constexpr bool CAN_UP = W + H <= 16;
    const int up_a = (CAN_UP && filt && need_above) ? bip_use_upsample(w, h, p_angle - 90, ft) : 0;
    const int up_l = (CAN_UP && filt && need_left) ? bip_use_upsample(h, w, p_angle - 180, ft) : 0;
    if (w + h <= 16 && __builtin_amdgcn_ballot_w64((up_a | up_l) != 0) != 0) {
        const int sza = w + (need_right ? h : 0), szl = h + (need_bottom ? w : 0);                 // <= 16 when up-sampling
        int ia[2][4], il[2][4];                                 // two rounds of LPB lanes: sz <= 16 <= 2 * LPB wherever the edge is up-sampled
        auto in_at = [](const uint16_t* p, int k, int sz) { return (int)p[k < 2 ? -1 : (k < sz + 2 ? k - 2 : sz - 1)]; };
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int i = lane + LPB * t;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                ia[t][k] = (up_a && i < sza) ? in_at(A, i + k, sza) : 0;
                il[t][k] = (up_l && i < szl) ? in_at(L, i + k, szl) : 0;
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int i = lane + LPB * t;
            if (up_a && i < sza) {
                if (i == 0) A[-2] = (uint16_t)ia[t][0];
                A[2 * i - 1] = (uint16_t)min(max((-ia[t][0] + 9 * ia[t][1] + 9 * ia[t][2] - ia[t][3] + 8) >> 4, 0), maxv);
                A[2 * i] = (uint16_t)ia[t][2];
            }
            if (up_l && i < szl) {
                if (i == 0) L[-2] = (uint16_t)il[t][0];
                L[2 * i - 1] = (uint16_t)min(max((-il[t][0] + 9 * il[t][1] + 9 * il[t][2] - il[t][3] + 8) >> 4, 0), maxv);
                L[2 * i] = (uint16_t)il[t][2];
            }
        }
    wave_lds_fence();
    }
    // ---- stage 5: prediction ---------------------------------------------------------------------------------------------
    // resolve to one of the predictor kinds
    int kind;
    if (const_fill) kind = IM_DC_128;
    else if (is_dr) kind = p_angle == 90 ? IM_V : (p_angle == 180 ? IM_H : (p_angle < 90 ? IM_Z1 : (p_angle < 180 ? IM_Z2 : IM_Z3))); else if (mode == 0) kind = n_left > 0 ? (n_top > 0 ? IM_DC : IM_DC_LEFT) : (n_top > 0 ? IM_DC_TOP : IM_DC_128);       // dc_pred[left][top], :3851
    else kind = mode == 9 ? IM_SMOOTH : (mode == 10 ? IM_SMOOTH_V : (mode == 11 ? IM_SMOOTH_H : IM_PAETH));
    // pair form of the edges for the angled kinds (the wave skips it when none of its blocks is one)
    const bool is_z = kind == IM_Z1 || kind == IM_Z2 || kind == IM_Z3;
    // last valid staged sample of each edge: without up-sampling [-1, need - 1], with it [-2, 2 need - 2] (av1_upsample_intra_edge);
    // for zone 1 / 3 that is max_base = (w + h - 1) << up
    const int need_a_n = need_above ? w + (need_right ? h : 0) : 0, need_l_n = need_left ? h + (need_bottom ? w : 0) : 0;
    const int last_a = up_a ? 2 * need_a_n - 2 : need_a_n - 1, last_l = up_l ? 2 * need_l_n - 2 : need_l_n - 1;
    if (__builtin_amdgcn_ballot_w64(is_z) != 0) {
        if (is_z && kind != IM_Z3)
            for (int i = lane - 2; i <= last_a; i += LPB) PA[i] = (uint32_t)A[i] | ((uint32_t)A[min(i + 1, last_a)] << 16);
        if (is_z && kind != IM_Z1)
            for (int i = lane - 2; i <= last_l; i += LPB) PL[i] = (uint32_t)L[i] | ((uint32_t)L[min(i + 1, last_l)] << 16);
        wave_lds_fence();
    }
    int dc = const_fill ? const_val : base;
    if (kind == IM_DC || kind == IM_DC_TOP || kind == IM_DC_LEFT) {
        const int na = kind != IM_DC_LEFT ? w : 0, nl = kind != IM_DC_TOP ? h : 0;
        int sum = 0;
        for (int i = lane; i < na + nl; i += LPB) sum += i < na ? (int)A[i] : (int)L[i - na];
        for (int m = LPB >> 1; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
        dc = (sum + ((na + nl) >> 1)) / (na + nl);
    }
    const size_t boff = dst_offsets ? (size_t)dst_offsets[b] : (size_t)b * dst_block_pitch;
    PixT* __restrict__ out = dst + boff;
    // A lane writes PPL = min(W, 16) samples of one row per step: one address, one 4 .. 32-byte store (16 bytes per lane is the store
    // shape the HBM likes) and a quarter of the loop overhead of the first version's 4 samples per step.
    constexpr int PPL = W >= 16 ? 16 : W, CPR = W / PPL, items = CPR * H;
    const int bl_s = (int)L[h - 1], tr_s = (int)A[w - 1], tl_s = (int)A[-1];
    // KC::value >= 0: every block of the wave has this kind - the switch below is resolved at compile time; -1: mixed wave
    auto predict = [&](auto KC) {
    constexpr int KU = decltype(KC)::value;
#pragma unroll
    for (int q0 = 0; q0 < items; q0 += LPB) {
        const int q = q0 + lane;
        if (items % LPB != 0 && q >= items) break;
        const int r = q / CPR, c0 = (q % CPR) * PPL;
        const int kk = KU >= 0 ? KU : kind;
        // the row's samples of the above edge as 16-bit pairs (aligned: c0 is a multiple of PPL, the edge arrays start 16-byte aligned)
        uint32_t aw[PPL / 2];
        if (KU < 0 || KU == IM_V || KU == IM_SMOOTH || KU == IM_SMOOTH_V || KU == IM_PAETH) {
            const uint32_t* ap = reinterpret_cast<const uint32_t*>(A + c0);
#pragma unroll
            for (int k = 0; k < PPL / 2; k++) aw[k] = ap[k];
        }
        const int lr = (int)L[r < H ? r : 0];
        const int whr = kSmWeights[h + r];
        // zone 1 / 2: the row's position on the above edge and its weight pair
        const int z_x = (kk == IM_Z2 ? -dx : dx) * (r + 1);
        const int z_b = z_x >> (6 - up_a);
        const uint32_t z_sh = (uint32_t)(((z_x * (1 << up_a)) & 0x3f) >> 1), z_w = z_sh * 0xffffu + 32u;
        int px[PPL];
#pragma unroll
        for (int k = 0; k < PPL; k++) {
            const int c = c0 + k;
            const int ac = (int)((aw[k >> 1] >> (16 * (k & 1))) & 0xffffu);
            int v;
            switch (kk) {
            case IM_V: v = ac; break;
            case IM_H: v = lr; break;
            case IM_SMOOTH: {
                const int ww = kSmWeights[w + c];
                v = (whr * ac + (256 - whr) * bl_s + ww * lr + (256 - ww) * tr_s + 256) >> 9;
            } break;
            case IM_SMOOTH_V: v = (whr * ac + (256 - whr) * bl_s + 128) >> 8; break;
            case IM_SMOOTH_H: { const int ww = kSmWeights[w + c]; v = (ww * lr + (256 - ww) * tr_s + 128) >> 8; } break;
            case IM_PAETH: {
                const int t = ac, l = lr, pb = t + l - tl_s;
                const int pl = abs(pb - l), pt = abs(pb - t), ptl = abs(pb - tl_s);
                v = (pl <= pt && pl <= ptl) ? l : (pt <= ptl ? t : tl_s);
            } break;
            case IM_Z1: {     // av1_dr_prediction_z1 (:370 / :3394): base >= max_base reads the pair (e, e) at max_base
                v = (int)dir_lerp2(PA[min(z_b + (c << up_a), last_a)], z_w);
                if (sizeof(PixT) == 2) v = min(v, maxv);
            } break;
            case IM_Z3: {     // av1_dr_prediction_z3 (:447 / :3475)
                const int y = dy * (c + 1), bs = (y >> (6 - up_l)) + (r << up_l);
                const uint32_t sh = (uint32_t)(((y << up_l) & 0x3f) >> 1);
                v = (int)dir_lerp2(PL[min(bs, last_l)], sh * 0xffffu + 32u);          // (32 - sh) | sh << 16
                if (sizeof(PixT) == 2) v = min(v, maxv);
            } break;
            case IM_Z2: {     // av1_dr_prediction_z2 (:405 / :3431)
                const int base1 = z_b + (c << up_a);
                const bool ab = base1 >= -(1 << up_a);
                const int y = (r << 6) - dy * (c + 1), base2 = y >> (6 - up_l);
                const uint32_t sh2 = (uint32_t)(((y * (1 << up_l)) & 0x3f) >> 1);
                const uint32_t* pp = ab ? PA + base1 : PL + base2;
                v = (int)dir_lerp2(*pp, ab ? z_w : sh2 * 0xffffu + 32u);
                if (sizeof(PixT) == 2) v = min(v, maxv);
            } break;
            default: v = dc; break;          // IM_DC, IM_DC_TOP, IM_DC_LEFT, IM_DC_128 and the constant fill
            }
            px[k] = v;
        }
        if (live) {
            PixT* o = out + (size_t)r * dst_stride + c0;
            constexpr int NB = PPL * (int)sizeof(PixT);          // 4 .. 32 bytes
            uint32_t pw[NB / 4];
#pragma unroll
            for (int k = 0; k < NB / 4; k++) {
                if (sizeof(PixT) == 1) pw[k] = (uint32_t)px[4 * k] | ((uint32_t)px[4 * k + 1] << 8) | ((uint32_t)px[4 * k + 2] << 16) | ((uint32_t)px[4 * k + 3] << 24);
                else pw[k] = (uint32_t)px[2 * k] | ((uint32_t)px[2 * k + 1] << 16);
            }
            constexpr int AL = NB >= 16 ? 16 : NB;               // widest store unit
            if ((reinterpret_cast<uintptr_t>(o) & (AL - 1)) == 0) {
                if constexpr (NB == 4) *reinterpret_cast<uint32_t*>(o) = pw[0];
                else if constexpr (NB == 8) *reinterpret_cast<uint2*>(o) = make_uint2(pw[0], pw[1]);
                else {
#pragma unroll
                    for (int k = 0; k < NB / 16; k++) reinterpret_cast<uint4*>(o)[k] = make_uint4(pw[4 * k], pw[4 * k + 1], pw[4 * k + 2], pw[4 * k + 3]);
                }
            } else if ((reinterpret_cast<uintptr_t>(o) & 3) == 0) {
#pragma unroll
                for (int k = 0; k < NB / 4; k++) reinterpret_cast<uint32_t*>(o)[k] = pw[k];
            } else {
#pragma unroll
                for (int k = 0; k < PPL; k++) o[k] = (PixT)px[k];
            }
        }
    }
    };
    const int kind0 = __builtin_amdgcn_readfirstlane(kind);
    if (__builtin_amdgcn_ballot_w64(kind != kind0) == 0) {
        switch (kind0) {
#define BIP_CASE(K) case K: predict(std::integral_constant<int, K>{}); break;
        BIP_CASE(IM_DC) BIP_CASE(IM_V) BIP_CASE(IM_H) BIP_CASE(IM_SMOOTH) BIP_CASE(IM_SMOOTH_V) BIP_CASE(IM_SMOOTH_H) BIP_CASE(IM_PAETH)
        BIP_CASE(IM_DC_TOP) BIP_CASE(IM_DC_LEFT) BIP_CASE(IM_DC_128) BIP_CASE(IM_Z1) BIP_CASE(IM_Z2)
#undef BIP_CASE
        default: predict(std::integral_constant<int, IM_Z3>{}); break;
        }
    } else {
        predict(std::integral_constant<int, -1>{});
    }
}

}  // namespace svtdev
