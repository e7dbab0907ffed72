// kernel_ois.h — open-loop intra search (reference open_loop_intra_search_sb,
// EbMotionEstimation.c:8694-8850) around the intra prediction kernels of kernel_intra.h:
//   ois_gather_kernel  update_neighbor_samples_array_open_loop (EbIntraPrediction.c:4707-4773) for every block of
//                      a batch straight from the source picture, in the neighbour layout svt_hip_intra_pred_batch
//                      reads, plus each block's DC value under the reference's availability rule
//                      (dc_pred[x > 0][y > 0], EbIntraPrediction.c:4801);
//   ois_sad_kernel     SAD of every candidate's dense prediction batch (or of the constant DC prediction) against
//                      the source blocks in the picture -> the [block][candidate] distortion matrix and the best
//                      index (first strict minimum below 64*64*255, EbMotionEstimation.c:8756, 8800-8803), one launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_common.h"

namespace svtdev {

constexpr int OIS_NB_ORIGIN = 16;      // == NB_ORIGIN of kernel_intra.h (position p of an edge at index 16 + p)

__device__ __forceinline__ void ois_gather_body(const uint8_t* __restrict__ pic, uint32_t stride, uint32_t width,
                                                uint32_t height, const uint32_t* __restrict__ xy, uint32_t bsize,
                                                uint8_t* __restrict__ above, uint8_t* __restrict__ left, uint32_t nb_pitch,
                                                uint8_t* __restrict__ dc, uint32_t nblocks, const uint32_t bid) {
    __shared__ int s_sum[16];
    const uint32_t lpb = 2 * bsize;                      // 16 .. 128 lanes per block
    const uint32_t lsh = __builtin_ctz(lpb);
    const uint32_t slot = threadIdx.x >> lsh, l = threadIdx.x & (lpb - 1);
    const uint32_t slots = 256u >> lsh;
    const uint32_t blk = bid * slots + slot;
    if (threadIdx.x < 16) s_sum[threadIdx.x] = 0;
    __syncthreads();
    const bool valid = blk < nblocks;
    uint32_t x = 0, y = 0, av_dc = 0, lv_dc = 0;
    if (valid) { const uint32_t q = xy[blk]; x = q & 0xffffu; y = q >> 16; }
    if (valid) {
        const uint8_t* src = pic + (size_t)y * stride + x;
        // fill values 129 / 127 / 128: EbIntraPrediction.c:4733-4746
        const uint32_t lv = (x != 0 && y + l < height) ? src[(size_t)l * stride - 1] : 129u;
        const uint32_t av = (y != 0 && x + l < width) ? src[(ptrdiff_t)l - (ptrdiff_t)stride] : 127u;
        uint8_t* ab = above + (size_t)blk * nb_pitch + OIS_NB_ORIGIN;
        uint8_t* lb = left + (size_t)blk * nb_pitch + OIS_NB_ORIGIN;
        ab[l] = (uint8_t)av;
        lb[l] = (uint8_t)lv;
        av_dc = av; lv_dc = lv;
        if (l == 0) {
            const uint8_t tl = (x != 0 && y != 0) ? src[-(ptrdiff_t)stride - 1] : (uint8_t)128;
            ab[-1] = tl; lb[-1] = tl;
            ab[-2] = 0; lb[-2] = 0;          // position -2 is staged with the rest (it only matters with up-sampling, which the open loop never has)
        }
    }
    // DC sum over the first bsize samples of each available edge: summed inside the wave first (DPP), one LDS atomic per (block,
    // wave) - a block's lanes span two waves only at 64x64.  (One atomic per lane on the block's word serialised the whole wave:
    // 70 % of this kernel's LDS cycles were bank conflicts.)  Every lane takes part in the DPP steps; dead lanes add 0.
    {
        const uint32_t part = (valid && l < bsize) ? (y != 0 ? av_dc : 0u) + (x != 0 ? lv_dc : 0u) : 0u;
        const uint32_t glanes = lpb < 64u ? lpb : 64u;
        const uint32_t sum = group_sum_rt(part, glanes);
        if (valid && (l & (glanes - 1)) == 0) atomicAdd(&s_sum[slot], (int)sum);
    }
    __syncthreads();
    if (valid && l == 0) {
        const int sum = s_sum[slot];
        const uint32_t lg = __builtin_ctz(bsize);
        int v;
        if (x != 0 && y != 0) v = (sum + (int)bsize) >> (lg + 1);
        else if (x != 0 || y != 0) v = (sum + (int)(bsize >> 1)) >> lg;
        else v = 128;
        dc[blk] = (uint8_t)v;
    }
}
__global__ __launch_bounds__(256) void ois_gather_kernel(const uint8_t* __restrict__ pic, uint32_t stride, uint32_t width,
                                                         uint32_t height, const uint32_t* __restrict__ xy, uint32_t bsize,
                                                         uint8_t* __restrict__ above, uint8_t* __restrict__ left, uint32_t nb_pitch,
                                                         uint8_t* __restrict__ dc, uint32_t nblocks) {
    ois_gather_body(pic, stride, width, height, xy, bsize, above, left, nb_pitch, dc, nblocks, blockIdx.x);
}
// the gathers of every block size of a picture in one launch (svt_hip_ois_search_frame)
constexpr int OIS_GATHER_MAX_GROUPS = 4;
struct OisGatherGroup {
    const uint32_t* xy; uint8_t* above; uint8_t* left; uint8_t* dc;
    uint32_t bsize, nb_pitch, nblocks, wg_end;
};
struct OisGatherMulti { int32_t ngroups; OisGatherGroup g[OIS_GATHER_MAX_GROUPS]; };
__global__ __launch_bounds__(256) void ois_gather_multi_kernel(const uint8_t* __restrict__ pic, uint32_t stride, uint32_t width, uint32_t height,
                                                               const OisGatherMulti m) {
    int gi = 0;
    uint32_t start = 0;
#pragma unroll 1
    for (int i = 0; i < m.ngroups; i++) {
        if (blockIdx.x >= m.g[i].wg_end) { gi = i + 1; start = m.g[i].wg_end; }
    }
    if (gi >= m.ngroups) return;
    const OisGatherGroup& G = m.g[gi];
    ois_gather_body(pic, stride, width, height, G.xy, G.bsize, G.above, G.left, G.nb_pitch, G.dc, G.nblocks, blockIdx.x - start);
}

// One launch for ALL candidates: pred_all holds ncand dense prediction batches back to back (a candidate whose bit
// is set in const_mask is the block's DC value instead).  Lanes per block = B*B / min(B, 16).  A lane loads its
// source chunk once and loops over the candidates; the block's ncand sums go through LDS so that they leave as one
// contiguous run of dist[block][0 .. ncand) (candidates in fold_mask were already summed by intra_dir_kernel's SAD mode
// and are only picked up), and the best index (first strict minimum below 64*64*255,
// EbMotionEstimation.c:8756, 8800-8803) is taken from the same LDS row - no second pass over the matrix.
constexpr int OIS_MAX_CAND = 61;       // MAX_OIS_CANDIDATES, EbCodingUnit.h:43
__global__ __launch_bounds__(256) void ois_sad_kernel(const uint8_t* __restrict__ pic, uint32_t stride, const uint32_t* __restrict__ xy,
                                                      uint32_t bsize, const uint8_t* __restrict__ pred_all, size_t pred_cand_pitch,
                                                      const uint8_t* __restrict__ dc, unsigned long long const_mask,
                                                      unsigned long long fold_mask, uint32_t* dist, int8_t* __restrict__ best_index, uint32_t ncand,
                                                      uint32_t nblocks) {
    extern __shared__ uint32_t s_dist[];                  // [slots][ncand] (+ [4][ncand] wave partials for 64x64)
    const uint32_t cs = bsize < 16 ? 8u : 16u;            // pixels per lane
    const uint32_t lpb = bsize * bsize / cs;              // 8, 16, 64, 256
    const uint32_t lsh = __builtin_ctz(lpb);
    const uint32_t slots = 256u >> lsh;
    const uint32_t slot = threadIdx.x >> lsh, l = threadIdx.x & (lpb - 1);
    const uint32_t blk = blockIdx.x * slots + slot;
    const bool valid = blk < nblocks;
    uint32_t sv[4] = {0, 0, 0, 0};
    uint32_t dcw = 0;
    if (valid) {
        const uint32_t q = xy[blk];
        const uint32_t cpr_sh = __builtin_ctz(bsize / cs);
        const uint32_t row = l >> cpr_sh, col = (l & ((1u << cpr_sh) - 1)) * cs;
        const uint8_t* s = pic + (size_t)((q >> 16) + row) * stride + (q & 0xffffu) + col;
        if (cs == 16) __builtin_memcpy(sv, s, 16); else __builtin_memcpy(sv, s, 8);
        dcw = dc[blk] * 0x01010101u;
    }
    const uint8_t* p = pred_all + (valid ? (size_t)blk * bsize * bsize + (size_t)l * cs : 0);
    const uint32_t span = lpb < 64 ? lpb : 64;
    uint32_t* row_out = s_dist + (size_t)slot * ncand;
    uint32_t* wave_part = s_dist + (size_t)slots * ncand;            // 64x64 only: [wave][cand]
    for (uint32_t c = 0; c < ncand; c++) {
        if ((fold_mask >> c) & 1ull) {                     // already in dist: the directional kernels' SAD mode (blocks <= 64 lanes)
            if (valid && l == 0) row_out[c] = dist[(size_t)blk * ncand + c];
            continue;
        }
        uint32_t pv[4];
        if ((const_mask >> c) & 1ull) { pv[0] = pv[1] = pv[2] = pv[3] = dcw; }
        else if (cs == 16) { const uint4 v = *reinterpret_cast<const uint4*>(p + (size_t)c * pred_cand_pitch); pv[0] = v.x; pv[1] = v.y; pv[2] = v.z; pv[3] = v.w; }
        else { const uint2 v = *reinterpret_cast<const uint2*>(p + (size_t)c * pred_cand_pitch); pv[0] = v.x; pv[1] = v.y; pv[2] = pv[3] = 0; }
        uint32_t sad = 0;
        sad = __builtin_amdgcn_sad_u8(sv[0], pv[0], sad);
        sad = __builtin_amdgcn_sad_u8(sv[1], pv[1], sad);
        if (cs == 16) { sad = __builtin_amdgcn_sad_u8(sv[2], pv[2], sad); sad = __builtin_amdgcn_sad_u8(sv[3], pv[3], sad); }
        sad = group_sum_rt(sad, span);
        if (lpb <= 64) { if (l == 0) row_out[c] = sad; }
        else if ((threadIdx.x & 63) == 0) wave_part[(threadIdx.x >> 6) * ncand + c] = sad;
    }
    __syncthreads();
    if (lpb > 64) {                                        // 64x64: one block per workgroup, add the four waves' partials
        for (uint32_t c = threadIdx.x; c < ncand; c += 256) row_out[c] = wave_part[c] + wave_part[ncand + c] + wave_part[2 * ncand + c] + wave_part[3 * ncand + c];
        __syncthreads();
    }
    // contiguous store of the workgroup's slots x ncand sums
    const uint32_t first = blockIdx.x * slots;
    const uint32_t nb_here = first < nblocks ? (nblocks - first < slots ? nblocks - first : slots) : 0;
    for (uint32_t i = threadIdx.x; i < nb_here * ncand; i += 256) dist[(size_t)first * ncand + i] = s_dist[i];
    if (threadIdx.x < nb_here) {
        const uint32_t* r = s_dist + (size_t)threadIdx.x * ncand;
        uint32_t best = 64u * 64u * 255u;
        int bi = 0;
        for (uint32_t c = 0; c < ncand; c++) {
            const uint32_t d = r[c];
            if (d < best) { best = d; bi = (int)c; }
        }
        best_index[first + threadIdx.x] = (int8_t)bi;
    }
}

// ---------------------------------------------------------------------------
// ois_nd_kernel: every NON-directional candidate of the list (DC under the availability rule, V / H - the directional modes at 90 /
// 180 degrees -, SMOOTH, SMOOTH_V, SMOOTH_H, PAETH) predicted and compared with the source block in one pass, nothing written
// but the sums: a lane takes its chunk of the source block and the neighbour samples that chunk needs STRAIGHT FROM THE
// PICTURE (update_neighbor_samples_array_open_loop's rules: 127 above / 129 left / 128 corner when the picture has no such
// sample, EbIntraPrediction.c:4707-4773), forms each candidate's prediction of its 8 or 16 pixels in registers and adds
// |source - prediction| with v_sad_u8.  Candidates of kind OIS_K_FOLDED were summed by the directional kernels (SAD mode) and
// are picked up from dist.  Then, as in ois_sad_kernel: the block's row of sums through LDS, one contiguous store, best index =
// first strict minimum.  With a list that has no directional candidate (every 32x32 / 64x64 list, EbMotionEstimation.c:8747)
// this is the whole open-loop search in ONE launch - no neighbour arrays, no prediction scratch.
// ---------------------------------------------------------------------------
enum { OIS_K_DC = 0, OIS_K_V, OIS_K_H, OIS_K_SMOOTH, OIS_K_SMOOTH_V, OIS_K_SMOOTH_H, OIS_K_PAETH, OIS_K_FOLDED };
struct OisKinds {
    uint8_t k[OIS_MAX_CAND + 3];             // kind of candidate c; [OIS_MAX_CAND + 2] = the list has folded candidates
    uint8_t n_nd, nd_c[15], nd_kind[15];     // the candidates ois_nd_kernel computes itself, in list order (host-built: the kernel's loop
};                                           // then runs 7 times, not 45 with a scalar load and a branch per folded candidate)

__device__ constexpr uint8_t kOisSmWeights[128] = {          // sm_weight_arrays (ASM_AVX2/EbIntraPrediction_AVX2.h:19-38), index [bs + i]
    0, 0, 255, 128, 255, 149, 85, 64, 255, 197, 146, 105, 73, 50, 37, 32,
    255, 225, 196, 170, 145, 123, 102, 84, 68, 54, 43, 33, 26, 20, 17, 16,
    255, 240, 225, 210, 196, 182, 169, 157, 145, 133, 122, 111, 101, 92, 83, 74,
    66, 59, 52, 45, 39, 34, 29, 25, 21, 17, 14, 12, 10, 9, 8, 8,
    255, 248, 240, 233, 225, 218, 210, 203, 196, 189, 182, 176, 169, 163, 156, 150,
    144, 138, 133, 127, 121, 116, 111, 106, 101, 96, 91, 86, 82, 77, 73, 69,
    65, 61, 57, 54, 50, 47, 44, 41, 38, 35, 32, 29, 27, 25, 22, 20,
    18, 16, 15, 13, 12, 10, 9, 8, 7, 6, 6, 5, 5, 4, 4, 4};

// (a device body: ois_nd_kernel runs it for one block size, ois_nd_multi_kernel for every size of a picture in one launch; `bid` is
// the workgroup's index inside its group, `kinds` a reference into the kernel arguments)
template <int CS>            // pixels per lane: 8 (8x8 blocks) or 16
__device__ __forceinline__ void ois_nd_body(const uint8_t* __restrict__ pic, uint32_t stride, uint32_t width, uint32_t height,
                                            const uint32_t* __restrict__ xy, uint32_t bsize, const OisKinds& kinds, uint32_t* dist,
                                            int8_t* __restrict__ best_index, uint32_t ncand, uint32_t nblocks, const uint32_t bid) {
    extern __shared__ uint32_t s_dist[];                  // [slots][ncand] (+ [4][ncand] wave partials and 4 DC partials for 64x64)
    const uint32_t lpb = bsize * bsize / CS;              // 8, 16, 64, 256
    const uint32_t lsh = __builtin_ctz(lpb);
    const uint32_t slots = 256u >> lsh;
    const uint32_t slot = threadIdx.x >> lsh, l = threadIdx.x & (lpb - 1);
    const uint32_t blk = bid * slots + slot;
    const bool valid = blk < nblocks;
    const uint32_t q = xy[valid ? blk : 0];
    const uint32_t x = q & 0xffffu, y = q >> 16;
    const uint32_t cpr_sh = __builtin_ctz(bsize / CS);
    const uint32_t row = l >> cpr_sh, col = (l & ((1u << cpr_sh) - 1)) * CS;
    const uint8_t* sblk = pic + (size_t)y * stride + x;
    const bool has_a = y != 0, has_l = x != 0;
    // ---- the sums the directional kernels left in dist (kind OIS_K_FOLDED): the workgroup's contiguous run of rows in one coalesced
    // pass, issued before everything else.  (Fetching them one candidate at a time inside the loop below was a chain of up to 38
    // dependent global loads per wave - two thirds of this kernel's time on the 8x8 list.)  The loop overwrites the other entries.
    const uint32_t first = bid * slots;
    const uint32_t nb_here = first < nblocks ? (nblocks - first < slots ? nblocks - first : slots) : 0;
    if (kinds.k[OIS_MAX_CAND + 2])                         // host: set when the list has a folded candidate
        for (uint32_t i = threadIdx.x; i < nb_here * ncand; i += 256) s_dist[i] = dist[(size_t)first * ncand + i];
    // ---- this lane's samples: source chunk, the above segment over it, its row's left sample, the three corners --------------
    uint32_t sv[4] = {0, 0, 0, 0}, av[4];
    __builtin_memcpy(sv, sblk + (size_t)row * stride + col, CS);
    if (has_a && x + col + CS <= width) {
        av[2] = av[3] = 0;
        __builtin_memcpy(av, sblk - (ptrdiff_t)stride + col, CS);
    } else {
        // (a block that reaches past the picture, or no row above: never the encoder's own blocks.  A rolled byte loop: unrolled, its
        // sixteen conditional loads and their 64-bit addresses set the whole kernel's register count - 132 VGPRs, 3 waves per SIMD)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t wv = j < CS / 4 ? 0x7f7f7f7fu : 0u;
            if (j < CS / 4 && has_a) {
#pragma unroll 1
                for (uint32_t b = 0; b < 4; b++)
                    if (x + col + 4 * j + b < width) wv = (wv & ~(0xffu << (8 * b))) | ((uint32_t)sblk[(ptrdiff_t)(col + 4 * j + b) - (ptrdiff_t)stride] << (8 * b));
            }
            av[j] = wv;
        }
    }
    auto left_at = [&](uint32_t r) { return (has_l && y + r < height) ? (int)sblk[(size_t)r * stride - 1] : 129; };
    auto above_at = [&](uint32_t c) { return (has_a && x + c < width) ? (int)sblk[(ptrdiff_t)c - (ptrdiff_t)stride] : 127; };
    const int lf = left_at(row), bl = left_at(bsize - 1), tr = above_at(bsize - 1);
    const int tl = (has_a && has_l) ? (int)sblk[-(ptrdiff_t)stride - 1] : 128;
    // ---- DC value: sum of the first bsize samples of each edge the picture has (dc_pred[x > 0][y > 0], :4801) ------------------
    int dsum = 0;
    for (uint32_t i = l; i < 2 * bsize; i += lpb) dsum += i < bsize ? (has_a ? above_at(i) : 0) : (has_l ? left_at(i - bsize) : 0);
    const uint32_t span = lpb < 64 ? lpb : 64;
    dsum = (int)group_sum_rt((uint32_t)dsum, span);
    uint32_t* row_out = s_dist + (size_t)slot * ncand;
    uint32_t* wave_part = s_dist + (size_t)slots * ncand;            // 64x64 only: [wave][cand], then 4 DC partials
    if (lpb > 64) {
        if ((threadIdx.x & 63) == 0) wave_part[4 * ncand + (threadIdx.x >> 6)] = (uint32_t)dsum;
        __syncthreads();
        dsum = (int)(wave_part[4 * ncand] + wave_part[4 * ncand + 1] + wave_part[4 * ncand + 2] + wave_part[4 * ncand + 3]);
    }
    const uint32_t lg = __builtin_ctz(bsize);
    const int dcv = (has_a && has_l) ? (dsum + (int)bsize) >> (lg + 1) : ((has_a || has_l) ? (dsum + (int)(bsize >> 1)) >> lg : 128);
    const int wh = kOisSmWeights[bsize + row];
    uint32_t wwv[4] = {0, 0, 0, 0};                        // this lane's column weights, one byte each: one load instead of one per pixel and kind
    __builtin_memcpy(wwv, kOisSmWeights + bsize + col, CS);
    if (lpb <= 64) __syncthreads();                        // the copy above before this loop's stores (64x64: the barrier above)
    // ---- candidates -------------------------------------------------------------------------------------------------------------
#pragma unroll 1
    for (uint32_t i = 0; i < kinds.n_nd; i++) {
        const uint32_t kind = kinds.nd_kind[i], c = kinds.nd_c[i];       // uniform (kernel arguments)
        // the packed above samples and column weights are re-read as "changed" every round: left to itself the compiler unpacks all
        // 16 + 16 of them (and their complements) once, outside this loop, and the 16-pixel form then needs 122 VGPRs - 4 waves per
        // SIMD in a kernel that is one wave's latency chain
#pragma unroll
        for (int q = 0; q < CS / 4; q++) asm volatile("" : "+v"(av[q]), "+v"(wwv[q]));
        uint32_t pv[4] = {0, 0, 0, 0};
        if (kind == OIS_K_DC) { pv[0] = pv[1] = pv[2] = pv[3] = (uint32_t)dcv * 0x01010101u; }
        else if (kind == OIS_K_V) { pv[0] = av[0]; pv[1] = av[1]; pv[2] = av[2]; pv[3] = av[3]; }
        else if (kind == OIS_K_H) { pv[0] = pv[1] = pv[2] = pv[3] = (uint32_t)lf * 0x01010101u; }
        else {
#pragma unroll
            for (int k = 0; k < CS; k++) {
                const int t = (int)((av[k >> 2] >> (8 * (k & 3))) & 0xffu);
                const int ww = (int)((wwv[k >> 2] >> (8 * (k & 3))) & 0xffu);
                int v;
                if (kind == OIS_K_SMOOTH) v = (wh * t + (256 - wh) * bl + ww * lf + (256 - ww) * tr + 256) >> 9;
                else if (kind == OIS_K_SMOOTH_V) v = (wh * t + (256 - wh) * bl + 128) >> 8;
                else if (kind == OIS_K_SMOOTH_H) v = (ww * lf + (256 - ww) * tr + 128) >> 8;
                else {                                     // PAETH: nearest of left / top / top-left to top + left - topleft
                    const int pb = t + lf - tl, pl = abs(pb - lf), pt = abs(pb - t), ptl = abs(pb - tl);
                    v = (pl <= pt && pl <= ptl) ? lf : (pt <= ptl ? t : tl);
                }
                pv[k >> 2] |= (uint32_t)v << (8 * (k & 3));
            }
        }
        uint32_t sad = 0;
        sad = __builtin_amdgcn_sad_u8(sv[0], pv[0], sad);
        sad = __builtin_amdgcn_sad_u8(sv[1], pv[1], sad);
        if (CS == 16) { sad = __builtin_amdgcn_sad_u8(sv[2], pv[2], sad); sad = __builtin_amdgcn_sad_u8(sv[3], pv[3], sad); }
        sad = group_sum_rt(sad, span);
        if (lpb <= 64) { if (l == 0) row_out[c] = sad; }
        else if ((threadIdx.x & 63) == 0) wave_part[(threadIdx.x >> 6) * ncand + c] = sad;
    }
    __syncthreads();
    if (lpb > 64) {                                        // 64x64: one block per workgroup, add the four waves' partials
        for (uint32_t c = threadIdx.x; c < ncand; c += 256)
            if (kinds.k[c] != OIS_K_FOLDED) row_out[c] = wave_part[c] + wave_part[ncand + c] + wave_part[2 * ncand + c] + wave_part[3 * ncand + c];
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < nb_here * ncand; i += 256) dist[(size_t)first * ncand + i] = s_dist[i];
    if (lpb <= 64) {
        // best index = first strict minimum: the block's lanes each scan every lpb-th candidate, key = sum << 6 | index (sums
        // < 2^20, at most 61 candidates), minimum over the group
        uint32_t key = 0xffffffffu;
        for (uint32_t c = l; c < ncand; c += lpb) key = min(key, (row_out[c] << 6) | c);
        key = group_min_rt(key, span);
        if (valid && l == 0) best_index[blk] = (int8_t)(key & 63u);
    } else if (threadIdx.x < nb_here) {
        const uint32_t* r = s_dist + (size_t)threadIdx.x * ncand;
        uint32_t best = 64u * 64u * 255u;
        int bi = 0;
        for (uint32_t c = 0; c < ncand; c++) {
            const uint32_t d = r[c];
            if (d < best) { best = d; bi = (int)c; }
        }
        best_index[first + threadIdx.x] = (int8_t)bi;
    }
}

template <int CS>
__global__ __launch_bounds__(256) void ois_nd_kernel(const uint8_t* __restrict__ pic, uint32_t stride, uint32_t width, uint32_t height,
                                                     const uint32_t* __restrict__ xy, uint32_t bsize, OisKinds kinds, uint32_t* dist,
                                                     int8_t* __restrict__ best_index, uint32_t ncand, uint32_t nblocks) {
    ois_nd_body<CS>(pic, stride, width, height, xy, bsize, kinds, dist, best_index, ncand, nblocks, blockIdx.x);
}

// The non-directional launches of EVERY block size of a picture in one (svt_hip_ois_search_frame): as four launches the 32x32 and
// 64x64 lists (five candidates, ~ 500 workgroups, one wave's latency chain: ~ 20 us each) neither hid behind the 8x8 / 16x16 chains on
// a side stream (those fill the GPU; the call cost the SUM of its groups) nor amortised anything.  Group table in the kernel arguments.
constexpr int OIS_ND_MAX_GROUPS = 4;
struct OisNdGroup {
    const uint32_t* xy; uint32_t* dist; int8_t* best_index;
    uint32_t bsize, ncand, nblocks, wg_end;
    OisKinds kinds;
};
struct OisNdMulti { int32_t ngroups; OisNdGroup g[OIS_ND_MAX_GROUPS]; };
__global__ __launch_bounds__(256) void ois_nd_multi_kernel(const uint8_t* __restrict__ pic, uint32_t stride, uint32_t width, uint32_t height,
                                                           const OisNdMulti m) {
    int gi = 0;
    uint32_t start = 0;
#pragma unroll 1
    for (int i = 0; i < m.ngroups; i++) {
        if (blockIdx.x >= m.g[i].wg_end) { gi = i + 1; start = m.g[i].wg_end; }
    }
    if (gi >= m.ngroups) return;
    const OisNdGroup& G = m.g[gi];
    if (G.bsize < 16) ois_nd_body<8>(pic, stride, width, height, G.xy, G.bsize, G.kinds, G.dist, G.best_index, G.ncand, G.nblocks, blockIdx.x - start);
    else ois_nd_body<16>(pic, stride, width, height, G.xy, G.bsize, G.kinds, G.dist, G.best_index, G.ncand, G.nblocks, blockIdx.x - start);
}

}  // namespace svtdev
