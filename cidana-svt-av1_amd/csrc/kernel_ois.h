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

__global__ __launch_bounds__(256) void ois_gather_kernel(const uint8_t* __restrict__ pic, uint32_t stride, uint32_t width,
                                                         uint32_t height, const uint32_t* __restrict__ xy, uint32_t bsize,
                                                         uint8_t* __restrict__ above, uint8_t* __restrict__ left, uint32_t nb_pitch,
                                                         uint8_t* __restrict__ dc, uint32_t nblocks) {
    __shared__ int s_sum[16];
    const uint32_t lpb = 2 * bsize;                      // 16 .. 128 lanes per block
    const uint32_t lsh = __builtin_ctz(lpb);
    const uint32_t slot = threadIdx.x >> lsh, l = threadIdx.x & (lpb - 1);
    const uint32_t slots = 256u >> lsh;
    const uint32_t blk = blockIdx.x * slots + slot;
    if (threadIdx.x < 16) s_sum[threadIdx.x] = 0;
    __syncthreads();
    const bool valid = blk < nblocks;
    uint32_t x = 0, y = 0;
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
        if (l == 0) {
            const uint8_t tl = (x != 0 && y != 0) ? src[-(ptrdiff_t)stride - 1] : (uint8_t)128;
            ab[-1] = tl; lb[-1] = tl;
        }
        // DC sum over the first bsize samples of each available edge (a block's lanes may span two waves: LDS)
        if (l < bsize) atomicAdd(&s_sum[slot], (y != 0 ? (int)av : 0) + (x != 0 ? (int)lv : 0));
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
        for (uint32_t m = span >> 1; m >= 1; m >>= 1) sad += __shfl_xor(sad, (int)m, 64);
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

}  // namespace svtdev
