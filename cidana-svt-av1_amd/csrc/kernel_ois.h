// kernel_ois.h — open-loop intra search (reference open_loop_intra_search_sb,
// EbMotionEstimation.c:8694-8850) around the intra prediction kernels of kernel_intra.h:
//   ois_gather_kernel  update_neighbor_samples_array_open_loop (EbIntraPrediction.c:4707-4773) for every block of
//                      a batch straight from the source picture, in the neighbour layout svt_hip_intra_pred_batch
//                      reads, plus each block's DC value under the reference's availability rule
//                      (dc_pred[x > 0][y > 0], EbIntraPrediction.c:4801);
//   ois_sad_kernel     SAD of every candidate's dense prediction batch (or of the constant DC prediction) against
//                      the source blocks in the picture -> the [block][candidate] distortion matrix, one launch;
//   ois_best_kernel    first strict minimum below 64*64*255 per block (EbMotionEstimation.c:8756, 8800-8803).
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

// One launch for ALL candidates (blockIdx.y = candidate): pred holds ncand dense prediction batches back to back;
// a candidate whose bit is set in const_mask is the block's DC value instead.  Lanes per block = B*B / min(B, 16).
__global__ __launch_bounds__(256) void ois_sad_kernel(const uint8_t* __restrict__ pic, uint32_t stride, const uint32_t* __restrict__ xy,
                                                      uint32_t bsize, const uint8_t* __restrict__ pred_all, size_t pred_cand_pitch,
                                                      const uint8_t* __restrict__ dc, unsigned long long const_mask,
                                                      uint32_t* __restrict__ dist, uint32_t ncand, uint32_t nblocks) {
    const uint32_t cand = blockIdx.y;
    const bool CONST = (const_mask >> cand) & 1ull;
    const uint8_t* pred = pred_all + (size_t)cand * pred_cand_pitch;
    __shared__ uint32_t s_part[4];
    const uint32_t cs = bsize < 16 ? 8u : 16u;            // pixels per lane
    const uint32_t lpb = bsize * bsize / cs;
    const uint32_t lsh = __builtin_ctz(lpb);
    const uint32_t slot = threadIdx.x >> lsh, l = threadIdx.x & (lpb - 1);
    const uint32_t blk = blockIdx.x * (256u >> lsh) + slot;
    const bool valid = blk < nblocks;
    uint32_t sad = 0;
    if (valid) {
        const uint32_t q = xy[blk];
        const uint32_t cpr_sh = __builtin_ctz(bsize / cs);
        const uint32_t row = l >> cpr_sh, col = (l & ((1u << cpr_sh) - 1)) * cs;
        const uint8_t* s = pic + (size_t)((q >> 16) + row) * stride + (q & 0xffffu) + col;
        uint32_t sv[4] = {0, 0, 0, 0}, pv[4] = {0, 0, 0, 0};
        if (cs == 16) __builtin_memcpy(sv, s, 16); else __builtin_memcpy(sv, s, 8);
        if (CONST) {
            const uint32_t d = dc[blk] * 0x01010101u;
            pv[0] = pv[1] = d;
            if (cs == 16) pv[2] = pv[3] = d;
        } else {
            const uint8_t* p = pred + (size_t)blk * bsize * bsize + (size_t)l * cs;
            if (cs == 16) pv[0] = reinterpret_cast<const uint4*>(p)->x, pv[1] = reinterpret_cast<const uint4*>(p)->y,
                          pv[2] = reinterpret_cast<const uint4*>(p)->z, pv[3] = reinterpret_cast<const uint4*>(p)->w;
            else pv[0] = reinterpret_cast<const uint2*>(p)->x, pv[1] = reinterpret_cast<const uint2*>(p)->y;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) sad = __builtin_amdgcn_sad_u8(sv[i], pv[i], sad);
    }
    const uint32_t span = lpb < 64 ? lpb : 64;
    for (uint32_t m = span >> 1; m >= 1; m >>= 1) sad += __shfl_xor(sad, (int)m, 64);
    if (lpb <= 64) {
        if (valid && l == 0) dist[(size_t)blk * ncand + cand] = sad;
    } else {                                              // 64x64: four waves per block
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sad;
        __syncthreads();
        if (valid && threadIdx.x == 0) dist[(size_t)blk * ncand + cand] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    }
}

__global__ __launch_bounds__(256) void ois_best_kernel(const uint32_t* __restrict__ dist, uint32_t ncand, int8_t* __restrict__ best_index,
                                                       uint32_t nblocks) {
    const uint32_t blk = blockIdx.x * 256u + threadIdx.x;
    if (blk >= nblocks) return;
    uint32_t best = 64u * 64u * 255u;
    int bi = 0;
    for (uint32_t c = 0; c < ncand; c++) {
        const uint32_t d = dist[(size_t)blk * ncand + c];
        if (d < best) { best = d; bi = (int)c; }
    }
    best_index[blk] = (int8_t)bi;
}

}  // namespace svtdev
