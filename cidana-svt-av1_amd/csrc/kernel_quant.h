// kernel_quant.h — K3: stand-alone quantize / dequantize / eob kernel (aom_highbd_quantize_b*, EbFullLoop.c:239-333).
#pragma once
#include "dev_common.h"

namespace svtdev {

// ---------------------------------------------------------------------------
// quantize_b on already-transformed coefficients (aom_highbd_quantize_b*,
// EbFullLoop.c:239-333; AVX2 highbd_quantize_intrin_avx2.c:127-484).
// n coefficients per block (dense), LPB = min(64, n/4) lanes per block, each
// lane handles int4 chunks at coalesced positions.
// ---------------------------------------------------------------------------
template <int LPB>
__global__ __launch_bounds__(256) void quantize_b_kernel(
    const int32_t* __restrict__ coeff, int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff,
    uint16_t* __restrict__ eob, const int16_t* __restrict__ iscan, QParams qp, int n, int skip_block,
    uint32_t nblocks) {
    constexpr int BPW = 64 / LPB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPB, l = lane % LPB;
    const uint32_t blk = (blockIdx.x * 4 + wave) * BPW + sub;
    const bool valid = blk < nblocks;
    const int4* c4 = reinterpret_cast<const int4*>(coeff + (size_t)blk * n);
    int4* q4 = reinterpret_cast<int4*>(qcoeff + (size_t)blk * n);
    int4* d4 = reinterpret_cast<int4*>(dqcoeff + (size_t)blk * n);
    int eob_acc = 0;
    for (int i = l; i < n / 4; i += LPB) {
        int4 c = {0, 0, 0, 0}, q = c, d = c;
        if (valid && !skip_block) {
            c = c4[i];
            const uint2 is = *reinterpret_cast<const uint2*>(iscan + i * 4);
            quant_one<0>(c.x, i == 0 ? 0 : 1, qp, q.x, d.x);
            quant_one<0>(c.y, 1, qp, q.y, d.y);
            quant_one<0>(c.z, 1, qp, q.z, d.z);
            quant_one<0>(c.w, 1, qp, q.w, d.w);
            const int e0 = q.x ? (int)(is.x & 0xffffu) + 1 : 0, e1 = q.y ? (int)(is.x >> 16) + 1 : 0;
            const int e2 = q.z ? (int)(is.y & 0xffffu) + 1 : 0, e3 = q.w ? (int)(is.y >> 16) + 1 : 0;
            eob_acc = max(eob_acc, max(max(e0, e1), max(e2, e3)));
        }
        if (valid) { q4[i] = q; d4[i] = d; }
    }
    eob_acc = group_max<LPB>(eob_acc);
    if (valid && l == 0) eob[blk] = (uint16_t)eob_acc;
}


}  // namespace svtdev
