// kernel_frame.h — enc_frame_kernel: the encode pass of EVERY (plane, transform size) group of a picture in ONE launch
// (SURVEY 8f n3 "one fused frame kernel"; BASELINE configs[3]).
//
// The launch covers the workgroups of all groups back to back; a workgroup finds its group in a table that rides in the
// kernel arguments (no device-side descriptor memory: the call stays a pure enqueue and is graph-capturable), takes its index
// inside the group and runs that size's body - the same bodies the per-size kernels run (enc4_body, enc_staged_body<8,8> /
// <16,16>, enc32_body, enc64_body), so results are identical by construction.  A workgroup is homogeneous (one size); groups
// are ordered largest blocks first so the long workgroups start early.
//
// What one launch costs: the register file and LDS of the launch are the maximum over the bodies (the 64x64 body), so the
// small sizes run at its occupancy instead of their own.  For ONE picture that is cheaper than 13 dispatches; for a GOP the
// per-size launches win.  3 waves / SIMD: at that budget (168 VGPRs) the 64x64 body spills 12 B (8-bit) / 36 B
// (10-bit) to scratch; measured against 2 waves / SIMD without scratch (192 VGPRs): 5 350 against 4 560 4K 10-bit pictures/s.  Covered: square sizes 4 .. 64, every transform type the reference defines for them, qcoeff + eob + recon outputs, 8 / 10 bit,
// power-of-two quant_shift tables.
#pragma once
#include "kernel_enc64.h"
#include "kernel_fused32.h"
#include "kernel_txfm_staged.h"

namespace svtdev {

constexpr int FRAME_MAX_GROUPS = 16;
struct FrameGroupDev {
    const void* src; const void* pred; void* recon;
    int32_t* qcoeff; uint16_t* eob;
    const uint32_t* xy; const int16_t* iscan;
    uint32_t src_stride, pred_stride, recon_stride, nblocks;
    int32_t tx_size;                 // SVT_TX_4X4 .. SVT_TX_64X64 (0 .. 4)
    int32_t tx_type;                 // any type the reference defines for the size (32x32: DCT_DCT / IDTX, 64x64: DCT_DCT)
    uint32_t wg_end;                 // one past the last workgroup of this group in the launch
    QParams qp;                      // per group: log_scale differs with the size
};
struct FrameDesc {
    int32_t ngroups;
    FrameGroupDev g[FRAME_MAX_GROUPS];
};
static_assert(sizeof(FrameDesc) <= 4000, "kernel arguments");

constexpr int FRAME_LDS_BYTES = cmax(cmax(ENC32_LDS_BYTES, E64_WAVES * E64_WAVE_LDS),
                                     cmax(EncStagedLds<16, 16, uint16_t>::BYTES, EncStagedLds<8, 8, uint16_t>::BYTES));

template <typename PixT, int BD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void enc_frame_kernel(const FrameDesc fd) {
    __shared__ __attribute__((aligned(16))) char lds[FRAME_LDS_BYTES];
    // which group (uniform: scalar compares against the table in the kernel arguments)
    int gi = 0;
    uint32_t start = 0;
#pragma unroll 1
    for (int i = 0; i < fd.ngroups; i++) {
        if (blockIdx.x >= fd.g[i].wg_end) { gi = i + 1; start = fd.g[i].wg_end; }
    }
    if (gi >= fd.ngroups) return;
    const FrameGroupDev& G = fd.g[gi];
    const uint32_t bid = blockIdx.x - start;
    const PixT* src = reinterpret_cast<const PixT*>(G.src);
    const PixT* pred = reinterpret_cast<const PixT*>(G.pred);
    PixT* recon = reinterpret_cast<PixT*>(G.recon);
    switch (G.tx_size) {
    case 0:
        enc4_body<PixT, BD, false>(src, pred, recon, nullptr, G.qcoeff, nullptr, G.eob, nullptr, G.iscan, G.qp, 1, G.tx_type, G.nblocks, G.xy, G.src_stride,
                                   G.pred_stride, G.recon_stride, bid);
        break;
    case 1:
        enc_staged_body<8, 8, false, PixT, BD>(src, pred, recon, nullptr, G.qcoeff, nullptr, G.eob, nullptr, G.iscan, G.qp, G.tx_type, G.nblocks, G.xy,
                                               G.src_stride, G.pred_stride, G.recon_stride, bid, lds);
        break;
    case 2:
        enc_staged_body<16, 16, false, PixT, BD>(src, pred, recon, nullptr, G.qcoeff, nullptr, G.eob, nullptr, G.iscan, G.qp, G.tx_type, G.nblocks, G.xy,
                                                 G.src_stride, G.pred_stride, G.recon_stride, bid, lds);
        break;
    case 3:
        enc32_body<PixT, BD, false, false>(src, pred, recon, nullptr, G.qcoeff, nullptr, G.eob, nullptr, G.iscan, G.qp, G.tx_type == 9 /* IDTX */ ? 1 : 0, G.nblocks, G.xy, G.src_stride,
                                           G.pred_stride, G.recon_stride, bid, reinterpret_cast<int32_t*>(lds));
        break;
    default:
        enc64_body<PixT, BD, false>(src, pred, recon, nullptr, G.qcoeff, nullptr, G.eob, nullptr, G.iscan, G.qp, G.nblocks, G.xy, G.src_stride, G.pred_stride,
                                    G.recon_stride, bid, lds);
        break;
    }
}

}  // namespace svtdev
