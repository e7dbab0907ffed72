// kernel_frame.h — enc_frame_kernel: the encode pass of the (plane, transform size) groups of a picture in ONE launch, or in one
// launch per REGISTER CLASS (SURVEY 8f n3 "one fused frame kernel"; BASELINE configs[3]).
//
// A launch covers the workgroups of all its groups back to back; a workgroup finds its group in a table that rides in the
// kernel arguments (no device-side descriptor memory: the call stays a pure enqueue and is graph-capturable), takes its index
// inside the group and runs that size's body - the same bodies the per-size kernels run (enc4_body, enc_staged_body<W, H>,
// enc32_body, enc64_body), so results are identical by construction.  A workgroup is homogeneous (one size); groups are ordered
// largest blocks first so the long workgroups start early.
//
// Registers.  The register file and the LDS of a launch are the maximum over the bodies it contains.  Round 2's single launch ran at
// the 64x64 body's budget (168 VGPRs at 3 waves / SIMD) and that body still spilled 20 B (8-bit) / 36 B (10-bit) per lane to
// SCRATCH.  With this round's LDS layouts (kernel_txfm_staged.h) every body fits 3 waves / SIMD without scratch, and the sizes can
// also be launched by register class (tools/kernel_resources.py):
//     class 0  every size up to 16 in both dimensions (4x4 .. 16x16, 4x8 .. 16x4)     <= 80 VGPRs, 6 waves / SIMD, 17 KiB LDS
//     class 1  32x32 and every rectangle with a 32- or 64-sample side                   <= 139 VGPRs, 3 - 4 waves / SIMD, 33 KiB LDS
//     class 2  64x64 (two blocks per wave, pruned 64-point networks)                    169 / 192 VGPRs, 2 waves / SIMD
//     class 3  the five square sizes in ONE launch at 3 waves / SIMD                    147 / 160 VGPRs
// One picture takes class 3 (three back-to-back launches each drain before the next fills the GPU: 0.066 against 0.049 ms for a
// 1080p picture, A/B on one box); a call with many pictures takes the class launches, where the small sizes - three fifths of a
// picture's pixel passes in configs[3] - run at twice the occupancy.  No kernel of this file uses scratch
// (tests/test_kernel_resources.py).  Covered: all 19 transform sizes, every transform type the reference defines for them,
// qcoeff + eob + recon outputs, 8 / 10 bit, power-of-two quant_shift tables.
#pragma once
#include "kernel_enc64.h"
#include "kernel_fused32.h"
#include "kernel_txfm_staged.h"

namespace svtdev {

constexpr int FRAME_MAX_GROUPS = 22;        // per launch: what fits the 4 KiB of kernel arguments
struct FrameGroupDev {
    const void* src; const void* pred; void* recon;
    int32_t* qcoeff; uint16_t* eob;
    const uint32_t* xy; const int16_t* iscan;
    uint32_t src_stride, pred_stride, recon_stride, nblocks;
    int32_t tx_size;                 // SVT_TX_4X4 .. SVT_TX_64X16 (0 .. 18)
    int32_t tx_type;                 // any type the reference defines for the size (32-point sides: DCT_DCT / IDTX, 64: DCT_DCT)
    uint32_t wg_end;                 // one past the last workgroup of this group in the launch
    QParams qp;                      // per group: log_scale differs with the size
};
struct FrameDesc {
    int32_t ngroups;
    FrameGroupDev g[FRAME_MAX_GROUPS];
};
static_assert(sizeof(FrameDesc) <= 4000, "kernel arguments");

// register class of a transform size (TxSize numbering of the reference, EbDefinitions.h:615-650)
__host__ __device__ constexpr int frame_class_of(int tx_size) {
    // sizes with both sides <= 16: TX_4X4 0, 8X8 1, 16X16 2, 4X8 5, 8X4 6, 8X16 7, 16X8 8, 4X16 13, 16X4 14 (a bit mask, not a table:
    // the device evaluates this with a run-time size)
    constexpr unsigned small = (1u << 0) | (1u << 1) | (1u << 2) | (1u << 5) | (1u << 6) | (1u << 7) | (1u << 8) | (1u << 13) | (1u << 14);
    return tx_size == 4 ? 2 : (((small >> tx_size) & 1u) ? 0 : 1);
}
// blocks one 256-thread workgroup of the body takes: 4x4 one block per lane; staged bodies WAVES x BPW; 32x32 F32_WAVES x 2;
// 64x64 E64_WAVES x 2
inline uint32_t frame_blocks_per_wg(int tx_size) {        // host only
    static const int kW[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64};
    static const int kH[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};
    if (tx_size == 0) return 256;
    if (tx_size == 3) return F32_WAVES * 2;
    if (tx_size == 4) return E64_WAVES * 2;
    const int w = kW[tx_size], h = kH[tx_size], m = w > h ? w : h;
    return (uint32_t)((w * h >= 4096 ? 2 : 4) * (64 / m));
}

template <typename PixT, int CLS> struct FrameLds;
template <typename PixT> struct FrameLds<PixT, 0> {
    static constexpr int BYTES = cmax(cmax(cmax(EncStagedLds<16, 16, PixT>::BYTES, EncStagedLds<8, 8, PixT>::BYTES),
                                           cmax(EncStagedLds<8, 16, PixT>::BYTES, EncStagedLds<16, 8, PixT>::BYTES)),
                                      cmax(cmax(EncStagedLds<4, 8, PixT>::BYTES, EncStagedLds<8, 4, PixT>::BYTES),
                                           cmax(EncStagedLds<4, 16, PixT>::BYTES, EncStagedLds<16, 4, PixT>::BYTES)));
};
template <typename PixT> struct FrameLds<PixT, 1> {
    static constexpr int BYTES = cmax(cmax(cmax(ENC32_LDS_BYTES, EncStagedLds<16, 32, PixT>::BYTES), cmax(EncStagedLds<32, 16, PixT>::BYTES, EncStagedLds<8, 32, PixT>::BYTES)),
                                      cmax(cmax(EncStagedLds<32, 8, PixT>::BYTES, EncStagedLds<32, 64, PixT>::BYTES),
                                           cmax(EncStagedLds<64, 32, PixT>::BYTES, cmax(EncStagedLds<16, 64, PixT>::BYTES, EncStagedLds<64, 16, PixT>::BYTES))));
};
template <typename PixT> struct FrameLds<PixT, 2> { static constexpr int BYTES = E64_WAVES * E64_WAVE_LDS; };
template <typename PixT> struct FrameLds<PixT, 3> {       // every size in one launch
    static constexpr int BYTES = cmax(cmax(FrameLds<PixT, 0>::BYTES, FrameLds<PixT, 1>::BYTES), FrameLds<PixT, 2>::BYTES);
};

#define SVT_FRAME_STAGED(W, H)                                                                                                         \
    enc_staged_body<W, H, false, PixT, BD>(src, pred, recon, nullptr, G.qcoeff, nullptr, G.eob, nullptr, G.iscan, G.qp, G.tx_type, G.nblocks, G.xy, \
                                           G.src_stride, G.pred_stride, G.recon_stride, bid, lds)

// CLS 0 / 1 / 2: one register class; CLS 3: every size in ONE launch at 3 waves / SIMD (<= 168 registers: with the LDS layouts of
// this round the 64x64 body fits, 147 / 160 VGPRs, no scratch) - what a single picture takes: measured A/B on one box (configs[3],
// 1080p, five sizes) 0.049 ms against 0.066 ms for three class launches back to back, each of which drains before the next fills
// the GPU; the classes are for calls large enough to amortise that (a GOP per call).
template <int CLS> struct FrameWaves { static constexpr int MIN = CLS == 3 ? 3 : 1; };
template <typename PixT, int BD, int CLS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FrameWaves<CLS>::MIN))) void enc_frame_kernel(const FrameDesc fd) {
    __shared__ __attribute__((aligned(16))) char lds[FrameLds<PixT, CLS>::BYTES];
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
#define SVT_FRAME_CASE(N, W, H) case N: SVT_FRAME_STAGED(W, H); break;
#define SVT_FRAME_ENC4 enc4_body<PixT, BD, false>(src, pred, recon, nullptr, G.qcoeff, nullptr, G.eob, nullptr, G.iscan, G.qp, 1, G.tx_type, G.nblocks, G.xy, G.src_stride, \
                                                  G.pred_stride, G.recon_stride, bid)
#define SVT_FRAME_ENC32 enc32_body<PixT, BD, false, false>(src, pred, recon, nullptr, G.qcoeff, nullptr, G.eob, nullptr, G.iscan, G.qp, G.tx_type == 9 /* IDTX */ ? 1 : 0, \
                                                           G.nblocks, G.xy, G.src_stride, G.pred_stride, G.recon_stride, bid, reinterpret_cast<int32_t*>(lds))
#define SVT_FRAME_ENC64 enc64_body<PixT, BD, false>(src, pred, recon, nullptr, G.qcoeff, nullptr, G.eob, nullptr, G.iscan, G.qp, G.nblocks, G.xy, G.src_stride, \
                                                    G.pred_stride, G.recon_stride, bid, lds)
    if constexpr (CLS == 3) {
        // the five SQUARE sizes only: with all nineteen bodies inlined into one function the compiler left their register arrays
        // in scratch (2.7 KiB per lane); a call with rectangular groups is launched by class
        switch (G.tx_size) {
        case 0: SVT_FRAME_ENC4; break;
        SVT_FRAME_CASE(1, 8, 8) SVT_FRAME_CASE(2, 16, 16)
        case 3: SVT_FRAME_ENC32; break;
        default: SVT_FRAME_ENC64; break;
        }
    } else if constexpr (CLS == 0) {
        switch (G.tx_size) {
        case 0: SVT_FRAME_ENC4; break;
        SVT_FRAME_CASE(1, 8, 8) SVT_FRAME_CASE(2, 16, 16) SVT_FRAME_CASE(5, 4, 8) SVT_FRAME_CASE(6, 8, 4)
        SVT_FRAME_CASE(7, 8, 16) SVT_FRAME_CASE(8, 16, 8) SVT_FRAME_CASE(13, 4, 16)
        default: SVT_FRAME_STAGED(16, 4); break;          // 14
        }
    } else if constexpr (CLS == 1) {
        switch (G.tx_size) {
        case 3: SVT_FRAME_ENC32; break;
        SVT_FRAME_CASE(9, 16, 32) SVT_FRAME_CASE(10, 32, 16) SVT_FRAME_CASE(11, 32, 64) SVT_FRAME_CASE(12, 64, 32)
        SVT_FRAME_CASE(15, 8, 32) SVT_FRAME_CASE(16, 32, 8) SVT_FRAME_CASE(17, 16, 64)
        default: SVT_FRAME_STAGED(64, 16); break;         // 18
        }
    } else {
        SVT_FRAME_ENC64;
    }
#undef SVT_FRAME_CASE
#undef SVT_FRAME_ENC4
#undef SVT_FRAME_ENC32
#undef SVT_FRAME_ENC64
}
#undef SVT_FRAME_STAGED

}  // namespace svtdev
