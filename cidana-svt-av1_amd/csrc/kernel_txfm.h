// kernel_txfm.h — generic batched 2-D forward / inverse AV1 transforms, all 19
// sizes x 16 types, bit depth 8/10.  One lane owns one column (first pass) then
// one row (second pass) of a block; LPB = max(W,H) lanes per block, 64/LPB
// blocks per wave; the block's wave-private LDS tile does the transpose.
//
// Forward  == Av1TranformTwoDCore_c      (EbTransforms.c:3701-3780)
// Inverse  == inv_txfm2d_add_c           (EbTransforms.c:8180-8265)
#pragma once
#include "dev_common.h"
#include "gen/txfm1d_gen.h"

namespace svtdev {

enum { K1D_DCT = 0, K1D_ADST = 1, K1D_FLIPADST = 2, K1D_IDTX = 3 };

// vtx_tab / htx_tab (EbTransforms.h:87-98)
__device__ constexpr uint8_t kVKind[16] = {0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3};
__device__ constexpr uint8_t kHKind[16] = {0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2};

constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n >> 1); }
// fwd_cos_bit_col / fwd_cos_bit_row [log2(w)-2][log2(h)-2] (EbTransforms.h:141-156)
constexpr int kFwdCosCol[5][5] = {{13, 13, 13, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 13, 12, 13}, {0, 13, 13, 12, 13}, {0, 0, 13, 12, 13}};
constexpr int kFwdCosRow[5][5] = {{13, 13, 12, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 12, 13, 12}, {0, 12, 13, 12, 11}, {0, 0, 12, 11, 10}};
constexpr int fwd_cos_col(int w, int h) { return kFwdCosCol[ilog2c(w) - 2][ilog2c(h) - 2]; }
constexpr int fwd_cos_row(int w, int h) { return kFwdCosRow[ilog2c(w) - 2][ilog2c(h) - 2]; }
// fwd_shift_WxH (EbTransforms.h:120-138): {col up-shift, mid round, final round}
constexpr int fwd_shift(int w, int h, int i) {
    if (w == 64 && h == 64) return i == 0 ? 0 : -2;
    if (w == 32 && h == 64) return i == 0 ? 0 : -2;
    if (w == 64 && h == 32) return i == 0 ? 2 : (i == 1 ? -4 : -2);
    if (w == 16 && h == 64) return i == 0 ? 0 : (i == 1 ? -2 : 0);
    if (w == 64 && h == 16) return i == 0 ? 2 : (i == 1 ? -4 : 0);
    const int m = w > h ? w : h, mn = w < h ? w : h;
    if (i == 0) return 2;
    if (i == 2) return 0;
    if (m == 4) return 0;
    if (m == 8) return -1;
    if (m == 16) return (mn == 4) ? -1 : -2;
    if (m == 32) return (mn == 8) ? -2 : -4;
    return 0;
}
// inv_shift_WxH (EbTransforms.h:268-286): {after rows, after cols}
constexpr int inv_shift0(int w, int h) {
    if (w == h) return w == 4 ? 0 : (w == 8 ? -1 : -2);
    const int m = w > h ? w : h, mn = w < h ? w : h;
    if (m == 8) return 0;                       // 4x8 8x4
    if (m == 16) return -1;                     // 8x16 16x8 4x16 16x4
    if (m == 32) return mn == 8 ? -2 : -1;      // 8x32 32x8 : 16x32 32x16
    return mn == 16 ? -2 : -1;                  // 16x64 64x16 : 32x64 64x32
}

template <int N, int BIT>
__device__ __forceinline__ void fwd1d(int kind, int (&x)[N]) {
    using namespace svtgen;
    if constexpr (N == 4) {
        if (kind == K1D_DCT) svt_fdct4<BIT>(x); else if (kind == K1D_IDTX) svt_fidentity4<BIT>(x); else svt_fadst4<BIT>(x);
    } else if constexpr (N == 8) {
        if (kind == K1D_DCT) svt_fdct8<BIT>(x); else if (kind == K1D_IDTX) svt_fidentity8<BIT>(x); else svt_fadst8<BIT>(x);
    } else if constexpr (N == 16) {
        if (kind == K1D_DCT) svt_fdct16<BIT>(x); else if (kind == K1D_IDTX) svt_fidentity16<BIT>(x); else svt_fadst16<BIT>(x);
    } else if constexpr (N == 32) {
        if (kind == K1D_IDTX) svt_fidentity32<BIT>(x); else svt_fdct32<BIT>(x);
    } else {
        svt_fdct64<BIT>(x);
    }
}
// WIDE: half_btf sums in 64 bits (bd 12, see csrc/gen/txfm1d_gen.h)
template <int N, bool WIDE = false>
__device__ __forceinline__ void inv1d(int kind, int (&x)[N], int lo, int hi) {
    using namespace svtgen;
    constexpr int BIT = 12;   // inv_cos_bit_* are all INV_COS_BIT = 12 (EbTransforms.h:252-267)
    // Clamp-free fast path (N >= 16, where it pays): while gain * sum|x| + slack <= hi every stage clamp returns its argument
    // (txfm_net.clamp_free_bound; constants in gen/txfm1d_gen.h), so a wave whose active lanes all pass the test runs the
    // network without its v_med3_i32 (20-35 % of a pass's issue slots).  x is already clamped to the pass's input range.
    if constexpr (N >= 16 && !WIDE) {
        if (kind != K1D_IDTX) {
            constexpr int NL = N == 64 ? 32 : N;          // live inputs
            int lim;
            if constexpr (N == 16) lim = kind == K1D_DCT ? svt_clamp_free_l1(svt_idct16_gain_q10, svt_idct16_slack, hi) : svt_clamp_free_l1(svt_iadst16_gain_q10, svt_iadst16_slack, hi);
            else if constexpr (N == 32) lim = svt_clamp_free_l1(svt_idct32_gain_q10, svt_idct32_slack, hi);
            else lim = svt_clamp_free_l1(svt_idct64_low32_gain_q10, svt_idct64_low32_slack, hi);
            const int l1 = svt_l1<N>(x, NL);
            if (__builtin_amdgcn_ballot_w64(l1 > lim) == 0) {
                if constexpr (N == 16) { if (kind == K1D_DCT) svt_idct16<BIT, false, false>(x, 0, 0); else svt_iadst16<BIT, false, false>(x, 0, 0); }
                else if constexpr (N == 32) svt_idct32<BIT, false, false>(x, 0, 0);
                else svt_idct64_low32<BIT, false, false>(x, 0, 0);
                return;
            }
        }
    }
    lo = svt_vgpr(lo); hi = svt_vgpr(hi);     // clamp bounds live in two VGPRs (see svt_clamp)
    if constexpr (N == 4) {
        if (kind == K1D_DCT) svt_idct4<BIT, WIDE>(x, lo, hi); else if (kind == K1D_IDTX) svt_iidentity4<BIT, WIDE>(x, lo, hi); else svt_iadst4<BIT, WIDE>(x, lo, hi);
    } else if constexpr (N == 8) {
        if (kind == K1D_DCT) svt_idct8<BIT, WIDE>(x, lo, hi); else if (kind == K1D_IDTX) svt_iidentity8<BIT, WIDE>(x, lo, hi); else svt_iadst8<BIT, WIDE>(x, lo, hi);
    } else if constexpr (N == 16) {
        if (kind == K1D_DCT) svt_idct16<BIT, WIDE>(x, lo, hi); else if (kind == K1D_IDTX) svt_iidentity16<BIT, WIDE>(x, lo, hi); else svt_iadst16<BIT, WIDE>(x, lo, hi);
    } else if constexpr (N == 32) {
        if (kind == K1D_IDTX) svt_iidentity32<BIT, WIDE>(x, lo, hi); else svt_idct32<BIT, WIDE>(x, lo, hi);
    } else {
        // elements 32..63 of every 64-point inverse input are zero in AV1 (only 32x32 coefficients are
        // coded, EbTransforms.c:8226-8240; the callers zero-fill them): zero-propagated network
        svt_idct64_low32<BIT, WIDE>(x, lo, hi);
    }
}

template <int SH>
__device__ __forceinline__ int round_shift_c(int v) {   // av1_round_shift_array_c, bit = SH
    if constexpr (SH == 0) return v;
    else if constexpr (SH > 0) return (v + (1 << (SH - 1))) >> SH;
    else return v * (1 << (-SH));
}
__device__ __forceinline__ int mul_q12(int v, int k) {  // round_shift((int64)v * k, 12)
    return (int)(((long long)v * k + 2048) >> 12);
}

constexpr int TX_WAVES = 4;
template <int W, int H> struct TxGeom {
    static constexpr int LPB = W > H ? W : H;     // lanes per block
    static constexpr int BPW = 64 / LPB;          // blocks per wave
    static constexpr int PITCH = W + 1;           // odd word pitch: conflict-free transposes
    static constexpr int TILE = H * PITCH;
    static constexpr bool RECT2 = (W == 2 * H) || (H == 2 * W);
};

// ---------------------------------------------------------------------------
// forward: in = int16 residual blocks (block b at in + b*in_block_pitch, row
// stride in_stride), out = dense W*H int32 per block (reference layout).
// ---------------------------------------------------------------------------
template <int W, int H>
__global__ __launch_bounds__(TX_WAVES * 64) void fwd_txfm2d_kernel(
    const int16_t* __restrict__ in, int32_t* __restrict__ out, uint32_t in_stride,
    size_t in_block_pitch, int tx_type, uint32_t nblocks) {
    using G = TxGeom<W, H>;
    __shared__ int32_t lds[TX_WAVES * G::BPW * G::TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / G::LPB, l = lane % G::LPB;
    const uint32_t blk = (blockIdx.x * TX_WAVES + wave) * G::BPW + sub;
    const bool valid = blk < nblocks;
    int32_t* tile = lds + (wave * G::BPW + sub) * G::TILE;
    const int vk = kVKind[tx_type], hk = kHKind[tx_type];
    const bool ud = vk == K1D_FLIPADST, lr = hk == K1D_FLIPADST;
    constexpr int CBC = fwd_cos_col(W, H), CBR = fwd_cos_row(W, H);
    constexpr int S0 = fwd_shift(W, H, 0), S1 = fwd_shift(W, H, 1), S2 = fwd_shift(W, H, 2);

    if (l < W) {
        int x[H];
        const int16_t* p = in + (size_t)blk * in_block_pitch + l;
#pragma unroll
        for (int r = 0; r < H; r++) {
            const int v = valid ? (int)p[(size_t)(ud ? H - 1 - r : r) * in_stride] : 0;
            x[r] = round_shift_c<-S0>(v);
        }
        fwd1d<H, CBC>(vk, x);
        const int cdst = lr ? W - 1 - l : l;
#pragma unroll
        for (int r = 0; r < H; r++) tile[r * G::PITCH + cdst] = round_shift_c<-S1>(x[r]);
    }
    wave_lds_fence();
    if (l < H) {
        int y[W];
#pragma unroll
        for (int c = 0; c < W; c++) y[c] = tile[l * G::PITCH + c];
        fwd1d<W, CBR>(hk, y);
        if (valid) {
            int32_t* o = out + (size_t)blk * (W * H) + l * W;
#pragma unroll
            for (int c = 0; c < W; c += 4) {
                int4 v;
                int* vp = &v.x;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int t = round_shift_c<-S2>(y[c + j]);
                    if (G::RECT2) t = mul_q12(t, 5793);      // x sqrt(2) for 2:1 rectangles
                    vp[j] = t;
                }
                *reinterpret_cast<int4*>(o + c) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// generic fused chain, every size / type / bit depth:
//   residual(src, pred) -> fwd 2-D -> [64-pt: energy + pack] -> quantize_b -> eob (+ SAD)
// (Av1EncodeLoop, EbCodingLoop.c:617 -> :655 -> :673; av1_estimate_transform's
// 64-pt handling EbTransforms.c:4377-4408; 16-bit twin Av1EncodeLoop16bit :1020).
// src / pred are PLANES: block b starts at (xy[b] >> 16) * stride + (xy[b] & 0xffff)
// when xy != NULL, else at b * block_pitch.  Outputs are dense, packed
// KW*KH = min(W,32)*min(H,32) coefficients per block.  The tuned 32x32 8-bit
// kernel (kernel_fused32.h) serves the headline case; this one is the general path.
// ---------------------------------------------------------------------------
template <int W, int H, typename PixT>
__global__ __launch_bounds__(TX_WAVES * 64) void fwd_quant_generic_kernel(
    const PixT* __restrict__ src, uint32_t src_stride, size_t src_block_pitch, const PixT* __restrict__ pred,
    uint32_t pred_stride, size_t pred_block_pitch, const uint32_t* __restrict__ xy, int tx_type, QParams qp,
    const int16_t* __restrict__ iscan, int32_t* __restrict__ coeff, int32_t* __restrict__ qcoeff,
    int32_t* __restrict__ dqcoeff, uint16_t* __restrict__ eob, uint32_t* __restrict__ sad,
    unsigned long long* __restrict__ energy, uint32_t nblocks) {
    using G = TxGeom<W, H>;
    constexpr int KW = W > 32 ? 32 : W, KH = H > 32 ? 32 : H;
    constexpr int FAST = sizeof(PixT) == 1 ? 1 : 0;   // 24-bit quantiser arithmetic is proven for 8-bit only
    __shared__ int32_t lds[TX_WAVES * G::BPW * G::TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / G::LPB, l = lane % G::LPB;
    const uint32_t blk = (blockIdx.x * TX_WAVES + wave) * G::BPW + sub;
    const bool valid = blk < nblocks;
    int32_t* tile = lds + (wave * G::BPW + sub) * G::TILE;
    const int vk = kVKind[tx_type], hk = kHKind[tx_type];
    const bool ud = vk == K1D_FLIPADST, lr = hk == K1D_FLIPADST;
    constexpr int CBC = fwd_cos_col(W, H), CBR = fwd_cos_row(W, H);
    constexpr int S0 = fwd_shift(W, H, 0), S1 = fwd_shift(W, H, 1), S2 = fwd_shift(W, H, 2);

    unsigned sad_acc = 0;
    if (l < W) {
        int x[H];
        size_t so = 0, po = 0;
        if (valid) {
            if (xy) {
                const uint32_t v = xy[blk];
                so = (size_t)(v >> 16) * src_stride + (v & 0xffffu);
                po = (size_t)(v >> 16) * pred_stride + (v & 0xffffu);
            } else {
                so = (size_t)blk * src_block_pitch;
                po = (size_t)blk * pred_block_pitch;
            }
        }
#pragma unroll
        for (int r = 0; r < H; r++) {
            const int rr = ud ? H - 1 - r : r;
            const int d = valid ? (int)src[so + (size_t)rr * src_stride + l] - (int)pred[po + (size_t)rr * pred_stride + l] : 0;
            sad_acc += (unsigned)(d < 0 ? -d : d);
            x[r] = round_shift_c<-S0>(d);
        }
        fwd1d<H, CBC>(vk, x);
        const int cdst = lr ? W - 1 - l : l;
#pragma unroll
        for (int r = 0; r < H; r++) tile[r * G::PITCH + cdst] = round_shift_c<-S1>(x[r]);
    }
    wave_lds_fence();
    int eob_acc = 0;
    unsigned long long en = 0;
    if (l < H) {
        int y[W];
#pragma unroll
        for (int c = 0; c < W; c++) y[c] = tile[l * G::PITCH + c];
        fwd1d<W, CBR>(hk, y);
#pragma unroll
        for (int c = 0; c < W; c++) {
            int t = round_shift_c<-S2>(y[c]);
            if (G::RECT2) t = mul_q12(t, 5793);
            y[c] = t;
        }
        if (W > 32 || H > 32) {                  // three_quad_energy of the discarded region
#pragma unroll
            for (int c = 0; c < W; c++)
                if (l >= KH || c >= KW) { const long long v = y[c]; en += (unsigned long long)(v * v); }
        }
        if (l < KH && valid) {
            const size_t o = (size_t)blk * (KW * KH) + (size_t)l * KW;
#pragma unroll
            for (int c = 0; c < KW; c += 4) {
                int4 cv, qv, dv;
                int* cp = &cv.x; int* qp4 = &qv.x; int* dp = &dv.x;
                const uint2 is = *reinterpret_cast<const uint2*>(iscan + l * KW + c);
                const unsigned isv[4] = {is.x & 0xffffu, is.x >> 16, is.y & 0xffffu, is.y >> 16};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    cp[j] = y[c + j];
                    // power-of-two quant_shift (every av1_build_quantizer table): the one-product form is exact for
                    // |coeff| + round < 2^24, i.e. for 8- and 10-bit transform output alike; otherwise the general forms
                    if (qp.fast_ok) quant_one<2>(y[c + j], (l == 0 && c + j == 0) ? 0 : 1, qp, qp4[j], dp[j]);
                    else quant_one<FAST>(y[c + j], (l == 0 && c + j == 0) ? 0 : 1, qp, qp4[j], dp[j]);
                    eob_acc = max(eob_acc, qp4[j] ? (int)isv[j] + 1 : 0);
                }
                *reinterpret_cast<int4*>(coeff + o + c) = cv;
                *reinterpret_cast<int4*>(qcoeff + o + c) = qv;
                *reinterpret_cast<int4*>(dqcoeff + o + c) = dv;
            }
        }
    }
    eob_acc = group_max<G::LPB>(eob_acc);
    sad_acc = group_sum<G::LPB>(sad_acc);
    if (W > 32 || H > 32) en = group_sum64<G::LPB>(en);
    if (valid && l == 0) {
        eob[blk] = (uint16_t)eob_acc;
        if (sad) sad[blk] = sad_acc;
        if (energy) energy[blk] = en;
    }
}

// ---------------------------------------------------------------------------
// inverse + add: in = packed min(W,32) x min(H,32) int32 coefficients per block
// (dense), dst = PixT samples; block b at dst + (dst_offsets ? dst_offsets[b]
// : b * dst_block_pitch), row stride dst_stride (elements).
// ---------------------------------------------------------------------------
template <int W, int H, typename PixT, bool WIDE = false>
__global__ __launch_bounds__(TX_WAVES * 64) void inv_txfm2d_add_kernel(
    const int32_t* __restrict__ in, PixT* __restrict__ dst, int32_t dst_stride, size_t dst_block_pitch,
    const uint32_t* __restrict__ dst_offsets, int tx_type, int bd, uint32_t nblocks) {
    using G = TxGeom<W, H>;
    constexpr int KW = W > 32 ? 32 : W, KH = H > 32 ? 32 : H;
    __shared__ int32_t lds[TX_WAVES * G::BPW * G::TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / G::LPB, l = lane % G::LPB;
    const uint32_t blk = (blockIdx.x * TX_WAVES + wave) * G::BPW + sub;
    const bool valid = blk < nblocks;
    int32_t* tile = lds + (wave * G::BPW + sub) * G::TILE;
    const int vk = kVKind[tx_type], hk = kHKind[tx_type];
    const bool ud = vk == K1D_FLIPADST, lr = hk == K1D_FLIPADST;
    // av1_gen_inv_stage_range (EbTransforms.c:5404-5456)
    const int row_bits = bd == 8 ? 16 : (bd == 10 ? 18 : 20);
    const int col_bits = bd == 12 ? 18 : 16;
    const int in_bits = bd + 8;
    const int colin_bits = bd + 6 > 16 ? bd + 6 : 16;
    constexpr int S0 = inv_shift0(W, H);

    if (l < H) {   // row pass: lane l owns row l
        int x[W];
        if (l < KH) {
            const int32_t* p = in + (size_t)blk * (KW * KH) + l * KW;
#pragma unroll
            for (int c = 0; c < W; c++) {
                int v = (c < KW && valid) ? p[c] : 0;        // 64-pt: zero re-expansion (:8299-8315)
                if (G::RECT2) v = mul_q12(v, 2896);          // x 1/sqrt(2)
                x[c] = svtgen::svt_clamp(v, -(1 << (in_bits - 1)), (1 << (in_bits - 1)) - 1);
            }
            inv1d<W, WIDE>(hk, x, -(1 << (row_bits - 1)), (1 << (row_bits - 1)) - 1);
#pragma unroll
            for (int c = 0; c < W; c++) tile[l * G::PITCH + c] = round_shift_c<-S0>(x[c]);
        } else {
#pragma unroll
            for (int c = 0; c < W; c++) tile[l * G::PITCH + c] = 0;   // all-zero rows transform to zero
        }
    }
    wave_lds_fence();
    if (l < W) {   // column pass: lane l owns column l
        int y[H];
        const int csrc = lr ? W - 1 - l : l;
#pragma unroll
        for (int r = 0; r < H; r++)
            y[r] = svtgen::svt_clamp(tile[r * G::PITCH + csrc], -(1 << (colin_bits - 1)), (1 << (colin_bits - 1)) - 1);
        inv1d<H, WIDE>(vk, y, -(1 << (col_bits - 1)), (1 << (col_bits - 1)) - 1);
        if (valid) {
            const size_t base = dst_offsets ? (size_t)dst_offsets[blk] : (size_t)blk * dst_block_pitch;
            PixT* d = dst + base + l;
            const int maxpix = (1 << bd) - 1;
#pragma unroll
            for (int r = 0; r < H; r++) {
                const int res = round_shift_c<4>(y[ud ? H - 1 - r : r]);   // shift[1] = -4 for every size
                const int v = (int)d[(size_t)r * dst_stride] + res;
                d[(size_t)r * dst_stride] = (PixT)min(max(v, 0), maxpix);
            }
        }
    }
}

}  // namespace svtdev
