// svt_hip_txfm.hip — entry points of the transform family of libsvt_hip_dsp.so (include/svt_hip_dsp.h): forward / inverse
// 2-D transforms, 64-point packing, quantiser, the fused chains (headline, encode pass, planes) and their drop-ins.
#include "host_common.h"
#include "kernel_fused32.h"
#include "kernel_quant.h"
#include "kernel_txfm.h"
#include "kernel_txfm_staged.h"
#include "kernel_enc64.h"
#include "kernel_frame.h"

using namespace svtdev;
using namespace svthost;

namespace {

int rpot(int v, int n) { return n == 0 ? v : ((v + (1 << (n - 1))) >> n); }

QParams make_qparams(const int16_t* zbin, const int16_t* round, const int16_t* quant,
                     const int16_t* quant_shift, const int16_t* dequant, int log_scale) {
    QParams qp;
    for (int i = 0; i < 2; i++) {
        qp.zbin[i] = rpot(zbin[i], log_scale);       // EbFullLoop.c:248-249
        qp.round[i] = rpot(round[i], log_scale);     // :277
        qp.quant_m[i] = (uint32_t)((int)quant[i] + 65536);
        qp.quant_shift[i] = quant_shift[i];
        qp.dequant[i] = dequant[i];
    }
    qp.log_scale = log_scale;
    qp.fast_ok = 1;
    for (int i = 0; i < 2; i++) {
        const int qs = quant_shift[i];
        int k = -1;
        if (qs > 0 && (qs & (qs - 1)) == 0) { k = 0; while ((1 << k) != qs) k++; }
        const int sh = 32 - log_scale - k;
        qp.fast_sh[i] = sh;
        if (k < 0 || sh < 16 || sh > 31 || dequant[i] < 0 || qp.round[i] < 0 || qp.quant_m[i] >= (1u << 17)) qp.fast_ok = 0;
        qp.quant_hi[i] = (sh >= 16 && sh <= 31) ? qp.quant_m[i] << (31 - sh) : 0u;
        qp.zbin2[i] = 2 * qp.zbin[i];
        qp.round2[i] = 2 * qp.round[i];
    }
    return qp;
}


// ---------------------------------------------------------------------------
// 64-pt packing kernel (HandleTransform64x64_c & friends)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack64_kernel(int32_t* __restrict__ coeff,
                                                     unsigned long long* __restrict__ energy, int w, int h,
                                                     uint32_t nblocks) {
    // one workgroup per block; reads complete before any write (barrier)
    const uint32_t blk = blockIdx.x;
    if (blk >= nblocks) return;
    const int kw = w > 32 ? 32 : w, kh = h > 32 ? 32 : h;
    int32_t* c = coeff + (size_t)blk * w * h;
    __shared__ int32_t keep[1024];
    __shared__ unsigned long long part[4];
    unsigned long long e = 0;
    for (int i = threadIdx.x; i < w * h; i += 256) {
        const int r = i / w, cc = i - r * w;
        const long long v = c[i];
        if (r < kh && cc < kw) keep[r * kw + cc] = (int32_t)v;
        else e += (unsigned long long)(v * v);
    }
    e = group_sum64<64>(e);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = e;
    __syncthreads();
    for (int i = threadIdx.x; i < w * h; i += 256) c[i] = i < kw * kh ? keep[i] : 0;
    if (threadIdx.x == 0 && energy) energy[blk] = part[0] + part[1] + part[2] + part[3];
}

template <int W, int H>
int launch_fwd(const int16_t* in, uint32_t in_stride, size_t pitch, int32_t* out, size_t n, int tx_type,
               hipStream_t s) {
    constexpr int BPW = TxGeom<W, H>::BPW;
    const uint32_t per_wg = TX_WAVES * BPW;
    const uint32_t grid = (uint32_t)((n + per_wg - 1) / per_wg);
    hipLaunchKernelGGL((fwd_txfm2d_kernel<W, H>), dim3(grid), dim3(TX_WAVES * 64), 0, s, in, out, in_stride,
                       pitch, tx_type, (uint32_t)n);
    return launch_status("fwd_txfm2d");
}
template <int W, int H>
int launch_fwd_staged(const int16_t* in, int32_t* out, size_t n, int tx_type, hipStream_t s) {
    using SG = StagedGeom<W, H>;
    const uint32_t per_wg = SG::WAVES * TxGeom<W, H>::BPW;
    QParams qp = {};
    hipLaunchKernelGGL((fwd_staged_kernel<W, H, 0>), dim3((uint32_t)((n + per_wg - 1) / per_wg)), dim3(SG::WAVES * 64), 0, s,
                       (const void*)in, (const uint8_t*)nullptr, out, (int32_t*)nullptr, (int32_t*)nullptr, (uint16_t*)nullptr,
                       (uint32_t*)nullptr, (unsigned long long*)nullptr, (const int16_t*)nullptr, qp, tx_type, (uint32_t)n);
    return launch_status("fwd_staged");
}
template <int W, int H>
int launch_fq_staged(const void* src, const void* pred, int is16, const uint32_t* xy, uint32_t ss, uint32_t ps, size_t n, int tx_type,
                     const QParams& qp, const int16_t* iscan, int32_t* co, int32_t* q, int32_t* dq, uint16_t* eob, uint32_t* sad,
                     uint64_t* energy, hipStream_t s) {
    using SG = StagedGeom<W, H>;
    const uint32_t per_wg = SG::WAVES * TxGeom<W, H>::BPW;
    const dim3 grid((uint32_t)((n + per_wg - 1) / per_wg)), block(SG::WAVES * 64);
    if (is16)
        hipLaunchKernelGGL((fwd_staged_kernel<W, H, 1, uint16_t>), grid, block, 0, s, src, pred, co, q, dq, eob, sad,
                           (unsigned long long*)energy, iscan, qp, tx_type, (uint32_t)n, xy, ss, ps);
    else
        hipLaunchKernelGGL((fwd_staged_kernel<W, H, 1, uint8_t>), grid, block, 0, s, src, pred, co, q, dq, eob, sad,
                           (unsigned long long*)energy, iscan, qp, tx_type, (uint32_t)n, xy, ss, ps);
    return launch_status("fwd_quant_staged");
}
template <int W, int H>
int launch_enc_staged(const void* src, const void* pred, void* recon, int is16, int32_t* co, int32_t* q, int32_t* dq, uint16_t* eob,
                      uint32_t* sad, const int16_t* iscan, const QParams& qp, int tx_type, size_t n, const uint32_t* xy, uint32_t ss,
                      uint32_t ps, uint32_t rs, hipStream_t s) {
    using SG = StagedGeom<W, H>;
    const uint32_t per_wg = SG::WAVES * TxGeom<W, H>::BPW;
    const dim3 grid((uint32_t)((n + per_wg - 1) / per_wg)), block(SG::WAVES * 64);
#define ENCL(KEEP, T, B) hipLaunchKernelGGL((enc_staged_kernel<W, H, KEEP, T, B>), grid, block, 0, s, (const T*)src, (const T*)pred, (T*)recon, co, q, dq, \
                                          eob, sad, iscan, qp, tx_type, (uint32_t)n, xy, ss, ps, rs)
    if (is16) { if (co) ENCL(true, uint16_t, 10); else ENCL(false, uint16_t, 10); }
    else { if (co) ENCL(true, uint8_t, 8); else ENCL(false, uint8_t, 8); }
#undef ENCL
    return launch_status("encode_recon_staged");
}
template <int W, int H>
int launch_inv_staged(const int32_t* in, void* dst, int is16, size_t n, int tx_type, int bd, const uint32_t* offs, int32_t stride,
                      hipStream_t s) {
    using SG = StagedGeom<W, H>;
    const uint32_t per_wg = SG::WAVES * TxGeom<W, H>::BPW;
    const uint32_t grid = (uint32_t)((n + per_wg - 1) / per_wg);
    // 64-point sizes at bd <= 10: 16-bit transpose tile (the column input is clamped to 16 bits there anyway), which lifts
    // their LDS-limited 2 waves per SIMD to 4; smaller sizes are not LDS-limited and sub-dword LDS writes are slower
    constexpr bool T16 = W >= 64 || H >= 64;
    if (is16) hipLaunchKernelGGL((inv_staged_kernel<W, H, uint16_t, T16>), dim3(grid), dim3(SG::WAVES * 64), 0, s, in, (uint16_t*)dst, tx_type, bd, (uint32_t)n, offs, stride);
    else hipLaunchKernelGGL((inv_staged_kernel<W, H, uint8_t, T16>), dim3(grid), dim3(SG::WAVES * 64), 0, s, in, (uint8_t*)dst, tx_type, bd, (uint32_t)n, offs, stride);
    return launch_status("inv_staged");
}
template <int W, int H>
int launch_inv(const int32_t* in, void* dst, int is16, int32_t stride, size_t pitch, const uint32_t* offs,
               size_t n, int tx_type, int bd, hipStream_t s) {
    constexpr int BPW = TxGeom<W, H>::BPW;
    const uint32_t per_wg = TX_WAVES * BPW;
    const uint32_t grid = (uint32_t)((n + per_wg - 1) / per_wg);
    if (is16 && bd > 10)      // bd 12: half_btf sums need 64 bits (txfm1d_gen.h, WIDE)
        hipLaunchKernelGGL((inv_txfm2d_add_kernel<W, H, uint16_t, true>), dim3(grid), dim3(TX_WAVES * 64), 0, s, in,
                           (uint16_t*)dst, stride, pitch, offs, tx_type, bd, (uint32_t)n);
    else if (is16)
        hipLaunchKernelGGL((inv_txfm2d_add_kernel<W, H, uint16_t>), dim3(grid), dim3(TX_WAVES * 64), 0, s, in,
                           (uint16_t*)dst, stride, pitch, offs, tx_type, bd, (uint32_t)n);
    else
        hipLaunchKernelGGL((inv_txfm2d_add_kernel<W, H, uint8_t>), dim3(grid), dim3(TX_WAVES * 64), 0, s, in,
                           (uint8_t*)dst, stride, pitch, offs, tx_type, bd, (uint32_t)n);
    return launch_status("inv_txfm2d_add");
}
}  // namespace

// ===========================================================================
// (B) batched API
// ===========================================================================
extern "C" int svt_hip_fwd_txfm2d_batch(const int16_t* d_in, uint32_t in_stride, size_t in_block_pitch,
                                        int32_t* d_out, size_t nblocks, int tx_size, int tx_type, int bd,
                                        void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_in || !d_out) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (!txfm_allowed(tx_size, tx_type)) return set_err(SVT_HIP_ERR_INVALID, "tx_size %d / tx_type %d not defined by the reference", tx_size, tx_type);
    if (bd != 8 && bd != 10) return set_err(SVT_HIP_ERR_INVALID, "bit depth %d", bd);
    if (nblocks == 0) return SVT_HIP_OK;
    if (nblocks > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "nblocks too large");
    hipStream_t s = (hipStream_t)stream;
    if (tx_size == SVT_TX_32X32 && in_stride == 32 && in_block_pitch == 1024 && ((uintptr_t)d_in & 15) == 0 &&
        ((uintptr_t)d_out & 15) == 0 && (tx_type == SVT_DCT_DCT || tx_type == SVT_IDTX)) {
        const uint32_t npairs = (uint32_t)((nblocks + 1) / 2);
        QParams qp = {};
        hipLaunchKernelGGL((fwd32_kernel<0, false, false>), dim3((npairs + F32_WAVES - 1) / F32_WAVES), dim3(F32_WAVES * 64), 0,
                           s, (const void*)d_in, (const uint8_t*)nullptr, d_out, (int32_t*)nullptr, (int32_t*)nullptr,
                           (uint16_t*)nullptr, (uint32_t*)nullptr, (const int16_t*)nullptr, qp, tx_type == SVT_IDTX ? 1 : 0,
                           (uint32_t)nblocks);
        return launch_status("fwd32");
    }
    if (!g_tune_no_staged && in_stride == (uint32_t)kTxW[tx_size] && in_block_pitch == (size_t)kTxW[tx_size] * kTxH[tx_size] &&
        ((uintptr_t)d_in & 15) == 0 && ((uintptr_t)d_out & 15) == 0) {
#define CALLS(W, H) launch_fwd_staged<W, H>(d_in, d_out, nblocks, tx_type, s)
        TX_SWITCH(tx_size, CALLS)
#undef CALLS
    }
#define CALL(W, H) launch_fwd<W, H>(d_in, in_stride, in_block_pitch, d_out, nblocks, tx_type, s)
    TX_SWITCH(tx_size, CALL)
#undef CALL
}

extern "C" int svt_hip_pack64_batch(int32_t* d_coeff, uint64_t* d_energy, size_t nblocks, int tx_size,
                                    void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (tx_size < 0 || tx_size >= SVT_TX_SIZES_ALL || !d_coeff) return set_err(SVT_HIP_ERR_INVALID, "bad argument");
    const int w = kTxW[tx_size], h = kTxH[tx_size];
    if (nblocks == 0) return SVT_HIP_OK;
    if (w != 64 && h != 64) {
        if (d_energy) HIP_TRY(hipMemsetAsync(d_energy, 0, nblocks * sizeof(uint64_t), (hipStream_t)stream));
        return SVT_HIP_OK;
    }
    hipLaunchKernelGGL(pack64_kernel, dim3((uint32_t)nblocks), dim3(256), 0, (hipStream_t)stream, d_coeff,
                       (unsigned long long*)d_energy, w, h, (uint32_t)nblocks);
    return launch_status("pack64");
}

extern "C" int svt_hip_inv_txfm2d_add_batch(const int32_t* d_coeff, void* d_dst, int dst_is_16bit,
                                            int32_t dst_stride, size_t dst_block_pitch,
                                            const uint32_t* d_dst_offsets, size_t nblocks, int tx_size,
                                            int tx_type, int bd, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_coeff || !d_dst) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (!txfm_allowed(tx_size, tx_type)) return set_err(SVT_HIP_ERR_INVALID, "tx_size %d / tx_type %d not defined by the reference", tx_size, tx_type);
    if (bd != 8 && bd != 10 && bd != 12) return set_err(SVT_HIP_ERR_INVALID, "bit depth %d", bd);
    if (!dst_is_16bit && bd != 8) return set_err(SVT_HIP_ERR_INVALID, "8-bit destination needs bd = 8");
    if (nblocks == 0) return SVT_HIP_OK;
    hipStream_t s = (hipStream_t)stream;
    if (bd > 10) {
        // bd 12 (not an encoder configuration, only the C inverse kernels define it): the general kernel with 64-bit
        // half_btf sums; the tuned kernels' 32-bit multiply-accumulate chains are exact for bd <= 10 only
#define CALL(W, H) launch_inv<W, H>(d_coeff, d_dst, dst_is_16bit, dst_stride, dst_block_pitch, d_dst_offsets, nblocks, tx_type, bd, s)
        TX_SWITCH(tx_size, CALL)
#undef CALL
    }
    if (tx_size == SVT_TX_32X32 && ((uintptr_t)d_coeff & 15) == 0 && (tx_type == SVT_DCT_DCT || tx_type == SVT_IDTX)) {
        if (!dst_is_16bit && (g_tune_inv32_waves != 4 || g_tune_inv32_var != 0)) {     // tuning probes (tools/tune_inv32.py)
#define INVV(WV, VR) if (g_tune_inv32_waves == WV && g_tune_inv32_var == VR) { \
            hipLaunchKernelGGL((inv32_kernel<uint8_t, 8, WV, VR>), dim3((uint32_t)((nblocks + 2 * WV - 1) / (2 * WV))), dim3(WV * 64), 0, s, d_coeff, \
                               (uint8_t*)d_dst, dst_stride, dst_block_pitch, d_dst_offsets, tx_type == SVT_IDTX ? 1 : 0, (uint32_t)nblocks); \
            return launch_status("inv32 probe"); }
            INVV(4, 1) INVV(4, 2) INVV(4, 4) INVV(4, 5) INVV(2, 0) INVV(4, 8)
#undef INVV
            return set_err(SVT_HIP_ERR_INVALID, "inv32 probe variant not built");
        }
        const uint32_t grid = (uint32_t)((nblocks + 2 * F32_WAVES - 1) / (2 * F32_WAVES));
#define INV32(T, B) hipLaunchKernelGGL((inv32_kernel<T, B>), dim3(grid), dim3(F32_WAVES * 64), 0, s, d_coeff, (T*)d_dst, dst_stride, \
                                      dst_block_pitch, d_dst_offsets, tx_type == SVT_IDTX ? 1 : 0, (uint32_t)nblocks)
        if (dst_is_16bit) { if (bd == 8) INV32(uint16_t, 8); else INV32(uint16_t, 10); }
        else INV32(uint8_t, 8);
#undef INV32
        return launch_status("inv32");
    }
    const bool dense_dst = !d_dst_offsets && dst_stride == kTxW[tx_size] && dst_block_pitch == (size_t)kTxW[tx_size] * kTxH[tx_size] &&
                           ((uintptr_t)d_dst & 15) == 0 && (kTxW[tx_size] * kTxH[tx_size] * (dst_is_16bit ? 2 : 1)) % 16 == 0;
    // 4-sample-wide 8-bit rows would be 4-B chunks of an unaligned plane: leave those to the general kernel
    const bool plane_dst = d_dst_offsets && kTxW[tx_size] * (dst_is_16bit ? 2 : 1) >= 8 && !g_tune_no_inv_planes;
    if (!g_tune_no_staged && (dense_dst || plane_dst) && ((uintptr_t)d_coeff & 15) == 0) {
#define CALLS(W, H) launch_inv_staged<W, H>(d_coeff, d_dst, dst_is_16bit, nblocks, tx_type, bd, d_dst_offsets, dst_stride, s)
        TX_SWITCH(tx_size, CALLS)
#undef CALLS
    }
#define CALL(W, H) launch_inv<W, H>(d_coeff, d_dst, dst_is_16bit, dst_stride, dst_block_pitch, d_dst_offsets, nblocks, tx_type, bd, s)
    TX_SWITCH(tx_size, CALL)
#undef CALL
}

extern "C" int svt_hip_quantize_b_batch(const int32_t* d_coeff, size_t n_coeffs, int skip_block,
                                        const int16_t* zbin, const int16_t* round, const int16_t* quant,
                                        const int16_t* quant_shift, int32_t* d_qcoeff, int32_t* d_dqcoeff,
                                        const int16_t* dequant, uint16_t* d_eob, const int16_t* d_iscan,
                                        int log_scale, size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_coeff || !d_qcoeff || !d_dqcoeff || !d_eob || !d_iscan || !zbin || !round || !quant || !quant_shift || !dequant)
        return set_err(SVT_HIP_ERR_INVALID, "NULL argument");
    if (n_coeffs < 16 || n_coeffs > 4096 || (n_coeffs & 15)) return set_err(SVT_HIP_ERR_INVALID, "n_coeffs %zu", n_coeffs);
    if (log_scale < 0 || log_scale > 2) return set_err(SVT_HIP_ERR_INVALID, "log_scale %d", log_scale);
    if (nblocks == 0) return SVT_HIP_OK;
    const QParams qp = make_qparams(zbin, round, quant, quant_shift, dequant, log_scale);
    hipStream_t s = (hipStream_t)stream;
    const int n = (int)n_coeffs;
    const int lpb = n / 4 >= 64 ? 64 : n / 4;   // 4, 8, 16, 32 or 64 lanes per block
#define QL(L)                                                                                          \
    {                                                                                                  \
        const uint32_t per_wg = 4 * (64 / L);                                                          \
        hipLaunchKernelGGL((quantize_b_kernel<L>), dim3((uint32_t)((nblocks + per_wg - 1) / per_wg)), dim3(256), 0, \
                           s, d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_iscan, qp, n, skip_block, (uint32_t)nblocks); \
    }
    switch (lpb) {
    case 4: QL(4) break; case 8: QL(8) break; case 16: QL(16) break; case 32: QL(32) break;
    default: QL(64) break;
    }
#undef QL
    return launch_status("quantize_b");
}

extern "C" int svt_hip_fwd_quant_sad_batch(const uint8_t* d_src, const uint8_t* d_pred, size_t nblocks,
                                           int tx_size, int tx_type, const int16_t* zbin, const int16_t* round,
                                           const int16_t* quant, const int16_t* quant_shift,
                                           const int16_t* dequant, const int16_t* d_iscan, int32_t* d_coeff,
                                           int32_t* d_qcoeff, int32_t* d_dqcoeff, uint16_t* d_eob,
                                           uint32_t* d_sad, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_src || !d_pred || !d_coeff || !d_qcoeff || !d_dqcoeff || !d_eob || !d_iscan || !zbin || !round || !quant || !quant_shift || !dequant)
        return set_err(SVT_HIP_ERR_INVALID, "NULL argument");
    if (tx_size != SVT_TX_32X32 || (tx_type != SVT_DCT_DCT && tx_type != SVT_IDTX))
        return svt_hip_fwd_quant_planes_batch(d_src, 0, d_pred, 0, nullptr, nblocks, 0, 8, tx_size, tx_type, zbin, round, quant,
                                              quant_shift, dequant, d_iscan, d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_sad,
                                              nullptr, stream);
    if (nblocks == 0) return SVT_HIP_OK;
    if (nblocks > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "nblocks too large");
    const QParams qp = make_qparams(zbin, round, quant, quant_shift, dequant, 1);
    // FAST24 precondition of the fused kernel (dev_common.h quant_one<true>)
    for (int i = 0; i < 2; i++)
        if (qp.quant_shift[i] < 0 || qp.dequant[i] < 0 || qp.round[i] < 0)
            return set_err(SVT_HIP_ERR_INVALID, "negative quantizer table entry");
    const uint32_t npairs = (uint32_t)((nblocks + 1) / 2);
    uint32_t grid = (npairs + F32_WAVES - 1) / F32_WAVES;
    const uint32_t max_grid = (uint32_t)g_num_cu * (uint32_t)g_tune_f32_wg_per_cu;
    if (g_tune_f32_wg_per_cu > 0 && grid > max_grid) grid = max_grid;
    hipStream_t s = (hipStream_t)stream;
    // QMODE 2 needs power-of-two quant_shift (every av1_build_quantizer table); else the 24-bit general form
#define F32_LAUNCH_Q(SAD, MW, NT, QM)                                                                              \
    hipLaunchKernelGGL((fwd32_kernel<1, true, SAD, MW, NT, QM>), dim3(grid), dim3(F32_WAVES * 64), 0, s,           \
                       (const void*)d_src, d_pred, d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_sad, d_iscan, qp,           \
                       tx_type == SVT_IDTX ? 1 : 0, (uint32_t)nblocks)
    const bool fastq = qp.fast_ok && !g_tune_f32_qmode1;
    if (g_tune_f32_nt && fastq) {
        if (d_sad) F32_LAUNCH_Q(true, 1, true, 2); else F32_LAUNCH_Q(false, 1, true, 2);
    } else if (g_tune_f32_nt) {
        if (d_sad) F32_LAUNCH_Q(true, 1, true, 1); else F32_LAUNCH_Q(false, 1, true, 1);
    } else if (!fastq) {
        if (d_sad) F32_LAUNCH_Q(true, 1, false, 1); else F32_LAUNCH_Q(false, 1, false, 1);
    } else {
        if (d_sad) F32_LAUNCH_Q(true, 1, false, 2); else F32_LAUNCH_Q(false, 1, false, 2);
    }
#undef F32_LAUNCH_Q
#undef F32_LAUNCH
    return launch_status("fwd_quant_sad_32x32");
}

static int encode_recon_impl(const void* d_src_v, uint32_t src_stride, const void* d_pred_v, uint32_t pred_stride,
                             void* d_recon_v, uint32_t recon_stride, const uint32_t* d_xy, int is_16bit, int bd, size_t nblocks, int tx_size,
                             int tx_type, const int16_t* zbin, const int16_t* round, const int16_t* quant,
                             const int16_t* quant_shift, const int16_t* dequant, const int16_t* d_iscan,
                             int32_t* d_coeff, int32_t* d_qcoeff, int32_t* d_dqcoeff, uint16_t* d_eob,
                             uint32_t* d_sad, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    const uint8_t* d_src = (const uint8_t*)d_src_v; const uint8_t* d_pred = (const uint8_t*)d_pred_v; uint8_t* d_recon = (uint8_t*)d_recon_v;
    if (!d_src || !d_pred || !d_qcoeff || !d_eob || !d_recon || !d_iscan || !zbin || !round || !quant || !quant_shift || !dequant)
        return set_err(SVT_HIP_ERR_INVALID, "NULL argument");
    if (!txfm_allowed(tx_size, tx_type)) return set_err(SVT_HIP_ERR_INVALID, "tx_size %d / tx_type %d not defined by the reference", tx_size, tx_type);
    if (d_recon == d_src || (!d_xy && d_recon == d_pred)) return set_err(SVT_HIP_ERR_INVALID, "d_recon must not alias d_src (or, for dense batches, d_pred)");
    if ((is_16bit && bd != 10) || (!is_16bit && bd != 8)) return set_err(SVT_HIP_ERR_INVALID, "bit depth %d (%d-bit samples)", bd, is_16bit ? 16 : 8);
    if (is_16bit && d_sad) return set_err(SVT_HIP_ERR_INVALID, "SAD is defined for 8-bit planes only (the reference searches on the 8-bit MSB plane)");
    if (nblocks > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "nblocks too large");
    hipStream_t s = (hipStream_t)stream;
    if (tx_size == SVT_TX_32X32 && (tx_type == SVT_DCT_DCT || tx_type == SVT_IDTX)) {
        const QParams qp = make_qparams(zbin, round, quant, quant_shift, dequant, 1);
        bool ok = qp.fast_ok;
        for (int i = 0; i < 2; i++) ok = ok && qp.quant_shift[i] >= 0 && qp.dequant[i] >= 0 && qp.round[i] >= 0;
        if (ok && ((d_coeff != nullptr) == (d_dqcoeff != nullptr))) {
            const uint32_t grid = (uint32_t)((nblocks + 2 * F32_WAVES - 1) / (2 * F32_WAVES));
#define ENC32(T, B, KEEP, SAD) hipLaunchKernelGGL((enc32_kernel<T, B, KEEP, SAD>), dim3(grid), dim3(F32_WAVES * 64), 0, s, (const T*)d_src_v, \
                                         (const T*)d_pred_v, (T*)d_recon_v, d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_sad, d_iscan, qp,          \
                                         tx_type == SVT_IDTX ? 1 : 0, (uint32_t)nblocks, d_xy, src_stride, pred_stride, recon_stride)
            if (is_16bit) { if (d_coeff) ENC32(uint16_t, 10, true, false); else ENC32(uint16_t, 10, false, false); }
            else if (d_coeff) { if (d_sad) ENC32(uint8_t, 8, true, true); else ENC32(uint8_t, 8, true, false); }
            else { if (d_sad) ENC32(uint8_t, 8, false, true); else ENC32(uint8_t, 8, false, false); }
#undef ENC32
            return launch_status("encode_recon_32x32");
        }
    }
    if (tx_size == SVT_TX_4X4 && ((d_coeff != nullptr) == (d_dqcoeff != nullptr)) && !g_tune_no_enc_staged) {
        // one lane per block, everything in registers (enc4_kernel); any quantiser table
        const QParams qp = make_qparams(zbin, round, quant, quant_shift, dequant, 0);
        bool fast = qp.fast_ok;
        for (int i = 0; i < 2; i++) fast = fast && qp.quant_shift[i] >= 0 && qp.dequant[i] >= 0 && qp.round[i] >= 0;
        const dim3 grid((uint32_t)((nblocks + 255) / 256));
#define ENC4(T, B, KEEP) hipLaunchKernelGGL((enc4_kernel<T, B, KEEP>), grid, dim3(256), 0, s, (const T*)d_src_v, (const T*)d_pred_v, (T*)d_recon_v, \
                                           d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_sad, d_iscan, qp, fast ? 1 : 0, tx_type, (uint32_t)nblocks, d_xy,         \
                                           src_stride, pred_stride, recon_stride)
        if (is_16bit) { if (d_coeff) ENC4(uint16_t, 10, true); else ENC4(uint16_t, 10, false); }
        else { if (d_coeff) ENC4(uint8_t, 8, true); else ENC4(uint8_t, 8, false); }
#undef ENC4
        return launch_status("encode_recon_4x4");
    }
    {   // every other size: the staged fused kernel (dense 8-bit batches, power-of-two quant_shift tables)
        const int pels = kTxW[tx_size] * kTxH[tx_size];
        const QParams qp = make_qparams(zbin, round, quant, quant_shift, dequant, pels > 1024 ? 2 : (pels > 256 ? 1 : 0));
        bool ok = qp.fast_ok && pels > 16 && !g_tune_no_enc_staged && ((d_coeff != nullptr) == (d_dqcoeff != nullptr));
        for (int i = 0; i < 2; i++) ok = ok && qp.quant_shift[i] >= 0 && qp.dequant[i] >= 0 && qp.round[i] >= 0;
        ok = ok && (((uintptr_t)d_qcoeff | (uintptr_t)d_coeff | (uintptr_t)d_dqcoeff) & 15) == 0;
        ok = ok && (d_xy || (((uintptr_t)d_src | (uintptr_t)d_pred | (uintptr_t)d_recon) & 15) == 0);
        if (ok && tx_size == SVT_TX_64X64 && !g_tune_no_enc64) {
            // two blocks per wave, pruned 64-point networks (kernel_enc64.h)
            const dim3 grid((uint32_t)((nblocks + 2 * E64_WAVES - 1) / (2 * E64_WAVES)));
#define ENC64(T, B, KEEP, SAD) hipLaunchKernelGGL((enc64_kernel<T, B, KEEP, SAD>), grid, dim3(E64_WAVES * 64), 0, s, (const T*)d_src_v, (const T*)d_pred_v, (T*)d_recon_v, \
                                            d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_sad, d_iscan, qp, (uint32_t)nblocks, d_xy, src_stride, pred_stride, recon_stride)
            if (is_16bit) { if (d_coeff) ENC64(uint16_t, 10, true, true); else if (d_sad) ENC64(uint16_t, 10, false, true); else ENC64(uint16_t, 10, false, false); }
            else { if (d_coeff) ENC64(uint8_t, 8, true, true); else if (d_sad) ENC64(uint8_t, 8, false, true); else ENC64(uint8_t, 8, false, false); }
#undef ENC64
            return launch_status("encode_recon_64x64");
        }
        if (ok) {
#define ENCS(W, H) launch_enc_staged<W, H>(d_src_v, d_pred_v, d_recon_v, is_16bit, d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_sad, d_iscan, qp, tx_type, nblocks, d_xy, src_stride, pred_stride, recon_stride, s)
            TX_SWITCH(tx_size, ENCS)
#undef ENCS
        }
    }
    // composed path: the two batched stages around a device copy of the prediction
    if (!d_coeff || !d_dqcoeff) return set_err(SVT_HIP_ERR_INVALID, "this size/type/quantizer needs d_coeff and d_dqcoeff");
    if (d_xy || is_16bit) return set_err(SVT_HIP_ERR_INVALID, "no fused kernel for this case (4x4, non-power-of-two quant_shift or misaligned coefficient buffers); use svt_hip_fwd_quant_planes_batch + svt_hip_inv_txfm2d_add_batch");
    if (int rc = svt_hip_fwd_quant_sad_batch(d_src, d_pred, nblocks, tx_size, tx_type, zbin, round, quant, quant_shift, dequant,
                                             d_iscan, d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_sad, stream)) return rc;
    const size_t pels = (size_t)kTxW[tx_size] * kTxH[tx_size];
    HIP_TRY(hipMemcpyAsync(d_recon, d_pred, pels * nblocks, hipMemcpyDeviceToDevice, s));
    return svt_hip_inv_txfm2d_add_batch(d_dqcoeff, d_recon, 0, kTxW[tx_size], pels, nullptr, nblocks, tx_size, tx_type, 8, stream);
}

extern "C" int svt_hip_encode_recon_batch(const uint8_t* d_src, const uint8_t* d_pred, size_t nblocks, int tx_size,
                                          int tx_type, const int16_t* zbin, const int16_t* round, const int16_t* quant,
                                          const int16_t* quant_shift, const int16_t* dequant, const int16_t* d_iscan,
                                          int32_t* d_coeff, int32_t* d_qcoeff, int32_t* d_dqcoeff, uint16_t* d_eob,
                                          uint32_t* d_sad, uint8_t* d_recon, void* stream) {
    return encode_recon_impl(d_src, 0, d_pred, 0, d_recon, 0, nullptr, 0, 8, nblocks, tx_size, tx_type, zbin, round, quant, quant_shift,
                             dequant, d_iscan, d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_sad, stream);
}
extern "C" int svt_hip_encode_recon_planes_batch(const void* d_src, uint32_t src_stride, const void* d_pred,
                                                 uint32_t pred_stride, void* d_recon, uint32_t recon_stride,
                                                 const uint32_t* d_xy, size_t nblocks, int is_16bit, int bd, int tx_size, int tx_type,
                                                 const int16_t* zbin, const int16_t* round, const int16_t* quant,
                                                 const int16_t* quant_shift, const int16_t* dequant, const int16_t* d_iscan,
                                                 int32_t* d_coeff, int32_t* d_qcoeff, int32_t* d_dqcoeff, uint16_t* d_eob,
                                                 uint32_t* d_sad, void* stream) {
    if (nblocks && !d_xy) { if (int rc = require_init()) return rc; return set_err(SVT_HIP_ERR_INVALID, "NULL origin table"); }
    return encode_recon_impl(d_src, src_stride, d_pred, pred_stride, d_recon, recon_stride, d_xy, is_16bit, bd, nblocks, tx_size, tx_type, zbin,
                             round, quant, quant_shift, dequant, d_iscan, d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_sad, stream);
}

template <int W, int H>
int launch_fq(const void* src, uint32_t ss, const void* pred, uint32_t ps, const uint32_t* xy, size_t n, int is16, int tx_type,
              const QParams& qp, const int16_t* iscan, int32_t* co, int32_t* q, int32_t* dq, uint16_t* eob, uint32_t* sad,
              uint64_t* energy, hipStream_t s) {
    constexpr int BPW = TxGeom<W, H>::BPW;
    const uint32_t per_wg = TX_WAVES * BPW;
    const uint32_t grid = (uint32_t)((n + per_wg - 1) / per_wg);
    const uint32_t sstr = xy ? ss : (uint32_t)W, pstr = xy ? ps : (uint32_t)W;
    if (is16)
        hipLaunchKernelGGL((fwd_quant_generic_kernel<W, H, uint16_t>), dim3(grid), dim3(TX_WAVES * 64), 0, s, (const uint16_t*)src,
                           sstr, (size_t)W * H, (const uint16_t*)pred, pstr, (size_t)W * H, xy, tx_type, qp, iscan, co, q, dq,
                           eob, sad, (unsigned long long*)energy, (uint32_t)n);
    else
        hipLaunchKernelGGL((fwd_quant_generic_kernel<W, H, uint8_t>), dim3(grid), dim3(TX_WAVES * 64), 0, s, (const uint8_t*)src,
                           sstr, (size_t)W * H, (const uint8_t*)pred, pstr, (size_t)W * H, xy, tx_type, qp, iscan, co, q, dq,
                           eob, sad, (unsigned long long*)energy, (uint32_t)n);
    return launch_status("fwd_quant_generic");
}

extern "C" int svt_hip_fwd_quant_planes_batch(const void* d_src, uint32_t src_stride, const void* d_pred,
                                              uint32_t pred_stride, const uint32_t* d_xy, size_t nblocks, int is_16bit,
                                              int bd, int tx_size, int tx_type, const int16_t* zbin, const int16_t* round,
                                              const int16_t* quant, const int16_t* quant_shift, const int16_t* dequant,
                                              const int16_t* d_iscan, int32_t* d_coeff, int32_t* d_qcoeff,
                                              int32_t* d_dqcoeff, uint16_t* d_eob, uint32_t* d_sad, uint64_t* d_energy,
                                              void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_src || !d_pred || !d_coeff || !d_qcoeff || !d_dqcoeff || !d_eob || !d_iscan || !zbin || !round || !quant || !quant_shift || !dequant)
        return set_err(SVT_HIP_ERR_INVALID, "NULL argument");
    if (!txfm_allowed(tx_size, tx_type)) return set_err(SVT_HIP_ERR_INVALID, "tx_size %d / tx_type %d not defined by the reference", tx_size, tx_type);
    if ((is_16bit && bd != 10) || (!is_16bit && bd != 8)) return set_err(SVT_HIP_ERR_INVALID, "bit depth %d (%d-bit planes)", bd, is_16bit ? 16 : 8);
    if (is_16bit && d_sad) return set_err(SVT_HIP_ERR_INVALID, "SAD is defined for 8-bit planes only (the reference searches on the 8-bit MSB plane)");
    if (nblocks > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "nblocks too large");
    const int pels = kTxW[tx_size] * kTxH[tx_size];
    const QParams qp = make_qparams(zbin, round, quant, quant_shift, dequant, pels > 1024 ? 2 : (pels > 256 ? 1 : 0));
    for (int i = 0; i < 2; i++)
        if (qp.quant_shift[i] < 0 || qp.dequant[i] < 0 || qp.round[i] < 0) return set_err(SVT_HIP_ERR_INVALID, "negative quantizer table entry");
    hipStream_t s = (hipStream_t)stream;
    if (tx_size == SVT_TX_32X32 && (tx_type == SVT_DCT_DCT || tx_type == SVT_IDTX) && qp.fast_ok && !d_energy && !g_tune_no_f32p &&
        (d_xy || is_16bit) && ((uintptr_t)d_coeff & 15) == 0 && ((uintptr_t)d_qcoeff & 15) == 0 && ((uintptr_t)d_dqcoeff & 15) == 0) {
        // the tuned 32x32 kernel on planes / 10-bit samples (dense 16-bit batches are "planes" of stride 32 with a NULL table)
        const uint32_t npairs = (uint32_t)((nblocks + 1) / 2);
        const uint32_t grid = (npairs + F32_WAVES - 1) / F32_WAVES;
        const int idtx = tx_type == SVT_IDTX ? 1 : 0;
#define F32P(INM, SAD, PL) hipLaunchKernelGGL((fwd32_kernel<INM, true, SAD, 1, false, 2, PL>), dim3(grid), dim3(F32_WAVES * 64), 0, s, d_src, d_pred, \
                                          d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_sad, d_iscan, qp, idtx, (uint32_t)nblocks, src_stride, pred_stride, d_xy)
        if (is_16bit) { if (d_xy) F32P(2, false, true); else F32P(2, false, false); }
        else { if (d_sad) F32P(1, true, true); else F32P(1, false, true); }
#undef F32P
        return launch_status("fwd_quant_32x32_planes");
    }
    if (tx_size == SVT_TX_64X64 && !g_tune_no_staged && !g_tune_no_enc64 && qp.fast_ok && !d_energy &&
        (d_xy || (((uintptr_t)d_src & 15) == 0 && ((uintptr_t)d_pred & 15) == 0)) &&
        ((uintptr_t)d_coeff & 15) == 0 && ((uintptr_t)d_qcoeff & 15) == 0 && ((uintptr_t)d_dqcoeff & 15) == 0) {
        // two blocks per wave, forward networks pruned to the 32 kept outputs (the forward half of enc64_kernel); callers that
        // want three_quad_energy keep the one-block kernel, which forms the discarded coefficients
        const dim3 grid((uint32_t)((nblocks + 2 * E64_WAVES - 1) / (2 * E64_WAVES)));
        const uint32_t ss = d_xy ? src_stride : 64u, ps = d_xy ? pred_stride : 64u;
        if (is_16bit)
            hipLaunchKernelGGL((fq64_kernel<uint16_t, 10>), grid, dim3(E64_WAVES * 64), 0, s, (const uint16_t*)d_src, (const uint16_t*)d_pred, d_coeff,
                               d_qcoeff, d_dqcoeff, d_eob, d_sad, d_iscan, qp, (uint32_t)nblocks, d_xy, ss, ps);
        else
            hipLaunchKernelGGL((fq64_kernel<uint8_t, 8>), grid, dim3(E64_WAVES * 64), 0, s, (const uint8_t*)d_src, (const uint8_t*)d_pred, d_coeff,
                               d_qcoeff, d_dqcoeff, d_eob, d_sad, d_iscan, qp, (uint32_t)nblocks, d_xy, ss, ps);
        return launch_status("fwd_quant_64x64");
    }
    // staged (coalesced) kernels: dense batches need 16-B aligned inputs, plane-addressed blocks do not
    if (!g_tune_no_staged && qp.fast_ok && pels > 16 && (d_xy || (((uintptr_t)d_src & 15) == 0 && ((uintptr_t)d_pred & 15) == 0)) &&
        ((uintptr_t)d_coeff & 15) == 0 && ((uintptr_t)d_qcoeff & 15) == 0 && ((uintptr_t)d_dqcoeff & 15) == 0) {
#define CALLS(W, H) launch_fq_staged<W, H>(d_src, d_pred, is_16bit, d_xy, src_stride, pred_stride, nblocks, tx_type, qp, d_iscan, d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_sad, d_energy, s)
        TX_SWITCH(tx_size, CALLS)
#undef CALLS
    }
#define CALL(W, H) launch_fq<W, H>(d_src, src_stride, d_pred, pred_stride, d_xy, nblocks, is_16bit, tx_type, qp, d_iscan, d_coeff, d_qcoeff, d_dqcoeff, d_eob, d_sad, d_energy, s)
    TX_SWITCH(tx_size, CALL)
#undef CALL
}

// ---- frame-level fan-out: independent groups on internal streams, forked from and joined into the caller's stream ----
namespace {
}  // namespace

// argument checks of a frame call's groups (also used by svt_hip_encode_recon_frame_ex before it enqueues its first phase)
int svthost::frame_groups_check(const svt_hip_frame_group* groups, int ngroups) {
    if (ngroups == 0) return SVT_HIP_OK;
    if (!groups || ngroups < 0) return set_err(SVT_HIP_ERR_INVALID, "NULL group list");
    if (ngroups > 256) return set_err(SVT_HIP_ERR_INVALID, "more than 256 groups in one call");
    for (int g = 0; g < ngroups; g++) {
        const svt_hip_frame_group& G = groups[g];
        if (G.nblocks == 0) continue;
        if (!G.d_src || !G.d_pred || !G.d_recon || !G.d_xy || !G.d_iscan || !G.d_qcoeff || !G.d_eob)
            return set_err(SVT_HIP_ERR_INVALID, "group %d: NULL member", g);
        if (!txfm_allowed(G.tx_size, G.tx_type)) return set_err(SVT_HIP_ERR_INVALID, "group %d: tx_size %d / tx_type %d", g, G.tx_size, G.tx_type);
        if ((G.d_coeff != nullptr) != (G.d_dqcoeff != nullptr)) return set_err(SVT_HIP_ERR_INVALID, "group %d: d_coeff and d_dqcoeff go together", g);
        if (G.tx_size == SVT_TX_4X4 && g_tune_no_enc_staged && (!G.d_coeff || !G.d_offsets || G.d_recon != G.d_pred || G.recon_stride != G.pred_stride))
            return set_err(SVT_HIP_ERR_INVALID, "group %d: with no_enc_staged set, 4x4 groups take the two-stage path and need d_coeff, d_dqcoeff, d_offsets and in-place reconstruction", g);
    }
    return SVT_HIP_OK;
}

extern "C" int svt_hip_encode_recon_frame(const svt_hip_frame_group* groups, int ngroups, int is_16bit, int bd,
                                          const int16_t* zbin, const int16_t* round, const int16_t* quant,
                                          const int16_t* quant_shift, const int16_t* dequant, void* stream) {
    if (int rc = require_init()) return rc;
    if (ngroups == 0) return SVT_HIP_OK;
    if (int rc = frame_groups_check(groups, ngroups)) return rc;            // validate everything before anything is enqueued
    // ---- one launch per REGISTER CLASS (enc_frame_kernel, kernel_frame.h) when every group is one the fused bodies cover: any of the
    // 19 sizes and their types, qcoeff + recon outputs, power-of-two quant_shift tables.  Measured (tools/bench_frame.py,
    // tools/bench_c5.py): a 1080p picture as 13 per-size launches on 8 streams 0.160 ms, those captured into a graph 0.101, ONE
    // launch at the 64x64 body's register budget (round 2) 0.044; by class see DESIGN 4.17.  A call above 2^27 pixel passes (a GOP
    // of 4K pictures) takes the launches by register class.
    size_t call_pixels = 0;
    for (int g = 0; g < ngroups; g++)
        if (groups[g].nblocks) call_pixels += (size_t)groups[g].nblocks * kTxW[groups[g].tx_size] * kTxH[groups[g].tx_size];      // (validated above)
    // frame_single_launch: 1 one launch (class 3), 2 one launch per register class, 0 per-size launches, -1 (default): one launch up to
    // 2^25 pixel passes per call (two 1080p pictures with five sizes; measured cross-over between 2 and 4, tools/bench_frame.py), class launches above
    // ... and per-size launches on the fan-out streams above 2^28 (a GOP of 4K pictures: every size fills the GPU on its own at its
    // own kernel's occupancy; configs[4], 30 pictures per call, tools/bench_c5.py: 6 162 pictures/s against 5 756 by class and 4 888 in one launch)
    int mode = g_tune_frame_single_launch >= 0 ? g_tune_frame_single_launch
                                               : (call_pixels <= ((size_t)1 << 25) ? 1 : (call_pixels <= ((size_t)1 << 28) ? 2 : 0));
    for (int g = 0; g < ngroups; g++)
        if (mode == 1 && groups[g].nblocks && groups[g].tx_size > SVT_TX_64X64) mode = 2;      // the one-launch kernel holds the square sizes
    if (mode > 0) {
        bool ok = true;
        int per_class[4] = {0, 0, 0, 0};
        for (int g = 0; g < ngroups && ok; g++) {
            const svt_hip_frame_group& G = groups[g];
            if (G.nblocks == 0) continue;
            ok = !G.d_coeff && G.d_recon != G.d_src && (((uintptr_t)G.d_qcoeff) & 15) == 0;      // (type / size validated above)
            per_class[mode == 1 ? 3 : frame_class_of(G.tx_size)]++;
        }
        if ((is_16bit && bd != 10) || (!is_16bit && bd != 8)) ok = false;
        for (int c = 0; c < 4; c++) ok = ok && per_class[c] <= FRAME_MAX_GROUPS;
        FrameDesc fd[4];
        uint32_t total[4] = {0, 0, 0, 0};
        if (ok) {
            memset(fd, 0, sizeof(fd));
            int order[256];
            for (int i = 0; i < ngroups; i++) order[i] = i;
            for (int i = 1; i < ngroups; i++) {          // largest blocks first: the long workgroups start early
                const int v = order[i];
                const int pv = kTxW[groups[v].tx_size] * kTxH[groups[v].tx_size];
                int j = i - 1;
                while (j >= 0 && kTxW[groups[order[j]].tx_size] * kTxH[groups[order[j]].tx_size] < pv) { order[j + 1] = order[j]; j--; }
                order[j + 1] = v;
            }
            for (int k = 0; k < ngroups && ok; k++) {
                const svt_hip_frame_group& G = groups[order[k]];
                if (G.nblocks == 0) continue;
                const int pels = kTxW[G.tx_size] * kTxH[G.tx_size], c = mode == 1 ? 3 : frame_class_of(G.tx_size);
                FrameGroupDev& D = fd[c].g[fd[c].ngroups];
                D.qp = make_qparams(zbin, round, quant, quant_shift, dequant, pels > 1024 ? 2 : (pels > 256 ? 1 : 0));
                for (int i = 0; i < 2; i++) ok = ok && D.qp.quant_shift[i] >= 0 && D.qp.dequant[i] >= 0 && D.qp.round[i] >= 0;
                ok = ok && D.qp.fast_ok;
                D.src = G.d_src; D.pred = G.d_pred; D.recon = G.d_recon; D.qcoeff = G.d_qcoeff; D.eob = G.d_eob; D.xy = G.d_xy; D.iscan = G.d_iscan;
                D.src_stride = G.src_stride; D.pred_stride = G.pred_stride; D.recon_stride = G.recon_stride; D.nblocks = G.nblocks; D.tx_size = G.tx_size; D.tx_type = G.tx_type;
                const uint32_t per_wg = frame_blocks_per_wg(G.tx_size);
                total[c] += (G.nblocks + per_wg - 1) / per_wg;
                D.wg_end = total[c];
                fd[c].ngroups++;
            }
        }
        if (ok) {
            // (class launches: the 64x64 class first, the small sizes last)
            hipStream_t hs = (hipStream_t)stream;
#define FRAME_LAUNCH(C)                                                                                                                         \
            if (fd[C].ngroups) {                                                                                                                \
                if (is_16bit) hipLaunchKernelGGL((enc_frame_kernel<uint16_t, 10, C>), dim3(total[C]), dim3(256), 0, hs, fd[C]);                 \
                else hipLaunchKernelGGL((enc_frame_kernel<uint8_t, 8, C>), dim3(total[C]), dim3(256), 0, hs, fd[C]);                            \
                if (int rc = launch_status("enc_frame")) return rc;                                                                             \
            }
            if (fd[3].ngroups) return launch_enc_frame_one(&fd[3], total[3], is_16bit, hs);      // (svt_hip_frame.hip)
            FRAME_LAUNCH(2) FRAME_LAUNCH(1) FRAME_LAUNCH(0)
#undef FRAME_LAUNCH
            return SVT_HIP_OK;
        }
    }
    if (int rc = t_fan.ensure()) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int nstreams = ngroups < kFanStreams ? ngroups : kFanStreams;
    // From the fork on nothing returns early: a side stream that has waited on the fork event must be joined back, or a
    // caller capturing this call into a graph is left with dangling branches.  Failures become rc and fall through to the join.
    HIP_TRY(hipEventRecord(t_fan.fork, s));
    int rc = SVT_HIP_OK, forked = 0;
    for (; forked < nstreams; forked++)
        if (hipStreamWaitEvent(t_fan.s[forked], t_fan.fork, 0) != hipSuccess) { rc = set_err(SVT_HIP_ERR_RUNTIME, "stream fork failed"); break; }
    // largest groups first, round-robin: the long kernels start early and the small ones fill in beside them
    int order[256];
    const int ng = ngroups;                        // <= 256, checked with the arguments
    for (int i = 0; i < ng; i++) order[i] = i;
    for (int i = 1; i < ng; i++) {                 // insertion sort by work (pixels), descending
        const int v = order[i];
        const size_t wv = (size_t)groups[v].nblocks * kTxW[groups[v].tx_size] * kTxH[groups[v].tx_size];
        int j = i - 1;
        while (j >= 0 && (size_t)groups[order[j]].nblocks * kTxW[groups[order[j]].tx_size] * kTxH[groups[order[j]].tx_size] < wv) { order[j + 1] = order[j]; j--; }
        order[j + 1] = v;
    }
    for (int k = 0; k < ng && rc == SVT_HIP_OK; k++) {
        const svt_hip_frame_group& G = groups[order[k]];
        if (G.nblocks == 0) continue;
        hipStream_t gs = t_fan.s[k % nstreams];
        if (G.tx_size == SVT_TX_4X4 && g_tune_no_enc_staged) {
            rc = svt_hip_fwd_quant_planes_batch(G.d_src, G.src_stride, G.d_pred, G.pred_stride, G.d_xy, G.nblocks, is_16bit, bd, G.tx_size, G.tx_type,
                                                zbin, round, quant, quant_shift, dequant, G.d_iscan, G.d_coeff, G.d_qcoeff, G.d_dqcoeff, G.d_eob,
                                                nullptr, nullptr, gs);
            if (rc == SVT_HIP_OK)
                rc = svt_hip_inv_txfm2d_add_batch(G.d_dqcoeff, G.d_recon, is_16bit, (int32_t)G.recon_stride, 0, G.d_offsets, G.nblocks, G.tx_size,
                                                  G.tx_type, bd, gs);
        } else {
            rc = encode_recon_impl(G.d_src, G.src_stride, G.d_pred, G.pred_stride, G.d_recon, G.recon_stride, G.d_xy, is_16bit, bd, G.nblocks,
                                   G.tx_size, G.tx_type, zbin, round, quant, quant_shift, dequant, G.d_iscan, G.d_coeff, G.d_qcoeff, G.d_dqcoeff,
                                   G.d_eob, nullptr, gs);
        }
    }
    // always join, also after an error: the caller's stream (or capture) must not be left with dangling branches
    for (int i = 0; i < forked; i++) {
        if (hipEventRecord(t_fan.join[i], t_fan.s[i]) != hipSuccess || hipStreamWaitEvent(s, t_fan.join[i], 0) != hipSuccess)
            if (rc == SVT_HIP_OK) rc = set_err(SVT_HIP_ERR_RUNTIME, "stream join failed");
    }
    return rc;
}

extern "C" int svt_hip_fwd_quant_batch(const int16_t* d_residual, size_t nblocks, int tx_size, int tx_type, int bd,
                                       const int16_t* zbin, const int16_t* round, const int16_t* quant,
                                       const int16_t* quant_shift, const int16_t* dequant, const int16_t* d_iscan,
                                       int32_t* d_coeff, int32_t* d_qcoeff, int32_t* d_dqcoeff, uint16_t* d_eob,
                                       void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_residual || !d_coeff || !d_qcoeff || !d_dqcoeff || !d_eob || !d_iscan || !zbin || !round || !quant || !quant_shift || !dequant)
        return set_err(SVT_HIP_ERR_INVALID, "NULL argument");
    if (!txfm_allowed(tx_size, tx_type)) return set_err(SVT_HIP_ERR_INVALID, "tx_size %d / tx_type %d not defined by the reference", tx_size, tx_type);
    const int w = kTxW[tx_size], h = kTxH[tx_size];
    const int pels = w * h, ls = pels > 1024 ? 2 : (pels > 256 ? 1 : 0);
    hipStream_t s = (hipStream_t)stream;
    const QParams qp = make_qparams(zbin, round, quant, quant_shift, dequant, ls);
    if (tx_size == SVT_TX_32X32 && bd == 8 && qp.fast_ok && ((uintptr_t)d_residual & 15) == 0 &&
        (tx_type == SVT_DCT_DCT || tx_type == SVT_IDTX)) {
        // 24-bit quantiser arithmetic needs 8-bit-range residuals (|coeff| < 2^17)
        const uint32_t npairs = (uint32_t)((nblocks + 1) / 2);
        hipLaunchKernelGGL((fwd32_kernel<0, true, false>), dim3((npairs + F32_WAVES - 1) / F32_WAVES), dim3(F32_WAVES * 64), 0, s,
                           (const void*)d_residual, (const uint8_t*)nullptr, d_coeff, d_qcoeff, d_dqcoeff, d_eob,
                           (uint32_t*)nullptr, d_iscan, qp, tx_type == SVT_IDTX ? 1 : 0, (uint32_t)nblocks);
        return launch_status("fwd32_quant");
    }
    // general sizes: transform into d_coeff, pack 64-pt outputs, quantise (two more passes over d_coeff)
    if (w == 64 || h == 64) return set_err(SVT_HIP_ERR_UNSUPPORTED, "use svt_hip_fwd_quant_planes_batch for 64-pt sizes (packed coefficient layout)");
    if (int rc = svt_hip_fwd_txfm2d_batch(d_residual, (uint32_t)w, (size_t)w * h, d_coeff, nblocks, tx_size, tx_type, bd, stream)) return rc;
    return svt_hip_quantize_b_batch(d_coeff, (size_t)w * h, 0, zbin, round, quant, quant_shift, d_qcoeff, d_dqcoeff, dequant, d_eob,
                                    d_iscan, ls, nblocks, stream);
}

// ===========================================================================
static void dropin_fwd(int tx_size, int16_t* input, int32_t* output, uint32_t stride, uint8_t tx_type, uint8_t bd,
                       const char* fn) {
    const int w = kTxW[tx_size], h = kTxH[tx_size];
    const size_t in_b = align256((size_t)w * h * 2), out_b = (size_t)w * h * 4;
    DROPIN_TRY(t_ctx.ensure(in_b + out_b), fn);
    int16_t* d_in = (int16_t*)t_ctx.dbuf;
    int32_t* d_out = (int32_t*)(t_ctx.dbuf + in_b);
    HIP_DIE(hipMemcpy2DAsync(d_in, (size_t)w * 2, input, (size_t)stride * 2, (size_t)w * 2, h, hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_fwd_txfm2d_batch(d_in, (uint32_t)w, (size_t)w * h, d_out, 1, tx_size, tx_type, bd, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(output, d_out, out_b, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
#define DEF_FWD(W, H, TS)                                                                                          \
    extern "C" void svt_hip_av1_fwd_txfm2d_##W##x##H(int16_t* input, int32_t* output, uint32_t input_stride,      \
                                                     svt_tx_type_t transform_type, uint8_t bit_depth) {           \
        dropin_fwd(TS, input, output, input_stride, transform_type, bit_depth, "svt_hip_av1_fwd_txfm2d_" #W "x" #H); \
    }
DEF_FWD(4, 4, SVT_TX_4X4) DEF_FWD(8, 8, SVT_TX_8X8) DEF_FWD(16, 16, SVT_TX_16X16) DEF_FWD(32, 32, SVT_TX_32X32)
DEF_FWD(64, 64, SVT_TX_64X64) DEF_FWD(4, 8, SVT_TX_4X8) DEF_FWD(8, 4, SVT_TX_8X4) DEF_FWD(8, 16, SVT_TX_8X16)
DEF_FWD(16, 8, SVT_TX_16X8) DEF_FWD(16, 32, SVT_TX_16X32) DEF_FWD(32, 16, SVT_TX_32X16) DEF_FWD(32, 64, SVT_TX_32X64)
DEF_FWD(64, 32, SVT_TX_64X32) DEF_FWD(4, 16, SVT_TX_4X16) DEF_FWD(16, 4, SVT_TX_16X4) DEF_FWD(8, 32, SVT_TX_8X32)
DEF_FWD(32, 8, SVT_TX_32X8) DEF_FWD(16, 64, SVT_TX_16X64) DEF_FWD(64, 16, SVT_TX_64X16)
#undef DEF_FWD

static void dropin_inv(int tx_size, const int32_t* input, void* output, int is16, int32_t stride, uint8_t tx_type,
                       int32_t bd, const char* fn) {
    const int w = kTxW[tx_size], h = kTxH[tx_size];
    const int kw = w > 32 ? 32 : w, kh = h > 32 ? 32 : h;
    const size_t es = is16 ? 2 : 1;
    const size_t in_b = align256((size_t)kw * kh * 4), px_b = (size_t)w * h * es;
    DROPIN_TRY(t_ctx.ensure(in_b + px_b), fn);
    int32_t* d_in = (int32_t*)t_ctx.dbuf;
    char* d_px = t_ctx.dbuf + in_b;
    HIP_DIE(hipMemcpyAsync(d_in, input, (size_t)kw * kh * 4, hipMemcpyHostToDevice, t_ctx.stream), fn);
    HIP_DIE(hipMemcpy2DAsync(d_px, (size_t)w * es, output, (size_t)stride * es, (size_t)w * es, h, hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_inv_txfm2d_add_batch(d_in, d_px, is16, w, (size_t)w * h, nullptr, 1, tx_size, tx_type, bd, t_ctx.stream), fn);
    HIP_DIE(hipMemcpy2DAsync(output, (size_t)stride * es, d_px, (size_t)w * es, (size_t)w * es, h, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
#define DEF_INV_SQ(W, H, TS)                                                                                       \
    extern "C" void svt_hip_av1_inv_txfm2d_add_##W##x##H(const int32_t* input, uint16_t* output, int32_t stride,  \
                                                         svt_tx_type_t tx_type, int32_t bd) {                     \
        dropin_inv(TS, input, output, 1, stride, tx_type, bd, "svt_hip_av1_inv_txfm2d_add_" #W "x" #H);           \
    }
#define DEF_INV_R1(W, H, TS)                                                                                       \
    extern "C" void svt_hip_av1_inv_txfm2d_add_##W##x##H(const int32_t* input, uint16_t* output, int32_t stride,  \
                                                         svt_tx_type_t tx_type, svt_tx_size_t, int32_t, int32_t bd) { \
        dropin_inv(TS, input, output, 1, stride, tx_type, bd, "svt_hip_av1_inv_txfm2d_add_" #W "x" #H);           \
    }
#define DEF_INV_R2(W, H, TS)                                                                                       \
    extern "C" void svt_hip_av1_inv_txfm2d_add_##W##x##H(const int32_t* input, uint16_t* output, int32_t stride,  \
                                                         svt_tx_type_t tx_type, svt_tx_size_t, int32_t bd) {      \
        dropin_inv(TS, input, output, 1, stride, tx_type, bd, "svt_hip_av1_inv_txfm2d_add_" #W "x" #H);           \
    }
DEF_INV_SQ(4, 4, SVT_TX_4X4) DEF_INV_SQ(8, 8, SVT_TX_8X8) DEF_INV_SQ(16, 16, SVT_TX_16X16)
DEF_INV_SQ(32, 32, SVT_TX_32X32) DEF_INV_SQ(64, 64, SVT_TX_64X64)
DEF_INV_R1(8, 16, SVT_TX_8X16) DEF_INV_R1(16, 8, SVT_TX_16X8) DEF_INV_R1(16, 32, SVT_TX_16X32)
DEF_INV_R1(32, 16, SVT_TX_32X16) DEF_INV_R1(32, 64, SVT_TX_32X64) DEF_INV_R1(64, 32, SVT_TX_64X32)
DEF_INV_R1(8, 32, SVT_TX_8X32) DEF_INV_R1(32, 8, SVT_TX_32X8) DEF_INV_R1(16, 64, SVT_TX_16X64)
DEF_INV_R1(64, 16, SVT_TX_64X16)
DEF_INV_R2(4, 8, SVT_TX_4X8) DEF_INV_R2(8, 4, SVT_TX_8X4) DEF_INV_R2(4, 16, SVT_TX_4X16) DEF_INV_R2(16, 4, SVT_TX_16X4)
#undef DEF_INV_SQ
#undef DEF_INV_R1
#undef DEF_INV_R2

extern "C" void svt_hip_av1_inv_txfm_add(const svt_tran_low_t* dqcoeff, uint8_t* dst, int32_t stride,
                                         const svt_txfm_param* p) {
    dropin_inv(p->tx_size, dqcoeff, dst, 0, stride, p->tx_type, 8, "svt_hip_av1_inv_txfm_add");
}

static void dropin_quant(int log_scale, const int32_t* coeff, intptr_t n, int32_t skip, const int16_t* zbin,
                         const int16_t* round, const int16_t* quant, const int16_t* qshift, int32_t* q, int32_t* dq,
                         const int16_t* dequant, uint16_t* eob, const int16_t* iscan, const char* fn) {
    const size_t cb = align256((size_t)n * 4), ib = align256((size_t)n * 2);
    DROPIN_TRY(t_ctx.ensure(3 * cb + ib + 256), fn);
    int32_t* d_c = (int32_t*)t_ctx.dbuf;
    int32_t* d_q = (int32_t*)(t_ctx.dbuf + cb);
    int32_t* d_dq = (int32_t*)(t_ctx.dbuf + 2 * cb);
    int16_t* d_is = (int16_t*)(t_ctx.dbuf + 3 * cb);
    uint16_t* d_eob = (uint16_t*)(t_ctx.dbuf + 3 * cb + ib);
    HIP_DIE(hipMemcpyAsync(d_c, coeff, (size_t)n * 4, hipMemcpyHostToDevice, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(d_is, iscan, (size_t)n * 2, hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_quantize_b_batch(d_c, (size_t)n, skip, zbin, round, quant, qshift, d_q, d_dq, dequant, d_eob, d_is, log_scale, 1, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(q, d_q, (size_t)n * 4, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(dq, d_dq, (size_t)n * 4, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(eob, d_eob, 2, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
#define DEF_QUANT(name, LS)                                                                                        \
    extern "C" void name(const svt_tran_low_t* coeff_ptr, intptr_t n_coeffs, int32_t skip_block,                  \
                         const int16_t* zbin_ptr, const int16_t* round_ptr, const int16_t* quant_ptr,             \
                         const int16_t* quant_shift_ptr, svt_tran_low_t* qcoeff_ptr, svt_tran_low_t* dqcoeff_ptr, \
                         const int16_t* dequant_ptr, uint16_t* eob_ptr, const int16_t* scan, const int16_t* iscan) { \
        (void)scan;                                                                                                \
        dropin_quant(LS, coeff_ptr, n_coeffs, skip_block, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr,         \
                     qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, iscan, #name);                                 \
    }
DEF_QUANT(svt_hip_aom_highbd_quantize_b, 0)
DEF_QUANT(svt_hip_aom_highbd_quantize_b_32x32, 1)
DEF_QUANT(svt_hip_aom_highbd_quantize_b_64x64, 2)
DEF_QUANT(svt_hip_aom_quantize_b, 0)
DEF_QUANT(svt_hip_aom_quantize_b_32x32, 1)
DEF_QUANT(svt_hip_aom_quantize_b_64x64, 2)
#undef DEF_QUANT

