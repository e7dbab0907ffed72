// svt_hip_dsp.hip — C-ABI shim (include/svt_hip_dsp.h) over the gfx950 kernels.
// Host logic only: argument checks, launch geometry, per-thread staging for the
// drop-in (one block per call) entry points.  NO CPU compute path exists here:
// if HIP is unusable the batched API returns an error and the drop-ins abort.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "../../include/svt_hip_dsp.h"
#include "kernel_cfl.h"
#include "kernel_fused32.h"
#include "kernel_intra.h"
#include "kernel_me.h"
#include "kernel_ois.h"
#include "kernel_pixel.h"
#include "kernel_txfm.h"
#include "kernel_txfm_staged.h"

using namespace svtdev;

namespace {

thread_local char g_err[512] = "";
std::atomic<int> g_inited{0};
std::mutex g_init_mu;
int g_device = -1;
char g_devname[300] = "";
int g_num_cu = 256;
// tuning knobs (svt_hip_tune): fused 32x32 kernel occupancy / grid
int g_tune_f32_min_waves = 1;
int g_tune_f32_wg_per_cu = 0;
int g_tune_f32_nt = 0;
int g_tune_f32_qmode1 = 0;
int g_tune_no_staged = 0;
int g_tune_no_qsad = 0;
int g_tune_no_q2 = 0;
int g_tune_no_q16 = 0;
int g_tune_q2_su4 = 0;
int g_tune_ois_no_fold = 0;
int g_tune_no_me16 = 0;
int g_tune_me_exact = 0;
int g_tune_no_f32p = 0;
int g_tune_no_inv_planes = 0;
int g_tune_no_enc_staged = 0;
int g_tune_inv32_waves = 4;
int g_tune_inv32_var = 0;

int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return set_err(SVT_HIP_ERR_RUNTIME, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                                  \
    } while (0)

// The HIP "current device" is a per-THREAD setting and the encoder calls from many pthreads (Threading):
// every entry point passes through here, so every thread is bound to the library's device once.
thread_local int t_device_bound = -1;
int require_init() {
    if (!g_inited.load(std::memory_order_acquire)) {
        if (int rc = svt_hip_init(0)) return rc;
    }
    if (t_device_bound != g_device) {
        HIP_TRY(hipSetDevice(g_device));
        t_device_bound = g_device;
    }
    return SVT_HIP_OK;
}
int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(SVT_HIP_ERR_RUNTIME, "launch %s: %s", what, hipGetErrorString(e));
    return SVT_HIP_OK;
}

const int kTxW[SVT_TX_SIZES_ALL] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64};
const int kTxH[SVT_TX_SIZES_ALL] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};

// is_txfm_allowed (test/TxfmCommon.h:172-181; av1_estimate_transform's switch)
bool txfm_allowed(int tx_size, int tx_type) {
    if (tx_size < 0 || tx_size >= SVT_TX_SIZES_ALL || tx_type < 0 || tx_type >= SVT_TX_TYPES) return false;
    const int m = kTxW[tx_size] > kTxH[tx_size] ? kTxW[tx_size] : kTxH[tx_size];
    if (m == 64) return tx_type == SVT_DCT_DCT;
    if (m == 32) return tx_type == SVT_DCT_DCT || tx_type == SVT_IDTX;
    return true;
}

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
        if (k < 0 || sh < 1 || sh > 31 || dequant[i] < 0 || qp.round[i] < 0) qp.fast_ok = 0;
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

#define TX_SWITCH(tx_size, CALL)                                                                  \
    switch (tx_size) {                                                                            \
    case SVT_TX_4X4: return CALL(4, 4); case SVT_TX_8X8: return CALL(8, 8);                        \
    case SVT_TX_16X16: return CALL(16, 16); case SVT_TX_32X32: return CALL(32, 32);                \
    case SVT_TX_64X64: return CALL(64, 64); case SVT_TX_4X8: return CALL(4, 8);                    \
    case SVT_TX_8X4: return CALL(8, 4); case SVT_TX_8X16: return CALL(8, 16);                      \
    case SVT_TX_16X8: return CALL(16, 8); case SVT_TX_16X32: return CALL(16, 32);                  \
    case SVT_TX_32X16: return CALL(32, 16); case SVT_TX_32X64: return CALL(32, 64);                \
    case SVT_TX_64X32: return CALL(64, 32); case SVT_TX_4X16: return CALL(4, 16);                  \
    case SVT_TX_16X4: return CALL(16, 4); case SVT_TX_8X32: return CALL(8, 32);                    \
    case SVT_TX_32X8: return CALL(32, 8); case SVT_TX_16X64: return CALL(16, 64);                  \
    case SVT_TX_64X16: return CALL(64, 16);                                                       \
    default: return set_err(SVT_HIP_ERR_INVALID, "bad tx_size %d", tx_size);                      \
    }

// ---------------------------------------------------------------------------
// per-thread context for the drop-in entry points
// ---------------------------------------------------------------------------
struct ThreadCtx {
    hipStream_t stream = nullptr;
    char* dbuf = nullptr;
    size_t cap = 0;
    ~ThreadCtx() {
        if (dbuf) (void)hipFree(dbuf);
        if (stream) (void)hipStreamDestroy(stream);
    }
    int ensure(size_t bytes) {
        if (require_init() != SVT_HIP_OK) return SVT_HIP_ERR_NO_DEVICE;
        if (!stream) {
            HIP_TRY(hipSetDevice(g_device));
            HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        }
        if (bytes > cap) {
            if (dbuf) HIP_TRY(hipFree(dbuf));
            dbuf = nullptr;
            cap = 0;
            const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
            HIP_TRY(hipMalloc((void**)&dbuf, want));
            cap = want;
        }
        return SVT_HIP_OK;
    }
};
thread_local ThreadCtx t_ctx;

[[noreturn]] void die(const char* fn) {
    fprintf(stderr, "libsvt_hip_dsp: %s: %s — no CPU fallback exists; aborting\n", fn, g_err);
    abort();
}
#define DROPIN_TRY(expr, fn) do { if ((expr) != SVT_HIP_OK) die(fn); } while (0)
#define HIP_DIE(expr, fn)                                                          \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) { set_err(SVT_HIP_ERR_RUNTIME, "%s: %s", #expr, hipGetErrorString(e_)); die(fn); } \
    } while (0)

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// ===========================================================================
// init / misc
// ===========================================================================
extern "C" int svt_hip_init(int device) {
    std::lock_guard<std::mutex> lk(g_init_mu);
    if (g_inited.load(std::memory_order_acquire)) {
        if (device != g_device)
            return set_err(SVT_HIP_ERR_INVALID, "already initialised on device %d (asked for %d): one device per process", g_device, device);
        return SVT_HIP_OK;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return set_err(SVT_HIP_ERR_NO_DEVICE, "no HIP device (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return set_err(SVT_HIP_ERR_INVALID, "device %d out of range (%d)", device, count);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(g_devname, sizeof(g_devname), "%s %s (%d CUs)", prop.gcnArchName, prop.name, prop.multiProcessorCount);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(SVT_HIP_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                       prop.gcnArchName);
    g_num_cu = prop.multiProcessorCount;
    g_device = device;
    g_inited.store(1, std::memory_order_release);
    return SVT_HIP_OK;
}
extern "C" void svt_hip_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_init_mu);
    g_inited.store(0, std::memory_order_release);
}
extern "C" const char* svt_hip_last_error(void) { return g_err; }
extern "C" int svt_hip_tune(const char* key, int value) {
    if (!key) return SVT_HIP_ERR_INVALID;
    if (!strcmp(key, "f32_min_waves")) { g_tune_f32_min_waves = value; return SVT_HIP_OK; }
    if (!strcmp(key, "f32_wg_per_cu")) { g_tune_f32_wg_per_cu = value; return SVT_HIP_OK; }
    if (!strcmp(key, "f32_nt")) { g_tune_f32_nt = value; return SVT_HIP_OK; }
    if (!strcmp(key, "f32_qmode1")) { g_tune_f32_qmode1 = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_staged")) { g_tune_no_staged = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_qsad")) { g_tune_no_qsad = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_q2")) { g_tune_no_q2 = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_q16")) { g_tune_no_q16 = value; return SVT_HIP_OK; }
    if (!strcmp(key, "q2_su4")) { g_tune_q2_su4 = value; return SVT_HIP_OK; }
    if (!strcmp(key, "ois_no_fold")) { g_tune_ois_no_fold = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_me16")) { g_tune_no_me16 = value; return SVT_HIP_OK; }
    if (!strcmp(key, "me_exact")) { g_tune_me_exact = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_f32p")) { g_tune_no_f32p = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_inv_planes")) { g_tune_no_inv_planes = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_enc_staged")) { g_tune_no_enc_staged = value; return SVT_HIP_OK; }
    if (!strcmp(key, "inv32_waves")) { g_tune_inv32_waves = value; return SVT_HIP_OK; }
    if (!strcmp(key, "inv32_var")) { g_tune_inv32_var = value; return SVT_HIP_OK; }
    return set_err(SVT_HIP_ERR_INVALID, "unknown tuning key %s", key);
}
extern "C" const char* svt_hip_device_name(void) { return g_devname; }

extern "C" void* svt_hip_malloc(size_t bytes) {
    if (require_init() != SVT_HIP_OK) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { set_err(SVT_HIP_ERR_RUNTIME, "hipMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
extern "C" void svt_hip_free(void* p) { if (p) (void)hipFree(p); }
extern "C" int svt_hip_memcpy_h2d(void* d, const void* h, size_t n, void* s) {
    HIP_TRY(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)s));
    return SVT_HIP_OK;
}
extern "C" int svt_hip_memcpy_d2h(void* h, const void* d, size_t n, void* s) {
    HIP_TRY(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)s));
    return SVT_HIP_OK;
}
extern "C" int svt_hip_stream_sync(void* s) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)s));
    return SVT_HIP_OK;
}

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
            INVV(4, 1) INVV(4, 2) INVV(4, 4) INVV(4, 5) INVV(2, 0)
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
    } else if (g_tune_f32_min_waves == 4) {
        if (d_sad) F32_LAUNCH_Q(true, 4, false, 2); else F32_LAUNCH_Q(false, 4, false, 2);
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
    {   // every other size: the staged fused kernel (dense 8-bit batches, power-of-two quant_shift tables)
        const int pels = kTxW[tx_size] * kTxH[tx_size];
        const QParams qp = make_qparams(zbin, round, quant, quant_shift, dequant, pels > 1024 ? 2 : (pels > 256 ? 1 : 0));
        bool ok = qp.fast_ok && pels > 16 && !g_tune_no_enc_staged && ((d_coeff != nullptr) == (d_dqcoeff != nullptr));
        for (int i = 0; i < 2; i++) ok = ok && qp.quant_shift[i] >= 0 && qp.dequant[i] >= 0 && qp.round[i] >= 0;
        ok = ok && (((uintptr_t)d_qcoeff | (uintptr_t)d_coeff | (uintptr_t)d_dqcoeff) & 15) == 0;
        ok = ok && (d_xy || (((uintptr_t)d_src | (uintptr_t)d_pred | (uintptr_t)d_recon) & 15) == 0);
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

static int sad_sse_common(bool sse, const uint8_t* a, uint32_t as, size_t ap, const uint8_t* b, uint32_t bs,
                          size_t bp, uint32_t w, uint32_t h, void* out, size_t n, void* stream) {
    if (int rc = require_init()) return rc;
    if (n == 0) return SVT_HIP_OK;
    if (!a || !b || !out) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (w == 0 || h == 0 || w > 128 || h > 128) return set_err(SVT_HIP_ERR_INVALID, "block %ux%u", w, h);
    if (n == 0) return SVT_HIP_OK;
    const uint32_t grid = (uint32_t)((n + 15) / 16);
    if (sse)
        hipLaunchKernelGGL((sad_sse_kernel<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, as, ap, b, bs, bp, w, h, out, (uint32_t)n);
    else
        hipLaunchKernelGGL((sad_sse_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, as, ap, b, bs, bp, w, h, out, (uint32_t)n);
    return launch_status(sse ? "sse" : "sad");
}
extern "C" int svt_hip_sad_batch(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch,
                                 const uint8_t* d_ref, uint32_t ref_stride, size_t ref_block_pitch, uint32_t width,
                                 uint32_t height, uint32_t* d_out, size_t nblocks, void* stream) {
    return sad_sse_common(false, d_src, src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, width, height, d_out, nblocks, stream);
}
extern "C" int svt_hip_sse_batch(const uint8_t* d_a, uint32_t a_stride, size_t a_block_pitch, const uint8_t* d_b,
                                 uint32_t b_stride, size_t b_block_pitch, uint32_t width, uint32_t height,
                                 uint64_t* d_out, size_t nblocks, void* stream) {
    return sad_sse_common(true, d_a, a_stride, a_block_pitch, d_b, b_stride, b_block_pitch, width, height, d_out, nblocks, stream);
}
extern "C" int svt_hip_residual_batch(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch,
                                      const uint8_t* d_pred, uint32_t pred_stride, size_t pred_block_pitch,
                                      int16_t* d_res, uint32_t res_stride, size_t res_block_pitch, uint32_t width,
                                      uint32_t height, size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_src || !d_pred || !d_res) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (width == 0 || height == 0) return set_err(SVT_HIP_ERR_INVALID, "empty block");
    if (nblocks == 0) return SVT_HIP_OK;
    const uint32_t rcs = (width & 15) == 0 ? 16u : ((width & 7) == 0 ? 8u : ((width & 3) == 0 ? 4u : 1u));
    const size_t total = (size_t)(width / rcs) * height * nblocks;
    const size_t grid = (total + 255) / 256;
    if (grid > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many blocks for one launch");
    const uint32_t cpr = width / rcs;
    const size_t per = (size_t)cpr * height;
    const bool pow2 = (cpr & (cpr - 1)) == 0 && (per & (per - 1)) == 0;
#define RESL(CS, P2) hipLaunchKernelGGL((residual_kernel<CS, P2>), dim3((uint32_t)grid), dim3(256), 0, (hipStream_t)stream, d_src, src_stride, \
                       src_block_pitch, d_pred, pred_stride, pred_block_pitch, d_res, res_stride, res_block_pitch,   \
                       width, height, (uint32_t)nblocks)
#define RESC(CS) if (pow2) RESL(CS, true); else RESL(CS, false)
    if (rcs == 16) { RESC(16); } else if (rcs == 8) { RESC(8); } else if (rcs == 4) { RESC(4); } else { RESC(1); }
#undef RESC
#undef RESL
    return launch_status("residual");
}

static int sad_search_impl(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch, const uint32_t* d_src_offs,
                           const uint8_t* d_ref, uint32_t ref_stride, uint32_t ref_stride_raw,
                           size_t ref_block_pitch, const uint32_t* d_ref_offs, uint32_t width, uint32_t height,
                           int16_t search_area_width, int16_t search_area_height,
                           uint64_t* d_best_sad, int16_t* d_x, int16_t* d_y, size_t nblocks,
                           void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_src || !d_ref || !d_best_sad || !d_x || !d_y) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (width == 0 || height == 0 || width > 64 || height > 64) return set_err(SVT_HIP_ERR_INVALID, "block %ux%u", width, height);
    if (search_area_width <= 0 || search_area_height <= 0) return set_err(SVT_HIP_ERR_INVALID, "empty search area");
    if (nblocks == 0) return SVT_HIP_OK;
    const uint32_t win_w = width + search_area_width - 1;
    const bool plain = ref_stride == ref_stride_raw;
    const uint32_t nrows = plain ? (uint32_t)(search_area_height + height - 1) : (uint32_t)search_area_height * height;
    const bool q16_for_16x16 = !g_tune_no_q16 && width == 16 && search_area_width % 16 == 0;      // see the q16 routing below
    if (plain && !g_tune_no_qsad && !g_tune_no_q2 && ((width == 16 && height == 16 && !q16_for_16x16) || (width == 8 && height == 8))) {
        // small blocks: source block in registers, 4 x 2 candidates per lane
        const uint32_t wpitch = ((win_w + 15) & ~15u) + 16;
        const uint32_t src_bytes = (width * height + 15) & ~15u;
        const uint32_t groups = (uint32_t)((search_area_width + 3) / 4) * (uint32_t)((search_area_height + 1) / 2);
        uint32_t lpb = 1;
        while (lpb < groups && lpb < 64) lpb <<= 1;
        // window + one spare row + 16 spare bytes per lane, padded to 8 (mod 32) bytes (LDS bank spread, see the kernel)
        uint32_t ref_bytes = wpitch * (nrows + 1) + 16 * lpb;
        ref_bytes = ((ref_bytes + 31) & ~31u) + 8;
        const uint32_t per_blk = src_bytes + ref_bytes;
        if (per_blk <= 64 * 1024) {
            uint32_t threads = 256;
            while (threads > lpb && (size_t)(threads / lpb) * per_blk > 64 * 1024) threads >>= 1;
            const uint32_t slots = threads / lpb;
            const uint32_t grid = (uint32_t)((nblocks + slots - 1) / slots);
#define SSQ2(CW, CH, SU)                                                                                                \
    hipLaunchKernelGGL((sad_search_q2_kernel<CW, CH, SU>), dim3(grid), dim3(threads), (size_t)slots * per_blk, (hipStream_t)stream, \
                       d_src, src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, (int)search_area_width,         \
                       (int)search_area_height, (unsigned long long*)d_best_sad, d_x, d_y, ref_bytes, lpb, cpr_magic,           \
                       d_src_offs, d_ref_offs, (uint32_t)nblocks)
            // exact j / cpr for j < 2^16 chunks (cpr <= 8): floor(2^32 / cpr) + 1
            const uint32_t cpr_magic = (uint32_t)(0x100000000ull / ((win_w + 15) >> 4)) + 1u;
            // staging depth: all of a lane's chunks in ONE batch of loads when that takes at most 8 per lane (one memory
            // latency per block instead of two: the kernel is latency-bound at 3 waves per SIMD)
            const uint32_t nchunk = ((win_w + 15) >> 4) * nrows;
            const bool deep = !g_tune_q2_su4 && nchunk > 4 * lpb;
            if (width == 16) { if (deep) SSQ2(16, 16, 8); else SSQ2(16, 16, 4); }
            else { if (deep) SSQ2(8, 8, 8); else SSQ2(8, 8, 4); }
#undef SSQ2
            return launch_status("sad_search_q2");
        }
    }
    // 16-wide blocks: 16x32 / 16x64 always (measured 2.1-2.3x over sad_search_q_kernel); 16x16 when no candidate
    // of a lane is masked (search width % 16 == 0: 10 % over q2, equal otherwise)
    if (plain && !g_tune_no_qsad && !g_tune_no_q16 && height % (256 / (width ? width : 1)) == 0 &&
        (width == 32 || width == 64 || (width == 16 && (height != 16 || search_area_width % 16 == 0)))) {
        // wide blocks: 16 candidates per lane on b128 LDS reads (sad_search_q16_kernel)
        uint32_t wpitch = (((uint32_t)search_area_width + 15) & ~15u) + width;
        if (((wpitch >> 4) & 1) == 0) wpitch += 16;            // odd multiple of 16 B: bank spread over search rows
        const uint32_t ref_bytes = wpitch * nrows;
        const uint32_t per_blk = width * height + ref_bytes;
        if (per_blk <= 64 * 1024) {
            const uint32_t tasks = (uint32_t)((search_area_width + 15) / 16) * (uint32_t)search_area_height;
            uint32_t tsh = 0;
            while ((1u << tsh) < tasks && tsh < 6) tsh++;
            const uint32_t row_groups = height / (256 / width);
            uint32_t lpb = 1u << tsh;
            while (lpb < 64 && (lpb >> tsh) * 2 <= row_groups) lpb <<= 1;
            uint32_t threads = 256;
            while (threads > lpb && (size_t)(threads / lpb) * per_blk > 64 * 1024) threads >>= 1;
            const uint32_t slots = threads / lpb;
            const uint32_t grid = (uint32_t)((nblocks + slots - 1) / slots);
            const uint32_t cpr_magic = (uint32_t)(0x100000000ull / ((win_w + 15) >> 4)) + 1u;
#define SSQ16(CW)                                                                                                       \
    hipLaunchKernelGGL((sad_search_q16_kernel<CW>), dim3(grid), dim3(threads), (size_t)slots * per_blk, (hipStream_t)stream, \
                       d_src, src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, height, (int)search_area_width,  \
                       (int)search_area_height, (unsigned long long*)d_best_sad, d_x, d_y, wpitch, ref_bytes, lpb, tsh,         \
                       cpr_magic, d_src_offs, d_ref_offs, (uint32_t)nblocks)
            if (width == 16) SSQ16(16); else if (width == 32) SSQ16(32); else SSQ16(64);
#undef SSQ16
            return launch_status("sad_search_q16");
        }
    }
    if ((width & 3) == 0 && !g_tune_no_qsad) {
        // quad-SAD kernel: 4 candidates per lane
        const uint32_t wpitch = ((win_w + 15) & ~15u) + 16;
        const uint32_t src_bytes = (width * height + 15) & ~15u;
        const uint32_t ref_bytes = (wpitch * nrows + 16 + 15) & ~15u;
        const uint32_t per_blk = src_bytes + ref_bytes;
        if (per_blk > 64 * 1024) return set_err(SVT_HIP_ERR_INVALID, "search window needs %u B of LDS per block (> 64 KiB)", per_blk);
        const uint32_t groups = (uint32_t)((search_area_width + 3) / 4) * (uint32_t)search_area_height;
        uint32_t lpb = 1;
        while (lpb < groups && lpb < 64) lpb <<= 1;
        uint32_t threads = 256;
        while (threads > lpb && (size_t)(threads / lpb) * per_blk > 64 * 1024) threads >>= 1;
        const uint32_t slots = threads / lpb;
        const uint32_t grid = (uint32_t)((nblocks + slots - 1) / slots);
#define SSQ(CW, CH)                                                                                                     \
    hipLaunchKernelGGL((sad_search_q_kernel<CW, CH>), dim3(grid), dim3(threads), (size_t)slots * per_blk, (hipStream_t)stream, \
                       d_src, src_stride, src_block_pitch, d_ref, ref_stride, ref_stride_raw, ref_block_pitch, width, height,  \
                       (int)search_area_width, (int)search_area_height, (unsigned long long*)d_best_sad, d_x, d_y, src_bytes, \
                       ref_bytes, lpb, d_src_offs, d_ref_offs, (uint32_t)nblocks)
        if (width == 16 && height == 16) SSQ(16, 16);
        else if (width == 8 && height == 8) SSQ(8, 8);
        else if (width == 32 && height == 32) SSQ(32, 32);
        else if (width == 64 && height == 64) SSQ(64, 64);
        else SSQ(0, 0);
#undef SSQ
        return launch_status("sad_search_q");
    }
    const uint32_t wpitch = (win_w + 3 + 8) & ~3u;
    const uint32_t spitch = (width + 3) & ~3u;
    const uint32_t src_bytes = (spitch * height + 15) & ~15u;
    const uint32_t ref_bytes = (wpitch * nrows + 16 + 15) & ~15u;
    const uint32_t per_wave = src_bytes + ref_bytes;
    if (per_wave > 64 * 1024) return set_err(SVT_HIP_ERR_INVALID, "search window needs %u B of LDS per block (> 64 KiB)", per_wave);
    uint32_t waves = (64 * 1024) / per_wave;
    if (waves > 4) waves = 4;
    const uint32_t grid = (uint32_t)((nblocks + waves - 1) / waves);
    hipLaunchKernelGGL(sad_search_kernel, dim3(grid), dim3(waves * 64), waves * per_wave, (hipStream_t)stream, d_src,
                       src_stride, src_block_pitch, d_ref, ref_stride, ref_stride_raw, ref_block_pitch, width, height,
                       (int)search_area_width, (int)search_area_height, (unsigned long long*)d_best_sad, d_x, d_y,
                       src_bytes, ref_bytes, d_src_offs, d_ref_offs, (uint32_t)nblocks);
    return launch_status("sad_search");
}

extern "C" int svt_hip_sad_search_batch(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch,
                                        const uint8_t* d_ref, uint32_t ref_stride, uint32_t ref_stride_raw,
                                        size_t ref_block_pitch, uint32_t width, uint32_t height,
                                        int16_t search_area_width, int16_t search_area_height,
                                        uint64_t* d_best_sad, int16_t* d_x, int16_t* d_y, size_t nblocks,
                                        void* stream) {
    return sad_search_impl(d_src, src_stride, src_block_pitch, nullptr, d_ref, ref_stride, ref_stride_raw, ref_block_pitch,
                           nullptr, width, height, search_area_width, search_area_height, d_best_sad, d_x, d_y, nblocks, stream);
}
extern "C" int svt_hip_sad_search_planes_batch(const uint8_t* d_src_plane, uint32_t src_stride, const uint32_t* d_src_offsets,
                                               const uint8_t* d_ref_plane, uint32_t ref_stride, uint32_t ref_stride_raw,
                                               const uint32_t* d_ref_offsets, uint32_t width, uint32_t height,
                                               int16_t search_area_width, int16_t search_area_height,
                                               uint64_t* d_best_sad, int16_t* d_x, int16_t* d_y, size_t nblocks,
                                               void* stream) {
    if (nblocks && (!d_src_offsets || !d_ref_offsets)) { if (int rc = require_init()) return rc; return set_err(SVT_HIP_ERR_INVALID, "NULL offset table"); }
    return sad_search_impl(d_src_plane, src_stride, 0, d_src_offsets, d_ref_plane, ref_stride, ref_stride_raw, 0, d_ref_offsets,
                           width, height, search_area_width, search_area_height, d_best_sad, d_x, d_y, nblocks, stream);
}

static int me_sb_search_impl(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch, const uint32_t* d_src_offs,
                             const uint8_t* d_ref, uint32_t ref_stride, size_t ref_block_pitch, const uint32_t* d_ref_offs,
                             int search_w, int search_h, const int16_t* d_origins, int x_origin,
                             int y_origin, uint32_t* d_best_sad, uint32_t* d_best_mv, size_t nblocks,
                             void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_src || !d_ref || !d_best_sad || !d_best_mv) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (search_w <= 0 || search_h <= 0 || search_w * search_h > 4096)
        return set_err(SVT_HIP_ERR_INVALID, "search area %dx%d (1..4096 points)", search_w, search_h);
    const uint32_t win_w = 64 + search_w - 1, win_h = 64 + search_h - 1;
    const uint32_t wpitch = ((win_w + 15) & ~15u) + 16;
    const size_t lds = 32 * 64 + (size_t)wpitch * win_h;
    if (lds > 60 * 1024) return set_err(SVT_HIP_ERR_INVALID, "search window needs %zu B of LDS (> 60 KiB)", lds);
    if (!g_tune_no_me16) {      // 16 points per lane; widths that are not a multiple of 16 mask the tail of each row
        if ((search_w & 15) == 0)
            hipLaunchKernelGGL(me_sb_search16_kernel<false>, dim3((uint32_t)nblocks), dim3(ME_THREADS), lds, (hipStream_t)stream, d_src,
                               src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, search_w, search_h, d_origins,
                               x_origin, y_origin, d_best_sad, d_best_mv, wpitch, d_src_offs, d_ref_offs, (uint32_t)nblocks);
        else
            hipLaunchKernelGGL(me_sb_search16_kernel<true>, dim3((uint32_t)nblocks), dim3(ME_THREADS), lds, (hipStream_t)stream, d_src,
                               src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, search_w, search_h, d_origins,
                               x_origin, y_origin, d_best_sad, d_best_mv, wpitch, d_src_offs, d_ref_offs, (uint32_t)nblocks);
        return launch_status("me_sb_search16");
    }
    hipLaunchKernelGGL(me_sb_search_kernel, dim3((uint32_t)nblocks), dim3(ME_THREADS), lds, (hipStream_t)stream, d_src,
                       src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, search_w, search_h, d_origins,
                       x_origin, y_origin, d_best_sad, d_best_mv, wpitch, d_src_offs, d_ref_offs, (uint32_t)nblocks);
    return launch_status("me_sb_search");
}

extern "C" int svt_hip_me_sb_search_batch(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch,
                                          const uint8_t* d_ref, uint32_t ref_stride, size_t ref_block_pitch,
                                          int search_w, int search_h, const int16_t* d_origins, int x_origin,
                                          int y_origin, uint32_t* d_best_sad, uint32_t* d_best_mv, size_t nblocks,
                                          void* stream) {
    return me_sb_search_impl(d_src, src_stride, src_block_pitch, nullptr, d_ref, ref_stride, ref_block_pitch, nullptr, search_w,
                             search_h, d_origins, x_origin, y_origin, d_best_sad, d_best_mv, nblocks, stream);
}
extern "C" int svt_hip_me_sb_search_planes_batch(const uint8_t* d_src_plane, uint32_t src_stride, const uint32_t* d_src_offsets,
                                                 const uint8_t* d_ref_plane, uint32_t ref_stride, const uint32_t* d_ref_offsets,
                                                 int search_w, int search_h, const int16_t* d_origins, int x_origin,
                                                 int y_origin, uint32_t* d_best_sad, uint32_t* d_best_mv, size_t nblocks,
                                                 void* stream) {
    if (nblocks && (!d_src_offsets || !d_ref_offsets)) { if (int rc = require_init()) return rc; return set_err(SVT_HIP_ERR_INVALID, "NULL offset table"); }
    return me_sb_search_impl(d_src_plane, src_stride, 0, d_src_offsets, d_ref_plane, ref_stride, 0, d_ref_offsets, search_w,
                             search_h, d_origins, x_origin, y_origin, d_best_sad, d_best_mv, nblocks, stream);
}

// K6 in the reference's result layout, both result flavours, square or all 209 PUs (include/svt_hip_dsp.h)
extern "C" int svt_hip_me_fullpel_search_batch(const uint8_t* d_src, uint32_t src_stride, size_t src_block_pitch,
                                               const uint32_t* d_src_offsets, const uint8_t* d_ref, uint32_t ref_stride,
                                               size_t ref_block_pitch, const uint32_t* d_ref_offsets, int search_w,
                                               int search_h, const int16_t* d_origins, int x_origin, int y_origin,
                                               int flavour, int nsq, uint32_t* d_best_sad, uint32_t* d_best_mv,
                                               uint32_t pu_pitch, size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_src || !d_ref || !d_best_sad || !d_best_mv) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (flavour != SVT_HIP_FLAVOUR_C && flavour != SVT_HIP_FLAVOUR_AVX2) return set_err(SVT_HIP_ERR_INVALID, "flavour %d", flavour);
    const uint32_t npus = nsq ? SVT_HIP_ME_PUS_ALL : SVT_HIP_ME_PUS;
    if (pu_pitch < npus) return set_err(SVT_HIP_ERR_INVALID, "pu_pitch %u < %u PUs", pu_pitch, npus);
    if (search_w <= 0 || search_h <= 0 || search_w * search_h > 4096)
        return set_err(SVT_HIP_ERR_INVALID, "search area %dx%d (1..4096 points)", search_w, search_h);
    if (!nsq && !g_tune_me_exact) {
        // square PUs: the 16-points-per-lane kernel, any width; the AVX2 flavour only re-labels the 32x32 keys
        const uint32_t win_w = 64 + search_w - 1, win_h = 64 + search_h - 1;
        const uint32_t wpitch = ((win_w + 15) & ~15u) + 16;
        const size_t lds = 32 * 64 + (size_t)wpitch * win_h;
        if (lds <= 60 * 1024) {
            const int w8q = flavour == SVT_HIP_FLAVOUR_AVX2 ? (search_w & ~7) : 0;
            if ((search_w & 15) == 0)
                hipLaunchKernelGGL(me_sb_search16_kernel<false>, dim3((uint32_t)nblocks), dim3(ME_THREADS), lds, (hipStream_t)stream, d_src,
                                   src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, search_w, search_h, d_origins,
                                   x_origin, y_origin, d_best_sad, d_best_mv, wpitch, d_src_offsets, d_ref_offsets, (uint32_t)nblocks,
                                   w8q, 1, pu_pitch);
            else
                hipLaunchKernelGGL(me_sb_search16_kernel<true>, dim3((uint32_t)nblocks), dim3(ME_THREADS), lds, (hipStream_t)stream, d_src,
                                   src_stride, src_block_pitch, d_ref, ref_stride, ref_block_pitch, search_w, search_h, d_origins,
                                   x_origin, y_origin, d_best_sad, d_best_mv, wpitch, d_src_offsets, d_ref_offsets, (uint32_t)nblocks,
                                   w8q, 1, pu_pitch);
            return launch_status("me_sb_search16 (reference layout)");
        }
    }
    hipLaunchKernelGGL(me_fullpel_exact_kernel, dim3((uint32_t)nblocks), dim3(ME_THREADS), 0, (hipStream_t)stream, d_src, src_stride,
                       src_block_pitch, d_src_offsets, d_ref, ref_stride, ref_block_pitch, d_ref_offsets, search_w, search_h,
                       d_origins, x_origin, y_origin, flavour, nsq ? 1 : 0, d_best_sad, d_best_mv, pu_pitch, (uint32_t)nblocks);
    return launch_status("me_fullpel_exact");
}

extern "C" int svt_hip_full_distortion32_batch(const int32_t* d_coeff, uint32_t coeff_stride, size_t coeff_block_pitch,
                                               const int32_t* d_recon, uint32_t recon_stride, size_t recon_block_pitch,
                                               uint32_t width, uint32_t height, int cbf_zero, uint64_t* d_out,
                                               size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_coeff || !d_out || (!cbf_zero && !d_recon)) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (width == 0 || height == 0 || width > 128 || height > 128) return set_err(SVT_HIP_ERR_INVALID, "area %ux%u", width, height);
    hipLaunchKernelGGL(full_distortion32_kernel, dim3((uint32_t)((nblocks + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                       d_coeff, coeff_stride, coeff_block_pitch, d_recon, recon_stride, recon_block_pitch, width, height,
                       cbf_zero, (unsigned long long*)d_out, (uint32_t)nblocks);
    return launch_status("full_distortion32");
}

// ---- K11 chroma-from-luma helpers + av1_txb_init_levels ----
static bool cfl_dim_ok(uint32_t v) { return v == 4 || v == 8 || v == 16 || v == 32; }

static int cfl_ac_launch(int in_mode, const void* d_luma, uint32_t luma_stride, size_t luma_block_pitch, const uint32_t* d_xy,
                         int16_t* d_q3, uint32_t q3_line, size_t q3_block_pitch, uint32_t w, uint32_t h, int subtract,
                         int round_offset, int num_pel_log2, size_t nblocks, void* stream, const char* what) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_q3 || (in_mode != 2 && !d_luma)) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (!cfl_dim_ok(w) || !cfl_dim_ok(h)) return set_err(SVT_HIP_ERR_INVALID, "chroma block %ux%u", w, h);
    if (q3_line < w || q3_block_pitch < (size_t)q3_line * (h - 1) + w) return set_err(SVT_HIP_ERR_INVALID, "q3 layout: line %u, block pitch %zu", q3_line, q3_block_pitch);
    if (num_pel_log2 < 0 || num_pel_log2 > 31) return set_err(SVT_HIP_ERR_INVALID, "num_pel_log2 %d", num_pel_log2);
    const uint32_t nchunks = (w / (w < 8 ? 4 : 8)) * h;
    const uint32_t lpb = nchunks < 64 ? nchunks : 64;
    const size_t lanes = nblocks * lpb;
    const size_t grid = (lanes + 255) / 256;
    if (grid > 0x7fffffffu || nblocks > 0x7fffffffu / 64) return set_err(SVT_HIP_ERR_INVALID, "too many blocks for one launch");
#define CFL_AC(IN)                                                                                                          \
    hipLaunchKernelGGL((cfl_ac_kernel<IN>), dim3((uint32_t)grid), dim3(256), 0, (hipStream_t)stream, d_luma, luma_stride,   \
                       luma_block_pitch, d_xy, d_q3, q3_line, q3_block_pitch, w, h, lpb, subtract, round_offset, num_pel_log2, \
                       (uint32_t)nblocks)
    if (in_mode == 0) CFL_AC(0); else if (in_mode == 1) CFL_AC(1); else CFL_AC(2);
#undef CFL_AC
    return launch_status(what);
}

extern "C" int svt_hip_cfl_luma_subsampling_420_batch(const void* d_luma, uint32_t luma_stride, size_t luma_block_pitch,
                                                      const uint32_t* d_xy, int is_16bit, int16_t* d_q3, uint32_t q3_line,
                                                      size_t q3_block_pitch, uint32_t width, uint32_t height,
                                                      int subtract_average, size_t nblocks, void* stream) {
    if ((width & 1) || (height & 1)) return set_err(SVT_HIP_ERR_INVALID, "luma block %ux%u", width, height);
    const uint32_t w = width >> 1, h = height >> 1;
    int lg = 0;
    while ((1u << lg) < w * h) lg++;
    return cfl_ac_launch(is_16bit ? 1 : 0, d_luma, luma_stride, luma_block_pitch, d_xy, d_q3, q3_line, q3_block_pitch, w, h,
                         subtract_average ? 1 : 0, (int)(w * h / 2), lg, nblocks, stream, "cfl_luma_subsampling_420");
}

extern "C" int svt_hip_subtract_average_batch(int16_t* d_q3, uint32_t q3_line, size_t q3_block_pitch, uint32_t width,
                                              uint32_t height, int32_t round_offset, int32_t num_pel_log2, size_t nblocks,
                                              void* stream) {
    return cfl_ac_launch(2, nullptr, 0, 0, nullptr, d_q3, q3_line, q3_block_pitch, width, height, 1, round_offset,
                         num_pel_log2, nblocks, stream, "subtract_average");
}

extern "C" int svt_hip_cfl_predict_batch(const int16_t* d_ac_q3, uint32_t q3_line, size_t q3_block_pitch, const void* d_pred,
                                         uint32_t pred_stride, void* d_dst, uint32_t dst_stride, const uint32_t* d_xy,
                                         const int32_t* d_alpha_q3, int bit_depth, uint32_t width, uint32_t height,
                                         int is_16bit, size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_ac_q3 || !d_pred || !d_dst || !d_alpha_q3) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (!cfl_dim_ok(width) || !cfl_dim_ok(height)) return set_err(SVT_HIP_ERR_INVALID, "chroma block %ux%u", width, height);
    if ((is_16bit && bit_depth != 8 && bit_depth != 10 && bit_depth != 12) || (!is_16bit && bit_depth != 8))
        return set_err(SVT_HIP_ERR_INVALID, "bit depth %d", bit_depth);
    if (q3_line < width || pred_stride < width || dst_stride < width) return set_err(SVT_HIP_ERR_INVALID, "stride smaller than the block");
    const uint32_t nchunks = (width / (width < 8 ? 4 : 8)) * height;
    uint32_t sh = 0;
    while ((1u << sh) < nchunks) sh++;
    const size_t grid = ((nblocks << sh) + 255) / 256;
    if (grid > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many blocks for one launch");
    const int hi = (1 << bit_depth) - 1;
    if (is_16bit)
        hipLaunchKernelGGL((cfl_predict_kernel<uint16_t>), dim3((uint32_t)grid), dim3(256), 0, (hipStream_t)stream, d_ac_q3, q3_line,
                           q3_block_pitch, (const uint16_t*)d_pred, pred_stride, (uint16_t*)d_dst, dst_stride, d_xy, d_alpha_q3, hi,
                           width, height, sh, (uint32_t)nblocks);
    else
        hipLaunchKernelGGL((cfl_predict_kernel<uint8_t>), dim3((uint32_t)grid), dim3(256), 0, (hipStream_t)stream, d_ac_q3, q3_line,
                           q3_block_pitch, (const uint8_t*)d_pred, pred_stride, (uint8_t*)d_dst, dst_stride, d_xy, d_alpha_q3, hi,
                           width, height, sh, (uint32_t)nblocks);
    return launch_status("cfl_predict");
}

extern "C" int svt_hip_txb_init_levels_batch(const int32_t* d_coeff, size_t coeff_block_pitch, uint8_t* d_levels_buf,
                                             size_t levels_block_pitch, uint32_t width, uint32_t height, size_t nblocks,
                                             void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_coeff || !d_levels_buf) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    auto ok = [](uint32_t v) { return v == 4 || v == 8 || v == 16 || v == 32 || v == 64; };
    if (!ok(width) || !ok(height)) return set_err(SVT_HIP_ERR_INVALID, "block %ux%u", width, height);
    const uint32_t bytes = (width + 4) * (height + 6) + 16;      // (W + TX_PAD_HOR) * (H + TX_PAD_VER) + TX_PAD_END
    if (levels_block_pitch < bytes || (levels_block_pitch & 3) || ((uintptr_t)d_levels_buf & 3))
        return set_err(SVT_HIP_ERR_INVALID, "levels buffer: %zu B per block (need >= %u, multiple of 4, 4-byte aligned)", levels_block_pitch, bytes);
    if (coeff_block_pitch < (size_t)width * height) return set_err(SVT_HIP_ERR_INVALID, "coeff_block_pitch %zu", coeff_block_pitch);
    const uint32_t ndw = bytes >> 2, dpr = (width + 4) >> 2;
    const bool wide = (levels_block_pitch & 15) == 0 && ((uintptr_t)d_levels_buf & 15) == 0;
    const uint32_t items = wide ? (ndw + 3) / 4 : ndw;
    uint32_t lpb = 1;
    while (lpb < items && lpb < 256) lpb <<= 1;
    const uint32_t slots = 256 / lpb;
    const size_t grid = (nblocks + slots - 1) / slots;
    if (grid > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many blocks for one launch");
    const uint32_t row_magic = (uint32_t)(0x100000000ull / dpr) + 1u;
    if (wide)
        hipLaunchKernelGGL(txb_init_levels_kernel<true>, dim3((uint32_t)grid), dim3(256), 0, (hipStream_t)stream, d_coeff, coeff_block_pitch,
                           d_levels_buf, levels_block_pitch, width, height, lpb, ndw, row_magic, (uint32_t)nblocks);
    else
        hipLaunchKernelGGL(txb_init_levels_kernel<false>, dim3((uint32_t)grid), dim3(256), 0, (hipStream_t)stream, d_coeff, coeff_block_pitch,
                           d_levels_buf, levels_block_pitch, width, height, lpb, ndw, row_magic, (uint32_t)nblocks);
    return launch_status("txb_init_levels");
}

static bool intra_size_ok(int bw, int bh) {
    auto ok1 = [](int v) { return v == 4 || v == 8 || v == 16 || v == 32 || v == 64; };
    if (!ok1(bw) || !ok1(bh)) return false;
    const int m = bw > bh ? bw : bh, mn = bw < bh ? bw : bh;
    return m <= 4 * mn;     // the 19 TX sizes
}

static int intra_pred_impl(void* d_dst, int32_t dst_stride, size_t dst_block_pitch,
                           const uint32_t* d_dst_offsets, const void* d_above, const void* d_left,
                           int32_t nb_pitch, int mode, int bw, int bh, int upsample_above,
                           int upsample_left, int dx, int dy, int is_16bit, int bd, size_t nblocks,
                           void* stream, const DirMulti* multi);

extern "C" int svt_hip_intra_pred_batch(void* d_dst, int32_t dst_stride, size_t dst_block_pitch,
                                        const uint32_t* d_dst_offsets, const void* d_above, const void* d_left,
                                        int32_t nb_pitch, int mode, int bw, int bh, int upsample_above,
                                        int upsample_left, int dx, int dy, int is_16bit, int bd, size_t nblocks,
                                        void* stream) {
    return intra_pred_impl(d_dst, dst_stride, dst_block_pitch, d_dst_offsets, d_above, d_left, nb_pitch, mode, bw, bh,
                           upsample_above, upsample_left, dx, dy, is_16bit, bd, nblocks, stream, nullptr);
}

// multi (directional modes only): several (dx, dy) of the same zone in one launch on edges staged once, see DirMulti
static int intra_pred_impl(void* d_dst, int32_t dst_stride, size_t dst_block_pitch,
                           const uint32_t* d_dst_offsets, const void* d_above, const void* d_left,
                           int32_t nb_pitch, int mode, int bw, int bh, int upsample_above,
                           int upsample_left, int dx, int dy, int is_16bit, int bd, size_t nblocks,
                           void* stream, const DirMulti* multi) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_dst || !d_above || !d_left) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (mode < 0 || mode >= SVT_INTRA_MODES) return set_err(SVT_HIP_ERR_INVALID, "intra mode %d", mode);
    if (!intra_size_ok(bw, bh)) return set_err(SVT_HIP_ERR_INVALID, "block %dx%d is not an AV1 transform size", bw, bh);
    if ((is_16bit && bd != 10 && bd != 12 && bd != 8) || (!is_16bit && bd != 8)) return set_err(SVT_HIP_ERR_INVALID, "bit depth %d", bd);
    if ((upsample_above | upsample_left) & ~1) return set_err(SVT_HIP_ERR_INVALID, "upsample flags");
    if (mode >= SVT_INTRA_Z1) {
        if (dx <= 0 || dy <= 0) return set_err(SVT_HIP_ERR_INVALID, "dx/dy must be positive");
        const int need = NB_ORIGIN + (((bw + bh) << 1) + 2);
        if (nb_pitch < need) return set_err(SVT_HIP_ERR_INVALID, "nb_pitch %d < %d", nb_pitch, need);
    } else if (nb_pitch < NB_ORIGIN + (bw > bh ? bw : bh)) {
        return set_err(SVT_HIP_ERR_INVALID, "nb_pitch %d too small", nb_pitch);
    }
    const int es = is_16bit ? 2 : 1;
    const int pxl = 16 / es, ppl = bw < pxl ? bw : pxl;
    const size_t per_block = (size_t)(bw / ppl) * bh;            // lanes per block: power of two, 4..512
    const size_t items = per_block * nblocks;
    size_t grid = (items + 255) / 256;
    if (grid > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "too many blocks for one launch");
    // grid * 256 must be a multiple of per_block so that a lane keeps its (row, column) across iterations
    if (per_block > 256) grid = (grid + 1) & ~(size_t)1;
    hipStream_t s = (hipStream_t)stream;
    if (mode >= SVT_INTRA_Z1) {
        // last edge sample the reference may read: index max_base = (bw + bh - 1) << upsample; the LDS copy
        // continues with copies of it for one lane-row (+2) so the pixel loop needs no bounds test
        const int lim_a = NB_ORIGIN + ((bw + bh - 1) << upsample_above), lim_l = NB_ORIGIN + ((bw + bh - 1) << upsample_left);
        const int up = upsample_above > upsample_left ? upsample_above : upsample_left;
        const int n_pad = (lim_a > lim_l ? lim_a : lim_l) + ((16 / es) << up) + 3;
        const size_t slots = per_block >= 256 ? 1 : 256 / per_block;
        const size_t shmem = slots * 2 * (size_t)((n_pad + 7) & ~7) * 4;            // pair dwords (see the kernel)
        DirMulti dm;
        if (multi) dm = *multi; else dm.n = 0;
#define IDL(T, M)                                                                                                     \
    hipLaunchKernelGGL((intra_dir_kernel<T, M>), dim3((uint32_t)grid), dim3(256), shmem, s, (T*)d_dst, dst_stride, \
                       dst_block_pitch, d_dst_offsets, (const T*)d_above, (const T*)d_left, nb_pitch, bw, bh,            \
                       upsample_above, upsample_left, dx, dy, lim_a, lim_l, n_pad, bd, (uint32_t)nblocks, dm)
#define IDM(T)                                                                                                        \
    switch (mode) {                                                                                                   \
    case SVT_INTRA_Z1: IDL(T, IM_Z1); break; case SVT_INTRA_Z2: IDL(T, IM_Z2); break; default: IDL(T, IM_Z3); break;  \
    }
        if (is_16bit) { IDM(uint16_t) } else { IDM(uint8_t) }
#undef IDM
#undef IDL
        return launch_status("intra_dir");
    }
    const int cnt = mode == SVT_INTRA_DC ? bw + bh : (mode == SVT_INTRA_DC_TOP ? bw : bh);
    const uint32_t dc_magic = (uint32_t)(0x100000000ull / (uint64_t)cnt) + 1u;
    constexpr int INTRA_IU = 2;                                    // blocks per lane in the wide kernels
    const size_t grid_w = (grid + INTRA_IU - 1) / INTRA_IU;
#define IPL(T, M)                                                                                                     \
    if (bw >= 16 / (int)sizeof(T))                                                                                    \
        hipLaunchKernelGGL((intra_pred_kernel<T, M, true, INTRA_IU>), dim3((uint32_t)(per_block > 256 ? (grid_w + 1) & ~(size_t)1 : grid_w)), dim3(256), 0, s, (T*)d_dst, dst_stride, \
                           dst_block_pitch, d_dst_offsets, (const T*)d_above, (const T*)d_left, nb_pitch, bw, bh, bd,    \
                           dc_magic, (uint32_t)nblocks);                                                              \
    else                                                                                                              \
        hipLaunchKernelGGL((intra_pred_kernel<T, M, false, 1>), dim3((uint32_t)grid), dim3(256), 0, s, (T*)d_dst, dst_stride, \
                           dst_block_pitch, d_dst_offsets, (const T*)d_above, (const T*)d_left, nb_pitch, bw, bh, bd,    \
                           dc_magic, (uint32_t)nblocks)
#define IPM(T)                                                                                                        \
    switch (mode) {                                                                                                   \
    case SVT_INTRA_DC: IPL(T, IM_DC); break; case SVT_INTRA_V: IPL(T, IM_V); break; case SVT_INTRA_H: IPL(T, IM_H); break; \
    case SVT_INTRA_SMOOTH: IPL(T, IM_SMOOTH); break; case SVT_INTRA_SMOOTH_V: IPL(T, IM_SMOOTH_V); break;             \
    case SVT_INTRA_SMOOTH_H: IPL(T, IM_SMOOTH_H); break; case SVT_INTRA_PAETH: IPL(T, IM_PAETH); break;               \
    case SVT_INTRA_DC_TOP: IPL(T, IM_DC_TOP); break; case SVT_INTRA_DC_LEFT: IPL(T, IM_DC_LEFT); break;               \
    default: IPL(T, IM_DC_128); break;                                                                                \
    }
    if (is_16bit) { IPM(uint16_t) } else { IPM(uint8_t) }
#undef IPM
#undef IPL
    return launch_status("intra_pred");
}

extern "C" int svt_hip_filter_intra_edge_batch(void* d_edges, int32_t nb_pitch, int sz, int strength, int is_16bit,
                                               size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0 || strength == 0) return SVT_HIP_OK;
    if (!d_edges) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (sz < 1 || sz > 129 || strength < 0 || strength > 3 || nb_pitch < NB_ORIGIN + sz)
        return set_err(SVT_HIP_ERR_INVALID, "edge sz %d strength %d pitch %d", sz, strength, nb_pitch);
    hipStream_t s = (hipStream_t)stream;
    if (is_16bit)
        hipLaunchKernelGGL((filter_edge_kernel<uint16_t>), dim3((uint32_t)nblocks), dim3(256), 0, s, (uint16_t*)d_edges, nb_pitch, NB_ORIGIN, sz, strength, (uint32_t)nblocks);
    else
        hipLaunchKernelGGL((filter_edge_kernel<uint8_t>), dim3((uint32_t)nblocks), dim3(256), 0, s, (uint8_t*)d_edges, nb_pitch, NB_ORIGIN, sz, strength, (uint32_t)nblocks);
    return launch_status("filter_intra_edge");
}
extern "C" int svt_hip_upsample_intra_edge_batch(void* d_edges, int32_t nb_pitch, int sz, int is_16bit, int bd,
                                                 size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_edges) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (sz < 1 || sz > 16 || nb_pitch < NB_ORIGIN + 2 * sz) return set_err(SVT_HIP_ERR_INVALID, "upsample sz %d pitch %d", sz, nb_pitch);
    hipStream_t s = (hipStream_t)stream;
    if (is_16bit)
        hipLaunchKernelGGL((upsample_edge_kernel<uint16_t>), dim3((uint32_t)nblocks), dim3(64), 0, s, (uint16_t*)d_edges, nb_pitch, NB_ORIGIN, sz, bd, (uint32_t)nblocks);
    else
        hipLaunchKernelGGL((upsample_edge_kernel<uint8_t>), dim3((uint32_t)nblocks), dim3(64), 0, s, (uint8_t*)d_edges, nb_pitch, NB_ORIGIN, sz, 8, (uint32_t)nblocks);
    return launch_status("upsample_intra_edge");
}

// ===========================================================================
// (A) drop-in entry points: host pointers, one block, synchronous
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

// copies a w x h u8 block with `stride` into dense device memory
static void h2d_block(void* d, const uint8_t* h, uint32_t stride, uint32_t w, uint32_t ht, const char* fn) {
    HIP_DIE(hipMemcpy2DAsync(d, w, h, stride, w, ht, hipMemcpyHostToDevice, t_ctx.stream), fn);
}

extern "C" uint32_t svt_hip_nxm_sad_kernel(const uint8_t* src, uint32_t src_stride, const uint8_t* ref,
                                           uint32_t ref_stride, uint32_t height, uint32_t width) {
    const char* fn = "svt_hip_nxm_sad_kernel";
    const size_t bb = align256((size_t)width * height);
    DROPIN_TRY(t_ctx.ensure(2 * bb + 256), fn);
    uint8_t* d_a = (uint8_t*)t_ctx.dbuf;
    uint8_t* d_b = d_a + bb;
    uint32_t* d_o = (uint32_t*)(d_a + 2 * bb);
    h2d_block(d_a, src, src_stride, width, height, fn);
    h2d_block(d_b, ref, ref_stride, width, height, fn);
    DROPIN_TRY(svt_hip_sad_batch(d_a, width, 0, d_b, width, 0, width, height, d_o, 1, t_ctx.stream), fn);
    uint32_t out = 0;
    HIP_DIE(hipMemcpyAsync(&out, d_o, 4, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
    return out;
}
extern "C" uint64_t svt_hip_spatial_full_distortion_kernel(uint8_t* input, uint32_t input_stride, uint8_t* recon,
                                                           uint32_t recon_stride, uint32_t area_width,
                                                           uint32_t area_height) {
    const char* fn = "svt_hip_spatial_full_distortion_kernel";
    const size_t bb = align256((size_t)area_width * area_height);
    DROPIN_TRY(t_ctx.ensure(2 * bb + 256), fn);
    uint8_t* d_a = (uint8_t*)t_ctx.dbuf;
    uint8_t* d_b = d_a + bb;
    uint64_t* d_o = (uint64_t*)(d_a + 2 * bb);
    h2d_block(d_a, input, input_stride, area_width, area_height, fn);
    h2d_block(d_b, recon, recon_stride, area_width, area_height, fn);
    DROPIN_TRY(svt_hip_sse_batch(d_a, area_width, 0, d_b, area_width, 0, area_width, area_height, d_o, 1, t_ctx.stream), fn);
    uint64_t out = 0;
    HIP_DIE(hipMemcpyAsync(&out, d_o, 8, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
    return out;
}
extern "C" void svt_hip_residual_kernel(uint8_t* input, uint32_t input_stride, uint8_t* pred, uint32_t pred_stride,
                                        int16_t* residual, uint32_t residual_stride, uint32_t area_width,
                                        uint32_t area_height) {
    const char* fn = "svt_hip_residual_kernel";
    const size_t bb = align256((size_t)area_width * area_height);
    DROPIN_TRY(t_ctx.ensure(4 * bb), fn);
    uint8_t* d_a = (uint8_t*)t_ctx.dbuf;
    uint8_t* d_b = d_a + bb;
    int16_t* d_r = (int16_t*)(d_a + 2 * bb);
    h2d_block(d_a, input, input_stride, area_width, area_height, fn);
    h2d_block(d_b, pred, pred_stride, area_width, area_height, fn);
    DROPIN_TRY(svt_hip_residual_batch(d_a, area_width, 0, d_b, area_width, 0, d_r, area_width, 0, area_width, area_height, 1, t_ctx.stream), fn);
    HIP_DIE(hipMemcpy2DAsync(residual, (size_t)residual_stride * 2, d_r, (size_t)area_width * 2, (size_t)area_width * 2,
                             area_height, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
extern "C" void svt_hip_sad_loop_kernel(uint8_t* src, uint32_t src_stride, uint8_t* ref, uint32_t ref_stride,
                                        uint32_t height, uint32_t width, uint64_t* best_sad, int16_t* x_search_center,
                                        int16_t* y_search_center, uint32_t src_stride_raw, int16_t search_area_width,
                                        int16_t search_area_height) {
    const char* fn = "svt_hip_sad_loop_kernel";
    // stage the touched source rows and the touched reference span as-is (strides kept)
    const size_t src_span = (size_t)(height - 1) * src_stride + width;
    const size_t ref_span = (size_t)(search_area_height - 1) * src_stride_raw + (size_t)(height - 1) * ref_stride +
                            width + search_area_width - 1;
    const size_t sb = align256(src_span), rb = align256(ref_span);
    DROPIN_TRY(t_ctx.ensure(sb + rb + 256), fn);
    uint8_t* d_s = (uint8_t*)t_ctx.dbuf;
    uint8_t* d_r = d_s + sb;
    uint64_t* d_best = (uint64_t*)(d_r + rb);
    int16_t* d_xy = (int16_t*)(d_best + 1);
    int16_t xy[2] = {*x_search_center, *y_search_center};
    HIP_DIE(hipMemcpyAsync(d_s, src, src_span, hipMemcpyHostToDevice, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(d_r, ref, ref_span, hipMemcpyHostToDevice, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(d_xy, xy, 4, hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_sad_search_batch(d_s, src_stride, 0, d_r, ref_stride, src_stride_raw, 0, width, height,
                                        search_area_width, search_area_height, d_best, d_xy, d_xy + 1, 1, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(best_sad, d_best, 8, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(xy, d_xy, 4, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
    *x_search_center = xy[0];
    *y_search_center = xy[1];
}

static void dropin_dist32(int cbf_zero, int32_t* coeff, uint32_t cs, int32_t* recon, uint32_t rs, uint64_t out[2],
                          uint32_t w, uint32_t h, const char* fn) {
    const size_t bb = align256((size_t)w * h * 4);
    DROPIN_TRY(t_ctx.ensure(2 * bb + 256), fn);
    int32_t* d_c = (int32_t*)t_ctx.dbuf;
    int32_t* d_r = (int32_t*)(t_ctx.dbuf + bb);
    uint64_t* d_o = (uint64_t*)(t_ctx.dbuf + 2 * bb);
    HIP_DIE(hipMemcpy2DAsync(d_c, (size_t)w * 4, coeff, (size_t)cs * 4, (size_t)w * 4, h, hipMemcpyHostToDevice, t_ctx.stream), fn);
    if (!cbf_zero)
        HIP_DIE(hipMemcpy2DAsync(d_r, (size_t)w * 4, recon, (size_t)rs * 4, (size_t)w * 4, h, hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_full_distortion32_batch(d_c, w, 0, d_r, w, 0, w, h, cbf_zero, d_o, 1, t_ctx.stream), fn);
    HIP_DIE(hipMemcpyAsync(out, d_o, 16, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
extern "C" void svt_hip_full_distortion_kernel32_bits(int32_t* coeff, uint32_t coeff_stride, int32_t* recon_coeff,
                                                      uint32_t recon_coeff_stride, uint64_t distortion_result[2],
                                                      uint32_t area_width, uint32_t area_height) {
    dropin_dist32(0, coeff, coeff_stride, recon_coeff, recon_coeff_stride, distortion_result, area_width, area_height,
                  "svt_hip_full_distortion_kernel32_bits");
}
extern "C" void svt_hip_full_distortion_kernel_cbf_zero32_bits(int32_t* coeff, uint32_t coeff_stride, int32_t* recon_coeff,
                                                               uint32_t recon_coeff_stride, uint64_t distortion_result[2],
                                                               uint32_t area_width, uint32_t area_height) {
    dropin_dist32(1, coeff, coeff_stride, recon_coeff, recon_coeff_stride, distortion_result, area_width, area_height,
                  "svt_hip_full_distortion_kernel_cbf_zero32_bits");
}

// ---- open-loop intra search ----
static size_t ois_nb_pitch(uint32_t bsize) { return (size_t)NB_ORIGIN + 4 * bsize + 16; }     // multiple of 16
static size_t ois_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t svt_hip_ois_work_bytes(uint32_t bsize, int ncand, size_t nblocks) {
    if ((bsize != 8 && bsize != 16 && bsize != 32 && bsize != 64) || ncand <= 0 || ncand > 61) return 0;
    return 2 * ois_align(nblocks * ois_nb_pitch(bsize)) + ois_align(nblocks) + (size_t)ncand * ois_align(nblocks * (size_t)bsize * bsize);
}

// dr_intra_derivative (AV1 spec 7.11.2.4; reference EbIntraPrediction.c:299), non-zero entries
static int ois_dr_derivative(int angle) {
    static const uint16_t at[][2] = {{3, 1023}, {6, 547}, {9, 372}, {14, 273}, {17, 215}, {20, 178}, {23, 151}, {26, 132},
                                     {29, 116}, {32, 102}, {36, 90}, {39, 80}, {42, 71}, {45, 64}, {48, 57}, {51, 51},
                                     {54, 45}, {58, 40}, {61, 35}, {64, 31}, {67, 27}, {70, 23}, {73, 19}, {76, 15},
                                     {81, 11}, {84, 7}, {87, 3}};
    for (const auto& e : at)
        if (e[0] == angle) return e[1];
    return 0;
}

extern "C" int svt_hip_ois_search_batch(const uint8_t* d_pic, uint32_t stride, uint32_t width, uint32_t height,
                                        const uint32_t* d_xy, uint32_t bsize, const uint8_t* modes, const int8_t* angle_deltas,
                                        int ncand, uint32_t* d_distortion, int8_t* d_best_index, void* d_work,
                                        size_t work_bytes, size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (nblocks == 0) return SVT_HIP_OK;
    if (!d_pic || !d_xy || !d_distortion || !d_best_index || !d_work || !modes || !angle_deltas) return set_err(SVT_HIP_ERR_INVALID, "NULL buffer");
    if (bsize != 8 && bsize != 16 && bsize != 32 && bsize != 64) return set_err(SVT_HIP_ERR_INVALID, "block size %u", bsize);
    if (ncand <= 0 || ncand > 61) return set_err(SVT_HIP_ERR_INVALID, "%d candidates (1..61, MAX_OIS_CANDIDATES)", ncand);
    if (width == 0 || height == 0 || width > 0xffffu || height > 0xffffu || stride < width) return set_err(SVT_HIP_ERR_INVALID, "picture %ux%u stride %u", width, height, stride);
    if (work_bytes < svt_hip_ois_work_bytes(bsize, ncand, nblocks)) return set_err(SVT_HIP_ERR_INVALID, "work buffer: %zu B, need %zu", work_bytes, svt_hip_ois_work_bytes(bsize, ncand, nblocks));
    if (nblocks > 0x7fffffffu / 256) return set_err(SVT_HIP_ERR_INVALID, "too many blocks for one launch");
    static const int mode_angle[13] = {0, 90, 180, 45, 135, 113, 157, 203, 67, 0, 0, 0, 0};      // mode_to_angle_map, EbCodingUnit.h:129
    for (int c = 0; c < ncand; c++) {
        if (modes[c] > 12) return set_err(SVT_HIP_ERR_INVALID, "candidate %d: prediction mode %u", c, modes[c]);
        if (modes[c] >= 1 && modes[c] <= 8) {
            const int a = mode_angle[modes[c]] + 3 * angle_deltas[c];
            if (a <= 0 || a >= 270) return set_err(SVT_HIP_ERR_INVALID, "candidate %d: angle %d", c, a);
            if (a != 90 && a != 180) {
                const int d1 = a < 90 ? a : (a < 180 ? 180 - a : 270 - a), d2 = a < 180 && a > 90 ? a - 90 : d1;
                if (!ois_dr_derivative(d1) || !ois_dr_derivative(d2)) return set_err(SVT_HIP_ERR_INVALID, "candidate %d: angle %d has no derivative", c, a);
            }
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t pitch = ois_nb_pitch(bsize);
    char* w = (char*)d_work;
    uint8_t* d_above = (uint8_t*)w;
    uint8_t* d_left = d_above + ois_align(nblocks * pitch);
    uint8_t* d_dc = d_left + ois_align(nblocks * pitch);
    uint8_t* d_pred = d_dc + ois_align(nblocks);
    if (hipMemsetAsync(d_above, 0, 2 * ois_align(nblocks * pitch), st) != hipSuccess) return set_err(SVT_HIP_ERR_RUNTIME, "hipMemsetAsync");
    {
        const uint32_t slots = 256 / (2 * bsize);
        hipLaunchKernelGGL(ois_gather_kernel, dim3((uint32_t)((nblocks + slots - 1) / slots)), dim3(256), 0, st, d_pic, stride, width,
                           height, d_xy, bsize, d_above, d_left, (uint32_t)pitch, d_dc, (uint32_t)nblocks);
        if (int rc = launch_status("ois_gather")) return rc;
    }
    // every candidate's prediction into its own dense batch, then ONE SAD launch over (block, candidate).  The
    // directional candidates of one zone (up to 19) share one launch (DirMulti): 9 prediction launches for the
    // reference's 45-candidate list instead of 44.
    const size_t cand_pitch = ois_align(nblocks * (size_t)bsize * bsize);
    unsigned long long const_mask = 0;
    DirMulti zone[3];
    // 8x8 / 16x16 (a block's lanes share a wave): the directional kernels compare each angle's prediction with the source
    // block themselves (DirMulti SAD mode) - no prediction scratch round trip for 38 of the 45 candidates
    const bool fold = bsize <= 16 && !g_tune_ois_no_fold;
    unsigned long long fold_mask = 0;
    for (auto& z : zone) {
        z.n = 0; z.batch_pitch = cand_pitch;
        z.sad_pic = d_pic; z.sad_stride = stride; z.sad_xy = d_xy; z.sad_dist = fold ? d_distortion : nullptr; z.sad_ncand = (uint32_t)ncand;
    }
    for (int c = 0; c < ncand; c++) {
        const int m = modes[c];
        if (m == 0) { const_mask |= 1ull << c; continue; }     // DC_PRED under the availability rule: constant prediction
        int mode = -1;
        if (m >= 1 && m <= 8) {                                               // dr_predictor, EbIntraPrediction.c:3352-3383
            const int a = mode_angle[m] + 3 * angle_deltas[c];
            if (a == 90) mode = SVT_INTRA_V;
            else if (a == 180) mode = SVT_INTRA_H;
            else {
                const int zi = a < 90 ? 0 : (a < 180 ? 1 : 2);
                DirMulti& z = zone[zi];
                if (z.n == 20) {                                              // flush a full group (longer candidate lists)
                    if (int rc = intra_pred_impl(d_pred, (int32_t)bsize, (size_t)bsize * bsize, nullptr, d_above, d_left, (int32_t)pitch,
                                                 SVT_INTRA_Z1 + zi, (int)bsize, (int)bsize, 0, 0, 1, 1, 0, 8, nblocks, stream, &z))
                        return rc;
                    z.n = 0;
                }
                z.dx[z.n] = (int16_t)(zi == 0 ? ois_dr_derivative(a) : (zi == 1 ? ois_dr_derivative(180 - a) : 1));
                z.dy[z.n] = (int16_t)(zi == 0 ? 1 : (zi == 1 ? ois_dr_derivative(a - 90) : ois_dr_derivative(270 - a)));
                z.slot[z.n] = (uint8_t)c;
                z.n++;
                if (fold) fold_mask |= 1ull << c;
                continue;
            }
        } else {
            mode = m == 9 ? SVT_INTRA_SMOOTH : m == 10 ? SVT_INTRA_SMOOTH_V : m == 11 ? SVT_INTRA_SMOOTH_H : SVT_INTRA_PAETH;
        }
        if (int rc = svt_hip_intra_pred_batch(d_pred + (size_t)c * cand_pitch, (int32_t)bsize, (size_t)bsize * bsize, nullptr, d_above,
                                              d_left, (int32_t)pitch, mode, (int)bsize, (int)bsize, 0, 0, 1, 1, 0, 8, nblocks, stream))
            return rc;
    }
    for (int zi = 0; zi < 3; zi++)
        if (zone[zi].n)
            if (int rc = intra_pred_impl(d_pred, (int32_t)bsize, (size_t)bsize * bsize, nullptr, d_above, d_left, (int32_t)pitch,
                                         SVT_INTRA_Z1 + zi, (int)bsize, (int)bsize, 0, 0, 1, 1, 0, 8, nblocks, stream, &zone[zi]))
                return rc;
    {
        const uint32_t lpb = bsize * bsize / (bsize < 16 ? 8 : 16);
        const uint32_t sad_slots = 256 / lpb;
        const uint32_t sad_grid = (uint32_t)((nblocks + sad_slots - 1) / sad_slots);
        const size_t shmem = ((size_t)sad_slots + (lpb > 64 ? 4 : 0)) * (size_t)ncand * sizeof(uint32_t);
        hipLaunchKernelGGL(ois_sad_kernel, dim3(sad_grid), dim3(256), shmem, st, d_pic, stride, d_xy, bsize, d_pred, cand_pitch,
                           d_dc, const_mask, fold_mask, d_distortion, d_best_index, (uint32_t)ncand, (uint32_t)nblocks);
    }
    return launch_status("ois_sad");
}

// one intra block: stage [lo, hi) of above / left around the origin, predict, copy the block back
static void dropin_intra(int mode, int bw, int bh, void* dst, ptrdiff_t stride, const void* above, const void* left,
                         int a_lo, int a_hi, int l_lo, int l_hi, int ua, int ul, int dx, int dy, int is16, int bd,
                         const char* fn) {
    const size_t es = is16 ? 2 : 1;
    const int pitch = NB_ORIGIN + 2 * (bw + bh) + 16;
    const size_t nb_b = align256((size_t)pitch * es), px_b = (size_t)bw * bh * es;
    DROPIN_TRY(t_ctx.ensure(2 * nb_b + px_b), fn);
    char* d_a = t_ctx.dbuf;
    char* d_l = t_ctx.dbuf + nb_b;
    char* d_px = t_ctx.dbuf + 2 * nb_b;
    HIP_DIE(hipMemsetAsync(d_a, 0, 2 * nb_b, t_ctx.stream), fn);
    if (a_hi > a_lo)
        HIP_DIE(hipMemcpyAsync(d_a + (size_t)(NB_ORIGIN + a_lo) * es, (const char*)above + (ptrdiff_t)a_lo * (ptrdiff_t)es,
                               (size_t)(a_hi - a_lo) * es, hipMemcpyHostToDevice, t_ctx.stream), fn);
    if (l_hi > l_lo)
        HIP_DIE(hipMemcpyAsync(d_l + (size_t)(NB_ORIGIN + l_lo) * es, (const char*)left + (ptrdiff_t)l_lo * (ptrdiff_t)es,
                               (size_t)(l_hi - l_lo) * es, hipMemcpyHostToDevice, t_ctx.stream), fn);
    DROPIN_TRY(svt_hip_intra_pred_batch(d_px, bw, (size_t)bw * bh, nullptr, d_a, d_l, pitch, mode, bw, bh, ua, ul, dx, dy,
                                        is16, bd, 1, t_ctx.stream), fn);
    HIP_DIE(hipMemcpy2DAsync(dst, (size_t)stride * es, d_px, (size_t)bw * es, (size_t)bw * es, bh, hipMemcpyDeviceToHost, t_ctx.stream), fn);
    HIP_DIE(hipStreamSynchronize(t_ctx.stream), fn);
}
extern "C" void svt_hip_intra_predictor(int mode, int bw, int bh, uint8_t* dst, ptrdiff_t stride, const uint8_t* above,
                                        const uint8_t* left) {
    dropin_intra(mode, bw, bh, dst, stride, above, left, -1, bw, 0, bh, 0, 0, 1, 1, 0, 8, "svt_hip_intra_predictor");
}
extern "C" void svt_hip_highbd_intra_predictor(int mode, int bw, int bh, uint16_t* dst, ptrdiff_t stride,
                                               const uint16_t* above, const uint16_t* left, int32_t bd) {
    dropin_intra(mode, bw, bh, dst, stride, above, left, -1, bw, 0, bh, 0, 0, 1, 1, 1, bd, "svt_hip_highbd_intra_predictor");
}
#define DR_RANGES_Z1 0, (((bw + bh - 1) << upsample_above) + 2), 0, 0
#define DR_RANGES_Z3 0, 0, 0, (((bw + bh - 1) << upsample_left) + 2)
#define DR_RANGES_Z2 -(1 << upsample_above), (((bw - 1) << upsample_above) + 2), -(1 << upsample_left), (((bh - 1) << upsample_left) + 2)
extern "C" void svt_hip_av1_dr_prediction_z1(uint8_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh, const uint8_t* above,
                                             const uint8_t* left, int32_t upsample_above, int32_t dx, int32_t dy) {
    dropin_intra(SVT_INTRA_Z1, bw, bh, dst, stride, above, left, DR_RANGES_Z1, upsample_above, 0, dx, dy, 0, 8, "svt_hip_av1_dr_prediction_z1");
}
extern "C" void svt_hip_av1_dr_prediction_z2(uint8_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh, const uint8_t* above,
                                             const uint8_t* left, int32_t upsample_above, int32_t upsample_left,
                                             int32_t dx, int32_t dy) {
    dropin_intra(SVT_INTRA_Z2, bw, bh, dst, stride, above, left, DR_RANGES_Z2, upsample_above, upsample_left, dx, dy, 0, 8, "svt_hip_av1_dr_prediction_z2");
}
extern "C" void svt_hip_av1_dr_prediction_z3(uint8_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh, const uint8_t* above,
                                             const uint8_t* left, int32_t upsample_left, int32_t dx, int32_t dy) {
    dropin_intra(SVT_INTRA_Z3, bw, bh, dst, stride, above, left, DR_RANGES_Z3, 0, upsample_left, dx, dy, 0, 8, "svt_hip_av1_dr_prediction_z3");
}
extern "C" void svt_hip_av1_highbd_dr_prediction_z1(uint16_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh,
                                                    const uint16_t* above, const uint16_t* left, int32_t upsample_above,
                                                    int32_t dx, int32_t dy, int32_t bd) {
    dropin_intra(SVT_INTRA_Z1, bw, bh, dst, stride, above, left, DR_RANGES_Z1, upsample_above, 0, dx, dy, 1, bd, "svt_hip_av1_highbd_dr_prediction_z1");
}
extern "C" void svt_hip_av1_highbd_dr_prediction_z2(uint16_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh,
                                                    const uint16_t* above, const uint16_t* left, int32_t upsample_above,
                                                    int32_t upsample_left, int32_t dx, int32_t dy, int32_t bd) {
    dropin_intra(SVT_INTRA_Z2, bw, bh, dst, stride, above, left, DR_RANGES_Z2, upsample_above, upsample_left, dx, dy, 1, bd, "svt_hip_av1_highbd_dr_prediction_z2");
}
extern "C" void svt_hip_av1_highbd_dr_prediction_z3(uint16_t* dst, ptrdiff_t stride, int32_t bw, int32_t bh,
                                                    const uint16_t* above, const uint16_t* left, int32_t upsample_left,
                                                    int32_t dx, int32_t dy, int32_t bd) {
    dropin_intra(SVT_INTRA_Z3, bw, bh, dst, stride, above, left, DR_RANGES_Z3, 0, upsample_left, dx, dy, 1, bd, "svt_hip_av1_highbd_dr_prediction_z3");
}

extern "C" int svt_hip_rtcd_override(const svt_hip_rtcd_table* t) {
    if (!t) return set_err(SVT_HIP_ERR_INVALID, "NULL table");
    if (int rc = require_init()) return rc;
    void* fwd[SVT_TX_SIZES_ALL] = {
        (void*)svt_hip_av1_fwd_txfm2d_4x4, (void*)svt_hip_av1_fwd_txfm2d_8x8, (void*)svt_hip_av1_fwd_txfm2d_16x16,
        (void*)svt_hip_av1_fwd_txfm2d_32x32, (void*)svt_hip_av1_fwd_txfm2d_64x64, (void*)svt_hip_av1_fwd_txfm2d_4x8,
        (void*)svt_hip_av1_fwd_txfm2d_8x4, (void*)svt_hip_av1_fwd_txfm2d_8x16, (void*)svt_hip_av1_fwd_txfm2d_16x8,
        (void*)svt_hip_av1_fwd_txfm2d_16x32, (void*)svt_hip_av1_fwd_txfm2d_32x16, (void*)svt_hip_av1_fwd_txfm2d_32x64,
        (void*)svt_hip_av1_fwd_txfm2d_64x32, (void*)svt_hip_av1_fwd_txfm2d_4x16, (void*)svt_hip_av1_fwd_txfm2d_16x4,
        (void*)svt_hip_av1_fwd_txfm2d_8x32, (void*)svt_hip_av1_fwd_txfm2d_32x8, (void*)svt_hip_av1_fwd_txfm2d_16x64,
        (void*)svt_hip_av1_fwd_txfm2d_64x16};
    void* inv[SVT_TX_SIZES_ALL] = {
        (void*)svt_hip_av1_inv_txfm2d_add_4x4, (void*)svt_hip_av1_inv_txfm2d_add_8x8, (void*)svt_hip_av1_inv_txfm2d_add_16x16,
        (void*)svt_hip_av1_inv_txfm2d_add_32x32, (void*)svt_hip_av1_inv_txfm2d_add_64x64, (void*)svt_hip_av1_inv_txfm2d_add_4x8,
        (void*)svt_hip_av1_inv_txfm2d_add_8x4, (void*)svt_hip_av1_inv_txfm2d_add_8x16, (void*)svt_hip_av1_inv_txfm2d_add_16x8,
        (void*)svt_hip_av1_inv_txfm2d_add_16x32, (void*)svt_hip_av1_inv_txfm2d_add_32x16, (void*)svt_hip_av1_inv_txfm2d_add_32x64,
        (void*)svt_hip_av1_inv_txfm2d_add_64x32, (void*)svt_hip_av1_inv_txfm2d_add_4x16, (void*)svt_hip_av1_inv_txfm2d_add_16x4,
        (void*)svt_hip_av1_inv_txfm2d_add_8x32, (void*)svt_hip_av1_inv_txfm2d_add_32x8, (void*)svt_hip_av1_inv_txfm2d_add_16x64,
        (void*)svt_hip_av1_inv_txfm2d_add_64x16};
    for (int i = 0; i < SVT_TX_SIZES_ALL; i++) {
        if (t->av1_fwd_txfm2d[i]) *t->av1_fwd_txfm2d[i] = fwd[i];
        if (t->av1_inv_txfm2d_add[i]) *t->av1_inv_txfm2d_add[i] = inv[i];
    }
    if (t->av1_inv_txfm_add) *t->av1_inv_txfm_add = (void*)svt_hip_av1_inv_txfm_add;
    if (t->aom_quantize_b) *t->aom_quantize_b = (void*)svt_hip_aom_quantize_b;
    if (t->aom_quantize_b_32x32) *t->aom_quantize_b_32x32 = (void*)svt_hip_aom_quantize_b_32x32;
    if (t->aom_quantize_b_64x64) *t->aom_quantize_b_64x64 = (void*)svt_hip_aom_quantize_b_64x64;
    if (t->aom_highbd_quantize_b) *t->aom_highbd_quantize_b = (void*)svt_hip_aom_highbd_quantize_b;
    if (t->aom_highbd_quantize_b_32x32) *t->aom_highbd_quantize_b_32x32 = (void*)svt_hip_aom_highbd_quantize_b_32x32;
    if (t->aom_highbd_quantize_b_64x64) *t->aom_highbd_quantize_b_64x64 = (void*)svt_hip_aom_highbd_quantize_b_64x64;
    if (t->ResidualKernel) *t->ResidualKernel = (void*)svt_hip_residual_kernel;
    return SVT_HIP_OK;
}
