// svt_hip_core.hip — library state, initialisation, tuning knobs, device-memory helpers and the RTCD override table of
// libsvt_hip_dsp.so (C ABI: include/svt_hip_dsp.h).  The only kernels here are the three bandwidth probes of svt_hip_membw_probe.
#include "host_common.h"

namespace svthost {


std::atomic<int> g_inited{0};
std::mutex g_init_mu;
int g_device = -1;
char g_devname[300] = "";
int g_num_cu = 256;
// tuning knobs (svt_hip_tune): fused 32x32 kernel occupancy / grid
int g_tune_f32_wg_per_cu = 0;
int g_tune_f32_nt = 0;
int g_tune_f32_qmode1 = 0;
int g_tune_no_staged = 0;
int g_tune_no_qsad = 0;
int g_tune_no_q2 = 0;
int g_tune_no_q2p = 0;
int g_tune_no_q16 = 0;
int g_tune_q2_su4 = 0;
int g_tune_ois_no_fold = 0;
int g_tune_ois_no_dir3 = 0;
int g_tune_ois_no_nd_multi = 0;
int g_tune_ois_no_nd = 0;
int g_tune_dir_no_split = 0;
int g_tune_dir_split_target = 4;         // workgroups per CU below which the directional kernels spread their angles over grid.y (swept 1 .. 16 on the 1080p search: 3-4 is the minimum for 8x8 and 16x16)
int g_tune_me_exact = 0;
int g_tune_no_f32p = 0;
int g_tune_no_inv_planes = 0;
int g_tune_no_enc_staged = 0;
int g_tune_no_enc64 = 0;
int g_tune_frame_single_launch = -1;      // -1: by call size (svt_hip_encode_recon_frame), 0 never, 1 always
int g_tune_inv32_waves = 4;
int g_tune_inv32_var = 0;

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


thread_local ThreadCtx t_ctx;

[[noreturn]] void die(const char* fn) {
    fprintf(stderr, "libsvt_hip_dsp: %s: %s — no CPU fallback exists; aborting\n", fn, g_err);
    abort();
}

}  // namespace svthost
using namespace svthost;

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
extern "C" int svt_hip_tune(const char* key, int value) {
    if (!key) return SVT_HIP_ERR_INVALID;
    if (!strcmp(key, "f32_wg_per_cu")) { g_tune_f32_wg_per_cu = value; return SVT_HIP_OK; }
    if (!strcmp(key, "f32_nt")) { g_tune_f32_nt = value; return SVT_HIP_OK; }
    if (!strcmp(key, "f32_qmode1")) { g_tune_f32_qmode1 = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_staged")) { g_tune_no_staged = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_qsad")) { g_tune_no_qsad = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_q2")) { g_tune_no_q2 = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_q2p")) { g_tune_no_q2p = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_q16")) { g_tune_no_q16 = value; return SVT_HIP_OK; }
    if (!strcmp(key, "q2_su4")) { g_tune_q2_su4 = value; return SVT_HIP_OK; }
    if (!strcmp(key, "ois_no_fold")) { g_tune_ois_no_fold = value; return SVT_HIP_OK; }
    if (!strcmp(key, "ois_no_dir3")) { g_tune_ois_no_dir3 = value; return SVT_HIP_OK; }
    if (!strcmp(key, "ois_no_nd_multi")) { g_tune_ois_no_nd_multi = value; return SVT_HIP_OK; }
    if (!strcmp(key, "ois_no_nd")) { g_tune_ois_no_nd = value; return SVT_HIP_OK; }
    if (!strcmp(key, "dir_no_split")) { g_tune_dir_no_split = value; return SVT_HIP_OK; }
    if (!strcmp(key, "dir_split_target")) { g_tune_dir_split_target = value > 0 ? value : 1; return SVT_HIP_OK; }
    if (!strcmp(key, "me_exact")) { g_tune_me_exact = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_f32p")) { g_tune_no_f32p = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_inv_planes")) { g_tune_no_inv_planes = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_enc_staged")) { g_tune_no_enc_staged = value; return SVT_HIP_OK; }
    if (!strcmp(key, "no_enc64")) { g_tune_no_enc64 = value; return SVT_HIP_OK; }
    if (!strcmp(key, "frame_single_launch")) { g_tune_frame_single_launch = value; return SVT_HIP_OK; }
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

// ---- box calibration: one 16-byte access per lane, grid as large as the job (the store shape of DESIGN 4.0) -----------
typedef int v4i_probe __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void membw_fill_kernel(v4i_probe* __restrict__ dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(v4i_probe{(int)i, 1, 2, 3}, &dst[i]);
}
__global__ __launch_bounds__(256) void membw_copy_kernel(v4i_probe* __restrict__ dst, const v4i_probe* __restrict__ src, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}
// 1 byte read : 6 bytes written (the fused 32x32 kernel: 2 KiB in, 12 KiB out per block) in the store shape that fills fastest - ONE
// 16-byte store per lane, a grid as large as the job; every sixth workgroup also loads the 4 KiB it then stores.  (The
// first form - a lane loaded once and stored six times - measured 5.2 TB/s on boxes where the fused kernel itself moves 6.5: six
// stores per lane is the slow shape, so that probe was no ceiling.)
__global__ __launch_bounds__(256) void membw_mix_kernel(v4i_probe* __restrict__ dst, const v4i_probe* __restrict__ src, size_t n16_written) {
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= n16_written) return;
    v4i_probe v = {(int)w, 1, 2, 3};
    if (blockIdx.x % 6 == 0) v = __builtin_nontemporal_load(&src[(size_t)(blockIdx.x / 6) * 256 + threadIdx.x]);
    __builtin_nontemporal_store(v, &dst[w]);
}
// The fused 32x32 chain's traffic and nothing else: per block two 1 KiB reads and three 4 KiB runs written as 1 KiB stores of
// one wave, into the caller's three arrays - what THIS memory system gives that stream pattern in THIS placement.
__global__ __launch_bounds__(256) void membw_chain_kernel(const v4i_probe* __restrict__ in0, const v4i_probe* __restrict__ in1, v4i_probe* __restrict__ o0,
                                                          v4i_probe* __restrict__ o1, v4i_probe* __restrict__ o2, size_t nblocks) {
    const size_t b = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave per block
    if (b >= nblocks) return;
    const int lane = threadIdx.x & 63;
    v4i_probe a = __builtin_nontemporal_load(&in0[b * 64 + lane]);
    const v4i_probe c = __builtin_nontemporal_load(&in1[b * 64 + lane]);
    a += c;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        a.x += q;
        __builtin_nontemporal_store(a, &o0[b * 256 + q * 64 + lane]);
        __builtin_nontemporal_store(a, &o1[b * 256 + q * 64 + lane]);
        __builtin_nontemporal_store(a, &o2[b * 256 + q * 64 + lane]);
    }
}
extern "C" int svt_hip_membw_probe_chain(const void* d_in0, const void* d_in1, void* d_out0, void* d_out1, void* d_out2, size_t nblocks, void* stream) {
    if (int rc = require_init()) return rc;
    if (!d_in0 || !d_in1 || !d_out0 || !d_out1 || !d_out2 || (((uintptr_t)d_in0 | (uintptr_t)d_in1 | (uintptr_t)d_out0 | (uintptr_t)d_out1 | (uintptr_t)d_out2) & 15))
        return set_err(SVT_HIP_ERR_INVALID, "membw_probe_chain: bad arguments");
    if (!nblocks) return SVT_HIP_OK;
    const size_t grid = (nblocks + 3) / 4;
    if (grid > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "membw_probe_chain: too many blocks");
    membw_chain_kernel<<<dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream>>>((const v4i_probe*)d_in0, (const v4i_probe*)d_in1, (v4i_probe*)d_out0,
                                                                                   (v4i_probe*)d_out1, (v4i_probe*)d_out2, nblocks);
    return launch_status("membw_probe_chain");
}

extern "C" int svt_hip_malloc_spread(const size_t* bytes, int n, size_t gap_bytes, void** ptrs) {
    if (int rc = require_init()) return rc;
    if (!bytes || !ptrs || n < 0 || n > 64) return set_err(SVT_HIP_ERR_INVALID, "malloc_spread: bad arguments");
    const size_t gap = gap_bytes ? gap_bytes : ((size_t)32 << 30);
    void* spacer[64];
    int nsp = 0, rc = SVT_HIP_OK;
    for (int i = 0; i < n; i++) ptrs[i] = nullptr;
    for (int i = 0; i < n && rc == SVT_HIP_OK; i++) {
        if (bytes[i] && hipMalloc(&ptrs[i], bytes[i]) != hipSuccess) { ptrs[i] = nullptr; rc = set_err(SVT_HIP_ERR_RUNTIME, "hipMalloc of %zu bytes failed", bytes[i]); break; }
        if (i + 1 < n) {
            void* sp = nullptr;
            if (hipMalloc(&sp, gap) == hipSuccess) spacer[nsp++] = sp;
            else (void)hipGetLastError();                  // no room for a spacer: the next buffer follows directly
        }
    }
    for (int i = 0; i < nsp; i++) (void)hipFree(spacer[i]);
    if (rc != SVT_HIP_OK)
        for (int i = 0; i < n; i++) if (ptrs[i]) { (void)hipFree(ptrs[i]); ptrs[i] = nullptr; }
    return rc;
}

extern "C" int svt_hip_membw_probe(int mode, void* dst, const void* src, size_t bytes, void* stream) {
    if (int rc = require_init()) return rc;
    if (!dst || (mode != 0 && !src) || mode < 0 || mode > 2 || (bytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15))
        return set_err(SVT_HIP_ERR_INVALID, "membw_probe: bad arguments");
    if (mode == 2 && (bytes & 4095)) return set_err(SVT_HIP_ERR_INVALID, "membw_probe: mode 2 takes a multiple of 4096 bytes");
    const size_t n16 = mode == 2 ? 6 * (bytes >> 4) : bytes >> 4;          // mode 2: `bytes` are read, six times as many written
    if (!n16) return SVT_HIP_OK;
    const size_t grid = (n16 + 255) / 256;
    if (grid > 0x7fffffffu) return set_err(SVT_HIP_ERR_INVALID, "membw_probe: more than 2^31 workgroups");
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) membw_fill_kernel<<<dim3((unsigned)grid), dim3(256), 0, s>>>((v4i_probe*)dst, n16);
    else if (mode == 1) membw_copy_kernel<<<dim3((unsigned)grid), dim3(256), 0, s>>>((v4i_probe*)dst, (const v4i_probe*)src, n16);
    else membw_mix_kernel<<<dim3((unsigned)grid), dim3(256), 0, s>>>((v4i_probe*)dst, (const v4i_probe*)src, n16);
    return launch_status("membw_probe");
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


// ---- slot registry: reference RTCD global name -> drop-in of the same signature --------------------------------------
namespace {
struct Slot { const char* name; void* fn; };
#define SLOT_FWD(A, B, W, H) {"av1_fwd_txfm2d_" #W "x" #H, (void*)svt_hip_av1_fwd_txfm2d_##W##x##H},
#define SLOT_INV(A, B, W, H) {"av1_inv_txfm2d_add_" #W "x" #H, (void*)svt_hip_av1_inv_txfm2d_add_##W##x##H},
#define SLOT_PRED(mode, MODE, W, H)                                                           \
    {"aom_" #mode "_predictor_" #W "x" #H, (void*)svt_hip_aom_##mode##_predictor_##W##x##H}, \
    {"aom_highbd_" #mode "_predictor_" #W "x" #H, (void*)svt_hip_aom_highbd_##mode##_predictor_##W##x##H},
#define SLOT_SAD(W, H) {"aom_sad" #W "x" #H, (void*)svt_hip_aom_sad##W##x##H}, {"aom_sad" #W "x" #H "x4d", (void*)svt_hip_aom_sad##W##x##H##x4d},
const Slot kSlots[] = {
    SVT_HIP_BLOCK_SIZES_2(SLOT_FWD, 0, 0)
    SVT_HIP_BLOCK_SIZES_2(SLOT_INV, 0, 0)
    {"av1_inv_txfm_add", (void*)svt_hip_av1_inv_txfm_add},
    {"aom_quantize_b", (void*)svt_hip_aom_quantize_b},
    {"aom_quantize_b_32x32", (void*)svt_hip_aom_quantize_b_32x32},
    {"aom_quantize_b_64x64", (void*)svt_hip_aom_quantize_b_64x64},
    {"aom_highbd_quantize_b", (void*)svt_hip_aom_highbd_quantize_b},
    {"aom_highbd_quantize_b_32x32", (void*)svt_hip_aom_highbd_quantize_b_32x32},
    {"aom_highbd_quantize_b_64x64", (void*)svt_hip_aom_highbd_quantize_b_64x64},
    {"ResidualKernel", (void*)svt_hip_residual_kernel},
    SVT_HIP_INTRA_MODES(SVT_HIP_BLOCK_SIZES_2, SLOT_PRED)
    {"eb_smooth_v_predictor", (void*)svt_hip_eb_smooth_v_predictor},
    {"eb_smooth_h_predictor", (void*)svt_hip_eb_smooth_h_predictor},
    {"av1_dr_prediction_z1", (void*)svt_hip_av1_dr_prediction_z1},
    {"av1_dr_prediction_z2", (void*)svt_hip_av1_dr_prediction_z2},
    {"av1_dr_prediction_z3", (void*)svt_hip_av1_dr_prediction_z3},
    {"av1_highbd_dr_prediction_z1", (void*)svt_hip_av1_highbd_dr_prediction_z1},
    {"av1_highbd_dr_prediction_z2", (void*)svt_hip_av1_highbd_dr_prediction_z2},
    {"av1_highbd_dr_prediction_z3", (void*)svt_hip_av1_highbd_dr_prediction_z3},
    {"av1_filter_intra_edge", (void*)svt_hip_av1_filter_intra_edge},
    {"av1_filter_intra_edge_high", (void*)svt_hip_av1_filter_intra_edge_high},
    {"av1_upsample_intra_edge", (void*)svt_hip_av1_upsample_intra_edge},
    {"av1_upsample_intra_edge_high", (void*)svt_hip_av1_upsample_intra_edge_high},
    {"subtract_average", (void*)svt_hip_subtract_average},
    {"cfl_predict_lbd", (void*)svt_hip_cfl_predict_lbd},
    {"cfl_predict_hbd", (void*)svt_hip_cfl_predict_hbd},
    {"av1_txb_init_levels", (void*)svt_hip_av1_txb_init_levels},
    SVT_HIP_SAD_SIZES(SLOT_SAD)
};
#undef SLOT_FWD
#undef SLOT_INV
#undef SLOT_PRED
#undef SLOT_SAD
constexpr int kNumSlots = (int)(sizeof(kSlots) / sizeof(kSlots[0]));
}  // namespace

extern "C" int svt_hip_rtcd_slot_count(void) { return kNumSlots; }
extern "C" const char* svt_hip_rtcd_slot_name(int index) { return index >= 0 && index < kNumSlots ? kSlots[index].name : nullptr; }
extern "C" void* svt_hip_rtcd_slot_function(const char* name) {
    if (!name) return nullptr;
    for (int i = 0; i < kNumSlots; i++)
        if (!strcmp(kSlots[i].name, name)) return kSlots[i].fn;
    return nullptr;
}
extern "C" int svt_hip_rtcd_override_slot(const char* name, void** slot) {
    if (!name || !slot) return set_err(SVT_HIP_ERR_INVALID, "NULL argument");
    void* fn = svt_hip_rtcd_slot_function(name);
    if (!fn) return set_err(SVT_HIP_ERR_INVALID, "no drop-in for slot '%s'", name);
    if (int rc = require_init()) return rc;      // a slot is only handed out when the device is usable: there is no CPU fallback
    *slot = fn;
    return SVT_HIP_OK;
}
