// host_common.h — host-side state and helpers shared by the translation units of libsvt_hip_dsp.so
// (svt_hip_core.hip defines them; svt_hip_txfm.hip / svt_hip_pixel.hip / svt_hip_intra.hip hold the entry points of one
// kernel family each, so that the library builds as four parallel hipcc jobs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "../../include/svt_hip_dsp.h"
#include "host_err.h"

namespace svthost {


extern std::atomic<int> g_inited;
extern std::mutex g_init_mu;
extern int g_device;
extern char g_devname[300];
extern int g_num_cu;
// tuning knobs (svt_hip_tune): fused 32x32 kernel occupancy / grid
extern int g_tune_f32_wg_per_cu;
extern int g_tune_f32_nt;
extern int g_tune_f32_qmode1;
extern int g_tune_no_staged;
extern int g_tune_no_qsad;
extern int g_tune_no_q2;
extern int g_tune_no_q2p;
extern int g_tune_no_q16;
extern int g_tune_q2_su4;
extern int g_tune_ois_no_fold;
extern int g_tune_ois_no_dir3;
extern int g_tune_ois_no_nd_multi;
extern int g_tune_ois_no_nd;
extern int g_tune_dir_no_split, g_tune_dir_split_target;
extern int g_tune_me_exact;
extern int g_tune_no_f32p;
extern int g_tune_no_inv_planes;
extern int g_tune_no_enc_staged;
extern int g_tune_no_enc64;
extern int g_tune_frame_single_launch;
extern int g_tune_inv32_waves;
extern int g_tune_inv32_var;

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return set_err(SVT_HIP_ERR_RUNTIME, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                                  \
    } while (0)

int require_init();
int launch_status(const char* what);
}  // namespace svthost
namespace svtdev { struct FrameDesc; }
namespace svthost {
// svt_hip_frame.hip: the one-launch form of svt_hip_encode_recon_frame (its kernel lives in a translation unit of its own)
int launch_enc_frame_one(const svtdev::FrameDesc* fd, uint32_t total_wgs, int is_16bit, hipStream_t s);

extern const int kTxW[SVT_TX_SIZES_ALL];
extern const int kTxH[SVT_TX_SIZES_ALL];
bool txfm_allowed(int tx_size, int tx_type);
int frame_groups_check(const svt_hip_frame_group* groups, int ngroups);      // svt_hip_txfm.hip

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
extern thread_local ThreadCtx t_ctx;

[[noreturn]] void die(const char* fn);
#define DROPIN_TRY(expr, fn) do { if ((expr) != SVT_HIP_OK) die(fn); } while (0)
#define HIP_DIE(expr, fn)                                                          \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) { set_err(SVT_HIP_ERR_RUNTIME, "%s: %s", #expr, hipGetErrorString(e_)); die(fn); } \
    } while (0)

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }


// Internal streams a frame-level call fans its independent groups out to (forked from, and joined back into, the caller's
// stream: the call stays a pure enqueue and is graph-capturable).  One set per calling thread.
constexpr int kFanStreams = 8;
struct FanOut {
    hipStream_t s[kFanStreams] = {};
    hipEvent_t fork = nullptr, join[kFanStreams] = {};
    bool ready = false;
    int ensure() {
        if (ready) return SVT_HIP_OK;
        HIP_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        for (int i = 0; i < kFanStreams; i++) {
            // (plain streams: creating stream 0 with hipStreamCreateWithPriority(.., greatest) made svt_hip_ois_search_frame FOUR times
            // slower on this runtime - 0.55 against 0.13 ms per 1080p picture, A/B on one box)
            HIP_TRY(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&join[i], hipEventDisableTiming));
        }
        ready = true;
        return SVT_HIP_OK;
    }
    ~FanOut() {
        if (!ready) return;
        for (int i = 0; i < kFanStreams; i++) { (void)hipStreamDestroy(s[i]); (void)hipEventDestroy(join[i]); }
        (void)hipEventDestroy(fork);
    }
};
inline thread_local FanOut t_fan;

}  // namespace svthost
