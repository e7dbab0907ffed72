// dev_common.h — device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svtdev {

// Quantizer scalars for one launch: index 0 = DC (coefficient 0), 1 = AC.
// Derived on the host from the reference's five int16[8] tables
// (EbFullLoop.c:239-296): zbin/round already ROUND_POWER_OF_TWO'ed by
// log_scale; quant_m = quant + 65536 (so ((t*quant)>>16)+t == (t*quant_m)>>16).
struct QParams {
    int32_t zbin[2];
    int32_t round[2];
    uint32_t quant_m[2];
    int32_t quant_shift[2];
    int32_t dequant[2];
    int32_t log_scale;
    // POW2 fast path (host-checked): quant_shift = 2^k, so the two-step reference
    // formula collapses to  aq = (t1 * quant_m) >> fast_sh  with fast_sh = 32 - log_scale - k
    int32_t fast_sh[2];
    int32_t fast_ok;
    // the same collapse as ONE multiply: with quant_hi = quant_m << (31 - fast_sh) (fits 32 bits: quant_m < 2^17, fast_sh >= 16),
    // aq = mulhi_u32(2 * t1, quant_hi) = floor(t1 * quant_m / 2^fast_sh) exactly; zbin2 / round2 are the doubled thresholds
    uint32_t quant_hi[2];
    int32_t zbin2[2];
    int32_t round2[2];
};

// Orders this wave's LDS traffic: LDS instructions of one wave execute in
// issue order, so a compiler-level fence is all that is needed between a
// phase that writes a tile and a phase in which OTHER lanes of the same wave
// read it.  (Tiles are never shared between waves.)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// (a * b) >> sh for a, b < 2^24, 0 < sh < 32, result < 2^32: two full-rate
// 24-bit multiplies + v_alignbit instead of a 64-bit multiply.
__device__ __forceinline__ uint32_t mul24_shr(uint32_t a, uint32_t b, int sh) {
    const uint64_t p = (uint64_t)(a & 0xffffffu) * (uint64_t)(b & 0xffffffu);
    return (uint32_t)(p >> sh);
}

// One coefficient through highbd_quantize_b_helper_c (EbFullLoop.c:239-296), flat quant
// matrix.  Reference:  tw = (|c| + round) * 32;  t2 = ((tw * quant) >> 16) + tw;
//                      aq = (t2 * quant_shift) >> (21 - log_scale);  adq = (aq * dequant) >> log_scale
// MODE 0: exact 64-bit arithmetic (any table, any coefficient).
// MODE 1: 24-bit multiplies, operands proven < 2^24 (8-bit transform output).
// MODE 2: MODE 1 + quant_shift is a power of two (true for every table av1_build_quantizer
//         emits: invert_quant, EbModeDecisionConfigurationProcess.c:322-330):
//         t2 = floor(t1 * m / 2^11) with m = quant + 65536, so aq = floor(t1 * m / 2^fast_sh) —
//         one 48-bit product (v_mul_u32_u24 + v_mul_hi_u32_u24) and one v_alignbit.
template <int MODE>
__device__ __forceinline__ void quant_one(int c, int ac, const QParams& qp, int& q, int& dq) {
    const int s = c >> 31;
    const int a = (c ^ s) - s;
    int aq, adq;
    if (MODE == 2) {
        // (v_mul_u32_u24 + v_mul_hi_u32_u24 + v_alignbit before: three slow-rate instructions; now one add and one v_mul_hi_u32)
        const int a2 = a + a;
        aq = (int)__umulhi((uint32_t)(a2 + qp.round2[ac]), qp.quant_hi[ac]);
        aq = a2 >= qp.zbin2[ac] ? aq : 0;
        adq = (int)(((uint32_t)aq & 0xffffffu) * ((uint32_t)qp.dequant[ac] & 0xffffffu)) >> qp.log_scale;
        q = (aq ^ s) - s;
        dq = (adq ^ s) - s;
        return;
    }
    const bool keep = a >= qp.zbin[ac];
    if (MODE == 1) {
        const uint32_t tw = (uint32_t)(a + qp.round[ac]) << 5;
        const uint32_t t2 = mul24_shr(tw, qp.quant_m[ac], 16);
        aq = (int)mul24_shr(t2, (uint32_t)qp.quant_shift[ac], 21 - qp.log_scale);
        adq = (int)(((uint32_t)aq & 0xffffffu) * ((uint32_t)qp.dequant[ac] & 0xffffffu)) >> qp.log_scale;
    } else {
        const long long tw = ((long long)a + qp.round[ac]) * 32;
        const long long t2 = ((tw * (long long)((int)qp.quant_m[ac] - 65536)) >> 16) + tw;
        aq = (int)((t2 * (long long)qp.quant_shift[ac]) >> (21 - qp.log_scale));
        adq = (int)((uint32_t)aq * (uint32_t)qp.dequant[ac]) >> qp.log_scale;
    }
    q = keep ? (aq ^ s) - s : 0;
    dq = keep ? (adq ^ s) - s : 0;
}

// max / sum over the 32 lanes of each half-wave (lanes 0-31 and 32-63 reduce
// independently); result valid in every lane of the half.
__device__ __forceinline__ int half_wave_max(int v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ unsigned half_wave_sum(unsigned v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// reductions over an aligned group of G lanes (G power of two <= 64)
template <int G>
__device__ __forceinline__ int group_max(int v) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return v;
}
template <int G>
__device__ __forceinline__ unsigned group_sum(unsigned v) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// sum / minimum over an aligned group of `lanes` lanes (a wave-uniform power of two, 4 .. 64), valid in every lane of the group.
// The steps inside a row of 16 lanes are DPP operands of the add itself (quad permutes, then the mirrors: lane i <-> 7 - i and
// i <-> 15 - i pair up what the previous steps left equal); only the two steps across rows go through ds_bpermute.  A runtime
// loop of __shfl_xor is one LDS round trip per step.
__device__ __forceinline__ uint32_t group_sum_rt(uint32_t v, uint32_t lanes) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
    if (lanes >= 8) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);      // row_half_mirror
    if (lanes >= 16) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);     // row_mirror
    if (lanes >= 32) v += (uint32_t)__shfl_xor((int)v, 16, 64);
    if (lanes >= 64) v += (uint32_t)__shfl_xor((int)v, 32, 64);
    return v;
}
__device__ __forceinline__ uint32_t group_min_rt(uint32_t v, uint32_t lanes) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, true));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, true));
    if (lanes >= 8) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, true));
    if (lanes >= 16) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, true));
    if (lanes >= 32) v = min(v, (uint32_t)__shfl_xor((int)v, 16, 64));
    if (lanes >= 64) v = min(v, (uint32_t)__shfl_xor((int)v, 32, 64));
    return v;
}
template <int G>
__device__ __forceinline__ unsigned long long group_sum64(unsigned long long v) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// packed 16-bit helpers of the reconstruction stages (v_pk_add_i16, v_pk_max/min_i16, v_sat_pk_u8_i16)
typedef short svt_v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add_i16(uint32_t a, uint32_t b) {
    union { uint32_t u; svt_v2s v; } x, y, z; x.u = a; y.u = b; z.v = x.v + y.v; return z.u;
}
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) {
    union { uint32_t u; svt_v2s v; } x, y, z; x.u = a; y.u = b; z.v = x.v - y.v; return z.u;
}
__device__ __forceinline__ uint32_t pk_shl2_i16(uint32_t a) {                // both lanes << 2
    union { uint32_t u; svt_v2s v; } x, z; x.u = a; z.v = x.v << 2; return z.u;
}
__device__ __forceinline__ uint32_t pk_clamp_i16(uint32_t a, int hi) {     // lanes clamped to [0, hi]
    union { uint32_t u; svt_v2s v; } x, z; x.u = a;
    const svt_v2s zero = {0, 0}, top = {(short)hi, (short)hi};
    z.v = __builtin_elementwise_min(__builtin_elementwise_max(x.v, zero), top);
    return z.u;
}
__device__ __forceinline__ uint32_t sat_pk_u8_i16(uint32_t a) {             // {sat_u8(a.lo), sat_u8(a.hi)} in bits 15:0
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(a));
    return r;
}

}  // namespace svtdev
