// valu_ops.hip — issue cost table of integer VALU instructions on gfx950 (wave-instr per clock per SIMD),
// tight loop of 16 independent chains, 8 waves/SIMD and 4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define OPS(X) \
    X(0, "v_add_u32 %0, %1, %2", 2) \
    X(1, "v_add_u32 %0, %4, %1", 4) \
    X(2, "v_add_u32 %0, 0x800, %1", 1) \
    X(3, "v_add_u32 %0, 8, %1", 1) \
    X(4, "v_lshlrev_b32 %0, 3, %1", 1) \
    X(5, "v_lshrrev_b32 %0, 3, %1", 1) \
    X(6, "v_lshlrev_b32 %0, %2, %1", 2) \
    X(7, "v_and_b32 %0, %1, %2", 2) \
    X(8, "v_and_b32 %0, 0xffff, %1", 1) \
    X(9, "v_or_b32 %0, %1, %2", 2) \
    X(10, "v_min_f32 %0, %1, %2", 2) \
    X(11, "v_max_f32 %0, %1, %2", 2) \
    X(12, "v_add_f32 %0, %1, %2", 2) \
    X(13, "v_add_f32 %0, %1, %2 clamp", 2) \
    X(14, "v_med3_f32 %0, %1, %2, %3", 3) \
    X(15, "v_perm_b32 %0, %1, %2, %3", 3) \
    X(16, "v_pack_b32_f16 %0, %1, %2", 2) \
    X(17, "v_bfi_b32 %0, %1, %2, %3", 3) \
    X(18, "v_and_or_b32 %0, %1, %2, %3", 3) \
    X(19, "v_lshl_or_b32 %0, %1, 16, %2", 2) \
    X(20, "v_qsad_pk_u16_u8 %0, %0, %1, %0", 21) \
    X(21, "v_mqsad_pk_u16_u8 %0, %0, %1, %0", 21) \
    X(22, "v_sad_u16 %0, %1, %2, %3", 3) \
    X(23, "v_msad_u8 %0, %1, %2, %3", 3) \
    X(24, "v_cndmask_b32 %0, %1, %2, %4", 24) \
    X(25, "v_cmp_lt_i32 vcc, %1, %2", 2) \
    X(26, "v_subrev_u32 %0, %1, %2", 2) \
    X(27, "v_mul_u32_u24 %0, 0xb50, %1", 1) \
    X(28, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", 1) \
    X(29, "v_add_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 2) \
    X(30, "v_max_u32 %0, %1, %2", 2) \
    X(31, "v_min_u32 %0, %1, %2", 2) \
    X(32, "v_max_i16 %0, %1, %2", 2) \
    X(33, "v_cvt_f32_ubyte0 %0, %1", 1) \
    X(34, "v_fmac_f32 %0, %1, %2", 2) \
    X(35, "v_mad_u32_u16 %0, %1, %2, %3", 3) \
    X(36, "v_sub_u32 %0, %1, %2", 2) \
    X(37, "v_not_b32 %0, %1", 1) \
    X(38, "v_bfe_u32 %0, %1, 8, 8", 1) \
    X(39, "v_ashrrev_i32 %0, %2, %1", 2) \
    X(40, "v_add_co_u32 %0, vcc, %1, %2", 2) \
    X(41, "v_xad_u32 %0, %1, %2, %3", 3) \
    X(42, "v_add_u32 %0, %1, %2\n v_ashrrev_i32 %0, 12, %0", 2) \
    X(43, "v_sat_pk_u8_i16 %0, %1", 1)

// kind: number = how operands are bound (see op1)
template <int OP> __device__ __forceinline__ void op1(int& a, int b, int c, int sb, int sc, long long& w, long long w2) {
    const long long w2s = __builtin_amdgcn_readfirstlane((int)w2) | ((long long)__builtin_amdgcn_readfirstlane((int)(w2 >> 32)) << 32);
#define X(id, str, kind) \
    if (OP == id) { \
        if constexpr (kind == 24) asm volatile(str : "=v"(a) : "v"(a), "v"(b), "v"(c), "s"(w2s)); \
        else if constexpr (kind <= 6) asm volatile(str : "=v"(a) : "v"(a), "v"(b), "v"(c), "s"(sb), "s"(sc) : "vcc"); \
        else asm volatile(str : "+v"(w) : "v"(a), "v"(b), "v"(c), "v"(w2), "v"(w2) : "vcc"); \
    }
    OPS(X)
#undef X
}

template <int OP>
__global__ __launch_bounds__(256) void k_loop(int* out, int b, int c, int iters) {
    int a[16]; long long w[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { a[j] = threadIdx.x + j; w[j] = threadIdx.x * 3 + j; }
    const int vb = b + (threadIdx.x & 1), vc = c + (threadIdx.x & 3);
    const long long w2 = ((long long)vb << 32) | (unsigned)vc;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) op1<OP>(a[j], vb, vc, b, c, w[j], w2);
    }
    int s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) s ^= a[j] ^ (int)w[j] ^ (int)(w[j] >> 32);
    if (s == 0x7fffffff) out[0] = s;
}

static int* d_out;
template <int OP> static float run(int wpc, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_loop<OP>), dim3(256 * wpc), dim3(256), 0, 0, d_out, 3, 5, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_loop<OP>), dim3(256 * wpc), dim3(256), 0, 0, d_out, 3, 5, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 3;
}

int main() {
    CK(hipMalloc(&d_out, 64));
    const int iters = 2048;
    // clock estimate: v_add_u32 assumed 2 cycles/wave-instr at 8 waves/SIMD
    const float tadd = run<0>(8, iters);
    const double instr8 = 8.0 * 16 * iters;           // wave-instr per SIMD
    const double ghz = instr8 * 2 / (tadd * 1e-3) / 1e9;
    printf("v_add_u32 reference: %.3f ms -> clock estimate %.2f GHz if it issues every 2 cycles\n", tadd, ghz);
#define X(id, str, kind) { const float t8 = run<id>(8, iters), t4 = run<id>(4, iters); \
        printf("%-72s  %5.2f cyc/instr @8w  %5.2f @4w  (relative to v_add_u32 = 2)\n", str, 2.0 * t8 / tadd, 2.0 * (t4 * 2) / tadd); }
    OPS(X)
#undef X
    return 0;
}
