// valu_probe.hip — what limits a VALU-dense straight-line kernel on gfx950?
// Measures wave-instructions/s for (a) a tight loop and (b) straight-line code executed once per wave,
// for 4-byte (VOP2) and 8-byte (VOP3) encodings.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_probe valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int OP> __device__ __forceinline__ int op1(int a, int b, int c) {
    int r;
    if (OP == 0) asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));                 // VOP2, 4 B
    else if (OP == 1) asm volatile("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));        // VOP2, 4 B
    else if (OP == 2) asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // VOP3, 8 B
    else asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));            // VOP3, 8 B
    return r;
}

template <int OP, int NACC>
__global__ __launch_bounds__(256) void k_loop(int* out, int b, int c, int iters) {
    int a[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) a[j] = threadIdx.x + j;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) a[j] = op1<OP>(a[j], b, c);
    }
    int s = 0;
#pragma unroll
    for (int j = 0; j < NACC; j++) s ^= a[j];
    if (s == 0x7fffffff) out[0] = s;
}

// straight-line: REP x NACC instructions, no loop; b/c vary per step so nothing folds
template <int OP, int NACC, int REP>
__global__ __launch_bounds__(256) void k_straight(int* out, const int* __restrict__ kb, int c) {
    int a[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) a[j] = threadIdx.x + j;
    const int b = kb[0];
#pragma unroll
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) a[j] = op1<OP>(a[j], b + r, c + j);   // inline constants differ
    }
    int s = 0;
#pragma unroll
    for (int j = 0; j < NACC; j++) s ^= a[j];
    if (s == 0x7fffffff) out[0] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static float timeit(void (*launch)(void), int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

static int* d_out; static int* d_kb;
static int g_grid, g_iters;
template <int OP> static void l_loop() { hipLaunchKernelGGL((k_loop<OP, 16>), dim3(g_grid), dim3(256), 0, 0, d_out, 3, 5, g_iters); }
template <int OP> static void l_str() { hipLaunchKernelGGL((k_straight<OP, 16, 96>), dim3(g_grid), dim3(256), 0, 0, d_out, d_kb, 5); }

int main() {
    CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_kb, 64)); CK(hipMemset(d_kb, 0, 64));
    const double simds = 256.0 * 4;
    const char* names[4] = {"v_add_u32(4B)", "v_mul_i32_i24(4B)", "v_mad_i32_i24(8B)", "v_med3_i32(8B)"};
    for (int wpc : {1, 2, 4, 8}) {
        g_grid = 256 * wpc; g_iters = 4096;
        float t[4] = {timeit(l_loop<0>, 5), timeit(l_loop<1>, 5), timeit(l_loop<2>, 5), timeit(l_loop<3>, 5)};
        for (int o = 0; o < 4; o++) {
            double instr = (double)g_grid * 4 * 16 * g_iters;
            printf("loop     waves/SIMD=%d %-18s %.3f ms  %.3f instr/clk/SIMD@2.4GHz\n", wpc, names[o], t[o], instr / (t[o] * 1e-3) / simds / 2.4e9);
        }
    }
    g_grid = 131072;
    float t[4] = {timeit(l_str<0>, 5), timeit(l_str<1>, 5), timeit(l_str<2>, 5), timeit(l_str<3>, 5)};
    for (int o = 0; o < 4; o++) {
        double instr = (double)g_grid * 4 * 16 * 96;
        printf("straight 131072 WGs        %-18s %.3f ms  %.3f instr/clk/SIMD@2.4GHz\n", names[o], t[o], instr / (t[o] * 1e-3) / simds / 2.4e9);
    }
    return 0;
}
