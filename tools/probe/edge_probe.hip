// edge_probe.hip — a write-bound kernel (1 KiB per wave) that first needs a small per-block read:
// which shape keeps the store stream at full speed on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// vector load of 16 B per lane (two distinct 16-B segments per block), IU blocks per lane
template <int IU, bool NT>
__global__ __launch_bounds__(256) void k_vec(uint4* out, const uint8_t* edge, int pitch, size_t nblk) {
    const size_t wave0 = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * 256) >> 6;
    const int lane = threadIdx.x & 63;
    uint4 v[IU];
#pragma unroll
    for (int u = 0; u < IU; u++) {
        const size_t b = wave0 + u * nw;
        const uint8_t* p = edge + (b < nblk ? b : 0) * pitch + 16 + (lane & 1) * 16;
        if (NT) { typedef int v4i __attribute__((ext_vector_type(4))); v4i t; v4i tt; __builtin_memcpy(&tt, p, 0); t = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p)); v[u] = make_uint4(t.x, t.y, t.z, t.w); }
        else __builtin_memcpy(&v[u], p, 16);
    }
#pragma unroll
    for (int u = 0; u < IU; u++) {
        const size_t b = wave0 + u * nw;
        if (b < nblk) out[b * 64 + lane] = v[u];
    }
}
// scalar load: the block index is wave-uniform
template <int IU>
__global__ __launch_bounds__(256) void k_sc(uint4* out, const uint8_t* edge, int pitch, size_t nblk) {
    const size_t wave0 = __builtin_amdgcn_readfirstlane((int)(((size_t)blockIdx.x * 256 + threadIdx.x) >> 6)), nw = ((size_t)gridDim.x * 256) >> 6;
    const int lane = threadIdx.x & 63;
    uint4 v[IU];
#pragma unroll
    for (int u = 0; u < IU; u++) {
        const size_t b = wave0 + u * nw;
        const uint32_t* p = reinterpret_cast<const uint32_t*>(edge + (b < nblk ? b : 0) * pitch + 16);
        uint32_t w[8];
#pragma unroll
        for (int q = 0; q < 8; q++) w[q] = p[q];                 // s_load_dwordx8 (uniform address)
        v[u] = (lane & 1) ? make_uint4(w[4], w[5], w[6], w[7]) : make_uint4(w[0], w[1], w[2], w[3]);
    }
#pragma unroll
    for (int u = 0; u < IU; u++) {
        const size_t b = wave0 + u * nw;
        if (b < nblk) out[b * 64 + lane] = v[u];
    }
}
template <typename F> static float timeit(F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; i++) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5;
}
int main() {
    const size_t nblk = (size_t)1 << 21, bytes = nblk * 1024;
    uint4* d; CK(hipMalloc(&d, bytes));
    uint8_t* e; CK(hipMalloc(&e, nblk * 160 + 64)); CK(hipMemset(e, 3, nblk * 160 + 64));
    auto rep = [&](const char* name, float ms) { printf("%-52s %.3f ms  %.2f TB/s written\n", name, ms, bytes / (ms * 1e-3) / 1e12); };
#define RUNV(IU, NT, P, NAME) rep(NAME, timeit([&] { hipLaunchKernelGGL((k_vec<IU, NT>), dim3(nblk / 4 / IU), dim3(256), 0, 0, d, e, P, nblk); }))
#define RUNS(IU, P, NAME) rep(NAME, timeit([&] { hipLaunchKernelGGL((k_sc<IU>), dim3(nblk / 4 / IU), dim3(256), 0, 0, d, e, P, nblk); }))
    RUNV(1, false, 160, "vector load, pitch 160, 1 block/lane");
    RUNV(2, false, 160, "vector load, pitch 160, 2 blocks/lane");
    RUNV(4, false, 160, "vector load, pitch 160, 4 blocks/lane");
    RUNV(8, false, 160, "vector load, pitch 160, 8 blocks/lane");
    RUNV(1, true, 160, "vector load nontemporal, pitch 160, 1 block/lane");
    RUNV(4, true, 160, "vector load nontemporal, pitch 160, 4 blocks/lane");
    RUNV(1, false, 32, "vector load, pitch 32 (dense), 1 block/lane");
    RUNV(4, false, 32, "vector load, pitch 32 (dense), 4 blocks/lane");
    RUNV(1, false, 0, "vector load, pitch 0 (always cached), 1 block/lane");
    RUNS(1, 160, "scalar load, pitch 160, 1 block/wave");
    RUNS(2, 160, "scalar load, pitch 160, 2 blocks/wave");
    RUNS(4, 160, "scalar load, pitch 160, 4 blocks/wave");
    return 0;
}
