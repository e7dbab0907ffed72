// store_probe.hip — how should a write-only kernel be shaped on gfx950?  2 GiB buffer, 16 B per lane per store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// A: one store per lane, consecutive lanes consecutive 16 B
__global__ __launch_bounds__(256) void k_one(uint4* out, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) out[i] = make_uint4(i, 1, 2, 3);
}
// B: U stores per lane, each wave writes U KiB contiguous (lane stride 16 B inside 1 KiB rows)
template <int U> __global__ __launch_bounds__(256) void k_wave_contig(uint4* out, size_t n16) {
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t i = (wave * U + u) * 64 + lane;
        if (i < n16) out[i] = make_uint4(i, 1, 2, 3);
    }
}
// C: U stores per lane, far apart (grid-size stride) — the intra kernel's shape
template <int U> __global__ __launch_bounds__(256) void k_far(uint4* out, size_t n16) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t i = i0 + u * stride;
        if (i < n16) out[i] = make_uint4(i, 1, 2, 3);
    }
}
// D: persistent grid-stride loop
__global__ __launch_bounds__(256) void k_loop(uint4* out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = make_uint4(i, 1, 2, 3);
}
// E: nontemporal
__global__ __launch_bounds__(256) void k_one_nt(uint4* out, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i v = {(int)i, 1, 2, 3};
    if (i < n16) __builtin_nontemporal_store(v, reinterpret_cast<v4i*>(out) + i);
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
    const size_t bytes = (size_t)2 << 30, n16 = bytes / 16;
    uint4* d; CK(hipMalloc(&d, bytes));
    auto rep = [&](const char* name, float ms) { printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12); };
    rep("hipMemsetAsync", timeit([&] { CK(hipMemsetAsync(d, 1, bytes, 0)); }));
    rep("A one store/lane", timeit([&] { hipLaunchKernelGGL(k_one, dim3(n16 / 256), dim3(256), 0, 0, d, n16); }));
    rep("E one store/lane nontemporal", timeit([&] { hipLaunchKernelGGL(k_one_nt, dim3(n16 / 256), dim3(256), 0, 0, d, n16); }));
    rep("B 4 stores/lane, wave-contiguous 4 KiB", timeit([&] { hipLaunchKernelGGL(k_wave_contig<4>, dim3(n16 / 256 / 4), dim3(256), 0, 0, d, n16); }));
    rep("B 16 stores/lane, wave-contiguous 16 KiB", timeit([&] { hipLaunchKernelGGL(k_wave_contig<16>, dim3(n16 / 256 / 16), dim3(256), 0, 0, d, n16); }));
    rep("C 4 stores/lane, grid-stride apart", timeit([&] { hipLaunchKernelGGL(k_far<4>, dim3(n16 / 256 / 4), dim3(256), 0, 0, d, n16); }));
    rep("C 16 stores/lane, grid-stride apart", timeit([&] { hipLaunchKernelGGL(k_far<16>, dim3(n16 / 256 / 16), dim3(256), 0, 0, d, n16); }));
    for (int wg : {2048, 4096, 16384, 65536})
        { char nm[64]; snprintf(nm, 64, "D loop, %d workgroups", wg); rep(nm, timeit([&] { hipLaunchKernelGGL(k_loop, dim3(wg), dim3(256), 0, 0, d, n16); })); }
    return 0;
}
