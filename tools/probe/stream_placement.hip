// stream_placement.hip — is "three arrays written at the same time inside one allocation are slow" a property of the memory system
// alone?  No transform, no LDS: a workgroup reads 2 x 2 KiB (two input streams) and writes 3 x 4 KiB (three output streams), the byte
// mix and the 1 KiB-per-wave store shape of the fused 32x32 kernel, one 16-byte access per lane per instruction.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/stream_placement tools/probe/stream_placement.hip
//   run:   tools/probe/stream_placement            (prints GB/s for the three output arrays at different offsets of one 44 GiB pool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// block b: in0[b] 1 KiB, in1[b] 1 KiB (64 lanes x 16 B each), out0/out1/out2[b] 4 KiB each (4 stores of 1 KiB per wave)
__global__ __launch_bounds__(256) void k(const v4i* __restrict__ in0, const v4i* __restrict__ in1, v4i* __restrict__ o0, v4i* __restrict__ o1,
                                         v4i* __restrict__ o2, size_t nblocks) {
    const size_t b = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave per block
    if (b >= nblocks) return;
    const int lane = threadIdx.x & 63;
    v4i a = __builtin_nontemporal_load(&in0[b * 64 + lane]);
    v4i c = __builtin_nontemporal_load(&in1[b * 64 + lane]);
    a += c;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        a.x += q;
        __builtin_nontemporal_store(a, &o0[b * 256 + q * 64 + lane]);
        __builtin_nontemporal_store(a, &o1[b * 256 + q * 64 + lane]);
        __builtin_nontemporal_store(a, &o2[b * 256 + q * 64 + lane]);
    }
}

int main() {
    const size_t n = (size_t)1 << 20, G = (size_t)1 << 30;
    char* pool; v4i *in0, *in1;
    CHECK(hipMalloc((void**)&pool, 44 * G));
    CHECK(hipMalloc((void**)&in0, n * 1024)); CHECK(hipMalloc((void**)&in1, n * 1024));
    CHECK(hipMemset(in0, 1, n * 1024)); CHECK(hipMemset(in1, 2, n * 1024));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int offs[][3] = {{0, 4, 8}, {4, 8, 12}, {12, 16, 20}, {20, 24, 28}, {24, 28, 32}, {0, 16, 32}, {0, 4, 36}, {32, 36, 40}, {1, 14, 27}, {0, 8, 16}};
    for (auto& o : offs) {
        v4i* p0 = (v4i*)(pool + o[0] * G); v4i* p1 = (v4i*)(pool + o[1] * G); v4i* p2 = (v4i*)(pool + o[2] * G);
        for (int w = 0; w < 2; w++) k<<<dim3((unsigned)(n / 4)), dim3(256)>>>(in0, in1, p0, p1, p2, n);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int it = 0; it < 8; it++) k<<<dim3((unsigned)(n / 4)), dim3(256)>>>(in0, in1, p0, p1, p2, n);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
        printf("{\"outputs_at_GiB\": [%d, %d, %d], \"ms\": %.4f, \"GBps\": %.0f}\n", o[0], o[1], o[2], ms, (double)n * 14336 / ms / 1e6);
        fflush(stdout);
    }
    return 0;
}
