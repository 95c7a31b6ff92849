// Read-bandwidth ceiling of one MI355X for a once-read stream: every lane loads 16 bytes per instruction, UNROLL
// instructions in flight, persistent grid.  Build: hipcc --offload-arch=gfx950 -O3 -o hbm_stream hbm_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ src, size_t n16, unsigned* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const u32x4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;   // practically never: keeps the loads alive
}

template <int UNROLL, bool NT>
static void run(const u32x4* src, size_t bytes, unsigned* sink, int blocks_per_cu) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((read_kernel<UNROLL, NT>), dim3(grid), dim3(256), 0, 0, src, bytes / 16, sink);
    (void)hipEventRecord(e0);
    const int iters = 20;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((read_kernel<UNROLL, NT>), dim3(grid), dim3(256), 0, 0, src, bytes / 16, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  %4zu MB unroll %d %s %d blocks/CU: %.4f ms  %.1f GB/s\n", bytes >> 20, UNROLL, NT ? "nt" : "  ", blocks_per_cu, ms / iters,
           bytes / (ms / iters) / 1e6);
}

int main() {
    const size_t max_bytes = (size_t)4 << 30;
    u32x4* src; unsigned* sink;
    if (hipMalloc(&src, max_bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(src, 1, max_bytes);
    for (size_t bytes : {(size_t)537 << 20, (size_t)2148 << 20, max_bytes}) {
        run<4, false>(src, bytes, sink, 8);
        run<8, false>(src, bytes, sink, 8);
        run<8, true>(src, bytes, sink, 8);
        run<8, true>(src, bytes, sink, 4);
        run<16, true>(src, bytes, sink, 4);
    }
    (void)hipFree(src); (void)hipFree(sink);
    return 0;
}
