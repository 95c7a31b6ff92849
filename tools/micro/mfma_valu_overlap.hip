// Do fp32 MFMAs of one wave run under the vector multiply-adds of another wave on the same SIMD?  One workgroup of 8 waves
// per CU (two waves per SIMD); waves 0-3 run a chain of v_mfma_f32_16x16x4_f32 over three accumulators, waves 4-7 a chain of
// packed fp32 multiply-adds (or LDS reads), each alone and both together.  Build: hipcc --offload-arch=gfx950 -O3 -o
// mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// mode bit 0: waves 0-3 do fp32 MFMAs (bit 3: bf16 16x16x32 MFMAs instead); bit 1: waves 4-7 do vector FMAs; bit 2: waves 4-7 do LDS reads instead
__global__ __launch_bounds__(512) void overlap_kernel(int mode, int iters, float* __restrict__ sink) {
    __shared__ f32x4 lds[1024];
    const int tid = threadIdx.x, wave = tid >> 6;
    lds[tid] = f32x4{1.f, 2.f, 3.f, 4.f};
    lds[tid + 512] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    float out = 0.f;
    if (wave < 4) {
        if (mode & 8) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
            bf16x8 x, y;
#pragma unroll
            for (int k = 0; k < 8; ++k) { x[k] = (__bf16)(tid * 1e-3f + k); y[k] = (__bf16)(1.f + k * 0.5f); }
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a2, 0, 0, 0);
                }
            }
            out = a0[0] + a1[1] + a2[2];
        } else if (mode & 1) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
            const float x = tid * 1e-3f, y = 1.f + tid * 1e-4f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
                }
            }
            out = a0[0] + a1[1] + a2[2];
        }
    } else if (mode & 2) {
        f32x2 t[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) t[k] = f32x2{(float)k, (float)tid};
        const f32x2 w = {1.0001f, 0.9999f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 12; ++k) t[k] = __builtin_elementwise_fma(t[k], w, w);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) out += t[k][0] + t[k][1];
    } else if (mode & 4) {
        f32x4 s = {0, 0, 0, 0};
        int idx = tid & 511;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                s += lds[(idx + 64 * u) & 1023];
            }
            idx = (idx + 7) & 511;
        }
        out = s[0] + s[1] + s[2] + s[3];
    }
    if (out == 12345.678f) sink[tid] = out;
}

int main() {
    float* sink;
    (void)hipMalloc(&sink, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    const char* names[] = {"", "mfma alone (24 per iteration)", "packed fma alone (24 per iteration)", "mfma + packed fma", "lds reads alone (6 x 16 B per iteration)",
                           "mfma + lds reads", "", "", "bf16 16x16x32 mfma alone (24 per iteration)", "", "bf16 mfma + packed fma", "", "bf16 mfma + lds reads"};
    const int modes[] = {1, 2, 3, 4, 5, 8, 10, 12};
    for (int m = 0; m < 8; ++m) {
        const int mode = modes[m];
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(overlap_kernel, dim3(256), dim3(512), 0, 0, mode, iters, sink);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(overlap_kernel, dim3(256), dim3(512), 0, 0, mode, iters, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-42s %.1f us per launch, %.1f cycles per iteration at 2.4 GHz\n", names[mode], ms / 5 * 1e3, ms / 5 * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
