// What does the bf16 codebook search's instruction mix sustain on this part when everything around the matrix
// instructions is taken away?  The search kernels (d-vqvae_amd/csrc/vq_argmin.hip) run, per wave, groups of 32 codes
// against four latent tiles: 16 items per group, each item one 1-KiB fragment read from LDS followed by four
// v_mfma_f32_16x16x32_bf16 on four accumulators, then 32 compare-selects per group.  This program runs that schedule
// on random bf16 data, one 8-wave workgroup per CU (two waves per SIMD) like the search, in stages:
//   mode 0  matrix instructions only (operands in registers)
//   mode 1  + the fragment read from LDS before every four matrix instructions (codebook resident, no copies)
//   mode 2  + the compare-select fold of every finished half (the search's argmax)
//   mode 3  + the codebook streamed global -> LDS in double-buffered 68-KiB stages with one barrier per stage
// and prints, per mode, the time and the fraction of the nominal 2.5 PFLOP/s.  Work = the search of 1 Mi x 256-d latents
// against 8192 codes (4.4 TFLOP).  Under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE` the
// same launches give the matrix pipe's busy share and the clock held.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_ceiling mfma_ceiling.hip ; run: ./mfma_ceiling [mode|-1] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int S = 8;                       // k-steps of 32 dims: D = 256
constexpr int NB = 4;                      // latent tiles per wave
constexpr int GROUP_BYTES = (2 * S + 1) * 1024;
constexpr int TS = 4;                      // groups per stage (68 KiB)
#ifndef CEIL_WAVES
#define CEIL_WAVES 8
#endif
constexpr int WAVES = CEIL_WAVES;   // 8: two waves per SIMD (the search); 4: one wave per SIMD

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(WAVES * 64) void ceiling_kernel(const unsigned short* __restrict__ z, const char* __restrict__ packed,
                                                             int n_groups, int sweeps, float zero, int* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4;
    bf16x8 zb[NB][S];
    const unsigned short* zr = z + ((size_t)(blockIdx.x * WAVES + wave) * 64 + lane) * 8 * NB * S;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = 0; s < S; ++s) zb[nb][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(zr + (nb * S + s) * 8));
    float best[NB];
    int bidx[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { best[nb] = -1e30f; bidx[nb] = 0; }
    auto issue = [&](int st, int buf) {
        const char* src = packed + (size_t)st * TS * GROUP_BYTES;
        char* dst = smem + buf * TS * GROUP_BYTES;
        for (int u = wave; u < TS * (2 * S + 1); u += WAVES) glds16(src + (size_t)u * 1024 + lane * 16, dst + u * 1024);
    };
    // resident part: the first stage (modes 0-2 keep re-using it; mode 3 cycles through the whole codebook)
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int n_stages = n_groups / TS;
    int g = 0;
    bf16x8 a_reg = zb[0][0];
    for (int sw = 0; sw < sweeps; ++sw) {
        for (int st = 0; st < n_stages; ++st, ++g) {
            const int buf = MODE >= 3 ? (g & 1) : 0;
            if (MODE >= 3) issue(st + 1 == n_stages ? 0 : st + 1, buf ^ 1);
            const char* eb = smem + buf * TS * GROUP_BYTES;
            for (int t = 0; t < TS; ++t) {
                const char* tb = eb + t * GROUP_BYTES;
                asm volatile("" ::: "memory");   // the resident modes must re-read their fragments like the search does
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    f32x4 acc[NB];
                    f32x4 e = {0.f, 0.f, 0.f, 0.f};
                    if (MODE >= 1) e = *reinterpret_cast<const f32x4*>(tb + 2 * S * 1024 + (16 * hh + 4 * q) * 4);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb] = e + best[nb] * zero;   // (a carried dependence: nothing is loop-invariant)
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        bf16x8 a = a_reg;
                        if (MODE >= 1) a = *reinterpret_cast<const bf16x8*>(tb + (hh * S + s) * 1024 + lane * 16);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, zb[nb][s], acc[nb], 0, 0, 0);
                    }
                    if (MODE >= 2) {
                        const int base = (st * TS + t) * 32 + 16 * hh + 4 * q;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float v = acc[nb][i];
                                const bool gt = v > best[nb];
                                bidx[nb] = gt ? base + i : bidx[nb];
                                best[nb] = gt ? v : best[nb];
                            }
                    } else {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) best[nb] += acc[nb][0] + acc[nb][1] + acc[nb][2] + acc[nb][3];
                    }
                }
            }
            if (MODE >= 3) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    }
    int out = 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) out += bidx[nb] + (int)best[nb];
    if (out == 0x7fffffff) sink[tid] = out;
}

int main(int argc, char** argv) {
    const int want = argc > 1 ? atoi(argv[1]) : -1;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const int K = 8192, D = 256, n_groups = K / 32;
    const long long n = 1 << 20;
    const int grid = 256;
    // every wave keeps its 64 latents for `sweeps` sweeps of the codebook: the same matrix work as n latents once
    const int sweeps = (int)(n / ((long long)grid * WAVES * NB * 16));
    const double flops = 2.0 * n * K * D;
    std::vector<unsigned short> hz((size_t)grid * WAVES * 64 * 8 * NB * S), hp((size_t)n_groups * GROUP_BYTES / 2);
    unsigned s = 12345u;
    auto rnd_bf16 = [&]() {   // roughly N(0,1): sum of 4 uniforms, as bf16 bits
        float v = 0.f;
        for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; v += (float)(s >> 8) / 16777216.f - 0.5f; }
        v *= 1.7320508f;
        unsigned b; __builtin_memcpy(&b, &v, 4);
        return (unsigned short)(b >> 16);
    };
    for (auto& x : hz) x = rnd_bf16();
    for (auto& x : hp) x = rnd_bf16();
    unsigned short* dz; char* dp; int* dsink;
    CHECK(hipMalloc(&dz, hz.size() * 2));
    CHECK(hipMalloc(&dp, hp.size() * 2));
    CHECK(hipMalloc(&dsink, 4096));
    CHECK(hipMemcpy(dz, hz.data(), hz.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
    const int lds = 2 * TS * GROUP_BYTES;
    void (*kern[4])(const unsigned short*, const char*, int, int, float, int*) = {ceiling_kernel<0>, ceiling_kernel<1>, ceiling_kernel<2>, ceiling_kernel<3>};
    const char* what[4] = {"matrix instructions only", "+ LDS fragment reads", "+ compare-select fold", "+ codebook streamed through LDS, barrier per stage"};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int m = 0; m < 4; ++m) {
        if (want >= 0 && m != want) continue;
        CHECK(hipFuncSetAttribute((const void*)kern[m], hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(kern[m], dim3(grid), dim3(WAVES * 64), lds, 0, dz, dp, n_groups, sweeps, 0.f, dsink);   // warm-up
        CHECK(hipDeviceSynchronize());
        float best_ms = 1e30f, sum = 0.f;
        for (int r = 0; r < reps; ++r) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern[m], dim3(grid), dim3(WAVES * 64), lds, 0, dz, dp, n_groups, sweeps, 0.f, dsink);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best_ms = ms < best_ms ? ms : best_ms;
            sum += ms;
        }
        const float avg = sum / reps;
        printf("mode %d (%s): avg %.3f ms, best %.3f ms, %.1f TFLOP/s = %.3f of 2.5 PFLOP/s (best %.3f)\n", m, what[m], avg, best_ms,
               flops / avg / 1e9, flops / avg / 1e9 / 2500.0, flops / best_ms / 1e9 / 2500.0);
    }
    return 0;
}
