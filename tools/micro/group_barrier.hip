// Cost and correctness of a barrier among the 8 workgroups that share one XCD-local row tile, for the ways a
// producer's stores can be made visible to the other workgroups of its group on gfx950:
//   mode 0  agent-scope release / acquire as the compiler emits them (L2 write-back + L2 invalidate)
//   mode 1  stores drained (vmcnt 0), relaxed device-scope counter, consumer drops its CU's vector L1 (buffer_inv sc0)
//   mode 2  as 1, consumer invalidates with buffer_inv sc1
//   mode 3  as 1, no invalidate at all (timing floor; expected to read stale data)
//   mode 4  as 3, but the consumer's loads carry the device-scope bit (sc1) instead
// Every phase each workgroup rewrites its 16-KiB slice of the group's buffer and, after the barrier, checks the slice
// of its neighbour in the group (which it also read in the phase before, so a stale L1 line shows).
// Build: hipcc --offload-arch=gfx950 -O3 -o group_barrier group_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kGroup = 8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 load_dev_scope(const u32x4* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void barrier_kernel(u32x4* buf, unsigned* ctr, unsigned* xcc, unsigned* errors, int phases, int groups) {
    const int lin = blockIdx.x;
    const int xcd = lin % 8, i = lin / 8;
    const int bx = i % kGroup, by = (i / kGroup) * 8 + xcd;   // group `by`: 8 workgroups with equal lin % 8
    if (by >= groups) return;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[lin] = id;
    }
    u32x4* mine = buf + ((size_t)by * kGroup + bx) * 1024;          // 1024 x 16 B
    const u32x4* theirs = buf + ((size_t)by * kGroup + (bx + 1) % kGroup) * 1024;
    unsigned bad = 0, spins = 0;
    for (int p = 0; p < phases; ++p) {
        for (int k = threadIdx.x; k < 1024; k += 256) {
            u32x4 v; v.x = p; v.y = k; v.z = by * kGroup + bx; v.w = p ^ k;
            mine[k] = v;
        }
        // ---- barrier
        if (MODE == 0) {
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(ctr + by, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr + by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(kGroup * (p + 1))) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 24)) { atomicAdd(errors, 1u << 20); break; } }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(ctr + by, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr + by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(kGroup * (p + 1))) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 24)) { atomicAdd(errors, 1u << 20); break; } }
            }
            __builtin_amdgcn_s_barrier();
            if (MODE == 1) asm volatile("buffer_inv sc0" ::: "memory");
            if (MODE == 2) asm volatile("buffer_inv sc1" ::: "memory");
        }
        // ---- check the neighbour's slice of THIS phase
        for (int k = threadIdx.x; k < 1024; k += 256) {
            const u32x4 v = (MODE == 4) ? load_dev_scope(theirs + k) : theirs[k];
            if (v.x != (unsigned)p || v.y != (unsigned)k || v.w != (unsigned)(p ^ k)) ++bad;
        }
        // the neighbour may not overwrite its slice before this workgroup has read it: second barrier of the phase
        // (counted on a second counter so that the two never mix)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr + 4096 + by, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr + 4096 + by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(kGroup * (p + 1))) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 24)) { atomicAdd(errors, 1u << 20); break; } }
        }
        __builtin_amdgcn_s_barrier();
    }
    if (bad) atomicAdd(errors, bad);
}

template <int MODE>
static void run(u32x4* buf, unsigned* ctr, unsigned* xcc, unsigned* errors, int groups, int phases) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 64 * ((groups + 7) / 8);
    float best = 1e9f;
    unsigned err_total = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(ctr, 0, 8192 * sizeof(unsigned));
        (void)hipMemset(errors, 0, sizeof(unsigned));
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((barrier_kernel<MODE>), dim3(grid), dim3(256), 0, 0, buf, ctr, xcc, errors, phases, groups);
        (void)hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("mode %d: launch failed\n", MODE); exit(1); }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        unsigned e = 0;
        (void)hipMemcpy(&e, errors, sizeof(e), hipMemcpyDeviceToHost);
        err_total += e;
    }
    printf("mode %d  %3d groups: %.3f us per phase (two barriers + 32 KiB of traffic per workgroup), wrong words %u\n", MODE, groups,
           best * 1e3f / phases, err_total);
}

int main(int argc, char** argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 300;
    u32x4* buf; unsigned *ctr, *xcc, *errors;
    (void)hipMalloc(&buf, (size_t)64 * kGroup * 1024 * 16);
    (void)hipMalloc(&ctr, 8192 * sizeof(unsigned));
    (void)hipMalloc(&xcc, 1024 * sizeof(unsigned));
    (void)hipMalloc(&errors, sizeof(unsigned));
    for (int groups : {32, 8, 1}) {
        run<0>(buf, ctr, xcc, errors, groups, phases);
        run<1>(buf, ctr, xcc, errors, groups, phases);
        run<2>(buf, ctr, xcc, errors, groups, phases);
        run<3>(buf, ctr, xcc, errors, groups, phases);
        run<4>(buf, ctr, xcc, errors, groups, phases);
    }
    unsigned h[256];
    (void)hipMemcpy(h, xcc, sizeof(h), hipMemcpyDeviceToHost);
    int mism = 0;
    for (int l = 0; l < 256; ++l) if ((h[l] & 15) != (h[l % 8] & 15)) ++mism;
    printf("XCC_ID of workgroups 0..15:");
    for (int l = 0; l < 16; ++l) printf(" %u", h[l] & 15);
    printf("\nworkgroups whose XCC differs from that of workgroup (id mod 8): %d of 256\n", mism);
    return 0;
}
