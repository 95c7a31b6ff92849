// PointNet shared per-point MLP (C -> 64 -> 128 -> 1024) + global max-pool, fused (gfx950).
//
// Replaces the Conv1d(k=1)+BatchNorm(+ReLU) x3 + torch.max of STN3d.forward and
// PointNetEncoder.forward (DVQ-VAE/network/pointnet_encoder.py:29-33, 146-164).  The
// reference materialises [B,64,N], [B,128,N] and [B,1024,N] activations; here a point's
// activations never leave the register file:
//
//   conv1 (C<=4 inputs)   fp32 VALU; a lane computes 32 of the 64 channels of ITS point,
//                         already in the layout of an MFMA B operand.
//   conv2 (64 -> 128)     MFMA with the weights as A (rows = output channel) and the
//                         points as B (columns = point): the 32x32 accumulator then has
//                         the POINT on the lane and the channel in the registers.
//   conv3 (128 -> 1024)   that accumulator, after bias/ReLU (and bf16 packing), is used
//                         AS IS as the A operand of the next MFMA (an accumulator tile
//                         X used as A computes X^T * B), with the conv3 weights as B
//                         (columns = output channel).  Its result has the point in the
//                         registers and the output channel on the lane, so
//   max over points       is a lane-local max over registers; one shuffle merges the
//                         two half-waves and one LDS atomic per (wave, 32 channels)
//                         folds it into the block's running maximum.
//
// One workgroup (8 waves) walks one cloud in chunks of 512 points.  The conv3
// weights (256 KiB bf16 / 512 KiB fp32, pre-packed in fragment order) stream through a
// double-buffered LDS stage by direct global->LDS loads while the matrix cores work on
// the previous stage; conv1/conv2 weights stay resident in LDS.  HBM traffic is the
// input cloud once (16 B per point) and 4 KiB of output per cloud.
//
// Two kernels: fp32 on the 32x32x2 MFMA (the description above, literally) and bf16 on the
// 16x16x32 MFMA (same fusion on 16-point tiles; its lane mapping is described at the kernel).
#include "common.h"

#include <type_traits>

namespace dvq {

constexpr int kPnWaves = 8;
constexpr int kPnThreads = kPnWaves * 64;
constexpr int kC1 = 64, kC2 = 128, kC3 = 1024;
constexpr int kCt = kC3 / 32;  // 32 output-channel tiles

template <int DT> struct PnCfg;
template <> struct PnCfg<DVQ_F32> {
    static constexpr int NP = 2;
    static constexpr int UPC = 16;        // 4 row tiles x 4 groups of 4 k-steps
    static constexpr int CPS = 2;         // 32 KiB per stage
    static constexpr int MIN_WAVES = 2;
    static constexpr int A2_UNITS = 32;   // 4 row tiles x 8 groups of 4 k-steps
    static constexpr int XR = 16;
};

// Order-preserving float <-> int map so that an integer max is a float max.
__device__ __forceinline__ int f32_key(float f) {
    const int b = __builtin_bit_cast(int, f);
    return b >= 0 ? b : (b ^ 0x7FFFFFFF);
}
__device__ __forceinline__ float key_f32(int k) {
    return __builtin_bit_cast(float, k >= 0 ? k : (k ^ 0x7FFFFFFF));
}

__device__ __forceinline__ void glds16(const void* gsrc_lane, void* lds_wave_base) {
    // 64 lanes x 16 B: lane l's bytes land at lds_wave_base + 16*l
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// conv3 weights [1024,128] fp32 -> fragment order of the fp32 kernel.
//   unit (ct, t2, g): lane (c,h) holds 4 f32 : q -> A3[32ct+c][32t2 + 8g + 4h + q]
// (the k order an accumulator tile presents when it is reused as an MFMA A operand).
template <int DT>
__global__ void pn_pack_w3_kernel(const float* __restrict__ a3, uint4* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = kCt * PnCfg<DT>::UPC * 64;
    if (t >= total) return;
    const int l = t & 63, unit = t >> 6;
    const int c = l & 31, h = l >> 5;
    const int ct = unit / PnCfg<DT>::UPC, u = unit % PnCfg<DT>::UPC;
    const float* row = a3 + (ct * 32 + c) * kC2;
    uint4 v;
    {
        const int t2 = u >> 2, g = u & 3;
        v = *reinterpret_cast<const uint4*>(row + 32 * t2 + 8 * g + 4 * h);
    }
    out[t] = v;
}

// conv1 (VALU) + conv2 (MFMA) for one tile of 32 points: lane (r,h) owns point `n`.
// Leaves relu(conv2) in `Xf_np` (64 values) in the k order conv3 consumes.
template <int DT, int C>
__device__ __forceinline__ void pn_point_tile(const float* __restrict__ xb_, int N, int pm, int n, bool has_trans, const float (&T)[9],
                                              const float* a1s, const float* c1s, const char* a2s, const float* c2s,
                                              int lane, int h, float (&Xf_np)[64]) {
    const float* xb = xb_;
    float p[4];
    // a cloud is C*N floats either way: channel-major [C][N] (the reference's layout) or point-major [N][C]
    const int sc = pm ? 1 : N, sn = pm ? C : 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) p[c] = c < C ? xb[(int64_t)c * sc + (int64_t)n * sn] : 0.f;
    if (has_trans) {  // xyz' = xyz * T  (pointnet_encoder.py:146)
        const float q0 = p[0], q1 = p[1], q2 = p[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v = q0 * T[j];
            v = __builtin_fmaf(q1, T[3 + j], v);
            v = __builtin_fmaf(q2, T[6 + j], v);
            p[j] = v;
        }
    }
    // conv1: channels 32h .. 32h+31 of this lane's point, 4 at a time
    auto conv1_4 = [&](int t4, float (&o)[4]) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(c1s + 32 * h + 4 * t4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(a1s + (32 * h + 4 * t4 + q) * 4);
            float v = w[0] * p[0];
            v = __builtin_fmaf(w[1], p[1], v);
            v = __builtin_fmaf(w[2], p[2], v);
            if (C > 3) v = __builtin_fmaf(w[3], p[3], v);
            v += bias[q];
            o[q] = v > 0.f ? v : 0.f;
        }
    };
    {
        float h1[32];
#pragma unroll
        for (int t4 = 0; t4 < 8; ++t4) {
            float o[4];
            conv1_4(t4, o);
            h1[4 * t4 + 0] = o[0]; h1[4 * t4 + 1] = o[1]; h1[4 * t4 + 2] = o[2]; h1[4 * t4 + 3] = o[3];
            // keep the scheduler from hoisting all 40 weight reads (160 registers) to the top
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bias = *reinterpret_cast<const f32x4*>(c2s + 32 * t2 + 8 * g + 4 * h);
                acc[4 * g + 0] = bias[0]; acc[4 * g + 1] = bias[1]; acc[4 * g + 2] = bias[2]; acc[4 * g + 3] = bias[3];
            }
#pragma unroll
            for (int t4 = 0; t4 < 8; ++t4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(a2s + ((t2 * 8 + t4) * 64 + lane) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], h1[4 * t4 + q], acc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = acc[i];
                Xf_np[t2 * 16 + i] = v > 0.f ? v : 0.f;
            }
        }
    }
}

// Arguments of up to kPnMaxGroups independent launches folded into one grid (dvq_pointnet_mlp_max_grouped): group g owns
// the workgroups block_start[g] .. block_start[g+1]-1.
constexpr int kPnMaxGroups = 8;
struct PnGroupsDev {
    const float* x[kPnMaxGroups];
    const float* trans[kPnMaxGroups];
    const float* a1[kPnMaxGroups];
    const float* c1[kPnMaxGroups];
    const float* a2[kPnMaxGroups];
    const float* c2[kPnMaxGroups];
    const char* w3p[kPnMaxGroups];
    const float* c3[kPnMaxGroups];
    float* feat[kPnMaxGroups];
    int N[kPnMaxGroups], B[kPnMaxGroups], cpb[kPnMaxGroups], relu3[kPnMaxGroups];
    int block_start[kPnMaxGroups + 1];
    int n;
};
__device__ __forceinline__ int pn_group_of(const PnGroupsDev& g, int block) {
    int k = 0;
    while (k + 1 < g.n && block >= g.block_start[k + 1]) ++k;
    return __builtin_amdgcn_readfirstlane(k);
}

template <int DT, int C>
__device__ __forceinline__ void pn_body(
    const float* __restrict__ x, const float* __restrict__ trans, int N, const float* __restrict__ a1,
    const float* __restrict__ c1, const float* __restrict__ a2, const float* __restrict__ c2,
    const char* __restrict__ w3p, const float* __restrict__ c3, int relu3, float* __restrict__ feat, int B, int cpb,
    int block_id, int pm) {
    typedef PnCfg<DT> Cfg;
    constexpr int NP = Cfg::NP;
    constexpr int STAGE_BYTES = Cfg::CPS * Cfg::UPC * 1024;
    constexpr int N_STAGES = kCt / Cfg::CPS;
    constexpr int A2_BYTES = Cfg::A2_UNITS * 1024;
    // Small clouds share a workgroup: cpb (1, 2 or 4) clouds per block, kPnWaves / cpb waves each, all fed by the
    // same stream of conv3 weights.
    const int wpc = kPnWaves / cpb;                       // waves per cloud
    const int PTS_PER_CHUNK = wpc * NP * 32;              // points of one cloud per chunk

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w3s = smem;                                                         // [2][STAGE_BYTES]
    char* a2s = smem + 2 * STAGE_BYTES;                                       // conv2 weights, fragment order
    float* a1s = reinterpret_cast<float*>(a2s + A2_BYTES);                    // [64][4]
    float* c1s = a1s + kC1 * 4;                                               // [64]
    float* c2s = c1s + kC1;                                                   // [128]
    int* bmax = reinterpret_cast<int*>(c2s + kC2);                            // [cpb][1024] running max (int keys)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps everything derived from it in SGPRs
    const int r = lane & 31;
    const int h = lane >> 5;
    const int cl = wave / wpc;                            // this wave's cloud within the block
    const int lw = wave - cl * wpc;
    int b = block_id * cpb + cl;
    if (b >= B) b = B - 1;                                // a surplus slot recomputes the last cloud; its result is not stored
    int* wmax = bmax + cl * kC3;

    // ---- resident tables
    for (int i = tid; i < kC1 * 4; i += kPnThreads) {
        const int ch = i >> 2, c = i & 3;
        a1s[i] = c < C ? a1[ch * C + c] : 0.f;
    }
    for (int i = tid; i < kC1; i += kPnThreads) c1s[i] = c1[i];
    for (int i = tid; i < kC2; i += kPnThreads) c2s[i] = c2[i];
    for (int i = tid; i < cpb * kC3; i += kPnThreads) bmax[i] = (int)0x80000000;
    // conv2 weights A2 [128][64] -> A-operand fragment order.  Channel order along k is
    // ch1(h, t) = 32h + t, the order conv1 leaves its outputs in.
    for (int i = tid; i < Cfg::A2_UNITS * 64; i += kPnThreads) {
        const int l = i & 63, u = i >> 6;
        const int rr = l & 31, hh = l >> 5;
        {
            const int t2 = u >> 3, t4 = u & 7;
            *reinterpret_cast<uint4*>(a2s + i * 16) =
                *reinterpret_cast<const uint4*>(a2 + (32 * t2 + rr) * kC1 + 32 * hh + 4 * t4);
        }
    }
    float T[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) T[i] = trans ? trans[b * 9 + i] : ((i % 4) == 0 ? 1.f : 0.f);
    const bool has_trans = trans != nullptr;
    __syncthreads();

    const float* xb = x + (int64_t)b * C * N;
    const int n_chunks = (N + PTS_PER_CHUNK - 1) / PTS_PER_CHUNK;

    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        // ---- kick off stage 0 of the conv3 weights; it lands while conv1/conv2 run
        {
            constexpr int UNITS = STAGE_BYTES / 1024;
            for (int u = wave; u < UNITS; u += kPnWaves) glds16(w3p + (int64_t)u * 1024 + lane * 16, w3s + u * 1024);
        }
        // ---- conv1 + conv2 for this wave's NP point tiles -> X (conv3 A fragments)
        float Xf[NP][64];
        // Tiles of 32 points are dealt to the waves round-robin (tile = chunk*wpc*NP + np*wpc + lw), so in the last
        // chunk of a cloud the tiles past its end fall into the higher np slots of most waves and are skipped
        // outright: nv = this wave's tiles that hold at least one point.
        const int tile0 = chunk * (wpc * NP) + lw;
        const int n_tiles_cloud = (N + 31) / 32;
        int nv = 0;
#pragma unroll
        for (int np = 0; np < NP; ++np) nv += (tile0 + np * wpc < n_tiles_cloud) ? 1 : 0;
        {
            // one call per point tile, spelled out so that every fragment index is a constant
#define DVQ_PN_TILE(np)                                                                                           \
    if constexpr (NP > np) {                                                                                      \
        if (np < nv) {                                                                                            \
            int n = (tile0 + np * wpc) * 32 + r;                                                                  \
            if (n >= N) n = N - 1; /* duplicate the last point: a duplicate cannot change a max */                \
            pn_point_tile<DT, C>(xb, N, pm, n, has_trans, T, a1s, c1s, a2s, c2s, lane, h, Xf[np]);                    \
        }                                                                                                         \
    }
            DVQ_PN_TILE(0) DVQ_PN_TILE(1) DVQ_PN_TILE(2) DVQ_PN_TILE(3)
#undef DVQ_PN_TILE
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own direct-to-LDS copies landed before they are published
        __syncthreads();  // stage 0 has landed (the barrier drains the direct-to-LDS loads)

        // ---- conv3 + max, one LDS stage of output-channel tiles at a time
#pragma unroll 1
        for (int st = 0; st < N_STAGES; ++st) {
            const int buf = st & 1;
            if (st + 1 < N_STAGES) {
                constexpr int UNITS = STAGE_BYTES / 1024;
                const char* src = w3p + (int64_t)(st + 1) * STAGE_BYTES;
                char* dst = w3s + (buf ^ 1) * STAGE_BYTES;
                for (int u = wave; u < UNITS; u += kPnWaves) glds16(src + (int64_t)u * 1024 + lane * 16, dst + u * 1024);
            }
            const char* sb = w3s + buf * STAGE_BYTES + lane * 16;
            // conv3 of one channel tile for the first NV point tiles of this wave
            auto conv3_tile = [&](auto nvc, int cc) {
                constexpr int NV = decltype(nvc)::value;
                f32x16 acc[NV];
#pragma unroll
                for (int np = 0; np < NV; ++np)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[np][i] = 0.f;
#pragma unroll
                for (int u = 0; u < Cfg::UPC; ++u) {
                    {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(sb + (cc * Cfg::UPC + u) * 1024);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int np = 0; np < NV; ++np)
                                acc[np] = __builtin_amdgcn_mfma_f32_32x32x2f32(Xf[np][4 * u + q], w[q], acc[np], 0, 0, 0);
                    }
                }
                float m = acc[0][0];
#pragma unroll
                for (int np = 0; np < NV; ++np)
#pragma unroll
                    for (int i = 0; i < 16; ++i) m = fmaxf(m, acc[np][i]);
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                if (h == 0) atomicMax(&wmax[(st * Cfg::CPS + cc) * 32 + r], f32_key(m));
            };
            static_assert(NP == 2, "the dispatch below spells out NP = 2");
            if (nv == 2) {
#pragma unroll 1
                for (int cc = 0; cc < Cfg::CPS; ++cc) conv3_tile(std::integral_constant<int, 2>{}, cc);
            } else if (nv == 1) {
#pragma unroll 1
                for (int cc = 0; cc < Cfg::CPS; ++cc) conv3_tile(std::integral_constant<int, 1>{}, cc);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own direct-to-LDS copies landed before they are published
            __syncthreads();  // next stage landed; this buffer is free for the stage after
        }
    }

    // ---- epilogue: bias (and ReLU for the STN trunk) after the max -- both are monotone,
    // so max_n act(h_n + c) == act(max_n h_n + c) exactly.
    for (int i = tid; i < cpb * kC3; i += kPnThreads) {
        const int cloud = block_id * cpb + i / kC3, ch = i % kC3;
        if (cloud >= B) break;
        float v = key_f32(bmax[i]) + c3[ch];
        if (relu3) v = v > 0.f ? v : 0.f;
        feat[(int64_t)cloud * kC3 + ch] = v;
    }
}

template <int DT, int C>
__global__ __launch_bounds__(kPnThreads, PnCfg<DT>::MIN_WAVES) void pointnet_mlp_max_kernel(
    const float* __restrict__ x, const float* __restrict__ trans, int N, const float* __restrict__ a1,
    const float* __restrict__ c1, const float* __restrict__ a2, const float* __restrict__ c2,
    const char* __restrict__ w3p, const float* __restrict__ c3, int relu3, float* __restrict__ feat, int B, int cpb, int pm) {
    pn_body<DT, C>(x, trans, N, a1, c1, a2, c2, w3p, c3, relu3, feat, B, cpb, blockIdx.x, pm);
}
template <int DT, int C>
__global__ __launch_bounds__(kPnThreads, PnCfg<DT>::MIN_WAVES) void pointnet_mlp_max_grouped_kernel(PnGroupsDev g) {
    const int k = pn_group_of(g, blockIdx.x);
    pn_body<DT, C>(g.x[k], g.trans[k], g.N[k], g.a1[k], g.c1[k], g.a2[k], g.c2[k], g.w3p[k], g.c3[k], g.relu3[k], g.feat[k],
                   g.B[k], g.cpb[k], blockIdx.x - g.block_start[k], 0);
}

// ------------------------------------------------------------------ bf16, 16x16x32 tiles
// The same fusion on the 16x16x32 MFMA (the shape that sustains the higher clock under a dense load).  A tile is 16
// points; lane (c = lane & 15, q = lane >> 4) owns point c of its tile and the k-group q:
//   conv1   the lane computes 16 of the 64 channels of its point: channels 32s + 8q + j (k-step s, j = 0..7), the
//           B fragment of conv2;
//   conv2   weights as A (rows = 16 output channels), points as B: the 16x16 accumulator has output channels
//           16t + 4q + i of point c on lane (c, q);
//   conv3   two such accumulators (t = 2ks, 2ks + 1), after ReLU and bf16 packing, are the A fragment of k-step ks
//           (k order: 32ks + 4q + i, then 32ks + 16 + 4q + i -- the conv3 weights are packed in that order); weights
//           as B (columns = 16 output channels): the result has output channel c on the lane and points 4q + i in
//           the registers;
//   max     over the registers of all point tiles, two lane exchanges across q, one LDS atomic per (wave, 16 channels).
constexpr int kCt16 = kC3 / 16;      // 64 output-channel tiles of 16
constexpr int kPnNP16 = 4;           // 16-point tiles per wave and chunk
constexpr int kPnUPC16 = 4;          // 1-KiB conv3 weight units per channel tile (4 k-steps of 32)
constexpr int kPnCPS16 = 4;          // channel tiles per LDS stage (16 KiB)

// conv3 weights [1024,128] fp32 -> unit (ct, ks): lane (c, q) holds 8 bf16, j -> A3[16ct + c][32ks + 16(j>>2) + 4q + (j&3)]
__global__ void pn_pack_w3_16_kernel(const float* __restrict__ a3, uint4* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= kCt16 * kPnUPC16 * 64) return;
    const int l = t & 63, unit = t >> 6;
    const int c = l & 15, q = l >> 4;
    const int ct = unit / kPnUPC16, ks = unit % kPnUPC16;
    const float* row = a3 + (ct * 16 + c) * kC2;
    unsigned short e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = f32_to_bf16_bits(row[32 * ks + 16 * (j >> 2) + 4 * q + (j & 3)]);
    uint4 v;
    v.x = e[0] | (e[1] << 16); v.y = e[2] | (e[3] << 16); v.z = e[4] | (e[5] << 16); v.w = e[6] | (e[7] << 16);
    out[t] = v;
}

// conv1 + conv2 for one tile of 16 points; leaves relu(conv2) as the four conv3 A fragments of the tile.
template <int C>
__device__ __forceinline__ void pn_point_tile16(const float* __restrict__ xb, int N, int pm, int n, bool has_trans, const float (&T)[9],
                                                const float* a1s, const float* c1s, const char* a2s, const float* c2s,
                                                int lane, int q, bf16x8 (&X)[4]) {
    float p[4];
    const int sc = pm ? 1 : N, sn = pm ? C : 1;   // channel-major [C][N] or point-major [N][C]
#pragma unroll
    for (int c = 0; c < 4; ++c) p[c] = c < C ? xb[(int64_t)c * sc + (int64_t)n * sn] : 0.f;
    if (has_trans) {  // xyz' = xyz * T  (pointnet_encoder.py:146)
        const float q0 = p[0], q1 = p[1], q2 = p[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v = q0 * T[j];
            v = __builtin_fmaf(q1, T[3 + j], v);
            v = __builtin_fmaf(q2, T[6 + j], v);
            p[j] = v;
        }
    }
    bf16x8 hb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int t4 = 0; t4 < 2; ++t4) {
            const int ch0 = 32 * s + 8 * q + 4 * t4;
            const f32x4 bias = *reinterpret_cast<const f32x4*>(c1s + ch0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(a1s + (ch0 + j) * 4);
                float v = w[0] * p[0];
                v = __builtin_fmaf(w[1], p[1], v);
                v = __builtin_fmaf(w[2], p[2], v);
                if (C > 3) v = __builtin_fmaf(w[3], p[3], v);
                v += bias[j];
                hb[s][4 * t4 + j] = static_cast<__bf16>(v > 0.f ? v : 0.f);
            }
        }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        f32x4 acc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = 2 * ks + u;
            acc[u] = *reinterpret_cast<const f32x4*>(c2s + 16 * t + 4 * q);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(a2s + ((t * 2 + s) * 64 + lane) * 16);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, hb[s], acc[u], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = acc[j >> 2][j & 3];
            X[ks][j] = static_cast<__bf16>(v > 0.f ? v : 0.f);
        }
    }
}

template <int C>
__device__ __forceinline__ void pn16_body(
    const float* __restrict__ x, const float* __restrict__ trans, int N, const float* __restrict__ a1,
    const float* __restrict__ c1, const float* __restrict__ a2, const float* __restrict__ c2,
    const char* __restrict__ w3p, const float* __restrict__ c3, int relu3, float* __restrict__ feat, int B, int cpb,
    int block_id, int pm) {
    constexpr int NP = kPnNP16;
    constexpr int STAGE_BYTES = kPnCPS16 * kPnUPC16 * 1024;
    constexpr int N_STAGES = kCt16 / kPnCPS16;
    constexpr int A2_BYTES = 16 * 1024;
    const int wpc = kPnWaves / cpb;                       // waves per cloud

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w3s = smem;                                                         // [2][STAGE_BYTES]
    char* a2s = smem + 2 * STAGE_BYTES;                                       // conv2 weights, fragment order
    float* a1s = reinterpret_cast<float*>(a2s + A2_BYTES);                    // [64][4]
    float* c1s = a1s + kC1 * 4;                                               // [64]
    float* c2s = c1s + kC1;                                                   // [128]
    float* bmax = c2s + kC2;                                                  // [cpb][1024] running max

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15;
    const int q = lane >> 4;
    const int cl = wave / wpc;                            // this wave's cloud within the block
    const int lw = wave - cl * wpc;
    int b = block_id * cpb + cl;
    if (b >= B) b = B - 1;                                // a surplus slot recomputes the last cloud; its result is not stored
    float* wmax = bmax + cl * kC3;

    for (int i = tid; i < kC1 * 4; i += kPnThreads) {
        const int ch = i >> 2, cc = i & 3;
        a1s[i] = cc < C ? a1[ch * C + cc] : 0.f;
    }
    for (int i = tid; i < kC1; i += kPnThreads) c1s[i] = c1[i];
    for (int i = tid; i < kC2; i += kPnThreads) c2s[i] = c2[i];
    for (int i = tid; i < cpb * kC3; i += kPnThreads) bmax[i] = -__builtin_inff();
    // conv2 weights A2 [128][64] -> unit (t, s): lane (cc, qq) holds A2[16t + cc][32s + 8qq .. +8]
    for (int i = tid; i < 16 * 64; i += kPnThreads) {
        const int l = i & 63, u = i >> 6;
        const int cc = l & 15, qq = l >> 4;
        const int t = u >> 1, s2 = u & 1;
        const float* src = a2 + (16 * t + cc) * kC1 + 32 * s2 + 8 * qq;
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = f32_to_bf16_bits(src[j]);
        uint4 v;
        v.x = e[0] | (e[1] << 16); v.y = e[2] | (e[3] << 16); v.z = e[4] | (e[5] << 16); v.w = e[6] | (e[7] << 16);
        *reinterpret_cast<uint4*>(a2s + i * 16) = v;
    }
    float T[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) T[i] = trans ? trans[b * 9 + i] : ((i % 4) == 0 ? 1.f : 0.f);
    const bool has_trans = trans != nullptr;
    __syncthreads();

    const float* xb = x + (int64_t)b * C * N;
    const int pts_per_chunk = wpc * NP * 16;
    const int n_chunks = (N + pts_per_chunk - 1) / pts_per_chunk;
    const int n_tiles_cloud = (N + 15) / 16;

    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        {
            constexpr int UNITS = STAGE_BYTES / 1024;
            for (int u = wave; u < UNITS; u += kPnWaves) glds16(w3p + (int64_t)u * 1024 + lane * 16, w3s + u * 1024);
        }
        // tiles are dealt to the waves round-robin, so the tiles past the end of the cloud fall into the higher slots
        bf16x8 X[NP][4];
        const int tile0 = chunk * (wpc * NP) + lw;
        int nv = 0;
#pragma unroll
        for (int np = 0; np < NP; ++np) nv += (tile0 + np * wpc < n_tiles_cloud) ? 1 : 0;
#define DVQ_PN_TILE16(np)                                                                              \
    if (np < nv) {                                                                                     \
        int n = (tile0 + np * wpc) * 16 + c;                                                           \
        if (n >= N) n = N - 1; /* duplicate the last point: a duplicate cannot change a max */         \
        pn_point_tile16<C>(xb, N, pm, n, has_trans, T, a1s, c1s, a2s, c2s, lane, q, X[np]);                \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    }
        DVQ_PN_TILE16(0) DVQ_PN_TILE16(1) DVQ_PN_TILE16(2) DVQ_PN_TILE16(3)
#undef DVQ_PN_TILE16
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own direct-to-LDS copies landed before they are published
        __syncthreads();  // stage 0 has landed

#pragma unroll 1
        for (int st = 0; st < N_STAGES; ++st) {
            const int buf = st & 1;
            if (st + 1 < N_STAGES) {
                constexpr int UNITS = STAGE_BYTES / 1024;
                const char* src = w3p + (int64_t)(st + 1) * STAGE_BYTES;
                char* dst = w3s + (buf ^ 1) * STAGE_BYTES;
                for (int u = wave; u < UNITS; u += kPnWaves) glds16(src + (int64_t)u * 1024 + lane * 16, dst + u * 1024);
            }
            const char* sb = w3s + buf * STAGE_BYTES + lane * 16;
            // conv3 of the stage's four channel tiles for the first NV point tiles of this wave.  The four maxima go to
            // LDS together after the last tile: an LDS atomic between the tiles would pin the order of the weight
            // reads around it (and the compiler drains the in-flight weight copies before each one).
            auto conv3_stage = [&](auto nvc) {
                constexpr int NV = decltype(nvc)::value;
                float m[kPnCPS16];
#pragma unroll
                for (int cc = 0; cc < kPnCPS16; ++cc) {
                    f32x4 acc[NV];
#pragma unroll
                    for (int np = 0; np < NV; ++np) acc[np] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const bf16x8 w = *reinterpret_cast<const bf16x8*>(sb + (cc * kPnUPC16 + ks) * 1024);
#pragma unroll
                        for (int np = 0; np < NV; ++np) acc[np] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X[np][ks], w, acc[np], 0, 0, 0);
                    }
                    float mm = acc[0][0];
#pragma unroll
                    for (int np = 0; np < NV; ++np)
#pragma unroll
                        for (int i = 0; i < 4; ++i) mm = fmaxf(mm, acc[np][i]);
                    m[cc] = mm;
                }
                // all 64 lanes fold into the block's running maximum: the four lanes of a channel (one per k-group, each
                // with its own four points of every tile) meet in the LDS atomic unit, which costs less than two lane
                // exchanges and the wait for them
#pragma unroll
                for (int cc = 0; cc < kPnCPS16; ++cc)
                    __builtin_amdgcn_ds_fmaxf((__attribute__((address_space(3))) float*)(wmax + (st * kPnCPS16 + cc) * 16 + c), m[cc], 0, 0, false);
            };
            if (nv == 4) conv3_stage(std::integral_constant<int, 4>{});
            else if (nv == 3) conv3_stage(std::integral_constant<int, 3>{});
            else if (nv == 2) conv3_stage(std::integral_constant<int, 2>{});
            else if (nv == 1) conv3_stage(std::integral_constant<int, 1>{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own direct-to-LDS copies landed before they are published
            __syncthreads();
        }
    }

    for (int i = tid; i < cpb * kC3; i += kPnThreads) {
        const int cloud = block_id * cpb + i / kC3, ch = i % kC3;
        if (cloud >= B) break;
        float v = bmax[i] + c3[ch];
        if (relu3) v = v > 0.f ? v : 0.f;
        feat[(int64_t)cloud * kC3 + ch] = v;
    }
}

template <int C>
__global__ __launch_bounds__(kPnThreads, 4) void pointnet_mlp_max_bf16_kernel(
    const float* __restrict__ x, const float* __restrict__ trans, int N, const float* __restrict__ a1,
    const float* __restrict__ c1, const float* __restrict__ a2, const float* __restrict__ c2,
    const char* __restrict__ w3p, const float* __restrict__ c3, int relu3, float* __restrict__ feat, int B, int cpb, int pm) {
    pn16_body<C>(x, trans, N, a1, c1, a2, c2, w3p, c3, relu3, feat, B, cpb, blockIdx.x, pm);
}
template <int C>
__global__ __launch_bounds__(kPnThreads, 4) void pointnet_mlp_max_bf16_grouped_kernel(PnGroupsDev g) {
    const int k = pn_group_of(g, blockIdx.x);
    pn16_body<C>(g.x[k], g.trans[k], g.N[k], g.a1[k], g.c1[k], g.a2[k], g.c2[k], g.w3p[k], g.c3[k], g.relu3[k], g.feat[k], g.B[k],
                 g.cpb[k], blockIdx.x - g.block_start[k], 0);
}

template <int DT>
static size_t pn_lds_bytes(int cpb) {
    typedef PnCfg<DT> Cfg;
    return 2 * (size_t)Cfg::CPS * Cfg::UPC * 1024 + (size_t)Cfg::A2_UNITS * 1024 + (kC1 * 4 + kC1 + kC2) * 4 + (size_t)cpb * kC3 * 4;
}

static int pn_cpb(int N, int pts_per_wave) {
    // clouds per workgroup: as many as fit a chunk of 8 waves x pts_per_wave points, at most 4 (LDS for the running maxima)
    int cpb = 1;
    while (cpb < 4 && N <= (kPnWaves / (2 * cpb)) * pts_per_wave) cpb *= 2;
    return cpb;
}
static size_t pn_lds_bytes16(int cpb) {
    return 2 * (size_t)kPnCPS16 * kPnUPC16 * 1024 + 16 * 1024 + (kC1 * 4 + kC1 + kC2) * 4 + (size_t)cpb * kC3 * 4;
}

template <int C>
static int pn_launch16(const float* x, const float* trans, int B, int N, int pm, const DvqPointMlp* w, int relu3, float* feat, hipStream_t st) {
    auto kern = pointnet_mlp_max_bf16_kernel<C>;
    const int cpb = pn_cpb(N, kPnNP16 * 16);
    const size_t lds = pn_lds_bytes16(cpb);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "pointnet_mlp_max")) return rc;
    hipLaunchKernelGGL(kern, dim3((B + cpb - 1) / cpb), dim3(kPnThreads), lds, st, x, trans, N, w->a1, w->c1, w->a2, w->c2,
                       reinterpret_cast<const char*>(w->w3p), w->c3, relu3, feat, B, cpb, pm);
    return check_launch("pointnet_mlp_max");
}

template <int DT, int C>
static int pn_launch(const float* x, const float* trans, int B, int N, int pm, const DvqPointMlp* w, int relu3, float* feat, hipStream_t st) {
    auto kern = pointnet_mlp_max_kernel<DT, C>;
    const int cpb = pn_cpb(N, PnCfg<DT>::NP * 32);
    const size_t lds = pn_lds_bytes<DT>(cpb);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "pointnet_mlp_max")) return rc;
    hipLaunchKernelGGL(kern, dim3((B + cpb - 1) / cpb), dim3(kPnThreads), lds, st, x, trans, N, w->a1, w->c1, w->a2, w->c2,
                       reinterpret_cast<const char*>(w->w3p), w->c3, relu3, feat, B, cpb, pm);
    return check_launch("pointnet_mlp_max");
}

// Vertex subsets of a point set, all parts in one launch: part k of cloud b is written as a contiguous [3, n_k] block
// of the k-th output segment.  With `center` the per-axis mean over ALL V points is subtracted first (DVQVAE.py:49-51).
__global__ __launch_bounds__(256) void gather_parts_kernel(const float* __restrict__ pts, int B, int V, const int* __restrict__ idx,
                                                           const int* __restrict__ part_ptr, int P, int center,
                                                           float* __restrict__ out) {
    __shared__ float red[3][4];
    __shared__ float mean[3];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = pts + (int64_t)b * 3 * V;
    if (center) {
        float acc[3] = {0.f, 0.f, 0.f};
        for (int v = tid; v < V; v += 256)
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[a] += src[a * V + v];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float t = acc[a];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane == 0) red[a][wave] = t;
        }
        __syncthreads();
        if (tid < 3) mean[tid] = (red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3]) / (float)V;
    } else if (tid < 3) {
        mean[tid] = 0.f;
    }
    __syncthreads();
    const int total = part_ptr[P];
    for (int t = tid; t < 3 * total; t += 256) {
        const int a = t / total, j = t - a * total;
        int k = 0;
        while (k + 1 < P && j >= part_ptr[k + 1]) ++k;
        const int n_k = part_ptr[k + 1] - part_ptr[k];
        out[(int64_t)3 * B * part_ptr[k] + ((int64_t)b * 3 + a) * n_k + (j - part_ptr[k])] = src[a * V + idx[j]] - mean[a];
    }
}

}  // namespace dvq

using namespace dvq;

extern "C" size_t dvq_pointnet_w3_bytes(int dtype) {
    if (dtype == DVQ_BF16) return (size_t)kC3 * kC2 * 2;
    if (dtype == DVQ_F32) return (size_t)kC3 * kC2 * 4;
    set_error("pointnet: unknown dtype %d", dtype);
    return 0;
}

extern "C" int dvq_pointnet_pack_w3(const float* a3, int dtype, void* w3p, void* stream) {
    if (!a3 || !w3p) return fail(DVQ_EINVAL, "pointnet_pack_w3: null pointer");
    if (((uintptr_t)a3 & 15) || ((uintptr_t)w3p & 15)) return fail(DVQ_EINVAL, "pointnet_pack_w3: pointers must be 16-byte aligned");
    if (dtype == DVQ_BF16) {
        const int total = kCt16 * kPnUPC16 * 64;
        hipLaunchKernelGGL(pn_pack_w3_16_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), a3, reinterpret_cast<uint4*>(w3p));
    } else if (dtype == DVQ_F32) {
        const int total = kCt * PnCfg<DVQ_F32>::UPC * 64;
        hipLaunchKernelGGL((pn_pack_w3_kernel<DVQ_F32>), dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), a3,
                           reinterpret_cast<uint4*>(w3p));
    } else {
        return fail(DVQ_EINVAL, "pointnet_pack_w3: unknown dtype %d", dtype);
    }
    return check_launch("pointnet_pack_w3");
}

extern "C" int dvq_pointnet_mlp_max(const float* x, const float* trans, int B, int C, int N, int point_major, const DvqPointMlp* w,
                                    int relu3, int dtype, float* feat_out, void* stream) {
    if (B < 0 || N < 1) return fail(DVQ_EINVAL, "pointnet: bad shape B=%d N=%d", B, N);
    if (C != 3 && C != 4) return fail(DVQ_EINVAL, "pointnet: C must be 3 or 4 (got %d)", C);
    if (B == 0) return DVQ_OK;
    if (!x || !w || !feat_out) return fail(DVQ_EINVAL, "pointnet: null pointer");
    if (!w->a1 || !w->c1 || !w->a2 || !w->c2 || !w->w3p || !w->c3) return fail(DVQ_EINVAL, "pointnet: null weight pointer");
    if (((uintptr_t)w->a2 & 15) || ((uintptr_t)w->w3p & 15)) return fail(DVQ_EINVAL, "pointnet: a2 and w3p must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const int pm = point_major ? 1 : 0;
    if (dtype == DVQ_BF16) {
        return C == 3 ? pn_launch16<3>(x, trans, B, N, pm, w, relu3, feat_out, st) : pn_launch16<4>(x, trans, B, N, pm, w, relu3, feat_out, st);
    } else if (dtype == DVQ_F32) {
        return C == 3 ? pn_launch<DVQ_F32, 3>(x, trans, B, N, pm, w, relu3, feat_out, st)
                      : pn_launch<DVQ_F32, 4>(x, trans, B, N, pm, w, relu3, feat_out, st);
    }
    return fail(DVQ_EINVAL, "pointnet: unknown dtype %d", dtype);
}

extern "C" int dvq_pointnet_mlp_max_grouped(const DvqPointGroup* groups, int n_groups, int C, int dtype, void* stream) {
    if (n_groups < 0 || n_groups > kPnMaxGroups) return fail(DVQ_EINVAL, "pointnet (grouped): 0..%d groups, got %d", kPnMaxGroups, n_groups);
    if (C != 3 && C != 4) return fail(DVQ_EINVAL, "pointnet: C must be 3 or 4 (got %d)", C);
    if (dtype != DVQ_BF16 && dtype != DVQ_F32) return fail(DVQ_EINVAL, "pointnet: unknown dtype %d", dtype);
    if (n_groups && !groups) return fail(DVQ_EINVAL, "pointnet: null pointer");
    PnGroupsDev g;
    g.n = 0;
    int blocks = 0, max_cpb = 1;
    const int pts_per_wave = dtype == DVQ_BF16 ? kPnNP16 * 16 : PnCfg<DVQ_F32>::NP * 32;
    for (int i = 0; i < n_groups; ++i) {
        const DvqPointGroup& s = groups[i];
        if (s.B < 0 || s.N < 1) return fail(DVQ_EINVAL, "pointnet: bad shape B=%d N=%d in group %d", s.B, s.N, i);
        if (s.B == 0) continue;
        if (!s.x || !s.feat_out) return fail(DVQ_EINVAL, "pointnet: null pointer in group %d", i);
        const DvqPointMlp& w = s.w;
        if (!w.a1 || !w.c1 || !w.a2 || !w.c2 || !w.w3p || !w.c3) return fail(DVQ_EINVAL, "pointnet: null weight pointer in group %d", i);
        if (((uintptr_t)w.a2 & 15) || ((uintptr_t)w.w3p & 15)) return fail(DVQ_EINVAL, "pointnet: a2 and w3p must be 16-byte aligned");
        const int k = g.n++;
        g.x[k] = s.x; g.trans[k] = s.trans; g.a1[k] = w.a1; g.c1[k] = w.c1; g.a2[k] = w.a2; g.c2[k] = w.c2;
        g.w3p[k] = reinterpret_cast<const char*>(w.w3p); g.c3[k] = w.c3; g.feat[k] = s.feat_out;
        g.N[k] = s.N; g.B[k] = s.B; g.relu3[k] = s.relu3;
        g.cpb[k] = pn_cpb(s.N, pts_per_wave);
        if (g.cpb[k] > max_cpb) max_cpb = g.cpb[k];
        g.block_start[k] = blocks;
        blocks += (s.B + g.cpb[k] - 1) / g.cpb[k];
    }
    if (g.n == 0) return DVQ_OK;
    g.block_start[g.n] = blocks;
    for (int k = g.n; k < kPnMaxGroups; ++k) {   // unused slots: defined values
        g.x[k] = nullptr; g.trans[k] = nullptr; g.a1[k] = g.c1[k] = g.a2[k] = g.c2[k] = g.c3[k] = nullptr; g.w3p[k] = nullptr;
        g.feat[k] = nullptr; g.N[k] = g.B[k] = g.relu3[k] = 0; g.cpb[k] = 1;
        if (k > g.n) g.block_start[k] = blocks;
    }
    hipStream_t st = as_stream(stream);
#define DVQ_PN_GROUPED(KERN, LDS)                                                                            \
    do {                                                                                                     \
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(KERN), LDS, "pointnet_mlp_max")) return rc; \
        hipLaunchKernelGGL(KERN, dim3(blocks), dim3(kPnThreads), LDS, st, g);                                \
    } while (0)
    if (dtype == DVQ_BF16) {
        const size_t lds = pn_lds_bytes16(max_cpb);
        if (C == 3) DVQ_PN_GROUPED(pointnet_mlp_max_bf16_grouped_kernel<3>, lds);
        else DVQ_PN_GROUPED(pointnet_mlp_max_bf16_grouped_kernel<4>, lds);
    } else {
        const size_t lds = pn_lds_bytes<DVQ_F32>(max_cpb);
        if (C == 3) DVQ_PN_GROUPED((pointnet_mlp_max_grouped_kernel<DVQ_F32, 3>), lds);
        else DVQ_PN_GROUPED((pointnet_mlp_max_grouped_kernel<DVQ_F32, 4>), lds);
    }
#undef DVQ_PN_GROUPED
    return check_launch("pointnet_mlp_max(grouped)");
}

extern "C" int dvq_gather_parts(const float* pts, int B, int V, const int* idx, const int* part_ptr, int n_parts, int center,
                                float* out, void* stream) {
    if (B < 0 || V < 1 || n_parts < 1) return fail(DVQ_EINVAL, "gather_parts: bad shape B=%d V=%d parts=%d", B, V, n_parts);
    if (B == 0) return DVQ_OK;
    if (!pts || !idx || !part_ptr || !out) return fail(DVQ_EINVAL, "gather_parts: null pointer");
    hipLaunchKernelGGL(gather_parts_kernel, dim3(B), dim3(256), 0, as_stream(stream), pts, B, V, idx, part_ptr, n_parts, center, out);
    return check_launch("gather_parts");
}
