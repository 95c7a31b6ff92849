// Matrix products of the GatedPixelCNN prior with fused gate / residual epilogues (gfx950).
//
// The prior (DVQ-VAE/network/pixelcnn/models.py:65-88, 161-198), evaluated cell by cell,
// is a chain of [batch, K] x [K, N] products: masked-convolution taps, 1x1 projections and
// the output head.  They run here in the transposed domain, out^T = W * x^T, on bf16 MFMA
// 32x32x16 with fp32 accumulation: the WEIGHTS are the A operand (rows = output feature)
// and the ACTIVATIONS the B operand (columns = sample).  The accumulator then holds a
// sample on the lane and 16 output features in the registers, which is exactly the
// B-operand layout of the next product of the chain (accumulator-as-operand), so
// activations are stored between kernels as "packed" 1-KiB units
//     unit(fb, sb, s) = 64 lanes x 8 bf16:  lane (c,h), j -> feature 32fb + 16s + 8(j>>2) + 4h + (j&3)
//                                                          of sample 32sb + c
// written and read as contiguous 16 B per lane (fully coalesced), never re-laid.
// Weights are packed once per checkpoint into A-fragment units with the same k order.
//
// One product sums over several (activation, weight) pairs ("taps": kernel positions of a
// masked convolution, or the horizontal stack's inputs) before its epilogue:
//   PACK  y = acc + bias (+ residual) (relu)          -> packed bf16
//   GATE  tanh(y[n]) * sigmoid(y[n+N/2]), y = p + class_embedding[label], p = acc + bias (+ pre_add[label])
//                                                      -> packed bf16 (N/2 features),
//         optionally also p (y without the class term) -> packed bf16 (N features)
//   ROWS  y = acc + bias (relu)                        -> fp32 [M, N] row-major (logits; heads)
//
// Tiling: a workgroup owns 4 sample tiles x 2..4 output-feature tiles; a wave multiplies NO weight
// tiles (shared through LDS) with one sample tile (four waves, or eight in two feature groups).
#include "common.h"

#ifndef DVQ_PG_RING8
#define DVQ_PG_RING8 3
#endif
namespace dvq {

constexpr int kPgWaves = 4;
constexpr int kPgRing = 3;      // stage buffers
constexpr int kPgMaxTaps = 10;

struct PgTapDev {
    const char* x;   // packed activations
    const char* w;   // packed weights of this tap
    int kb;          // k-blocks of 32 features
};
struct PgJobDev {
    PgTapDev taps[kPgMaxTaps];
    int n_taps;
    const float* bias;
    const float* cond;
    const int64_t* label;
    const char* resid;
    char* out_packed;
    char* out_pre;
    float* out_rows;
    const float* pre_add;
};
struct PgJobs {
    PgJobDev j[3];
};

__device__ __forceinline__ void pg_glds16(const void* gsrc_lane, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

__device__ __forceinline__ uint4 pack8(const float* v) {
    unsigned short e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = f32_to_bf16_bits(v[j]);
    uint4 o;
    o.x = e[0] | (e[1] << 16); o.y = e[2] | (e[3] << 16); o.z = e[4] | (e[5] << 16); o.w = e[6] | (e[7] << 16);
    return o;
}

enum { PG_PACK = 0, PG_GATE = 1, PG_ROWS = 2 };

// The same kernels run in exact fp32 (the parity configuration; DT = DVQ_F32) on the 32x32x2 f32 MFMA: a 1-KiB unit
// then holds 8 features instead of 16 (four floats per lane = four MFMA k-steps),
//     unit(fb, sb, q) = 64 lanes x 4 floats:  lane (c,h), j -> feature 32fb + 8q + 4h + j of sample 32sb + c,
// which is registers 4q .. 4q+3 of a 32x32 accumulator tile, so the accumulator is still stored as is.
template <int DT> struct PgT;
template <> struct PgT<DVQ_BF16> { static constexpr int UPB = 2; };   // units per block of 32 features
template <> struct PgT<DVQ_F32> { static constexpr int UPB = 4; };

__device__ __forceinline__ float pg_tanh(float x, bool exact) { return exact ? tanhf(x) : fast_tanh(x); }
__device__ __forceinline__ float pg_sigmoid(float x, bool exact) { return exact ? 1.f / (1.f + expf(-x)) : fast_sigmoid(x); }

// 32 features x 32 samples of a tile (y[16] of this lane) -> packed units at `base` (first unit of the block)
template <int DT>
__device__ __forceinline__ void pg_store_block(char* base, int lane, const float* y) {
    if (DT == DVQ_BF16) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) *reinterpret_cast<uint4*>(base + s2 * 1024 + lane * 16) = pack8(y + 8 * s2);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o; o[0] = y[4 * q]; o[1] = y[4 * q + 1]; o[2] = y[4 * q + 2]; o[3] = y[4 * q + 3];
            *reinterpret_cast<f32x4*>(base + q * 1024 + lane * 16) = o;
        }
    }
}
template <int DT>
__device__ __forceinline__ void pg_add_block(const char* base, int lane, float* y) {
    if (DT == DVQ_BF16) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const u16x8 rr = *reinterpret_cast<const u16x8*>(base + s2 * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) y[8 * s2 + j] += bf16_bits_to_f32(rr[j]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 rr = *reinterpret_cast<const f32x4*>(base + q * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) y[4 * q + j] += rr[j];
        }
    }
}

// Epilogue of one sample tile: acc[t] = tile of output-feature block ob[t] (for GATE: t < NO/2 the tanh half, the rest
// the matching sigmoid half).  Register i of a tile <-> feature 32*ob + (i&3) + 8*(i>>2) + 4h, lane c <-> sample 32*sb + c.
template <int DT, int EPI, int NO, bool ADD, typename Acc>
__device__ __forceinline__ void pg_epilogue_impl(const PgJobDev& job, const Acc& acc, const int* ob, int sb, int lane, int M,
                                                 int N, int SB, int relu) {
    constexpr int UPB = PgT<DT>::UPB;
    const int c = lane & 31, h = lane >> 5;
    int m = sb * 32 + c;
    const bool mvalid = m < M;
    if (!mvalid) m = M - 1;
    if (EPI == PG_GATE) {
        const float* crow = job.cond + job.label[m] * N;
        const float* arow = ADD ? job.pre_add + job.label[m] * N : nullptr;
#pragma unroll
        for (int t = 0; t < NO / 2; ++t) {
            float g[16], pre_lo[16], pre_hi[16];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int nlo = 32 * ob[t] + 8 * q4 + 4 * h, nhi = 32 * ob[t + NO / 2] + 8 * q4 + 4 * h;
                f32x4 blo = *reinterpret_cast<const f32x4*>(job.bias + nlo);
                f32x4 bhi = *reinterpret_cast<const f32x4*>(job.bias + nhi);
                if constexpr (ADD) { blo += *reinterpret_cast<const f32x4*>(arow + nlo); bhi += *reinterpret_cast<const f32x4*>(arow + nhi); }
                const f32x4 clo = *reinterpret_cast<const f32x4*>(crow + nlo);
                const f32x4 chi = *reinterpret_cast<const f32x4*>(crow + nhi);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 4 * q4 + q;
                    pre_lo[i] = acc[t][0][i] + blo[q];
                    pre_hi[i] = acc[t + NO / 2][0][i] + bhi[q];
                    g[i] = pg_tanh(pre_lo[i] + clo[q], DT == DVQ_F32) * pg_sigmoid(pre_hi[i] + chi[q], DT == DVQ_F32);
                }
            }
            pg_store_block<DT>(job.out_packed + ((int64_t)ob[t] * SB + sb) * UPB * 1024, lane, g);
            if (job.out_pre) {
                pg_store_block<DT>(job.out_pre + ((int64_t)ob[t] * SB + sb) * UPB * 1024, lane, pre_lo);
                pg_store_block<DT>(job.out_pre + ((int64_t)ob[t + NO / 2] * SB + sb) * UPB * 1024, lane, pre_hi);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NO; ++t) {
            float y[16];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int n0 = 32 * ob[t] + 8 * q4 + 4 * h;
                const f32x4 b = *reinterpret_cast<const f32x4*>(job.bias + n0);
#pragma unroll
                for (int q = 0; q < 4; ++q) y[4 * q4 + q] = acc[t][0][4 * q4 + q] + b[q];
                if (EPI == PG_ROWS) {
                    if (relu) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) y[4 * q4 + q] = y[4 * q4 + q] > 0.f ? y[4 * q4 + q] : 0.f;
                    }
                    if (mvalid) {
                        f32x4 o; o[0] = y[4 * q4]; o[1] = y[4 * q4 + 1]; o[2] = y[4 * q4 + 2]; o[3] = y[4 * q4 + 3];
                        *reinterpret_cast<f32x4*>(job.out_rows + (int64_t)m * N + n0) = o;
                    }
                }
            }
            if (EPI == PG_PACK) {
                const int64_t off = ((int64_t)ob[t] * SB + sb) * UPB * 1024;
                if (job.resid) pg_add_block<DT>(job.resid + off, lane, y);
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) y[j] = y[j] > 0.f ? y[j] : 0.f;
                }
                pg_store_block<DT>(job.out_packed + off, lane, y);
            }
        }
    }
}

// (two instantiations: with the optional per-class addend as a run-time branch the epilogue needed twice the registers)
template <int DT, int EPI, int NO, typename Acc>
__device__ __forceinline__ void pg_epilogue(const PgJobDev& job, const Acc& acc, const int* ob, int sb, int lane, int M, int N,
                                            int SB, int relu) {
    if (EPI == PG_GATE && job.pre_add) pg_epilogue_impl<DT, EPI, NO, true>(job, acc, ob, sb, lane, M, N, SB, relu);
    else pg_epilogue_impl<DT, EPI, NO, false>(job, acc, ob, sb, lane, M, N, SB, relu);
}

// Small batches (the reference samples one grid at a time; used up to 512 rows in bf16, 1024 in fp32): the product is
// a stream of weights with little arithmetic, so it is cut along k instead of along the samples.  A workgroup of
// eight waves owns one sample tile and one output-feature block (GATE: a block and its partner in the upper half);
// wave w takes the 1-KiB units
// u = w, w+8, ... of every tap -- weights and activations of a tap are both contiguous in u -- straight from
// global memory into MFMA operands, the eight partial tiles meet in LDS and are summed in wave order (the same
// order every time), and the first wave runs the epilogue.  N/32 (N/64) workgroups per job stream the weights
// instead of the handful that whole 128-feature tiles would give.
constexpr int kPgSmallWaves = 8;
template <int DT, int EPI>
__global__ __launch_bounds__(kPgSmallWaves * 64) void prior_gemm_small_kernel(PgJobs jobs, int M, int N, int relu) {
    constexpr int UPB = PgT<DT>::UPB;
    constexpr int NO = EPI == PG_GATE ? 2 : 1;
    __shared__ float part[kPgSmallWaves][NO][16][64];
    const PgJobDev& job = jobs.j[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_ob = N / 32;
    const int SB = gridDim.y, sb = blockIdx.y;               // one sample tile per workgroup
    int ob[NO];
    ob[0] = blockIdx.x;
    if (NO == 2) ob[1] = n_ob / 2 + blockIdx.x;
    f32x16 acc[NO][1];
#pragma unroll
    for (int t = 0; t < NO; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][0][i] = 0.f;
    for (int tp = 0; tp < job.n_taps; ++tp) {
        const int units = job.taps[tp].kb * UPB;
        // activation unit (k-block, q) of this sample tile sits at ((k-block * SB + sb) * UPB + q)
        const char* x = job.taps[tp].x + ((int64_t)sb * UPB) * 1024 + lane * 16;
        const int64_t xstep = (int64_t)(SB - 1) * UPB;       // extra units skipped per k-block
        const char* w0 = job.taps[tp].w + (int64_t)ob[0] * units * 1024 + lane * 16;
        const char* w1 = job.taps[tp].w + (int64_t)ob[NO - 1] * units * 1024 + lane * 16;
#pragma unroll 4
        for (int u = wave; u < units; u += kPgSmallWaves) {
            if constexpr (DT == DVQ_BF16) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(x + ((int64_t)u + (u / UPB) * xstep) * 1024);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(w0 + (int64_t)u * 1024), xf, acc[0][0], 0, 0, 0);
                if (NO == 2)
                    acc[NO - 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(w1 + (int64_t)u * 1024), xf, acc[NO - 1][0], 0, 0, 0);
            } else {
                const f32x4 xf = *reinterpret_cast<const f32x4*>(x + ((int64_t)u + (u / UPB) * xstep) * 1024);
                const f32x4 wa = *reinterpret_cast<const f32x4*>(w0 + (int64_t)u * 1024);
                f32x4 wb = wa;
                if (NO == 2) wb = *reinterpret_cast<const f32x4*>(w1 + (int64_t)u * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], xf[j], acc[0][0], 0, 0, 0);
                    if (NO == 2) acc[NO - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[j], xf[j], acc[NO - 1][0], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NO; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) part[wave][t][i][lane] = acc[t][0][i];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int t = 0; t < NO; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = part[0][t][i][lane];
#pragma unroll
            for (int w = 1; w < kPgSmallWaves; ++w) v += part[w][t][i][lane];
            acc[t][0][i] = v;
        }
    pg_epilogue<DT, EPI, NO>(job, acc, ob, sb, lane, M, N, SB, relu);
}

// grid: x = output-feature group (FG x NO tiles of 32; for GATE half of them + the matching half of the upper
//       half), y = group of 4 sample tiles, z = job.  4 x FG waves: wave (ws, wf) owns sample tile ws and the NO
//       tiles of feature group wf.
//
// Pipeline: every operand byte reaches LDS by direct global->LDS loads (no register-destination
// load inside the loop), in stages of KS k-blocks (weights FG x NO x KS x 2 KiB, activations 4 x KS x 2 KiB,
// dealt to the waves unit by unit) through a ring of R stage buffers.  A wave waits only for ITS
// loads of the stage it is about to read with a counted s_waitcnt vmcnt (the younger stages stay
// in flight), then one s_barrier makes every wave's part visible and frees the buffer read R-1
// stages ago, which is immediately refilled.
// A 128 x 128 output tile per CU is all a [4096 x 1024] product leaves for each of the 256
// CUs (64 FLOP per byte moved from L2).  No unit is saturated (matrix pipe 17 %, L2 request rate 25 % of its
// peak, waves waiting 41 % of their time by the counters): the loop is bound by the latency of a stage's
// barrier and load wait, which a second wave per SIMD hides and a deeper ring does not (R = 4, 5: +-1 %).
// Four waves with 128 features each (one wave per SIMD) are slower than two four-wave workgroups of 64
// features per CU although those move 1.5x the bytes; eight waves of 64 features in ONE workgroup (FG = 2)
// keep two waves per SIMD and share the activations: 6.86 -> 6.63 ms per 4096 grids.
// bf16 stages hold two k-blocks (half as many barriers per product: 7.19 -> 6.92 ms per 4096 grids),
// fp32 stages one (the same bytes).
template <int DT, int EPI, int NO, int KS, int FG>
__global__ __launch_bounds__(4 * FG * 64) void prior_gemm_kernel(PgJobs jobs, int M, int N, int relu) {
    // These products form a long dependent chain of short launches; when another stream's long-running kernel
    // (the point-cloud encoder) shares the CU, win the SIMD's issue arbitration so the chain keeps its pace.
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PgJobDev& job = jobs.j[blockIdx.z];
    constexpr int WV = 4 * FG;                                   // waves: 4 sample tiles x FG feature groups
    constexpr int UPB = PgT<DT>::UPB;
    constexpr int KU = KS * UPB;                                 // units per operand tile and stage
    constexpr int W_UNITS = FG * NO * KU;
    constexpr int X_UNITS = 4 * KU;
    constexpr int STAGE_BYTES = (W_UNITS + X_UNITS) * 1024;
    constexpr int R = FG == 2 ? DVQ_PG_RING8 : kPgRing;
    static_assert(W_UNITS % WV == 0 && X_UNITS % WV == 0, "a stage's units are dealt evenly to the waves");
    constexpr int GLDS_PER_STAGE = (W_UNITS + X_UNITS) / WV;     // per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wave & 3, wf = wave >> 2;                     // this wave's sample tile and feature group
    const int SB = (M + 31) / 32;
    const int n_ob = N / 32;

    // column tile bx (weights shared along it), row tile by (activations shared along it), dealt XCD-aware
    int bx, by;
    xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y, bx, by);
    // output-feature block of tile t of feature group g (GATE: the first half of a group's tiles are tanh blocks,
    // the second half their sigmoid partners in the upper half of the features)
    auto block_of = [&](int g, int t) {
        if (EPI == PG_GATE) return (t < NO / 2 ? 0 : n_ob / 2) + (FG * NO / 2) * bx + g * (NO / 2) + t % (NO / 2);
        return FG * NO * bx + g * NO + t;
    };
    int ob[NO];
#pragma unroll
    for (int t = 0; t < NO; ++t) ob[t] = block_of(wf, t);
    int sbw = by * 4 + ws;
    const bool live = sbw < SB;
    if (!live) sbw = SB - 1;           // a surplus tile recomputes the last one; its stores are skipped

    f32x16 acc[NO][1];
#pragma unroll
    for (int t = 0; t < NO; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][0][i] = 0.f;

    int total_stages = 0;
    for (int t = 0; t < job.n_taps; ++t) total_stages += job.taps[t].kb / KS;

    // operands of the tap whose stages are being issued (fetched from the job when the tap changes, not per stage)
    int itap = 0, ikb = 0;   // next stage to issue
    const char* tw = job.taps[0].w;
    const char* tx = job.taps[0].x;
    int tkb = job.taps[0].kb;
    // one stage: W units [group][t][k][s2] then X units [sample tile][k][s2], dealt to the waves unit by unit
    auto issue = [&](int buf) {
        char* dst = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < W_UNITS / WV; ++i) {
            const int u = wave + WV * i;
            const int gt = u / KU, q = u % KU;
            pg_glds16(tw + ((int64_t)(block_of(gt / NO, gt % NO) * tkb + ikb) * UPB + q) * 1024 + lane * 16, dst + u * 1024);
        }
#pragma unroll
        for (int i = 0; i < X_UNITS / WV; ++i) {
            const int u = wave + WV * i;
            const int tile = u / KU, q = u % KU;
            int sbt = by * 4 + tile;
            if (sbt >= SB) sbt = SB - 1;
            pg_glds16(tx + (((int64_t)(ikb + q / UPB) * SB + sbt) * UPB + (q % UPB)) * 1024 + lane * 16,
                      dst + (W_UNITS + u) * 1024);
        }
        ikb += KS;
        if (ikb >= tkb) {
            ikb = 0;
            ++itap;
            if (itap < job.n_taps) {
                tw = job.taps[itap].w; tx = job.taps[itap].x; tkb = job.taps[itap].kb;
                // the scalar loads complete here, at the tap change, not behind the next stage's fragment reads
                // (scalar loads and LDS reads share one counter)
                asm volatile("" ::"s"(tw), "s"(tx), "s"(tkb));
            }
        }
    };

    int issued = 0;
    for (; issued < R - 1 && issued < total_stages; ++issued) issue(issued % R);
    for (int st = 0; st < total_stages; ++st) {
        // this wave's loads of stage st have landed when at most `younger` stages' worth are outstanding
        const int younger = issued - st - 1;
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * GLDS_PER_STAGE) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(GLDS_PER_STAGE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // All fragment reads of the stage go out first, the refill of the buffer every wave finished reading before
        // this barrier is issued behind them (its issue time then covers the LDS latency), the products last.
        const char* sbuf = smem + (st % R) * STAGE_BYTES + lane * 16;
        if constexpr (DT == DVQ_BF16) {
            bf16x8 xf[KU], wfr[NO][KU];
#pragma unroll
            for (int q = 0; q < KU; ++q) {
                xf[q] = *reinterpret_cast<const bf16x8*>(sbuf + (W_UNITS + ws * KU + q) * 1024);
#pragma unroll
                for (int t = 0; t < NO; ++t) wfr[t][q] = *reinterpret_cast<const bf16x8*>(sbuf + ((wf * NO + t) * KU + q) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (issued < total_stages) { issue(issued % R); ++issued; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < KU; ++q)
#pragma unroll
                for (int t = 0; t < NO; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[t][q], xf[q], acc[t][0], 0, 0, 0);
        } else {
            f32x4 xf[KU], wfr[NO][KU];
#pragma unroll
            for (int q = 0; q < KU; ++q) {
                xf[q] = *reinterpret_cast<const f32x4*>(sbuf + (W_UNITS + ws * KU + q) * 1024);
#pragma unroll
                for (int t = 0; t < NO; ++t) wfr[t][q] = *reinterpret_cast<const f32x4*>(sbuf + ((wf * NO + t) * KU + q) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (issued < total_stages) { issue(issued % R); ++issued; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < KU; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < NO; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wfr[t][q][j], xf[q][j], acc[t][0], 0, 0, 0);
        }
    }

    if (live) pg_epilogue<DT, EPI, NO>(job, acc, ob, sbw, lane, M, N, SB, relu);
}

// W [N, K] (fp32, element (n,k) at W[n*ldn + k*ldk]) -> packed A-fragment units
//   bf16 unit (ob, fb, s): lane (r,h), j -> W[32ob + r][32fb + 16s + 8(j>>2) + 4h + (j&3)]
//   fp32 unit (ob, fb, q): lane (r,h), j -> W[32ob + r][32fb + 8q + 4h + j]
template <int DT>
__global__ void prior_pack_weight_kernel(const float* __restrict__ W, int64_t ldn, int64_t ldk, int N, int K, uint4* __restrict__ out) {
    constexpr int UPB = PgT<DT>::UPB;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int KB = K / 32;
    const int64_t total = (int64_t)(N / 32) * KB * UPB * 64;
    if (t >= total) return;
    const int l = (int)(t & 63);
    const int64_t unit = t >> 6;
    const int s = (int)(unit % UPB);
    const int fb = (int)((unit / UPB) % KB);
    const int ob = (int)((unit / UPB) / KB);
    const int r = l & 31, h = l >> 5;
    const float* row = W + (int64_t)(32 * ob + r) * ldn;
    if (DT == DVQ_BF16) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = row[(int64_t)(32 * fb + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)) * ldk];
        out[t] = pack8(v);
    } else {
        uint4 o;
        o.x = __builtin_bit_cast(unsigned, row[(int64_t)(32 * fb + 8 * s + 4 * h + 0) * ldk]);
        o.y = __builtin_bit_cast(unsigned, row[(int64_t)(32 * fb + 8 * s + 4 * h + 1) * ldk]);
        o.z = __builtin_bit_cast(unsigned, row[(int64_t)(32 * fb + 8 * s + 4 * h + 2) * ldk]);
        o.w = __builtin_bit_cast(unsigned, row[(int64_t)(32 * fb + 8 * s + 4 * h + 3) * ldk]);
        out[t] = o;
    }
}

// One lane's 16 bytes of packed unit (fb, s) for a row of an fp32 table: features as in the activation layouts above.
template <int DT>
__device__ __forceinline__ uint4 pg_pack_row_piece(const float* src_row, int fb, int s, int h) {
    if (DT == DVQ_BF16) {
        const float* src = src_row + 32 * fb + 16 * s + 4 * h;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[8 * (j >> 2) + (j & 3)];
        return pack8(v);
    }
    const float* src = src_row + 32 * fb + 8 * s + 4 * h;
    uint4 o;
    o.x = __builtin_bit_cast(unsigned, src[0]); o.y = __builtin_bit_cast(unsigned, src[1]);
    o.z = __builtin_bit_cast(unsigned, src[2]); o.w = __builtin_bit_cast(unsigned, src[3]);
    return o;
}

// packed[fb][sb][s][lane] <- table[idx[m]][feature]   (embedding rows of sampled indices)
template <int DT>
__global__ void prior_embed_kernel(const int64_t* __restrict__ idx, int64_t idx_stride, const float* __restrict__ table, int V, int M, int d,
                                   uint4* __restrict__ out) {
    constexpr int UPB = PgT<DT>::UPB;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int SB = (M + 31) / 32;
    const int64_t total = (int64_t)(d / 32) * SB * UPB * 64;
    if (t >= total) return;
    const int l = (int)(t & 63);
    const int64_t unit = t >> 6;
    const int s = (int)(unit % UPB);
    const int sbk = (int)((unit / UPB) % SB);
    const int fb = (int)((unit / UPB) / SB);
    const int c = l & 31, h = l >> 5;
    int m = sbk * 32 + c;
    if (m >= M) m = M - 1;
    int64_t row = idx[(int64_t)m * idx_stride];
    if (row < 0) row = 0;
    if (row >= V) row = V - 1;
    out[t] = pg_pack_row_piece<DT>(table + row * d, fb, s, h);
}

// out_b(packed)[m] <- table_b(packed)[idx[m]] for a list of buffers in one launch: rows of packed activations that
// were computed once per class, handed out to the rows of a batch by their class (grid.y = buffer).
constexpr int kPgGatherMax = 96;
struct PgGatherOne {
    const uint4* table;
    uint4* out;
    int features;
};
struct PgGatherList {
    PgGatherOne e[kPgGatherMax];
};

template <int DT>
__global__ __launch_bounds__(256) void prior_gather_rows_kernel(PgGatherList list, const int64_t* __restrict__ idx, int M, int V) {
    constexpr int UPB = PgT<DT>::UPB;
    const PgGatherOne& g = list.e[blockIdx.y];
    const int SB = (M + 31) / 32, SBV = (V + 31) / 32;
    const int64_t total = (int64_t)(g.features / 32) * SB * UPB * 64;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(t & 63);
        const int64_t unit = t >> 6;
        const int s = (int)(unit % UPB);
        const int sbk = (int)((unit / UPB) % SB);
        const int fb = (int)((unit / UPB) / SB);
        const int c = l & 31, h = l >> 5;
        int m = sbk * 32 + c;
        if (m >= M) m = M - 1;                      // the unused rows of the last block repeat the last row
        int64_t row = idx[m];
        if (row < 0) row = 0;
        if (row >= V) row = V - 1;
        g.out[t] = g.table[(((int64_t)fb * SBV + (row >> 5)) * UPB + s) * 64 + h * 32 + (row & 31)];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Vertical stack of one grid row in one launch (kernel-3 layers): the three cells of the row read the same input
// activations (cell j takes column c of input row r through tap db = c - j) and every tap matrix serves two or three
// cells, so a workgroup that owns a 128-sample x 128-feature tile of ALL THREE cells loads each activation unit
// and each weight unit once per k-block instead of once per cell: 149 FLOP per byte moved from L2 against 64 for
// the per-cell product above, which is what bounds it.  Stage = one input row x one k-block: 3 columns of
// activations (4 sample tiles x 2 units each) + 3 taps of weights (4 feature tiles x 2 units each), 48 KiB, ring of three.
// The tile is shared by 4 x FG waves: wave (ws, wf) owns sample tile ws and the 4 / FG feature tiles of group wf.
struct PvDesc {
    const char* x[2][3];
    const char* w[2][3];
    int n_rows;
    int kb;
    const float* bias;
    const float* cond;
    const float* pre_add[3];   // per cell, optional
    const int64_t* label;
    char* out_packed[3];
    char* out_pre[3];
};

template <int FG>
__global__ __launch_bounds__(4 * FG * 64) void prior_vrow_kernel(PvDesc d, int M, int N) {
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WV = 4 * FG;                               // waves: 4 sample tiles x FG feature groups
    constexpr int NO = 4 / FG;                               // feature tiles per wave (half tanh, half sigmoid blocks)
    constexpr int W_UNITS = 3 * 4 * 2;                       // [tap][group * NO + t][q]
    constexpr int X_UNITS = 3 * 4 * 2;                       // [col][sample tile][q]
    constexpr int STAGE_BYTES = (W_UNITS + X_UNITS) * 1024;
    constexpr int R = 3;
    constexpr int PW = W_UNITS / WV, PX = X_UNITS / WV;      // copies per wave and stage
    constexpr int GLDS_PER_STAGE = PW + PX;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wave & 3, wf = wave >> 2;
    const int c = lane & 31, h = lane >> 5;
    const int SB = (M + 31) / 32;
    const int n_ob = N / 32, half = n_ob / 2;
    int bx, by;
    xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y, bx, by);
    // output-feature block of tile gt (0..3) of the workgroup: group gt / NO, its tile gt % NO
    auto block_of = [&](int gt) {
        const int g = gt / NO, t = gt % NO;
        return (t < NO / 2 ? 0 : half) + 2 * bx + g * (NO / 2) + t % (NO / 2);
    };
    int ob[NO];
#pragma unroll
    for (int t = 0; t < NO; ++t) ob[t] = block_of(wf * NO + t);
    int sb = by * 4 + ws;
    const bool live = sb < SB;
    if (!live) sb = SB - 1;
    const int KB = d.kb;

    f32x16 acc[3][NO];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int t = 0; t < NO; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][t][i] = 0.f;

    const int total_stages = d.n_rows * KB;
    // operand pointers of both input rows, held in registers (indexing the descriptor by the row inside the loop
    // would fetch them with scalar loads every stage, and their wait also waits for the LDS reads in flight)
    const char* w0[3] = {d.w[0][0], d.w[0][1], d.w[0][2]};
    const char* w1[3] = {d.w[1][0], d.w[1][1], d.w[1][2]};
    const char* x0[3] = {d.x[0][0], d.x[0][1], d.x[0][2]};
    const char* x1[3] = {d.x[1][0], d.x[1][1], d.x[1][2]};
    int irow = 0, ikb = 0, issued = 0;       // next stage to issue
    // piece i of the stage being issued: this wave's share of the weight units, then of the activation units
    auto piece = [&](int i) {
        char* dst = smem + (issued % R) * STAGE_BYTES;
        if (i < PW) {
            const int u = wave + WV * i;                     // u = (tap * 4 + group * NO + t) * 2 + q
            const int tap = u / 8, gt = (u / 2) % 4, q = u & 1;
            const char* wp = irow ? (tap == 0 ? w1[0] : tap == 1 ? w1[1] : w1[2]) : (tap == 0 ? w0[0] : tap == 1 ? w0[1] : w0[2]);
            pg_glds16(wp + ((int64_t)(block_of(gt) * KB + ikb) * 2 + q) * 1024 + lane * 16, dst + u * 1024);
        } else {
            const int u = wave + WV * (i - PW);              // u = (col * 4 + sample tile) * 2 + q
            const int col = u / 8, tile = (u / 2) % 4, q = u & 1;
            int sbt = by * 4 + tile;
            if (sbt >= SB) sbt = SB - 1;
            const char* xp = irow ? (col == 0 ? x1[0] : col == 1 ? x1[1] : x1[2]) : (col == 0 ? x0[0] : col == 1 ? x0[1] : x0[2]);
            pg_glds16(xp + (((int64_t)ikb * SB + sbt) * 2 + q) * 1024 + lane * 16, dst + (W_UNITS + u) * 1024);
        }
    };
    auto issued_one = [&]() {
        ++issued;
        if (++ikb == KB) { ikb = 0; ++irow; }
    };

    for (int s0 = 0; s0 < R - 1 && s0 < total_stages; ++s0) {
#pragma unroll
        for (int i = 0; i < GLDS_PER_STAGE; ++i) piece(i);
        issued_one();
    }
    for (int st = 0; st < total_stages; ++st) {
        if (issued - st - 1 >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(GLDS_PER_STAGE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool refill = issued < total_stages;   // into the buffer every wave finished reading before this barrier
        const char* sbuf = smem + (st % R) * STAGE_BYTES + lane * 16;
        // A stage is 2 x 12 weight fragments, each feeding two or three MFMAs.  The fragments of the first k-half are
        // read together, those of the second one per item under the first half's MFMAs; the twelve refill copies are
        // dealt between the items, so that neither the LDS latency nor the issue time of a copy (60-180 cycles each)
        // stands before the matrix pipe with nothing in it.
        auto xaddr = [&](int q, int col) { return sbuf + (W_UNITS + (col * 4 + ws) * 2 + q) * 1024; };
        auto waddr = [&](int q, int r) { return sbuf + (((r / NO) * 4 + wf * NO + r % NO) * 2 + q) * 1024; };   // r = tap * NO + t
        bf16x8 xa[3], xb[3], wa[3 * NO], wb[3 * NO];
#pragma unroll
        for (int col = 0; col < 3; ++col) xa[col] = *reinterpret_cast<const bf16x8*>(xaddr(0, col));
#pragma unroll
        for (int r = 0; r < 3 * NO; ++r) wa[r] = *reinterpret_cast<const bf16x8*>(waddr(0, r));
        __builtin_amdgcn_sched_barrier(0);
        if (refill) {
#pragma unroll
            for (int i = 0; i < 3 && i < GLDS_PER_STAGE; ++i) piece(i);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 3 * NO; ++r) {
            const int tap = r / NO, t = r % NO;
            if (r < 3) xb[r] = *reinterpret_cast<const bf16x8*>(xaddr(1, r));
            wb[r] = *reinterpret_cast<const bf16x8*>(waddr(1, r));
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int col = j + tap - 1;       // tap <-> column offset tap - 1: cell j reads column j + tap - 1
                if (col >= 0 && col < 3) acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[r], xa[col], acc[j][t], 0, 0, 0);
            }
            if ((r & 1) && 3 + (r >> 1) < GLDS_PER_STAGE && refill) piece(3 + (r >> 1));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 3 * NO; ++r) {
            const int tap = r / NO, t = r % NO;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int col = j + tap - 1;
                if (col >= 0 && col < 3) acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[r], xb[col], acc[j][t], 0, 0, 0);
            }
            if (3 + (3 * NO) / 2 + r < GLDS_PER_STAGE && refill) piece(3 + (3 * NO) / 2 + r);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (refill) issued_one();
    }

    if (!live) return;
    int m = sb * 32 + c;
    if (m >= M) m = M - 1;
    const int64_t lab = d.label[m];
    const float* crow = d.cond + lab * N;
    auto epilogue = [&](auto add_const) {
        constexpr bool ADD = decltype(add_const)::value;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float* arow = ADD ? d.pre_add[j] + lab * N : nullptr;
#pragma unroll
            for (int t = 0; t < NO / 2; ++t) {
                float g[16], pre_lo[16], pre_hi[16];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int nlo = 32 * ob[t] + 8 * q4 + 4 * h, nhi = 32 * ob[t + NO / 2] + 8 * q4 + 4 * h;
                    f32x4 blo = *reinterpret_cast<const f32x4*>(d.bias + nlo);
                    f32x4 bhi = *reinterpret_cast<const f32x4*>(d.bias + nhi);
                    if constexpr (ADD) { blo += *reinterpret_cast<const f32x4*>(arow + nlo); bhi += *reinterpret_cast<const f32x4*>(arow + nhi); }
                    const f32x4 clo = *reinterpret_cast<const f32x4*>(crow + nlo);
                    const f32x4 chi = *reinterpret_cast<const f32x4*>(crow + nhi);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = 4 * q4 + q;
                        pre_lo[i] = acc[j][t][i] + blo[q];
                        pre_hi[i] = acc[j][t + NO / 2][i] + bhi[q];
                        g[i] = fast_tanh(pre_lo[i] + clo[q]) * fast_sigmoid(pre_hi[i] + chi[q]);
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    *reinterpret_cast<uint4*>(d.out_packed[j] + (((int64_t)ob[t] * SB + sb) * 2 + s2) * 1024 + lane * 16) = pack8(g + 8 * s2);
                    if (d.out_pre[j]) {
                        *reinterpret_cast<uint4*>(d.out_pre[j] + (((int64_t)ob[t] * SB + sb) * 2 + s2) * 1024 + lane * 16) = pack8(pre_lo + 8 * s2);
                        *reinterpret_cast<uint4*>(d.out_pre[j] + (((int64_t)ob[t + NO / 2] * SB + sb) * 2 + s2) * 1024 + lane * 16) = pack8(pre_hi + 8 * s2);
                    }
                }
            }
        }
    };
    // (the per-class addend comes for all three cells or for none)
    if (d.pre_add[0]) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

// One grid cell of the sampling loop in a single launch (DVQ-VAE/network/pixelcnn/models.py:189-196): softmax of the
// logits row, renormalisation, inverse-CDF draw with the row's uniform u in [0,1), the index written into the grid and
// its embedding row written as the packed input of the next cell.  One wave per row; lane l owns classes
// [l*per, (l+1)*per) so the running sum is a lane-local prefix plus one wave scan.  Probabilities are formed in fp32 as
// the softmax does, the cumulative sum in fp64; index = number of cdf entries <= u, clamped to the last class.
constexpr int kPsMaxPer = 16;   // classes per lane: up to 1024 classes

template <int DT>
__global__ __launch_bounds__(256) void prior_sample_kernel(const float* __restrict__ logits, int n, int M,
                                                           const double* __restrict__ u, int64_t u_stride,
                                                           int64_t* __restrict__ x_out, int64_t x_stride,
                                                           const float* __restrict__ table, int d, uint4* __restrict__ emb_out) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= M) return;
    const int per = (n + 63) / 64;
    const float* row = logits + (int64_t)m * n;
    float v[kPsMaxPer];
    float mx = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < kPsMaxPer; ++i) {
        const int k = lane * per + i;
        v[i] = (i < per && k < n) ? row[k] : -__builtin_inff();
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kPsMaxPer; ++i) {
        v[i] = (i < per) ? expf(v[i] - mx) : 0.f;
        sum += v[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    float sum2 = 0.f;
#pragma unroll
    for (int i = 0; i < kPsMaxPer; ++i) {
        v[i] = v[i] / sum;
        sum2 += v[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum2 += __shfl_xor(sum2, o, 64);
    double local = 0.0;
    double c[kPsMaxPer];
#pragma unroll
    for (int i = 0; i < kPsMaxPer; ++i) {
        local += (double)(v[i] / sum2);
        c[i] = local;
    }
    // inclusive wave scan of the lane totals
    double incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const double before = incl - local;
    const double total = __shfl(incl, 63, 64);
    const double uu = u[(int64_t)m * u_stride];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < kPsMaxPer; ++i) {
        const int k = lane * per + i;
        if (i < per && k < n && (before + c[i]) / total <= uu) ++cnt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    const int idx = cnt < n - 1 ? cnt : n - 1;
    if (lane == 0) x_out[(int64_t)m * x_stride] = (int64_t)idx;
    if (emb_out) {   // packed[fb][sb][s][lane (c,h)] <- table[idx][32fb + 16s + 8(j>>2) + 4h + (j&3)]
        const int SB = (M + 31) / 32;
        const int sbk = m >> 5, cc = m & 31;
        const float* src_row = table + (int64_t)idx * d;
        constexpr int UPB = PgT<DT>::UPB;
        for (int piece = lane; piece < (d / 32) * UPB * 2; piece += 64) {
            const int fb = piece / (UPB * 2), s2 = (piece >> 1) % UPB, hh = piece & 1;
            const uint4 pk = pg_pack_row_piece<DT>(src_row, fb, s2, hh);
            // the last row also fills the unused rows of its 32-row block, as prior_embed does
            const int reps = (m == M - 1) ? (SB * 32 - m) : 1;
            for (int rr = 0; rr < reps; ++rr) emb_out[((int64_t)(fb * SB + sbk) * UPB + s2) * 64 + hh * 32 + cc + rr] = pk;
        }
    }
}

// u[i][c] in [0,1): splitmix64 of the counter (first_id + i) * 1024 + c + seed * golden ratio, mixed twice; the top 53
// bits.  The draw of an object's cell is a pure function of (seed, global object id, cell), whatever the batch or rank.
__device__ __forceinline__ unsigned long long pg_splitmix(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void prior_uniforms_kernel(long long first_id, long long n, int n_cells, unsigned long long offset, double* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n_cells) return;
    const unsigned long long ctr = (unsigned long long)(first_id + t / n_cells) * 1024ull + (unsigned long long)(t % n_cells) + offset;
    const unsigned long long bits = pg_splitmix(pg_splitmix(ctr) + 0x632BE59BD9B4E019ull);
    out[t] = (double)(bits >> 11) * (1.0 / 9007199254740992.0);
}

}  // namespace dvq

using namespace dvq;

extern "C" int dvq_prior_uniforms(int64_t first_id, int64_t n, int n_cells, uint64_t seed, double* out, void* stream) {
    if (n < 0 || n_cells < 1 || n_cells > 1024 || (n > 0 && !out)) return fail(DVQ_EINVAL, "prior_uniforms: n >= 0, 1..1024 cells, an output buffer");
    if (n == 0) return DVQ_OK;
    const unsigned long long offset = (unsigned long long)seed * 0x9E3779B97F4A7C15ull;
    const long long total = (long long)n * n_cells;
    hipLaunchKernelGGL(prior_uniforms_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), (long long)first_id,
                       (long long)n, n_cells, offset, out);
    return check_launch("prior_uniforms");
}

static bool pg_dtype_ok(int dtype) { return dtype == DVQ_BF16 || dtype == DVQ_F32; }

extern "C" size_t dvq_prior_packed_bytes(int rows, int features, int dtype) {
    if (rows < 1 || features < 32 || (features % 32) != 0 || !pg_dtype_ok(dtype)) return 0;
    return (size_t)(features / 32) * ((rows + 31) / 32) * (dtype == DVQ_F32 ? 4096 : 2048);
}

extern "C" int dvq_prior_pack_weight(const float* W, int64_t ldn, int64_t ldk, int N, int K, int dtype, void* out, void* stream) {
    if (!W || !out || N < 32 || K < 32 || (N % 32) || (K % 32) || !pg_dtype_ok(dtype))
        return fail(DVQ_EINVAL, "prior_pack_weight: N and K must be multiples of 32, dtype DVQ_F32 or DVQ_BF16");
    const int64_t total = (int64_t)(N / 32) * (K / 32) * (dtype == DVQ_F32 ? 4 : 2) * 64;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == DVQ_F32)
        hipLaunchKernelGGL(prior_pack_weight_kernel<DVQ_F32>, grid, dim3(256), 0, as_stream(stream), W, ldn, ldk, N, K, reinterpret_cast<uint4*>(out));
    else
        hipLaunchKernelGGL(prior_pack_weight_kernel<DVQ_BF16>, grid, dim3(256), 0, as_stream(stream), W, ldn, ldk, N, K, reinterpret_cast<uint4*>(out));
    return check_launch("prior_pack_weight");
}

extern "C" int dvq_prior_embed(const int64_t* idx, int64_t idx_stride, const float* table, int V, int M, int d, int dtype, void* out,
                               void* stream) {
    if (!idx || !table || !out || M < 1 || d < 32 || (d % 32) || !pg_dtype_ok(dtype)) return fail(DVQ_EINVAL, "prior_embed: bad arguments");
    const int64_t total = (int64_t)(d / 32) * ((M + 31) / 32) * (dtype == DVQ_F32 ? 4 : 2) * 64;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == DVQ_F32)
        hipLaunchKernelGGL(prior_embed_kernel<DVQ_F32>, grid, dim3(256), 0, as_stream(stream), idx, idx_stride, table, V, M, d, reinterpret_cast<uint4*>(out));
    else
        hipLaunchKernelGGL(prior_embed_kernel<DVQ_BF16>, grid, dim3(256), 0, as_stream(stream), idx, idx_stride, table, V, M, d, reinterpret_cast<uint4*>(out));
    return check_launch("prior_embed");
}

extern "C" int dvq_prior_gather_rows(const DvqPriorGather* list, int n, const int64_t* idx, int M, int V, int dtype, void* stream) {
    if (!list || n < 1 || n > kPgGatherMax || !idx || M < 1 || V < 1)
        return fail(DVQ_EINVAL, "prior_gather_rows: 1..%d buffers, an index per row, M and V >= 1", kPgGatherMax);
    if (!pg_dtype_ok(dtype)) return fail(DVQ_EINVAL, "prior_gather_rows: dtype must be DVQ_F32 or DVQ_BF16");
    PgGatherList dev;
    int max_features = 0;
    for (int b = 0; b < n; ++b) {
        if (!list[b].table || !list[b].out || list[b].features < 32 || (list[b].features % 32))
            return fail(DVQ_EINVAL, "prior_gather_rows: buffer %d: table, out and a multiple of 32 features are required", b);
        dev.e[b].table = reinterpret_cast<const uint4*>(list[b].table);
        dev.e[b].out = reinterpret_cast<uint4*>(list[b].out);
        dev.e[b].features = list[b].features;
        if (list[b].features > max_features) max_features = list[b].features;
    }
    const int64_t total = (int64_t)(max_features / 32) * ((M + 31) / 32) * (dtype == DVQ_F32 ? 4 : 2) * 64;
    int64_t gx = (total + 255) / 256;
    if (gx > 4096) gx = 4096;                       // grid-stride beyond that
    const dim3 grid((unsigned)gx, (unsigned)n);
    if (dtype == DVQ_F32) hipLaunchKernelGGL(prior_gather_rows_kernel<DVQ_F32>, grid, dim3(256), 0, as_stream(stream), dev, idx, M, V);
    else hipLaunchKernelGGL(prior_gather_rows_kernel<DVQ_BF16>, grid, dim3(256), 0, as_stream(stream), dev, idx, M, V);
    return check_launch("prior_gather_rows");
}

extern "C" int dvq_prior_vrow(const DvqPriorVRow* v, int M, int N, void* stream) {
    if (!v || M < 1) return fail(DVQ_EINVAL, "prior_vrow: bad arguments");
    if (N < 256 || (N % 256) != 0) return fail(DVQ_EINVAL, "prior_vrow: N must be a multiple of 256 (got %d)", N);
    if (v->n_rows < 1 || v->n_rows > 2 || v->k_blocks < 1) return fail(DVQ_EINVAL, "prior_vrow: 1 or 2 input rows, k_blocks >= 1");
    if (!v->bias || !v->cond || !v->label) return fail(DVQ_EINVAL, "prior_vrow: bias, cond and label are required");
    PvDesc d;
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c) {
            if (r < v->n_rows && (!v->x[r][c] || !v->w[r][c])) return fail(DVQ_EINVAL, "prior_vrow: null operand (row %d, column %d)", r, c);
            d.x[r][c] = reinterpret_cast<const char*>(v->x[r][c]);
            d.w[r][c] = reinterpret_cast<const char*>(v->w[r][c]);
        }
    for (int j = 0; j < 3; ++j) {
        if (!v->out_packed[j]) return fail(DVQ_EINVAL, "prior_vrow: out_packed[%d] missing", j);
        d.out_packed[j] = reinterpret_cast<char*>(v->out_packed[j]);
        d.out_pre[j] = reinterpret_cast<char*>(v->out_pre[j]);
    }
    d.n_rows = v->n_rows; d.kb = v->k_blocks; d.bias = v->bias; d.cond = v->cond; d.label = v->label;
    for (int j = 0; j < 3; ++j) d.pre_add[j] = v->pre_add[j];
    if ((v->pre_add[0] != nullptr) != (v->pre_add[1] != nullptr) || (v->pre_add[0] != nullptr) != (v->pre_add[2] != nullptr))
        return fail(DVQ_EINVAL, "prior_vrow: pre_add must be given for all three cells or for none");
    const int SB = (M + 31) / 32;
    const dim3 grid(N / 128, (SB + 3) / 4);
    const size_t lds = (size_t)3 * (3 * 4 * 2 + 3 * 4 * 2) * 1024;
    // eight waves (two feature groups, 164 registers, two waves per SIMD): 69.5 -> 66 us per launch against four waves
    // with all four feature tiles each (FG = 1: 448 registers, one wave per SIMD)
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(prior_vrow_kernel<2>), lds, "prior_vrow")) return rc;
    hipLaunchKernelGGL(prior_vrow_kernel<2>, grid, dim3(512), lds, as_stream(stream), d, M, N);
    return check_launch("prior_vrow");
}

extern "C" int dvq_prior_sample(const float* logits, int n_classes, int M, const double* u, int64_t u_stride, int64_t* x_out,
                                int64_t x_stride, const float* table, int d, int emb_dtype, void* emb_out, void* stream) {
    if (!logits || !u || !x_out || M < 1) return fail(DVQ_EINVAL, "prior_sample: bad arguments");
    if (n_classes < 1 || n_classes > 64 * kPsMaxPer) return fail(DVQ_EINVAL, "prior_sample: 1..%d classes (got %d)", 64 * kPsMaxPer, n_classes);
    if (emb_out && (!table || d < 32 || (d % 32) || !pg_dtype_ok(emb_dtype)))
        return fail(DVQ_EINVAL, "prior_sample: embedding width must be a multiple of 32, emb_dtype DVQ_F32 or DVQ_BF16");
    if (emb_out && emb_dtype == DVQ_F32)
        hipLaunchKernelGGL(prior_sample_kernel<DVQ_F32>, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), logits, n_classes, M, u, u_stride,
                           x_out, x_stride, table, d, reinterpret_cast<uint4*>(emb_out));
    else
        hipLaunchKernelGGL(prior_sample_kernel<DVQ_BF16>, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), logits, n_classes, M, u, u_stride,
                           x_out, x_stride, table, d, reinterpret_cast<uint4*>(emb_out));
    return check_launch("prior_sample");
}

extern "C" int dvq_prior_gemm(const DvqPriorJob* jobs, int n_jobs, int M, int N, int epilogue, int relu, int dtype, void* stream) {
    if (!jobs || n_jobs < 1 || n_jobs > 3) return fail(DVQ_EINVAL, "prior_gemm: 1..3 jobs per launch");
    if (!pg_dtype_ok(dtype)) return fail(DVQ_EINVAL, "prior_gemm: dtype must be DVQ_F32 or DVQ_BF16");
    if (M < 1 || N < 128 || (N % 128) != 0) return fail(DVQ_EINVAL, "prior_gemm: N must be a multiple of 128 (got %d)", N);
    if (epilogue == PG_GATE && (N % 256) != 0) return fail(DVQ_EINVAL, "prior_gemm: the gate epilogue needs N %% 256 == 0");
    PgJobs dev;
    for (int j = 0; j < n_jobs; ++j) {
        const DvqPriorJob& s = jobs[j];
        if (s.n_taps < 1 || s.n_taps > kPgMaxTaps) return fail(DVQ_EINVAL, "prior_gemm: 1..%d taps per job", kPgMaxTaps);
        PgJobDev& d = dev.j[j];
        d.n_taps = s.n_taps;
        for (int t = 0; t < s.n_taps; ++t) {
            if (!s.taps[t].x || !s.taps[t].w || s.taps[t].k_blocks < 1) return fail(DVQ_EINVAL, "prior_gemm: bad tap %d", t);
            d.taps[t].x = reinterpret_cast<const char*>(s.taps[t].x);
            d.taps[t].w = reinterpret_cast<const char*>(s.taps[t].w);
            d.taps[t].kb = s.taps[t].k_blocks;
        }
        if (!s.bias) return fail(DVQ_EINVAL, "prior_gemm: bias is required");
        d.bias = s.bias; d.cond = s.cond; d.label = s.label;
        d.resid = reinterpret_cast<const char*>(s.resid);
        d.out_packed = reinterpret_cast<char*>(s.out_packed);
        d.out_pre = reinterpret_cast<char*>(s.out_pre);
        d.out_rows = s.out_rows;
        d.pre_add = s.pre_add;
        if (epilogue == PG_GATE && (!s.cond || !s.label || !s.out_packed)) return fail(DVQ_EINVAL, "prior_gemm: gate needs cond, label, out_packed");
        if (epilogue == PG_PACK && !s.out_packed) return fail(DVQ_EINVAL, "prior_gemm: out_packed missing");
        if (epilogue == PG_ROWS && !s.out_rows) return fail(DVQ_EINVAL, "prior_gemm: out_rows missing");
    }
    const int SB = (M + 31) / 32;
    hipStream_t st = as_stream(stream);
    // Few sample tiles: k-split kernel, one workgroup per (feature block, sample tile).  Whole tiles leave most CUs
    // without work there (a 256-row gate product is 32 workgroups); the price is the weights read once per sample
    // tile.  Per 9-cell sampling pass: 64..256 rows 4.0-4.1 -> 2.4-2.6 ms (bf16), 15.3 -> 5.6-6.3 ms (fp32);
    // 512 rows 4.2 -> 3.3 / 15.6 -> 7.7 ms; 1024 rows 4.4 -> 5.3 (bf16: tiles stay) / 16.7 -> 13.9 ms.
    const int small_sb = dtype == DVQ_F32 ? 32 : 16;
    if (SB <= small_sb) {
        const dim3 grid(epilogue == PG_GATE ? N / 64 : N / 32, SB, n_jobs), blk(kPgSmallWaves * 64);
#define DVQ_PG_SMALL(E)                                                                                                     \
    do {                                                                                                                    \
        if (dtype == DVQ_F32) hipLaunchKernelGGL((prior_gemm_small_kernel<DVQ_F32, E>), grid, blk, 0, st, dev, M, N, relu); \
        else hipLaunchKernelGGL((prior_gemm_small_kernel<DVQ_BF16, E>), grid, blk, 0, st, dev, M, N, relu);                 \
    } while (0)
        if (epilogue == PG_PACK) DVQ_PG_SMALL(PG_PACK);
        else if (epilogue == PG_GATE) DVQ_PG_SMALL(PG_GATE);
        else if (epilogue == PG_ROWS) DVQ_PG_SMALL(PG_ROWS);
        else return fail(DVQ_EINVAL, "prior_gemm: unknown epilogue %d", epilogue);
#undef DVQ_PG_SMALL
        return check_launch("prior_gemm(small)");
    }
    const int upb = dtype == DVQ_F32 ? 4 : 2;
    // k-blocks of 32 features per LDS stage (an fp32 stage of one block is as large as a bf16 stage of two)
    const int ks = dtype == DVQ_F32 ? 1 : 2;
    for (int j = 0; j < n_jobs; ++j)
        for (int t = 0; t < jobs[j].n_taps; ++t)
            if (jobs[j].taps[t].k_blocks % 2) return fail(DVQ_EINVAL, "prior_gemm: k_blocks must be even");
    // 128 output features per workgroup unless that leaves fewer than two workgroups per CU; then 64
    const int gy = (SB + kPgWaves - 1) / kPgWaves;
    // two workgroups per CU hide each other's barriers and load waits (one wave per SIMD each), at 1.5x the activation
    // traffic per FLOP: 8.68 -> 8.55 ms per 4096 grids, 6.59 -> 6.20 ms per 2048
    const int64_t want = 2 * kNumCU;
    const int no = ((int64_t)(N / 128) * gy * n_jobs >= want || (N % 64) != 0 || (epilogue == PG_GATE && (N % 128) != 0)) ? 4 : 2;
    // 64-wide tiles in pairs: one eight-wave workgroup per CU where that still gives every CU one (the gate products)
    const bool eight = no == 2 && dtype == DVQ_BF16 && (N % 128) == 0 && (epilogue != PG_GATE || (N % 256) == 0) &&
                       (int64_t)(N / 128) * gy * n_jobs >= kNumCU;
    // 128-wide tiles in pairs as well (256 features per eight-wave workgroup, 144 KiB of LDS) where that gives every CU
    // exactly its share of workgroups: the head's hidden layer (512 -> 2048 features) is then one round of 256 workgroups
    // instead of two rounds of 512 four-wave ones (19.5 -> 16 us per launch, 5.21 -> 5.18 ms per 4096 grids)
    const int64_t pairs4 = (int64_t)(N / 256) * gy * n_jobs;
    const bool eight4 = no == 4 && dtype == DVQ_BF16 && (N % 256) == 0 && (epilogue != PG_GATE || (N % 512) == 0) &&
                        pairs4 >= kNumCU && pairs4 % kNumCU == 0;
    const int fg = (eight || eight4) ? 2 : 1;
    // the residual 1x1 products are too small for two 64-wide workgroups per CU: 32-wide tiles (neutral at 4096 rows,
    // 5.46 -> 5.32 ms per 2048 grids)
    const bool one = !eight && no == 2 && dtype == DVQ_BF16 && epilogue == PG_PACK && (int64_t)(N / 64) * gy * n_jobs < want;
    // ... unless 64-wide tiles on eight waves (two groups of 32 features sharing the activations, two waves per SIMD) still
    // give every CU a workgroup: 40 % fewer operand bytes than the 32-wide tiles (5.33 -> 5.29 ms per 4096 grids)
    if (one && (int64_t)(N / 64) * gy * n_jobs >= kNumCU) {
        const dim3 grid2(N / 64, gy, n_jobs);
        const size_t lds2 = (size_t)DVQ_PG_RING8 * (2 * 1 * ks * upb + 4 * ks * upb) * 1024;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(prior_gemm_kernel<DVQ_BF16, PG_PACK, 1, 2, 2>), lds2, "prior_gemm")) return rc;
        hipLaunchKernelGGL((prior_gemm_kernel<DVQ_BF16, PG_PACK, 1, 2, 2>), grid2, dim3(512), lds2, st, dev, M, N, relu);
        return check_launch("prior_gemm");
    }
    if (one) {
        const dim3 grid1(N / 32, gy, n_jobs);
        const size_t lds1 = (size_t)kPgRing * (1 * ks * upb + 4 * ks * upb) * 1024;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(prior_gemm_kernel<DVQ_BF16, PG_PACK, 1, 2, 1>), lds1, "prior_gemm")) return rc;
        hipLaunchKernelGGL((prior_gemm_kernel<DVQ_BF16, PG_PACK, 1, 2, 1>), grid1, dim3(256), lds1, st, dev, M, N, relu);
        return check_launch("prior_gemm");
    }
    const dim3 grid(N / (32 * no * fg), gy, n_jobs);
    const size_t lds = (size_t)(fg == 2 ? DVQ_PG_RING8 : kPgRing) * (fg * no * ks * upb + 4 * ks * upb) * 1024;
    const dim3 blk(4 * fg * 64);
#define DVQ_PG_LAUNCH3(D, E, NO_, KS_, FG_)                                                                   \
    do {                                                                                                 \
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(prior_gemm_kernel<D, E, NO_, KS_, FG_>), lds, "prior_gemm")) return rc; \
        hipLaunchKernelGGL((prior_gemm_kernel<D, E, NO_, KS_, FG_>), grid, blk, lds, st, dev, M, N, relu);    \
    } while (0)
#define DVQ_PG_LAUNCH(E)                                                   \
    do {                                                                   \
        if (dtype == DVQ_F32) {                                            \
            if (no == 4) DVQ_PG_LAUNCH3(DVQ_F32, E, 4, 1, 1);              \
            else DVQ_PG_LAUNCH3(DVQ_F32, E, 2, 1, 1);                      \
        } else if (eight4) DVQ_PG_LAUNCH3(DVQ_BF16, E, 4, 2, 2);           \
        else if (no == 4) DVQ_PG_LAUNCH3(DVQ_BF16, E, 4, 2, 1);            \
        else if (eight) DVQ_PG_LAUNCH3(DVQ_BF16, E, 2, 2, 2);              \
        else DVQ_PG_LAUNCH3(DVQ_BF16, E, 2, 2, 1);                         \
    } while (0)
    if (epilogue == PG_PACK) DVQ_PG_LAUNCH(PG_PACK);
    else if (epilogue == PG_GATE) DVQ_PG_LAUNCH(PG_GATE);
    else if (epilogue == PG_ROWS) DVQ_PG_LAUNCH(PG_ROWS);
    else return fail(DVQ_EINVAL, "prior_gemm: unknown epilogue %d", epilogue);
#undef DVQ_PG_LAUNCH3
#undef DVQ_PG_LAUNCH
    return check_launch("prior_gemm");
}
