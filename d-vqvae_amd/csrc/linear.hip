// fp32 linear layers on the matrix cores and the BatchNorm fold (gfx950).
//
// dvq_linear replaces the nn.Linear (+ReLU) stacks of the reference: the STN head
// (DVQ-VAE/network/pointnet_encoder.py:35-43) and Encoder/Decoder
// (DVQ-VAE/network/DVQVAE.py:161-185).  y = act(x W^T + b), exact fp32: the
// 32x32x2 f32 MFMA is a k-ordered fmaf chain.
//
// Both operands are k-contiguous ([M,K] and [N,K]), so every lane loads 16 contiguous
// bytes of its own row for x and for W; the k order inside a group of 8 is permuted
// the same way for both (lane half h takes k0+4h..k0+4h+3), which a contraction does
// not care about.  One wave produces one 32x32 output tile.  Batches of up to 16 rows
// (the reference runs B=1) take a different kernel: one wave per output column streams
// that weight row with all 64 lanes along k and reduces across lanes, so the weight
// matrix is read once, coalesced, by N waves, and results do not depend on arrival order.
#include "common.h"

namespace dvq {

constexpr int kLinWaves = 4;

// Operands of up to kLinMaxBatch products of one shape folded into one grid (blockIdx.y picks the product).
constexpr int kLinMaxBatch = 8;
struct LinBatch {
    const float* x[kLinMaxBatch];
    const void* W[kLinMaxBatch];
    const float* b[kLinMaxBatch];
    float* y[kLinMaxBatch];
};

// RM x RN tiles of 32x32 per wave: every x fragment loaded feeds RN MFMA chains and every W fragment RM, so the
// L1/L2 traffic per FLOP falls by 2 RM RN / (RM + RN).  Each output element is the same k-ordered fmaf chain
// whatever the tiling, so the result does not depend on RM, RN.
template <int RM, int RN>
__global__ __launch_bounds__(kLinWaves * 64) void linear_kernel(
    LinBatch p, int64_t ldx, int M, int K, int N, int relu, int add_id3, int64_t ldy) {
    const float* __restrict__ x = p.x[blockIdx.y];
    const float* __restrict__ W = reinterpret_cast<const float*>(p.W[blockIdx.y]);
    const float* __restrict__ bias = p.b[blockIdx.y];
    float* __restrict__ y = p.y[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;
    const int n_tiles_n = (N + 32 * RN - 1) / (32 * RN);
    const int tile = blockIdx.x * kLinWaves + wave;
    const int tm = tile / n_tiles_n;
    const int tn = tile % n_tiles_n;
    if (tm * 32 * RM >= M) return;
    const float* xp[RM];
    const float* wp[RN];
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        int row = (tm * RM + i) * 32 + r;
        if (row >= M) row = M - 1;
        xp[i] = x + (int64_t)row * ldx;
    }
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        int row = (tn * RN + j) * 32 + r;
        if (row >= N) row = N - 1;
        wp[j] = W + (int64_t)row * K;
    }

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int k0 = 0;
    const int k_vec_end = (K / 8) * 8;
#pragma unroll 2
    for (; k0 < k_vec_end; k0 += 8) {
        f32x4 a[RM], b[RN];
#pragma unroll
        for (int i = 0; i < RM; ++i) a[i] = *reinterpret_cast<const f32x4*>(xp[i] + k0 + 4 * h);
#pragma unroll
        for (int j = 0; j < RN; ++j) b[j] = *reinterpret_cast<const f32x4*>(wp[j] + k0 + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
    }
    for (; k0 < K; k0 += 2) {  // tail: natural order, zero beyond K
        const int k = k0 + h;
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                const float a = k < K ? xp[i][k] : 0.f;
                const float b = k < K ? wp[j][k] : 0.f;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
            }
    }
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int col = (tn * RN + j) * 32 + r;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (tm * RM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < M) {
                    float v = acc[i][j][e] + bv;
                    if (relu) v = v > 0.f ? v : 0.f;
                    if (add_id3 && (col % 4) == 0) v += 1.f;
                    y[(int64_t)row * ldy + col] = v;
                }
            }
    }
}

// M <= kSkinnyM rows: one wave per output column.
constexpr int kSkinnyM = 16;
template <int MR>
__global__ __launch_bounds__(256) void linear_skinny_kernel(
    LinBatch p, int64_t ldx, int M, int K, int N, int relu, int add_id3, int64_t ldy) {
    const float* __restrict__ x = p.x[blockIdx.y];
    const float* __restrict__ W = reinterpret_cast<const float*>(p.W[blockIdx.y]);
    const float* __restrict__ bias = p.b[blockIdx.y];
    float* __restrict__ y = p.y[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= N) return;
    const float* wp = W + (int64_t)col * K;
    float acc[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(wp + k);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int mm = m < M ? m : M - 1;
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + (int64_t)mm * ldx + k);
            acc[m] = __builtin_fmaf(a[0], w[0], acc[m]);
            acc[m] = __builtin_fmaf(a[1], w[1], acc[m]);
            acc[m] = __builtin_fmaf(a[2], w[2], acc[m]);
            acc[m] = __builtin_fmaf(a[3], w[3], acc[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        float v = acc[m];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        acc[m] = v;
    }
    if (lane == 0) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m < M) {
                float v = acc[m] + bv;
                if (relu) v = v > 0.f ? v : 0.f;
                if (add_id3 && (col % 4) == 0) v += 1.f;
                y[(int64_t)m * ldy + col] = v;
            }
        }
    }
}

// bf16 operands, fp32 accumulation (the throughput configuration): y = act(x W^T + b) with x fp32 [M,K] rounded to
// bf16 as it is loaded and W already bf16 [N,K].  16x16x32 MFMA; lane (c = lane & 15, q = lane >> 4) supplies row c of
// its tile and k-group q (8 consecutive k) for both operands, so x costs two 16-byte loads and W one per fragment.
// A wave owns (16 RM) x (16 RN) outputs, a workgroup 2 x 2 waves; the fragments of the next k-step are loaded before
// the MFMAs of the current one.
template <int RM, int RN>
__global__ __launch_bounds__(256) void linear_bf16_kernel(
    LinBatch p, int64_t ldx, int M, int K, int N, int relu, int64_t ldy) {
    const float* __restrict__ x = p.x[blockIdx.y];
    const unsigned short* __restrict__ Wb = reinterpret_cast<const unsigned short*>(p.W[blockIdx.y]);
    const float* __restrict__ bias = p.b[blockIdx.y];
    float* __restrict__ y = p.y[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = lane & 15;
    const int q = lane >> 4;
    const int n_blk_n = (N + 32 * RN - 1) / (32 * RN);
    const int bm = blockIdx.x / n_blk_n, bn = blockIdx.x % n_blk_n;
    const int m0 = (bm * 2 + (wave >> 1)) * 16 * RM;
    const int n0 = (bn * 2 + (wave & 1)) * 16 * RN;
    if (m0 >= M || n0 >= N) return;
    const float* xp[RM];
    const unsigned short* wp[RN];
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        int row = m0 + 16 * i + c;
        if (row >= M) row = M - 1;
        xp[i] = x + (int64_t)row * ldx + 8 * q;
    }
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        int row = n0 + 16 * j + c;
        if (row >= N) row = N - 1;
        wp[j] = Wb + (int64_t)row * K + 8 * q;
    }
    f32x4 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a_lo[RM], a_hi[RM];
    u16x8 b_raw[RN];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            a_lo[i] = *reinterpret_cast<const f32x4*>(xp[i] + k0);
            a_hi[i] = *reinterpret_cast<const f32x4*>(xp[i] + k0 + 4);
        }
#pragma unroll
        for (int j = 0; j < RN; ++j) b_raw[j] = *reinterpret_cast<const u16x8*>(wp[j] + k0);
    };
    load(0);
    for (int k0 = 0; k0 < K; k0 += 32) {
        bf16x8 a[RM], b[RN];
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[i][e] = static_cast<__bf16>(a_lo[i][e]); a[i][4 + e] = static_cast<__bf16>(a_hi[i][e]); }
#pragma unroll
        for (int j = 0; j < RN; ++j) b[j] = __builtin_bit_cast(bf16x8, b_raw[j]);
        if (k0 + 32 < K) load(k0 + 32);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int col = n0 + 16 * j + c;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + 16 * i + 4 * q + e;
                if (row < M) {
                    float v = acc[i][j][e] + bv;
                    if (relu) v = v > 0.f ? v : 0.f;
                    y[(int64_t)row * ldy + col] = v;
                }
            }
    }
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f32_to_bf16_bits(in[i]);
}

__global__ void fold_bn_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var,
                               float eps, int out_f, int in_f, float* __restrict__ A, float* __restrict__ c) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)out_f * in_f) return;
    const int o = (int)(t / in_f);
    const float s = gamma[o] / sqrtf(var[o] + eps);
    A[t] = s * W[t];
    if (t % in_f == 0) c[o] = __builtin_fmaf(s, b[o] - mean[o], beta[o]);
}

}  // namespace dvq

using namespace dvq;

static int linear_impl(const LinBatch& p, int nb, int64_t ldx, int M, int K, int N, int relu, int add_identity3, int64_t ldy,
                       hipStream_t st) {
    if (M <= kSkinnyM) {
        const dim3 grid((N + 3) / 4, nb);
        if (M <= 4)
            hipLaunchKernelGGL((linear_skinny_kernel<4>), grid, dim3(256), 0, st, p, ldx, M, K, N, relu, add_identity3, ldy);
        else
            hipLaunchKernelGGL((linear_skinny_kernel<kSkinnyM>), grid, dim3(256), 0, st, p, ldx, M, K, N, relu, add_identity3, ldy);
    } else {
        // the largest per-wave tile that still leaves every SIMD a wave (4 x 256 CUs)
        auto waves = [&](int rm, int rn) { return (int64_t)((M + 32 * rm - 1) / (32 * rm)) * ((N + 32 * rn - 1) / (32 * rn)); };
        const int64_t enough = 4 * kNumCU;
#define DVQ_LIN_LAUNCH(RM, RN)                                                                                            \
    hipLaunchKernelGGL((linear_kernel<RM, RN>), dim3((unsigned)((waves(RM, RN) + kLinWaves - 1) / kLinWaves), nb),       \
                       dim3(kLinWaves * 64), 0, st, p, ldx, M, K, N, relu, add_identity3, ldy)
        if (waves(2, 2) * nb >= enough) DVQ_LIN_LAUNCH(2, 2);
        else if (waves(2, 1) * nb >= enough) DVQ_LIN_LAUNCH(2, 1);
        else DVQ_LIN_LAUNCH(1, 1);
#undef DVQ_LIN_LAUNCH
    }
    return check_launch("linear");
}

static int linear_bf16_impl(const LinBatch& p, int nb, int64_t ldx, int M, int K, int N, int relu, int64_t ldy, hipStream_t st) {
    // the largest wave tile that still gives every CU a workgroup (the kernel is bound by the L2 -> CU traffic of its
    // operands, which smaller tiles only raise: 64x64 per wave 0.125 ms at 4096x2560x1024, 32x64 0.157, 32x32 0.224)
    auto blocks = [&](int rm, int rn) { return (int64_t)((M + 32 * rm - 1) / (32 * rm)) * ((N + 32 * rn - 1) / (32 * rn)); };
#define DVQ_LINB_LAUNCH(RM, RN) \
    hipLaunchKernelGGL((linear_bf16_kernel<RM, RN>), dim3((unsigned)blocks(RM, RN), nb), dim3(256), 0, st, p, ldx, M, K, N, relu, ldy)
    if (blocks(4, 4) * nb >= kNumCU) DVQ_LINB_LAUNCH(4, 4);
    else if (blocks(2, 4) * nb >= kNumCU) DVQ_LINB_LAUNCH(2, 4);
    else DVQ_LINB_LAUNCH(2, 2);
#undef DVQ_LINB_LAUNCH
    return check_launch("linear_bf16");
}

static int linear_check(const float* x, const void* W, const float* y, int64_t ldx, int64_t ldy, int M, int K, int N, int add_identity3,
                        bool bf16) {
    if (!x || !W || !y) return fail(DVQ_EINVAL, "linear: null pointer");
    if (ldx < K || ldy < N) return fail(DVQ_EINVAL, "linear: row stride smaller than the row");
    if (bf16) {
        if ((K % 32) != 0 || (ldx % 4) != 0 || ((uintptr_t)x & 15) || ((uintptr_t)W & 15))
            return fail(DVQ_EINVAL, "linear_bf16: K must be a multiple of 32, ldx of 4, and x, W 16-byte aligned");
    } else {
        if ((K % 4) != 0 || (ldx % 4) != 0 || ((uintptr_t)x & 15) || ((uintptr_t)W & 15))
            return fail(DVQ_EINVAL, "linear: K and ldx must be multiples of 4 and x, W 16-byte aligned");
        if (add_identity3 && N != 9) return fail(DVQ_EINVAL, "linear: add_identity3 needs a 9-wide output");
    }
    return DVQ_OK;
}

extern "C" int dvq_linear(const float* x, int64_t ldx, const float* W, const float* b, int M, int K, int N, int relu,
                          int add_identity3, float* y, int64_t ldy, void* stream) {
    if (M < 0 || K < 1 || N < 1) return fail(DVQ_EINVAL, "linear: bad shape M=%d K=%d N=%d", M, K, N);
    if (M == 0) return DVQ_OK;
    if (int rc = linear_check(x, W, y, ldx, ldy, M, K, N, add_identity3, false)) return rc;
    LinBatch p = {};
    p.x[0] = x; p.W[0] = W; p.b[0] = b; p.y[0] = y;
    return linear_impl(p, 1, ldx, M, K, N, relu, add_identity3, ldy, as_stream(stream));
}

extern "C" int dvq_linear_batched(int n_batch, const float* const* x, int64_t ldx, const void* const* W, const float* const* b,
                                  int M, int K, int N, int relu, int add_identity3, float* const* y, int64_t ldy, int w_bf16,
                                  void* stream) {
    if (n_batch < 0 || n_batch > kLinMaxBatch) return fail(DVQ_EINVAL, "linear (batched): 0..%d products, got %d", kLinMaxBatch, n_batch);
    if (M < 0 || K < 1 || N < 1) return fail(DVQ_EINVAL, "linear: bad shape M=%d K=%d N=%d", M, K, N);
    if (M == 0 || n_batch == 0) return DVQ_OK;
    if (!x || !W || !y) return fail(DVQ_EINVAL, "linear: null pointer");
    if (w_bf16 && (add_identity3 || K < 32)) return fail(DVQ_EINVAL, "linear (batched): bf16 weights take K >= 32 and no identity");
    LinBatch p = {};
    for (int i = 0; i < n_batch; ++i) {
        if (int rc = linear_check(x[i], W[i], y[i], ldx, ldy, M, K, N, add_identity3, w_bf16 != 0)) return rc;
        p.x[i] = x[i]; p.W[i] = W[i]; p.b[i] = b ? b[i] : nullptr; p.y[i] = y[i];
    }
    hipStream_t st = as_stream(stream);
    return w_bf16 ? linear_bf16_impl(p, n_batch, ldx, M, K, N, relu, ldy, st)
                  : linear_impl(p, n_batch, ldx, M, K, N, relu, add_identity3, ldy, st);
}

extern "C" int dvq_fold_bn(const float* W, const float* b, const float* gamma, const float* beta, const float* mean,
                           const float* var, float eps, int out_f, int in_f, float* A, float* c, void* stream) {
    if (out_f < 1 || in_f < 1) return fail(DVQ_EINVAL, "fold_bn: bad shape");
    if (!W || !b || !gamma || !beta || !mean || !var || !A || !c) return fail(DVQ_EINVAL, "fold_bn: null pointer");
    const int64_t total = (int64_t)out_f * in_f;
    hipLaunchKernelGGL(fold_bn_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), W, b, gamma, beta, mean,
                       var, eps, out_f, in_f, A, c);
    return check_launch("fold_bn");
}

extern "C" int dvq_to_bf16(const float* in, void* out, int64_t n, void* stream) {
    if (n < 0) return fail(DVQ_EINVAL, "to_bf16: negative size");
    if (n == 0) return DVQ_OK;
    if (!in || !out) return fail(DVQ_EINVAL, "to_bf16: null pointer");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), in,
                       reinterpret_cast<unsigned short*>(out), n);
    return check_launch("to_bf16");
}

extern "C" int dvq_linear_bf16(const float* x, int64_t ldx, const void* Wb, const float* b, int M, int K, int N, int relu, float* y,
                               int64_t ldy, void* stream) {
    if (M < 0 || K < 32 || N < 1) return fail(DVQ_EINVAL, "linear_bf16: bad shape M=%d K=%d N=%d", M, K, N);
    if (M == 0) return DVQ_OK;
    if (int rc = linear_check(x, Wb, y, ldx, ldy, M, K, N, 0, true)) return rc;
    LinBatch p = {};
    p.x[0] = x; p.W[0] = Wb; p.b[0] = b; p.y[0] = y;
    return linear_bf16_impl(p, 1, ldx, M, K, N, relu, ldy, as_stream(stream));
}
