// Small kernels around the generation step (gfx950): the call sites' parameter assembly and the class ordering of a
// batch for the prior.  Each replaces a handful of framework launches (sort, index, copy, concatenate) by one.
#include "common.h"

namespace dvq {

// recon_param = [betas 10 | global_orient 3 | pose 45 | transl 3] from the decoder's [B,55] and the position decoder's
// [B,6] outputs (DVQ-VAE/gen_diverse_grasp_obman.py:243-247).
__global__ __launch_bounds__(256) void assemble_param_kernel(const float* __restrict__ recon, int64_t ld_recon,
                                                             const float* __restrict__ pos, int64_t ld_pos, int64_t B,
                                                             float* __restrict__ param) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * 61) return;
    const int64_t b = t / 61;
    const int c = (int)(t - b * 61);
    float v;
    if (c < 10) v = recon[b * ld_recon + c];
    else if (c < 13) v = pos[b * ld_pos + (c - 10)];
    else if (c < 58) v = recon[b * ld_recon + (c - 3)];
    else v = pos[b * ld_pos + (c - 55)];
    param[t] = v;
}

// Stable counting sort of the rows of a batch by class label, in ONE workgroup of 16 waves.
//   pass 1: wave w counts the classes of its contiguous share of the rows (its own LDS histogram);
//   scan:   per class, the first sorted position of every wave's rows (classes ascending, waves ascending);
//   pass 2: the waves walk their shares again, 64 rows at a time; a row's position is its wave's cursor for the class
//           plus the number of lower lanes of the batch-of-64 with the same class (found with one ballot per label bit).
// Equal labels keep their order, so the permutation is the one a stable sort gives and does not depend on timing.
// Writes the sorted labels and perm (sorted position -> original row); gather_rows_f64_kernel then moves the uniforms.
constexpr int kOrderWaves = 16;
constexpr int kOrderMaxClasses = 1024;
constexpr int kOrderKeep = 8;                     // labels of this many 64-row batches per wave stay in registers

__global__ __launch_bounds__(kOrderWaves * 64) void class_order_kernel(const int64_t* __restrict__ label, int B, int n_classes, int bits,
                                                                       int64_t* __restrict__ label_sorted, int* __restrict__ perm) {
    extern __shared__ int hist[];                 // [kOrderWaves][n_classes]: counts, then cursors
    __shared__ int class_start[kOrderMaxClasses];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kOrderWaves * n_classes; i += kOrderWaves * 64) hist[i] = 0;
    const int per = ((B + kOrderWaves - 1) / kOrderWaves + 63) / 64 * 64;   // rows per wave, whole batches of 64
    const int r0 = wave * per, r1 = min(B, r0 + per);
    int* mine = hist + wave * n_classes;
    auto clamp = [&](int64_t v) { return v < 0 ? 0 : (v >= n_classes ? n_classes - 1 : (int)v); };
    int kept[kOrderKeep];
#pragma unroll
    for (int k = 0; k < kOrderKeep; ++k) {        // all loads of the first batches in flight together
        const int r = r0 + k * 64 + lane;
        kept[k] = r < r1 ? clamp(label[r]) : -1;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kOrderKeep; ++k)
        if (kept[k] >= 0) atomicAdd(&mine[kept[k]], 1);            // order-free: only the totals matter here
    for (int r = r0 + kOrderKeep * 64 + lane; r < r1; r += 64) atomicAdd(&mine[clamp(label[r])], 1);
    __syncthreads();
    for (int c = tid; c < n_classes; c += kOrderWaves * 64) {
        int tot = 0;
        for (int w = 0; w < kOrderWaves; ++w) tot += hist[w * n_classes + c];
        class_start[c] = tot;
    }
    __syncthreads();
    if (wave == 0) {                              // exclusive scan of the class totals: a run of classes per lane, then across lanes
        const int run_len = (n_classes + 63) / 64;
        int sum = 0;
        for (int k = 0; k < run_len; ++k) { const int c = lane * run_len + k; if (c < n_classes) sum += class_start[c]; }
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
        int run = incl - sum;
        for (int k = 0; k < run_len; ++k) {
            const int c = lane * run_len + k;
            if (c < n_classes) { const int t = class_start[c]; class_start[c] = run; run += t; }
        }
    }
    __syncthreads();
    for (int c = tid; c < n_classes; c += kOrderWaves * 64) {
        int run = class_start[c];
        for (int w = 0; w < kOrderWaves; ++w) { const int t = hist[w * n_classes + c]; hist[w * n_classes + c] = run; run += t; }
    }
    __syncthreads();
    auto place = [&](int r, int c) {
        const bool live = r < r1;
        unsigned long long same = __ballot(live);
        for (int bit = 0; bit < bits; ++bit) {
            const unsigned long long has = __ballot(live && ((c >> bit) & 1));
            same &= ((c >> bit) & 1) ? has : ~has;
        }
        const unsigned long long below = same & ((1ull << lane) - 1ull);
        int pos = 0;
        if (live) pos = mine[c] + __popcll(below);
        __builtin_amdgcn_wave_barrier();
        if (live && below == 0) mine[c] += __popcll(same);        // the first lane of each class moves the cursor on
        __builtin_amdgcn_wave_barrier();
        if (live) {
            perm[pos] = r;
            label_sorted[pos] = (int64_t)c;
        }
    };
#pragma unroll
    for (int k = 0; k < kOrderKeep; ++k)
        if (r0 + k * 64 < r1) place(r0 + k * 64 + lane, kept[k] < 0 ? 0 : kept[k]);
    for (int base = r0 + kOrderKeep * 64; base < r1; base += 64) {
        const int r = base + lane;
        place(r, r < r1 ? clamp(label[r]) : 0);
    }
}

// out[r, :] = in[perm[r], :] for rows of n_cols doubles.
__global__ __launch_bounds__(256) void gather_rows_f64_kernel(const double* __restrict__ in, const int* __restrict__ perm, int64_t B,
                                                             int n_cols, double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n_cols) return;
    const int64_t r = t / n_cols;
    const int k = (int)(t - r * n_cols);
    out[t] = in[(int64_t)perm[r] * n_cols + k];
}

// out[perm[r]] = in[r] for rows of n_cols int64 values.
__global__ __launch_bounds__(256) void unpermute_rows_kernel(const int64_t* __restrict__ in, const int* __restrict__ perm, int64_t B,
                                                            int n_cols, int64_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n_cols) return;
    const int64_t r = t / n_cols;
    const int k = (int)(t - r * n_cols);
    out[(int64_t)perm[r] * n_cols + k] = in[t];
}

}  // namespace dvq

using namespace dvq;

extern "C" int dvq_assemble_param(const float* recon, int64_t ld_recon, const float* pos, int64_t ld_pos, int64_t B, float* param,
                                  void* stream) {
    if (B < 0 || ld_recon < 55 || ld_pos < 6) return fail(DVQ_EINVAL, "assemble_param: B >= 0, row strides >= 55 and >= 6");
    if (B == 0) return DVQ_OK;
    if (!recon || !pos || !param) return fail(DVQ_EINVAL, "assemble_param: null pointer");
    const int64_t total = B * 61;
    hipLaunchKernelGGL(assemble_param_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), recon, ld_recon, pos,
                       ld_pos, B, param);
    return check_launch("assemble_param");
}

extern "C" int dvq_prior_class_order(const int64_t* label, const double* u, int64_t B, int n_cols, int n_classes,
                                     int64_t* label_sorted, double* u_sorted, int* perm, void* stream) {
    if (B < 0 || B > (1 << 24) || n_classes < 1 || n_classes > kOrderMaxClasses || n_cols < 0)
        return fail(DVQ_EINVAL, "prior_class_order: 0 <= B <= 2^24, 1..%d classes", kOrderMaxClasses);
    if (B == 0) return DVQ_OK;
    if (!label || !label_sorted || !perm || (u && (!u_sorted || n_cols < 1))) return fail(DVQ_EINVAL, "prior_class_order: null pointer");
    int bits = 0;
    while ((1 << bits) < n_classes) ++bits;
    const size_t lds = (size_t)kOrderWaves * n_classes * sizeof(int);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(class_order_kernel), lds, "prior_class_order")) return rc;
    hipLaunchKernelGGL(class_order_kernel, dim3(1), dim3(kOrderWaves * 64), lds, as_stream(stream), label, (int)B, n_classes, bits,
                       label_sorted, perm);
    if (u) {
        const int64_t total = B * n_cols;
        hipLaunchKernelGGL(gather_rows_f64_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), u, perm, B, n_cols,
                           u_sorted);
    }
    return check_launch("prior_class_order");
}

extern "C" int dvq_unpermute_rows(const int64_t* in, const int* perm, int64_t B, int n_cols, int64_t* out, void* stream) {
    if (B < 0 || n_cols < 1) return fail(DVQ_EINVAL, "unpermute_rows: B >= 0, n_cols >= 1");
    if (B == 0) return DVQ_OK;
    if (!in || !perm || !out) return fail(DVQ_EINVAL, "unpermute_rows: null pointer");
    const int64_t total = B * n_cols;
    hipLaunchKernelGGL(unpermute_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), in, perm, B, n_cols, out);
    return check_launch("unpermute_rows");
}
