// Fused nearest-codebook search for the vector quantizer (gfx950).
//
// Replaces VectorQuantizer.forward(z, istrain=False) of the reference
// (DVQ-VAE/network/vqvae/quantizer.py:44-54).  For every latent row z_i the kernel
// evaluates d[i,k] = fl(fl(|z_i|^2 + |E_k|^2) - 2<z_i,E_k>) for all codes k on the
// matrix cores and keeps a running (min, first index) per latent in registers; the
// [n,K] distance matrix is never written anywhere.
//
// Mapping onto MFMA 32x32: the CODEBOOK tile is the A operand (32 codes = rows) and
// the LATENT tile is the B operand (32 latents = columns).  In the 32x32 accumulator
// layout a lane then owns ONE latent (column lane&31) and 16 codes (rows in its 16
// registers), so the argmin over codes is a lane-local scan with no cross-lane traffic;
// the two half-waves (rows +0..3 / +4..7 of each group of 8) are merged once at the end.
//
// Data layout
//   z      [n, D] row-major, streamed once from HBM straight into B-fragment registers
//          (the whole row of a latent lives in the registers of its two lanes).
//   E      re-laid once per codebook ("packed") into A-fragment order: a 1-KiB unit
//          holds, for one tile of 32 codes and one k-step group, the 16 bytes each of
//          the 64 lanes feeds to the MFMA, so staging into LDS is a linear copy and
//          every ds_read_b128 is conflict-free.
//   |E|^2  fp32 [Kpad]; padding codes get +inf and can never win.
#include "common.h"

#include <mutex>

// Workspace of dvq_vq_argmin_static: the packed codebook and |E|^2 of one search at a time.  One copy per GPU, at
// its own address on each, held until the process ends.  In host code this name is only a handle for
// hipGetSymbolAddress -- never a device address -- see vq_static_scratch().
static __device__ __align__(256) unsigned char scratch[DVQ_VQ_STATIC_WORKSPACE_BYTES];

namespace dvq {

template <int DT> struct VqTraits;
template <> struct VqTraits<DVQ_BF16> {
    typedef unsigned short elem;
    static constexpr int kDimsPerUnit = 16;  // one 32x32x16 k-step per 1-KiB unit
};
template <> struct VqTraits<DVQ_F32> {
    typedef float elem;
    static constexpr int kDimsPerUnit = 8;   // four 32x32x2 k-steps per 1-KiB unit
};

// Which dimension of the row a (unit s, lane half h, element j) triple stands for.
// Any bijection works as long as both operands use the same one; these make every
// lane's loads of z 16 contiguous bytes.
__device__ __host__ __forceinline__ int vq_dim_bf16(int s, int h, int j) { return 16 * s + 8 * h + j; }
__device__ __host__ __forceinline__ int vq_dim_f32(int s, int h, int i, int D) { return h * (D / 2) + 4 * s + i; }

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Operands of up to kVqMaxBatch searches of one shape folded into one grid (blockIdx.y picks the search).
constexpr int kVqMaxBatch = 8;
struct VqBatch {
    const void* z[kVqMaxBatch];
    const char* packed[kVqMaxBatch];
    const float* e2[kVqMaxBatch];
    int64_t* idx[kVqMaxBatch];
};

constexpr int kVqRotMinTiles = 9;          // (256 codes stay in LDS: one-sweep kernel)
constexpr int kVqRotMaxTiles = 64;
constexpr int64_t kVqRotMinRows = 65536;   // below this the one-sweep kernel has the lower fixed cost
constexpr int64_t kVqTwoTileMaxRows = (int64_t)3 << 20;

// ------------------------------------------------------------------ pack kernel
template <int DT>
__global__ void vq_pack_kernel(const typename VqTraits<DT>::elem* __restrict__ E, int K, int D, int Kpad,
                               uint4* __restrict__ packed, float* __restrict__ e2) {
    const int U = D / VqTraits<DT>::kDimsPerUnit;
    const int64_t nchunks = (int64_t)(Kpad / 32) * U * 64;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nchunks) {
        const int l = (int)(t & 63);
        const int64_t unit = t >> 6;
        const int s = (int)(unit % U);
        const int ct = (int)(unit / U);
        const int code = ct * 32 + (l & 31);
        const int h = l >> 5;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (code < K) {
            if (DT == DVQ_BF16) {
                v = *reinterpret_cast<const uint4*>(E + (int64_t)code * D + vq_dim_bf16(s, h, 0));
            } else {
                v = *reinterpret_cast<const uint4*>(E + (int64_t)code * D + vq_dim_f32(s, h, 0, D));
            }
        }
        packed[t] = v;
    }
    if (t < Kpad) {
        float acc = __builtin_inff();
        if (t < K) {
            acc = 0.f;
            const typename VqTraits<DT>::elem* row = E + t * D;
            for (int d = 0; d < D; ++d) {
                float x;
                if (DT == DVQ_BF16) x = bf16_bits_to_f32((unsigned short)row[d]); else x = (float)row[d];
                acc = __builtin_fmaf(x, x, acc);
            }
        }
        e2[t] = acc;
    }
}

// ------------------------------------------------------------- epilogue pieces
struct Best {
    float d;
    int k;
};

// Scan the 16 accumulator registers of one 32x32 tile.  Register i holds code
// tile_base + (i&3) + 8*(i>>2) + 4*h, increasing with i, so a strict '<' keeps the
// lowest index among equal distances (torch.argmin semantics, quantizer.py:49).
__device__ __forceinline__ void scan_tile(const f32x16& acc, const float* e2s /* LDS, this tile */, int h,
                                          int tile_base, float z2, Best& b) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 e = *reinterpret_cast<const f32x4*>(e2s + 8 * g + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float t1 = z2 + e[q];                               // fl(|z|^2 + |E_k|^2)
            const float d = __builtin_fmaf(-2.0f, acc[4 * g + q], t1);  // fl(t1 - 2*dot): 2*dot is exact
            const int code = tile_base + 8 * g + 4 * h + q;
            if (d < b.d) { b.d = d; b.k = code; }
        }
    }
}

__device__ __forceinline__ void merge_halves(Best& b) {
    const float od = __shfl_xor(b.d, 32, 64);
    const int ok = __shfl_xor(b.k, 32, 64);
    if (od < b.d || (od == b.d && ok < b.k)) { b.d = od; b.k = ok; }
}

// Linear copy of `bytes` (multiple of 16) global -> LDS by the whole block.
__device__ __forceinline__ void stage_copy(const char* __restrict__ src, char* dst, int bytes, int tid, int nthreads) {
    for (int off = tid * 16; off < bytes; off += nthreads * 16) {
        *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(src + off);
    }
}

__device__ __forceinline__ void vq_glds16(const void* gsrc_lane, void* lds_wave_base);
// one dword per lane straight into LDS (64 consecutive floats per wave-instruction)
__device__ __forceinline__ void vq_glds4(const void* gsrc_lane, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

// ------------------------------------------------- fast path: z rows in registers
// One wave owns NB*32 latents; their rows stay in registers for the whole sweep over
// the codebook.  The codebook streams through LDS in stages of TS tiles (double
// buffered when there is more than one stage).
template <int DT, int D, int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void vq_argmin_reg_kernel(VqBatch vb, int64_t n, int n_tiles, int TS) {
    const typename VqTraits<DT>::elem* __restrict__ z = reinterpret_cast<const typename VqTraits<DT>::elem*>(vb.z[blockIdx.y]);
    const char* __restrict__ packed = vb.packed[blockIdx.y];
    const float* __restrict__ e2 = vb.e2[blockIdx.y];
    int64_t* __restrict__ idx_out = vb.idx[blockIdx.y];
    constexpr int U = D / VqTraits<DT>::kDimsPerUnit;       // 1-KiB units per code tile
    constexpr int TILE_BYTES = U * 1024;
    constexpr int ZR = (DT == DVQ_BF16) ? U : U * 4;        // fragment registers-groups per latent block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n_stages = (n_tiles + TS - 1) / TS;
    const int stage_bytes = TS * TILE_BYTES;
    const int n_buf = n_stages > 1 ? 2 : 1;
    char* e_lds = smem;                                                  // [n_buf][stage_bytes]
    float* e2_lds = reinterpret_cast<float*>(smem + n_buf * stage_bytes);  // [n_buf][TS*32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;
    constexpr int LAT_PER_BLOCK = WAVES * NB * 32;
    const int64_t n_groups = (n + LAT_PER_BLOCK - 1) / LAT_PER_BLOCK;

    // A stage goes global -> LDS by direct copies (1-KiB units of codes dealt to the waves, then the stage's |E|^2, 64
    // floats per wave-instruction): no registers and no wait between the request and the products of the current stage.
    auto stage_dma = [&](int t_first, int t_count, int bufi) {
        const char* src = packed + (int64_t)t_first * TILE_BYTES;
        char* dst = e_lds + bufi * stage_bytes;
        const int units = t_count * U;
        for (int u = wave; u < units; u += WAVES) vq_glds16(src + (int64_t)u * 1024 + lane * 16, dst + u * 1024);
        const int ne = t_count * 32;
        for (int e0 = wave * 64; e0 < ne; e0 += WAVES * 64)
            if (e0 + lane < ne) vq_glds4(e2 + t_first * 32 + e0 + lane, e2_lds + bufi * TS * 32 + e0);
    };
    bool staged = false;  // the first stage of the coming sweep is in LDS (or on its way): single-stage codebooks are staged
                          // once per block, longer ones cycle -- the last stage of a sweep requests the first of the next
    int g = 0;            // stages consumed so far by a multi-stage block: stage g sits in buffer g & 1
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t lat0 = grp * LAT_PER_BLOCK + (int64_t)wave * NB * 32;
        // ---- load this wave's latent rows into B-fragment registers, and |z|^2
        bf16x8 zb[(DT == DVQ_BF16) ? NB : 1][(DT == DVQ_BF16) ? ZR : 1];
        f32x4 zf[(DT == DVQ_F32) ? NB : 1][(DT == DVQ_F32) ? U : 1];
        float z2[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            int64_t row = lat0 + nb * 32 + r;
            if (row >= n) row = n - 1;  // clamp: the result of such a lane is never stored
            float part = 0.f;
            if constexpr (DT == DVQ_BF16) {
                const unsigned short* zr = reinterpret_cast<const unsigned short*>(z) + row * D;
#pragma unroll
                for (int s = 0; s < U; ++s) {
                    const u16x8 raw = *reinterpret_cast<const u16x8*>(zr + vq_dim_bf16(s, h, 0));
                    zb[nb][s] = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = bf16_bits_to_f32(raw[j]);
                        part = __builtin_fmaf(x, x, part);
                    }
                }
            } else {
                const float* zr = reinterpret_cast<const float*>(z) + row * D;
#pragma unroll
                for (int s = 0; s < U; ++s) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(zr + vq_dim_f32(s, h, 0, D));
                    zf[nb][s] = v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) part = __builtin_fmaf(v[i], v[i], part);
                }
            }
            z2[nb] = part + __shfl_xor(part, 32, 64);
        }
        Best best[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { best[nb].d = __builtin_inff(); best[nb].k = 0; }

        // ---- sweep the codebook
        if (!staged) {
            stage_dma(0, n_tiles < TS ? n_tiles : TS, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own copies landed before they are published
            __syncthreads();
            staged = true;
        }
        for (int st = 0; st < n_stages; ++st) {
            const int buf = g & (n_buf - 1);
            const int t_first = st * TS;
            const int t_count = (n_tiles - t_first) < TS ? (n_tiles - t_first) : TS;
            if (n_stages > 1 && (st + 1 < n_stages || grp + gridDim.x < n_groups)) {
                // the next stage into the other buffer; behind the last one, the first stage of the next sweep
                const int nf = (st + 1 < n_stages ? st + 1 : 0) * TS;
                stage_dma(nf, (n_tiles - nf) < TS ? (n_tiles - nf) : TS, buf ^ 1);
            }
            const char* eb = e_lds + buf * stage_bytes;
            const float* e2b = e2_lds + buf * TS * 32;
            for (int t = 0; t < t_count; ++t) {
                f32x16 acc[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
                const char* tb = eb + t * TILE_BYTES + lane * 16;
#pragma unroll
                for (int s = 0; s < U; ++s) {
                    if constexpr (DT == DVQ_BF16) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(tb + s * 1024);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, zb[nb][s], acc[nb], 0, 0, 0);
                    } else {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(tb + s * 1024);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], zf[nb][s][i], acc[nb], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    scan_tile(acc[nb], e2b + t * 32, h, (t_first + t) * 32, z2[nb], best[nb]);
            }
            if (n_stages > 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own copies landed before they are published
                __syncthreads();  // next buffer landed; this one may be overwritten
                ++g;
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            merge_halves(best[nb]);
            const int64_t row = lat0 + nb * 32 + r;
            if (h == 0 && row < n) idx_out[row] = (int64_t)best[nb].k;
        }
    }
}


// ------------------------------------------ bf16 fast path: argmax of <z,E_k> - |E_k|^2/2
// |z|^2 is the same for every code of a row, so the nearest code is the one maximising
// <z,E_k> - |E_k|^2/2.  The accumulator of each 32x32 tile is INITIALISED with -|E_k|^2/2
// and the MFMA chain adds the dot product, so the per-element work after the matrix cores is
// three VALU operations (compare, select index, max) -- the matrix pipe stays the bottleneck.
// No term is rounded at the magnitude of |z|^2, so this is at least as accurate as the
// reference's formula; rows where the two can disagree are near-ties of the reference's
// own rounding.  A tile in the packed codebook is U units of E followed by one unit whose
// first 128 bytes hold -|E_k|^2/2 of its 32 codes (-inf for padding codes): the whole stage
// is one linear global->LDS stream.
__device__ __forceinline__ void vq_glds16(const void* gsrc_lane, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ void vq_pack_bf16x_kernel(const unsigned short* __restrict__ E, int K, int D, int Kpad, uint4* __restrict__ packed) {
    // group of 32 codes = 2 tiles of 16 codes x (D/32) k-steps of 32 dims, then one unit of -|E|^2/2.
    // E unit (hh, s): lane (c = l&15, q = l>>4) holds E[32g + 16hh + c][32s + 8q .. +8]  (16x16x32 A fragment)
    const int S = D / 32;
    const int UG = 2 * S + 1;
    const int64_t nchunks = (int64_t)(Kpad / 32) * UG * 64;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    const int l = (int)(t & 63);
    const int64_t unit = t >> 6;
    const int u = (int)(unit % UG);
    const int g = (int)(unit / UG);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (u < 2 * S) {
        const int hh = u / S, ss = u % S;
        const int code = g * 32 + 16 * hh + (l & 15);
        if (code < K) v = *reinterpret_cast<const uint4*>(E + (int64_t)code * D + 32 * ss + 8 * (l >> 4));
    } else if (l < 8) {  // lane l: codes 4l .. 4l+3 of the group
        float o[4];
        for (int q = 0; q < 4; ++q) {
            const int code = g * 32 + 4 * l + q;
            float acc = __builtin_inff();
            if (code < K) {
                acc = 0.f;
                const unsigned short* row = E + (int64_t)code * D;
                for (int d = 0; d < D; ++d) { const float x = bf16_bits_to_f32(row[d]); acc = __builtin_fmaf(x, x, acc); }
            }
            o[q] = -0.5f * acc;
        }
        v.x = __builtin_bit_cast(unsigned, o[0]); v.y = __builtin_bit_cast(unsigned, o[1]);
        v.z = __builtin_bit_cast(unsigned, o[2]); v.w = __builtin_bit_cast(unsigned, o[3]);
    }
    packed[t] = v;
}

// One wave owns NB tiles of 16 latents (MFMA 16x16x32: columns = latents, rows = codes); lane
// (c = lane&15, q = lane>>4) holds latent c's dims 32s+8q..+8 for every k-step s -- 64 contiguous
// bytes per latent row and instruction when loading z -- and, in the accumulator, codes 4q..4q+3
// of the 16-code tile.
template <int D, int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void vq_argmax_bf16_kernel(VqBatch vb, int64_t n, int n_tiles, int TS) {
    const unsigned short* __restrict__ z = reinterpret_cast<const unsigned short*>(vb.z[blockIdx.y]);
    const char* __restrict__ packed = vb.packed[blockIdx.y];
    int64_t* __restrict__ idx_out = vb.idx[blockIdx.y];
    constexpr int S = D / 32;
    constexpr int TILE_BYTES = (2 * S + 1) * 1024;   // a group of 32 codes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n_stages = (n_tiles + TS - 1) / TS;
    const int stage_bytes = TS * TILE_BYTES;
    const int n_buf = n_stages > 1 ? 2 : 1;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c = lane & 15;
    const int q = lane >> 4;
    constexpr int LAT_PER_BLOCK = WAVES * NB * 16;
    const int64_t n_groups = (n + LAT_PER_BLOCK - 1) / LAT_PER_BLOCK;

    auto issue_stage = [&](int st, int buf) {
        const int t_first = st * TS;
        const int t_count = (n_tiles - t_first) < TS ? (n_tiles - t_first) : TS;
        const int units = t_count * (2 * S + 1);
        const char* src = packed + (int64_t)t_first * TILE_BYTES;
        char* dst = smem + buf * stage_bytes;
        for (int u = wave; u < units; u += WAVES) vq_glds16(src + (int64_t)u * 1024 + lane * 16, dst + u * 1024);
    };

    bool resident = false;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t lat0 = grp * LAT_PER_BLOCK + (int64_t)wave * NB * 16;
        if (!resident) issue_stage(0, 0);   // lands while the latent rows are fetched
        bf16x8 zb[NB][S];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            int64_t row = lat0 + nb * 16 + c;
            if (row >= n) row = n - 1;
            const unsigned short* zr = z + row * D + 8 * q;
#pragma unroll
            for (int s = 0; s < S; ++s) zb[nb][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(zr + 32 * s));
        }
        float best[NB];
        int bidx[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { best[nb] = -__builtin_inff(); bidx[nb] = 0; }
        if (!resident) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's direct-to-LDS copies have landed ...
            __syncthreads();                                    // ... and so have everyone else's
            resident = (n_stages == 1);
        }
        for (int st = 0; st < n_stages; ++st) {
            const int buf = st & (n_buf - 1);
            if (st + 1 < n_stages) issue_stage(st + 1, buf ^ 1);
            const int t_first = st * TS;
            const int t_count = (n_tiles - t_first) < TS ? (n_tiles - t_first) : TS;
            const char* eb = smem + buf * stage_bytes;
            for (int t = 0; t < t_count; ++t) {
                const char* tb = eb + t * TILE_BYTES;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x4 e = *reinterpret_cast<const f32x4*>(tb + 2 * S * 1024 + (16 * hh + 4 * q) * 4);
                    f32x4 acc[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb] = e;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(tb + (hh * S + s) * 1024 + lane * 16);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, zb[nb][s], acc[nb], 0, 0, 0);
                    }
                    const int base = (t_first + t) * 32 + 16 * hh + 4 * q;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {   // codes increase with i: '>' keeps the lowest index among equals
                            const float v = acc[nb][i];
                            const bool gt = v > best[nb];
                            bidx[nb] = gt ? base + i : bidx[nb];
                            best[nb] = gt ? v : best[nb];
                        }
                }
            }
            if (n_stages > 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next stage's copies, before they are published
                __syncthreads();
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {   // merge the four lanes that share a latent
                const float ob = __shfl_xor(best[nb], off, 64);
                const int ok = __shfl_xor(bidx[nb], off, 64);
                if (ob > best[nb] || (ob == best[nb] && ok < bidx[nb])) { best[nb] = ob; bidx[nb] = ok; }
            }
            const int64_t row = lat0 + nb * 16 + c;
            if (q == 0 && row < n) idx_out[row] = (int64_t)bidx[nb];
        }
    }
}

// Long codebooks (four or more stages), many latents: the same search with the latent tiles of a wave at four
// different points of their sweep.  The codebook cycles through LDS without end (stage after stage, wrapping), split
// into four quarters; a wave holds four latent tiles ("slots"), and slot j starts its tiles at quarter j.  In every
// phase (one quarter of the codebook) all four slots use the same staged code tiles, exactly one slot finishes and is
// refilled from a one-tile prefetch buffer whose loads were issued a whole phase earlier.  The latent stream from
// HBM is therefore spread evenly over the run and hidden behind the matrix work, instead of a load burst between
// two sweeps that leaves the matrix cores idle.
// Every wave runs exactly cnt phases for its cnt tiles (cnt a multiple of 4), all four slots busy in each: the last
// tile of slot j = 1..3 would need j more phases at the end, so its quarters 0..j-1 are done in phases 0..j-1
// instead (the tile is fetched twice) and that partial winner waits in two registers until the end of the run.
// A slot that starts at quarter j > 0 meets the codes of quarters 0..j-1 after those of j..3: at the wrap its
// running winner is set aside, and when the two parts are merged the winner among the LOWER codes takes equal
// scores, so ties still resolve to the lowest index.
template <int D, int WAVES, bool STAGGER>
__global__ __launch_bounds__(WAVES * 64) void vq_argmax_bf16_rot_kernel(
    const unsigned short* __restrict__ z, const char* __restrict__ packed, int64_t n, int n_tiles, int TS,
    int64_t* __restrict__ idx_out) {
    constexpr int S = D / 32;
    constexpr int NB = 4;
    constexpr int TILE_BYTES = (2 * S + 1) * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n_stages = (n_tiles + TS - 1) / TS;   // >= 4 (host)
    const int stage_bytes = TS * TILE_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15;
    const int q = lane >> 4;
    const int64_t n_zt = (n + 15) / 16;                      // latent tiles of 16 rows
    const int64_t streams = (int64_t)gridDim.x * WAVES;
    const int cnt = (int)((n_zt + streams - 1) / streams + 3) / 4 * 4;   // tiles per wave, the same for every wave
    const int64_t my0 = (int64_t)blockIdx.x * WAVES + wave;  // tile i of this wave is latent tile i*streams + my0;
                                                             // tiles past the end are clamped dummies, never stored

    auto issue_stage = [&](int cs, int buf) {
        const int t_first = cs * TS;
        const int t_count = (n_tiles - t_first) < TS ? (n_tiles - t_first) : TS;
        const int units = t_count * (2 * S + 1);
        const char* src = packed + (int64_t)t_first * TILE_BYTES;
        char* dst = smem + buf * stage_bytes;
        for (int u = wave; u < units; u += WAVES) vq_glds16(src + (int64_t)u * 1024 + lane * 16, dst + u * 1024);
    };
    bf16x8 zb[NB][S], zp[S];
    auto load_tile = [&](int i) {
        int64_t row = ((int64_t)i * streams + my0) * 16 + c;
        if (row >= n) row = n - 1;
        const unsigned short* zr = z + row * D + 8 * q;
#pragma unroll
        // (plain loads: nontemporal ones measured 2-7 % slower here at 256..2048 codes, and 10 % slower in the one-sweep
        // kernel beyond 4 Mi rows, although the rows are read once)
        for (int s = 0; s < S; ++s) zp[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(zr + 32 * s));
    };
    auto store_tile = [&](int i, float bv, int bi) {
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {   // merge the four lanes that share a latent
            const float ob = __shfl_xor(bv, off, 64);
            const int ok = __shfl_xor(bi, off, 64);
            if (ob > bv || (ob == bv && ok < bi)) { bv = ob; bi = ok; }
        }
        const int64_t row = ((int64_t)i * streams + my0) * 16 + c;
        if (q == 0 && row < n) idx_out[row] = (int64_t)bi;
    };
    float best[NB], bestH[NB], bestE[NB];
    int bidx[NB], bidxH[NB], bidxE[NB];
    issue_stage(0, 0);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        best[j] = bestH[j] = bestE[j] = -__builtin_inff();
        bidx[j] = bidxH[j] = bidxE[j] = 0;
        load_tile(j == 0 ? 0 : cnt - 4 + j);   // slots 1..3 begin with the first quarters of their last tile
#pragma unroll
        for (int s = 0; s < S; ++s) zb[j][s] = zp[s];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // One group of 32 codes (two halves of 16) against the four slots, laid out so that one wave by itself keeps the
    // matrix pipe fed: the 2 x S code fragments are read from LDS one item ahead of the four matrix instructions that
    // use them, and the compare-selects that fold a finished half into the running winners (16 per half) are dealt
    // two at a time to the items of the NEXT half -- the second half of a group is folded under the first half of
    // the following group (`pend`), the last one of a stage by flush_pending().  Halves are folded in code order.
    f32x4 pend[NB];
    int pend_base = 0;
    auto fold_two = [&](const f32x4 (&a)[NB], int base, int pair) {
#pragma unroll
        for (int j = 2 * pair; j < 2 * pair + 2; ++j) {   // codes increase with i: '>' keeps the lowest index among equals
            const int nb = j >> 2, i = j & 3;
            const float v = a[nb][i];
            const bool gt = v > best[nb];
            bidx[nb] = gt ? base + i : bidx[nb];
            best[nb] = gt ? v : best[nb];
        }
    };
    auto flush_pending = [&]() {
#pragma unroll
        for (int pair = 0; pair < 2 * NB; ++pair) fold_two(pend, pend_base, pair);
    };
    auto code_tile = [&](auto has_pending, const char* tb, int base32) {
        constexpr bool PEND = decltype(has_pending)::value;
        static_assert(S == 2 * NB, "two compare-selects per item cover a half exactly when S == 2 * NB");
        f32x4 acc[2][NB];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x4 e = *reinterpret_cast<const f32x4*>(tb + 2 * S * 1024 + (16 * hh + 4 * q) * 4);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[hh][nb] = e;
        }
        bf16x8 a[2];
        a[0] = *reinterpret_cast<const bf16x8*>(tb + lane * 16);
        const int base0 = base32 + 4 * q;
#pragma unroll
        for (int item = 0; item < 2 * S; ++item) {
            const int hh = item / S, s = item % S;
            if (item + 1 < 2 * S) a[(item + 1) & 1] = *reinterpret_cast<const bf16x8*>(tb + (item + 1) * 1024 + lane * 16);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[hh][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[item & 1], zb[nb][s], acc[hh][nb], 0, 0, 0);
            if (hh == 0) {
                if (PEND) fold_two(pend, pend_base, s);
            } else {
                fold_two(acc[0], base0, s);
            }
            __builtin_amdgcn_sched_barrier(0);   // keeps this order: read of item + 1, products of item, two folds
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) pend[nb] = acc[1][nb];
        pend_base = base0 + 16;
    };

    int g = 0;               // stages consumed so far; stage g sits in buffer g&1
    // STAGGER: the two waves of a SIMD (wave w and w + 4) run half a stage apart.  Every wave meets a barrier twice per
    // stage, at its middle and at its end; for waves 0-3 ("early", GRP 0) a stage ends where it is half done for waves
    // 4-7 ("late", GRP 1), so whenever one wave of a SIMD is busy with what surrounds a stage boundary -- refilling a
    // slot, storing a tile, issuing the copies and the row prefetch -- the other is in the middle of its matrix
    // instructions and keeps the pipe fed (without this both waves of every SIMD do that work at the same time, right
    // behind the barrier).  Time in half stages h: the early waves compute stage s in h = 2s, 2s+1, the late ones in
    // h = 2s+1, 2s+2; all waves issue their share of the copies of stage s+1 at the start of an odd h = 2s+1 (the
    // buffer's previous stage s-1 was last read in h = 2s) and wait for them at its end (first read in h = 2s+2).
    // The late waves pass one barrier before their first stage, the early ones one after their last: same count.
    const int GRP = (STAGGER && wave >= WAVES / 2) ? 1 : 0;   // wave-uniform: 0 early wave, 1 late wave
    auto phase = [&](auto q_const, int p) {
        constexpr int Q = decltype(q_const)::value;
        // slot Q starts tile p ...
        if (p != 0) {
#pragma unroll
            for (int s = 0; s < S; ++s) zb[Q][s] = zp[s];
        }
        // ... and its previous tile, started four phases ago, has seen every code (the store comes after the copy, so
        // that waiting for the prefetched rows does not wait for this store as well); in the first round of slots 1..3
        // it is the early part of their last tile that ends here
        if (p >= 4) {
            float bv = best[Q];
            int bi = bidx[Q];
            if (Q != 0 && !(bv >= bestH[Q])) { bv = bestH[Q]; bi = bidxH[Q]; }   // best: after the wrap = lower codes
            store_tile(p - 4, bv, bi);
        } else if (Q != 0) {
            bestE[Q] = best[Q]; bidxE[Q] = bidx[Q];
        }
        best[Q] = -__builtin_inff();
        bidx[Q] = 0;
        if (Q == 0) {   // the codebook wraps here for the other three slots
#pragma unroll
            for (int j = 1; j < NB; ++j) {
                bestH[j] = best[j]; bidxH[j] = bidx[j];
                best[j] = -__builtin_inff(); bidx[j] = 0;
            }
        }
        const int s_begin = Q * n_stages / 4, s_end = (Q + 1) * n_stages / 4;
        for (int cs = s_begin; cs < s_end; ++cs, ++g) {
            const bool more = !(p == cnt - 1 && cs == s_end - 1);
            const bool zload = cs == s_begin && p + 1 < cnt;
            const char* eb = smem + (g & 1) * stage_bytes;
            const int t_first = cs * TS;
            const int t_count = (n_tiles - t_first) < TS ? (n_tiles - t_first) : TS;
            if constexpr (!STAGGER) {
                if (more) issue_stage(cs + 1 == n_stages ? 0 : cs + 1, (g + 1) & 1);
                if (zload) load_tile(p + 1);     // issued after the stage's copies: a counted wait can leave it in flight
                code_tile(std::false_type{}, eb, t_first * 32);
                for (int t = 1; t < t_count; ++t) code_tile(std::true_type{}, eb + t * TILE_BYTES, (t_first + t) * 32);
                flush_pending();
                if (zload) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(S) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                const int t_half = (t_count + 1) / 2;
                if (GRP == 1 && more) issue_stage(cs + 1 == n_stages ? 0 : cs + 1, (g + 1) & 1);   // start of an odd half stage
                if (zload) load_tile(p + 1);
                code_tile(std::false_type{}, eb, t_first * 32);
                for (int t = 1; t < t_half; ++t) code_tile(std::true_type{}, eb + t * TILE_BYTES, (t_first + t) * 32);
                if (GRP == 1) {                  // end of the odd half stage: this wave's copies have landed
                    if (zload) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(S) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                if (GRP == 0 && more) issue_stage(cs + 1 == n_stages ? 0 : cs + 1, (g + 1) & 1);   // start of an odd half stage
                for (int t = t_half; t < t_count; ++t) code_tile(std::true_type{}, eb + t * TILE_BYTES, (t_first + t) * 32);
                flush_pending();
                if (GRP == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the copies, and the rows requested before them
                __builtin_amdgcn_s_barrier();
            }
        }
    };
    if (STAGGER && GRP == 1) __builtin_amdgcn_s_barrier();       // the early waves' first half stage
    for (int p0 = 0; p0 < cnt; p0 += 4) {
        phase(std::integral_constant<int, 0>{}, p0);
        phase(std::integral_constant<int, 1>{}, p0 + 1);
        phase(std::integral_constant<int, 2>{}, p0 + 2);
        phase(std::integral_constant<int, 3>{}, p0 + 3);
    }
    if (STAGGER && GRP == 0) __builtin_amdgcn_s_barrier();       // the late waves' last half stage
    store_tile(cnt - 4, best[0], bidx[0]);
    // the last tiles of slots 1..3: quarters j..3 just now, quarters 0..j-1 (lower codes) in the first phases
#pragma unroll
    for (int j = 1; j < NB; ++j) {
        float bv = bestE[j];
        int bi = bidxE[j];
        if (!(bv >= best[j])) { bv = best[j]; bi = bidx[j]; }
        store_tile(cnt - 4 + j, bv, bi);
    }
}

// --------------------------------------------- generic path: any D % 16 == 0
// One wave owns 32 latents; z fragments are re-read (L1/L2) for every code tile and the
// codebook tile is staged through LDS in chunks of CH dims.  Slower; used for the shapes
// the register path does not cover (e.g. the 1024-wide object codebook).
// Winners found by several waves for one latent (vq_argmin_few_kernel: the code tiles are shared out) meet in idx_out
// through a 64-bit atomic min on (order-preserving distance bits, code); vq_unpack_key_kernel strips the distance
// afterwards.  idx_out must have been filled with 0xFF bytes.
__device__ __forceinline__ unsigned long long vq_pack_key(float d, int code) {
    const unsigned b = __builtin_bit_cast(unsigned, d);
    const unsigned ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)ord << 32) | (unsigned)code;
}
__global__ void vq_unpack_key_kernel(int64_t* __restrict__ idx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = idx[i] & 0xFFFFFFFFll;
}

template <int DT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void vq_argmin_generic_kernel(
    const typename VqTraits<DT>::elem* __restrict__ z, const char* __restrict__ packed,
    const float* __restrict__ e2, int64_t n, int D, int n_tiles, int64_t* __restrict__ idx_out) {
    constexpr int DPU = VqTraits<DT>::kDimsPerUnit;
    constexpr int CHU = 32;  // units (KiB) staged per chunk
    __shared__ __attribute__((aligned(16))) char e_lds[CHU * 1024];
    const int U = D / DPU;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31;
    const int h = lane >> 5;
    const int64_t n_groups = (n + WAVES * 32 - 1) / (WAVES * 32);
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        int64_t row = grp * (WAVES * 32) + wave * 32 + r;
        const bool live = row < n;
        if (!live) row = n - 1;
        const typename VqTraits<DT>::elem* zr = z + row * D;
        float part = 0.f;
        for (int d = h * (D / 2); d < (h + 1) * (D / 2); ++d) {
            float x;
            if (DT == DVQ_BF16) x = bf16_bits_to_f32((unsigned short)zr[d]); else x = (float)zr[d];
            part = __builtin_fmaf(x, x, part);
        }
        const float z2 = part + __shfl_xor(part, 32, 64);
        Best best; best.d = __builtin_inff(); best.k = 0;
        for (int t = 0; t < n_tiles; ++t) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            for (int u0 = 0; u0 < U; u0 += CHU) {
                const int nu = (U - u0) < CHU ? (U - u0) : CHU;
                __syncthreads();
                stage_copy(packed + ((int64_t)t * U + u0) * 1024, e_lds, nu * 1024, tid, WAVES * 64);
                __syncthreads();
                // four z fragment loads (global, L2-resident) are issued before the first of their MFMAs, so the
                // loop pays one load latency per four steps
                for (int s0 = 0; s0 < nu; s0 += 4) {
                    if constexpr (DT == DVQ_BF16) {
                        u16x8 raw[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int sj = (s0 + j < nu) ? s0 + j : nu - 1;
                            raw[j] = *reinterpret_cast<const u16x8*>(reinterpret_cast<const unsigned short*>(zr) + vq_dim_bf16(u0 + sj, h, 0));
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (s0 + j < nu) {
                                const bf16x8 a = *reinterpret_cast<const bf16x8*>(e_lds + (s0 + j) * 1024 + lane * 16);
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, raw[j]), acc, 0, 0, 0);
                            }
                        }
                    } else {
                        f32x4 bq[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int sj = (s0 + j < nu) ? s0 + j : nu - 1;
                            bq[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(zr) + vq_dim_f32(u0 + sj, h, 0, D));
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (s0 + j < nu) {
                                const f32x4 a = *reinterpret_cast<const f32x4*>(e_lds + (s0 + j) * 1024 + lane * 16);
#pragma unroll
                                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[j][i], acc, 0, 0, 0);
                            }
                        }
                    }
                }
            }
            // |E|^2 of this tile straight from global (L1-resident, 128 bytes)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 e = *reinterpret_cast<const f32x4*>(e2 + t * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t1 = z2 + e[q];
                    const float d = __builtin_fmaf(-2.0f, acc[4 * g + q], t1);
                    const int code = t * 32 + 8 * g + 4 * h + q;
                    if (d < best.d) { best.d = d; best.k = code; }
                }
            }
        }
        merge_halves(best);
        if (h == 0 && live) {
            idx_out[grp * (WAVES * 32) + wave * 32 + r] = (int64_t)best.k;
        }
    }
}

// Few latents (the object codebook is searched with one 1024-d row per object): one wave per (32 latents, share of the
// code tiles), no LDS and no barrier -- the packed codebook units and the latent fragments both go straight from
// global memory into MFMA operands, PF units requested ahead of the products that use them.  |z|^2 comes from the
// same fragments (fp32: the lane's half row in ascending order, the very fmaf chain of the generic kernel; bf16: its
// own pass, the fragments interleave the two halves).  Winners of the tile shares meet in idx_out through vq_pack_key
// (idx_out filled with 0xFF bytes, vq_unpack_key_kernel afterwards).
template <int DT>
__global__ __launch_bounds__(64) void vq_argmin_few_kernel(
    const typename VqTraits<DT>::elem* __restrict__ z, const char* __restrict__ packed,
    const float* __restrict__ e2, int64_t n, int D, int n_tiles, int64_t* __restrict__ idx_out) {
    constexpr int DPU = VqTraits<DT>::kDimsPerUnit;
    constexpr int PF = 8;
    const int U = D / DPU;
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    int64_t row = (int64_t)blockIdx.x * 32 + r;
    const bool live = row < n;
    if (!live) row = n - 1;
    const typename VqTraits<DT>::elem* zr = z + row * D;
    float part = 0.f;
    if (DT == DVQ_BF16) {
        for (int d = h * (D / 2); d < (h + 1) * (D / 2); d += 8) {
            const u16x8 v = *reinterpret_cast<const u16x8*>(reinterpret_cast<const unsigned short*>(zr) + d);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float x = bf16_bits_to_f32(v[j]); part = __builtin_fmaf(x, x, part); }
        }
    }
    Best best; best.d = __builtin_inff(); best.k = 0;
    bool first = true;
    float z2 = 0.f;
    for (int t = blockIdx.y; t < n_tiles; t += gridDim.y) {
        const char* eu = packed + (int64_t)t * U * 1024 + lane * 16;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        u32x4 a[PF], b[PF], an[PF], bn[PF];
        auto fetch = [&](int s0, u32x4 (&aa)[PF], u32x4 (&bb)[PF]) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int sj = (s0 + j < U) ? s0 + j : U - 1;
                aa[j] = *reinterpret_cast<const u32x4*>(eu + (int64_t)sj * 1024);
                if (DT == DVQ_BF16) bb[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(zr) + vq_dim_bf16(sj, h, 0));
                else bb[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(zr) + vq_dim_f32(sj, h, 0, D));
            }
        };
        fetch(0, a, b);
        for (int s0 = 0; s0 < U; s0 += PF) {
            if (s0 + PF < U) fetch(s0 + PF, an, bn);
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                if (s0 + j < U) {
                    if constexpr (DT == DVQ_BF16) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[j]), __builtin_bit_cast(bf16x8, b[j]), acc, 0, 0, 0);
                    } else {
                        const f32x4 af = __builtin_bit_cast(f32x4, a[j]), bf = __builtin_bit_cast(f32x4, b[j]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[i], acc, 0, 0, 0);
                            if (first) part = __builtin_fmaf(bf[i], bf[i], part);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < PF; ++j) { a[j] = an[j]; b[j] = bn[j]; }
        }
        if (first) { z2 = part + __shfl_xor(part, 32, 64); first = false; }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 e = *reinterpret_cast<const f32x4*>(e2 + t * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t1 = z2 + e[q];
                const float d = __builtin_fmaf(-2.0f, acc[4 * g + q], t1);
                const int code = t * 32 + 8 * g + 4 * h + q;
                if (d < best.d) { best.d = d; best.k = code; }
            }
        }
    }
    merge_halves(best);
    if (h == 0 && live) atomicMin(reinterpret_cast<unsigned long long*>(idx_out + (int64_t)blockIdx.x * 32 + r), vq_pack_key(best.d, best.k));
}

// ------------------------------------- wide rows (D a multiple of 256 / 128), many latents
// Rows too long for the register kernels (the 1024-wide object codebook).  The row is cut into chunks of 16 units
// (256 bf16 / 128 fp32 dims); a wave owns 32 latents and keeps, for a group of up to 128 codes, all four 32x32
// accumulators while the chunks go by, so every latent chunk is fetched once per code group and every codebook chunk
// (64 KiB for 128 codes) is staged once per 256 latents.  Both are prefetched one step ahead -- the codebook chunk by
// direct global->LDS copies into the other buffer, the latent chunk into a second set of registers -- so the latent
// stream from HBM, which bounds this shape at 128 codes, never stops.  Same packed layout, |E|^2 array and distance
// formula as the generic kernel.
template <int DT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void vq_argmin_wide_kernel(
    const typename VqTraits<DT>::elem* __restrict__ z, const char* __restrict__ packed, const float* __restrict__ e2,
    int64_t n, int D, int n_tiles, int64_t* __restrict__ idx_out) {
    constexpr int DPU = VqTraits<DT>::kDimsPerUnit;
    constexpr int CU_ = 16;                      // units per chunk
    constexpr int CT = 4;                        // code tiles (of 32) per group
    constexpr int STAGE_BYTES = CT * CU_ * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][STAGE_BYTES]
    typedef typename std::conditional<DT == DVQ_BF16, bf16x8, f32x4>::type frag;
    const int U = D / DPU;
    const int n_chunks = U / CU_;
    const int n_cg = (n_tiles + CT - 1) / CT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;
    const int64_t n_groups = (n + WAVES * 32 - 1) / (WAVES * 32);
    const int64_t my_groups = blockIdx.x < n_groups ? (n_groups - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int64_t steps = my_groups * n_cg * n_chunks;   // one step = one chunk of one code group for one latent group

    auto stage = [&](int cg, int c, int buf) {
        const int t_cnt = (n_tiles - cg * CT) < CT ? (n_tiles - cg * CT) : CT;
        for (int i = wave; i < t_cnt * CU_; i += WAVES) {
            const int ct = i / CU_, u = i % CU_;
            vq_glds16(packed + ((int64_t)(cg * CT + ct) * U + c * CU_ + u) * 1024 + lane * 16, smem + buf * STAGE_BYTES + i * 1024);
        }
    };
    auto load_z = [&](int64_t grp, int c, frag (&dst)[CU_]) {
        int64_t row = (blockIdx.x + grp * gridDim.x) * (WAVES * 32) + wave * 32 + r;
        if (row >= n) row = n - 1;
        const typename VqTraits<DT>::elem* zr = z + row * D;
#pragma unroll
        for (int s = 0; s < CU_; ++s) {
            if constexpr (DT == DVQ_BF16)
                dst[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8*>(reinterpret_cast<const unsigned short*>(zr) + vq_dim_bf16(c * CU_ + s, h, 0)));
            else
                dst[s] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(zr) + vq_dim_f32(c * CU_ + s, h, 0, D));
        }
    };

    frag zA[CU_], zB[CU_];
    f32x16 acc[CT];
    Best best;
    float z2part = 0.f;
    int64_t grp = 0;   // position of the CURRENT step
    int cg = 0, c = 0;
    int64_t k = 0;
    if (steps > 0) {
        stage(0, 0, 0);
        load_z(0, 0, zA);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __builtin_amdgcn_s_barrier();

    auto step = [&](frag (&zc)[CU_], frag (&zn)[CU_]) {
        // position of the next step
        int c1 = c + 1, cg1 = cg;
        int64_t grp1 = grp;
        if (c1 == n_chunks) { c1 = 0; if (++cg1 == n_cg) { cg1 = 0; ++grp1; } }
        if (k + 1 < steps) {
            stage(cg1, c1, (int)((k + 1) & 1));
            load_z(grp1, c1, zn);
        }
        if (c == 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
            if (cg == 0) { best.d = __builtin_inff(); best.k = 0; z2part = 0.f; }
        }
        const int t_cnt = (n_tiles - cg * CT) < CT ? (n_tiles - cg * CT) : CT;
        const char* lb = smem + (k & 1) * STAGE_BYTES + lane * 16;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if (ct < t_cnt) {
#pragma unroll
                for (int s = 0; s < CU_; ++s) {
                    if constexpr (DT == DVQ_BF16) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(lb + (ct * CU_ + s) * 1024);
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, zc[s], acc[ct], 0, 0, 0);
                    } else {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(lb + (ct * CU_ + s) * 1024);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], zc[s][i], acc[ct], 0, 0, 0);
                    }
                }
            }
        }
        if (cg == 0) {   // |z|^2 of this lane's half of the chunk
#pragma unroll
            for (int s = 0; s < CU_; ++s) {
                if constexpr (DT == DVQ_BF16) {
                    const u16x8 raw = __builtin_bit_cast(u16x8, zc[s]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float x = bf16_bits_to_f32(raw[j]); z2part = __builtin_fmaf(x, x, z2part); }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) z2part = __builtin_fmaf(zc[s][i], zc[s][i], z2part);
                }
            }
        }
        if (c == n_chunks - 1) {   // the group of codes is complete for these latents
            const float z2 = z2part + __shfl_xor(z2part, 32, 64);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                if (ct < t_cnt) scan_tile(acc[ct], e2 + (cg * CT + ct) * 32, h, (cg * CT + ct) * 32, z2, best);
            if (cg == n_cg - 1) {
                Best b = best;
                merge_halves(b);
                const int64_t row = (blockIdx.x + grp * gridDim.x) * (WAVES * 32) + wave * 32 + r;
                if (h == 0 && row < n) idx_out[row] = (int64_t)b.k;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // the prefetches have had the whole step to land
        __builtin_amdgcn_s_barrier();
        c = c1; cg = cg1; grp = grp1; ++k;
    };
    while (k < steps) {
        step(zA, zB);
        if (k < steps) step(zB, zA);
    }
}

// ------------------------------------------------------------------- row gather
// out[i, :] = E[idx[i], :] -- z_q of quantizer.py:50-53 and get_emb of :68-75 (the
// reference multiplies a one-hot matrix by E, which selects the row exactly).
template <typename T>
__global__ void vq_gather_kernel(const int64_t* __restrict__ idx, const T* __restrict__ E, int64_t n, int K, int D,
                                 T* __restrict__ out, int64_t ld_out) {
    const int vec = 16 / sizeof(T);
    const int chunks = D / vec;
    const int64_t total = n * chunks;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / chunks;
        const int c = (int)(t % chunks);
        int64_t k = idx[i];
        if (k < 0) k = 0;
        if (k >= K) k = K - 1;
        const uint4 v = *reinterpret_cast<const uint4*>(E + k * D + c * vec);
        *reinterpret_cast<uint4*>(out + i * ld_out + c * vec) = v;
    }
}

// ------------------------------------------------------------------ host side
struct VqPlan {
    int Kpad, n_tiles, U, tile_bytes;
    bool bf16x;   // bf16 fast-path layout: tiles of U+1 units with -|E|^2/2 embedded
    size_t packed_bytes, e2_off, total;
};

static bool vq_plan(int K, int D, int dtype, VqPlan* p) {
    if (K < 1 || K > 65536 || D < 16 || D > 4096 || (D % 16) != 0) return false;
    if (dtype != DVQ_F32 && dtype != DVQ_BF16) return false;
    p->Kpad = round_up(K, 32);
    p->n_tiles = p->Kpad / 32;
    p->U = D / (dtype == DVQ_BF16 ? 16 : 8);
    p->bf16x = dtype == DVQ_BF16 && (D == 64 || D == 128 || D == 256);
    p->tile_bytes = (p->U + (p->bf16x ? 1 : 0)) * 1024;
    p->packed_bytes = (size_t)p->n_tiles * p->tile_bytes;
    p->e2_off = (p->packed_bytes + 255) / 256 * 256;
    p->total = p->e2_off + (size_t)p->Kpad * 4;
    p->total = (p->total + 255) / 256 * 256;
    return true;
}

template <int DT, int D, int NB, int WAVES>
static int launch_reg(const VqBatch& vb, int nb, const VqPlan& p, int64_t n, hipStream_t st) {
    constexpr int U = D / VqTraits<DT>::kDimsPerUnit;
    constexpr int TILE_BYTES = U * 1024;
    int TS = 65536 / TILE_BYTES;  // 64 KiB of codebook per stage
    if (TS < 1) TS = 1;
    if (TS > p.n_tiles) TS = p.n_tiles;
    // a codebook of up to 144 KiB (fp32: the reference's 128 x 256) stays in LDS for the whole launch instead of being
    // re-staged in two halves, with two barriers, for every 256 latents
    if ((size_t)p.n_tiles * TILE_BYTES <= 144 * 1024) TS = p.n_tiles;
    const int n_stages = (p.n_tiles + TS - 1) / TS;
    const int n_buf = n_stages > 1 ? 2 : 1;
    const size_t lds = (size_t)n_buf * TS * TILE_BYTES + (size_t)n_buf * TS * 32 * 4;
    auto kern = vq_argmin_reg_kernel<DT, D, NB, WAVES>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "vq_argmin")) return rc;
    const int64_t groups = (n + WAVES * NB * 32 - 1) / (WAVES * NB * 32);
    int64_t max_blocks = ((lds > 80 * 1024) ? kNumCU : 2 * kNumCU) / nb;   // searches of a batch share the CUs
    if (max_blocks < 1) max_blocks = 1;
    const int grid = (int)(groups < max_blocks ? groups : max_blocks);
    hipLaunchKernelGGL(kern, dim3(grid, nb), dim3(WAVES * 64), lds, st, vb, n, p.n_tiles, TS);
    return check_launch("vq_argmin");
}

template <int D, int NB, int WAVES>
static int launch_bf16x(const VqBatch& vb, int nb, const VqPlan& p, int64_t n, hipStream_t st) {
    constexpr int TILE_BYTES = (D / 16 + 1) * 1024;
    int TS = (68 * 1024) / TILE_BYTES;   // 68-KiB stages: 4 groups of 32 codes at D=256
    if (TS < 1) TS = 1;
    if (TS > p.n_tiles) TS = p.n_tiles;
    if ((size_t)p.n_tiles * TILE_BYTES <= 144 * 1024) TS = p.n_tiles;   // up to 256 codes of 256 dims stay in LDS
    const int n_stages = (p.n_tiles + TS - 1) / TS;
    const size_t lds = (size_t)(n_stages > 1 ? 2 : 1) * TS * TILE_BYTES;
    auto kern = vq_argmax_bf16_kernel<D, NB, WAVES>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "vq_argmin")) return rc;
    const int64_t groups = (n + WAVES * NB * 16 - 1) / (WAVES * NB * 16);
    int64_t max_blocks = ((lds > 80 * 1024) ? kNumCU : 2 * kNumCU) / nb;
    if (max_blocks < 1) max_blocks = 1;
    const int grid = (int)(groups < max_blocks ? groups : max_blocks);
    hipLaunchKernelGGL(kern, dim3(grid, nb), dim3(WAVES * 64), lds, st, vb, n, p.n_tiles, TS);
    return check_launch("vq_argmin(bf16)");
}

template <int D, int WAVES>
static int launch_bf16_rot(const void* z, const char* ws, const VqPlan& p, int64_t n, int64_t* idx_out, hipStream_t st) {
    constexpr int TILE_BYTES = (D / 16 + 1) * 1024;
    int TS = (68 * 1024) / TILE_BYTES;
    if (TS > p.n_tiles / 4) TS = p.n_tiles / 4;
    const size_t lds = (size_t)2 * TS * TILE_BYTES;
    // the two waves of every SIMD half a stage apart: +1 to +1.3 points of the matrix peak at 1024 and 2048 codes, +0.5 to
    // +1 at 512 (same box, alternating runs), same results
    auto kern = vq_argmax_bf16_rot_kernel<D, WAVES, true>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "vq_argmin")) return rc;
    const int64_t n_zt = (n + 15) / 16;
    const int64_t blocks = (n_zt + WAVES * 4 - 1) / (WAVES * 4);   // at least one full rotation per wave
    const int grid = (int)(blocks < kNumCU ? blocks : kNumCU);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, st, reinterpret_cast<const unsigned short*>(z), ws, n, p.n_tiles, TS, idx_out);
    return check_launch("vq_argmin(bf16, rotating)");
}

// the register-resident kernels (D = 64, 128, 256), one search or a batch of searches of one shape
template <int DT>
static int launch_fast(const VqBatch& vb, int nb, const VqPlan& p, int64_t n, int D, hipStream_t st) {
    if (DT == DVQ_BF16) {
        // A codebook that stays in LDS (up to 128 codes: the reference's size) leaves the search bound by the latent
        // stream from HBM, i.e. by the bytes a CU keeps in flight: two latent tiles per wave need half the registers
        // of four, so two workgroups (16 waves) share a CU and one loads while the other computes -- 4.7 -> 5.0 TB/s
        // at 1 Mi latents, 3.4 -> 4.5 at 256 Ki; beyond a few Mi rows four tiles per wave are ahead again (5.2 vs 5.4).
        if (D == 256 && p.n_tiles <= 4 && n <= kVqTwoTileMaxRows) return launch_bf16x<256, 2, 8>(vb, nb, p, n, st);
        // up to 256 codes stay in LDS too (136 KiB: one workgroup per CU): sixteen waves of two tiles, so that some waves
        // load while others compute -- 1 Mi latents 0.148-0.184 ms with eight waves of four tiles, 0.132-0.156 ms with
        // sixteen of two, alternating in one process; the rotating sweep used before: 0.16-0.18 ms
        if (D == 256 && p.n_tiles <= 8) return launch_bf16x<256, 2, 16>(vb, nb, p, n, st);
        if (D == 256) return launch_bf16x<256, 4, 8>(vb, nb, p, n, st);
        if (D == 128) return launch_bf16x<128, 4, 8>(vb, nb, p, n, st);
        return launch_bf16x<64, 4, 8>(vb, nb, p, n, st);
    }
    if (D == 256) return launch_reg<DVQ_F32, 256, 1, 8>(vb, nb, p, n, st);
    if (D == 128) return launch_reg<DVQ_F32, 128, 2, 8>(vb, nb, p, n, st);
    return launch_reg<DVQ_F32, 64, 2, 8>(vb, nb, p, n, st);
}

template <int DT>
static int launch_search(const void* z, const char* ws, const VqPlan& p, int64_t n, int D, int64_t* idx_out, hipStream_t st) {
    if (n == 0) return DVQ_OK;
    if (DT == DVQ_BF16) {
        // 256 to 2048 codes and enough latents for every wave to keep four tiles in rotation (measured: 1.2x at 256
        // codes, 1.14x at 512, 1.07x at 1024, even at 2048, behind the one-sweep kernel at 8192)
        if (D == 256 && p.n_tiles >= kVqRotMinTiles && p.n_tiles <= kVqRotMaxTiles && n >= kVqRotMinRows) return launch_bf16_rot<256, 8>(z, ws, p, n, idx_out, st);
    }
    if (D == 256 || D == 128 || D == 64) {
        VqBatch vb = {};
        vb.z[0] = z; vb.packed[0] = ws; vb.e2[0] = reinterpret_cast<const float*>(ws + p.e2_off); vb.idx[0] = idx_out;
        return launch_fast<DT>(vb, 1, p, n, D, st);
    }
    // long rows, many latents
    if (n > (int64_t)32 * 4 * kNumCU && (D % (16 * VqTraits<DT>::kDimsPerUnit)) == 0) {
        constexpr int WAVES = 8;
        const size_t lds = 2 * 4 * 16 * 1024;
        auto kern = vq_argmin_wide_kernel<DT, WAVES>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "vq_argmin")) return rc;
        const int64_t groups = (n + WAVES * 32 - 1) / (WAVES * 32);
        const int grid = (int)(groups < kNumCU ? groups : kNumCU);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, st, reinterpret_cast<const typename VqTraits<DT>::elem*>(z), ws,
                           reinterpret_cast<const float*>(ws + p.e2_off), n, D, p.n_tiles, idx_out);
        return check_launch("vq_argmin(wide)");
    }
    // few latents (the object codebook is searched with one 1024-d row per object): one wave per workgroup, so that the
    // n/32 waves spread over as many CUs as possible instead of sharing a quarter of them
    if (n <= (int64_t)32 * 4 * kNumCU) {
        const int64_t groups = (n + 31) / 32;
        int split = (int)((4 * kNumCU + groups - 1) / groups);     // aim at four waves per CU
        if (split > p.n_tiles) split = p.n_tiles;
        if (split > 1) {
            if (hipMemsetAsync(idx_out, 0xFF, (size_t)n * sizeof(int64_t), st) != hipSuccess)
                return fail(DVQ_ELAUNCH, "vq_argmin: cannot initialise the index buffer");
            hipLaunchKernelGGL((vq_argmin_few_kernel<DT>), dim3((unsigned)groups, (unsigned)split), dim3(64), 0, st,
                               reinterpret_cast<const typename VqTraits<DT>::elem*>(z), ws, reinterpret_cast<const float*>(ws + p.e2_off),
                               n, D, p.n_tiles, idx_out);
            hipLaunchKernelGGL(vq_unpack_key_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, idx_out, n);
            return check_launch("vq_argmin(few latents)");
        }
        hipLaunchKernelGGL((vq_argmin_generic_kernel<DT, 1>), dim3((unsigned)groups), dim3(64), 0, st,
                           reinterpret_cast<const typename VqTraits<DT>::elem*>(z), ws, reinterpret_cast<const float*>(ws + p.e2_off),
                           n, D, p.n_tiles, idx_out);
        return check_launch("vq_argmin(generic)");
    }
    constexpr int WAVES = 4;
    const int64_t groups = (n + WAVES * 32 - 1) / (WAVES * 32);
    const int grid = (int)(groups < 8 * kNumCU ? groups : 8 * kNumCU);
    hipLaunchKernelGGL((vq_argmin_generic_kernel<DT, WAVES>), dim3(grid), dim3(WAVES * 64), 0, st,
                       reinterpret_cast<const typename VqTraits<DT>::elem*>(z), ws, reinterpret_cast<const float*>(ws + p.e2_off),
                       n, D, p.n_tiles, idx_out);
    return check_launch("vq_argmin(generic)");
}

static int vq_check(const void* z, const void* E, int64_t n, int K, int D, int dtype, const int64_t* idx_out,
                    const void* ws, VqPlan* p) {
    if (!vq_plan(K, D, dtype, p))
        return fail(DVQ_EINVAL, "vq: unsupported shape K=%d D=%d dtype=%d (need 1<=K<=65536, 16<=D<=4096, D%%16==0)", K, D, dtype);
    if (n < 0) return fail(DVQ_EINVAL, "vq: negative n");
    if (!E || !ws || (n > 0 && (!z || !idx_out))) return fail(DVQ_EINVAL, "vq: null pointer");
    if (((uintptr_t)ws & 255) != 0) return fail(DVQ_EINVAL, "vq: workspace must be 256-byte aligned");
    if (((uintptr_t)E & 15) != 0 || ((uintptr_t)z & 15) != 0) return fail(DVQ_EINVAL, "vq: z and E must be 16-byte aligned");
    return DVQ_OK;
}

static int vq_prepare_impl(const void* E, int K, int D, int dtype, void* ws, const VqPlan& p, hipStream_t st) {
    const int64_t nchunks = (int64_t)p.n_tiles * p.U * 64;
    const int64_t threads = nchunks > p.Kpad ? nchunks : p.Kpad;
    const int grid = (int)((threads + 255) / 256);
    char* w = reinterpret_cast<char*>(ws);
    if (p.bf16x) {
        const int64_t nch = (int64_t)p.n_tiles * (p.U + 1) * 64;   // U = D/16 = 2 tiles x D/32 k-steps
        hipLaunchKernelGGL(vq_pack_bf16x_kernel, dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const unsigned short*>(E), K, D, p.Kpad, reinterpret_cast<uint4*>(w));
        return check_launch("vq_prepare");
    }
    if (dtype == DVQ_BF16)
        hipLaunchKernelGGL((vq_pack_kernel<DVQ_BF16>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const unsigned short*>(E), K, D,
                           p.Kpad, reinterpret_cast<uint4*>(w), reinterpret_cast<float*>(w + p.e2_off));
    else
        hipLaunchKernelGGL((vq_pack_kernel<DVQ_F32>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const float*>(E), K, D, p.Kpad,
                           reinterpret_cast<uint4*>(w), reinterpret_cast<float*>(w + p.e2_off));
    return check_launch("vq_prepare");
}

static int vq_gather_impl(const int64_t* idx, const void* E, int64_t n, int K, int D, int dtype, void* out, int64_t ld,
                          hipStream_t st) {
    if (n == 0) return DVQ_OK;
    const int vec = dtype == DVQ_BF16 ? 8 : 4;
    const int64_t total = n * (D / vec);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == DVQ_BF16)
        hipLaunchKernelGGL((vq_gather_kernel<unsigned short>), dim3(grid), dim3(256), 0, st, idx, reinterpret_cast<const unsigned short*>(E),
                           n, K, D, reinterpret_cast<unsigned short*>(out), ld);
    else
        hipLaunchKernelGGL((vq_gather_kernel<float>), dim3(grid), dim3(256), 0, st, idx, reinterpret_cast<const float*>(E), n, K, D,
                           reinterpret_cast<float*>(out), ld);
    return check_launch("vq_gather");
}

// Device address of `scratch` on the current device, looked up once per device.
constexpr int kVqMaxDevices = 64;
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[kVqMaxDevices];

static int vq_static_scratch(void** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kVqMaxDevices)
        return fail(DVQ_ENODEV, "vq_argmin_static: no current HIP device (or device ordinal >= %d)", kVqMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(DVQ_ELAUNCH, "vq_argmin_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return DVQ_OK;
}

}  // namespace dvq

using namespace dvq;

extern "C" size_t dvq_vq_workspace_bytes(int K, int D, int dtype) {
    VqPlan p;
    if (!vq_plan(K, D, dtype, &p)) {
        set_error("vq: unsupported shape K=%d D=%d dtype=%d", K, D, dtype);
        return 0;
    }
    return p.total;
}

extern "C" int dvq_vq_prepare(const void* E, int K, int D, int dtype, void* workspace, void* stream) {
    VqPlan p;
    int rc = vq_check(nullptr, E, 0, K, D, dtype, nullptr, workspace, &p);
    if (rc) return rc;
    return vq_prepare_impl(E, K, D, dtype, workspace, p, as_stream(stream));
}

extern "C" int dvq_vq_argmin_prepared(const void* z, const void* E, int64_t n, int K, int D, int dtype, int64_t* idx_out,
                                      void* zq_out, const void* workspace, void* stream) {
    VqPlan p;
    int rc = vq_check(z, E, n, K, D, dtype, idx_out, workspace, &p);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    const char* ws = reinterpret_cast<const char*>(workspace);
    rc = (dtype == DVQ_BF16) ? launch_search<DVQ_BF16>(z, ws, p, n, D, idx_out, st) : launch_search<DVQ_F32>(z, ws, p, n, D, idx_out, st);
    if (rc) return rc;
    if (zq_out) return vq_gather_impl(idx_out, E, n, K, D, dtype, zq_out, D, st);
    return DVQ_OK;
}

extern "C" int dvq_vq_argmin_prepared_batched(int n_batch, const void* const* z, const void* const* E, int64_t n, int K, int D,
                                              int dtype, int64_t* const* idx_out, const void* const* workspace, void* stream) {
    if (n_batch < 0 || n_batch > kVqMaxBatch) return fail(DVQ_EINVAL, "vq (batched): 0..%d searches, got %d", kVqMaxBatch, n_batch);
    if (n_batch == 0) return DVQ_OK;
    if (!z || !E || !idx_out || !workspace) return fail(DVQ_EINVAL, "vq: null pointer");
    VqPlan p;
    VqBatch vb = {};
    for (int i = 0; i < n_batch; ++i) {
        if (int rc = vq_check(z[i], E[i], n, K, D, dtype, idx_out[i], workspace[i], &p)) return rc;
        const char* ws = reinterpret_cast<const char*>(workspace[i]);
        vb.z[i] = z[i]; vb.packed[i] = ws; vb.e2[i] = reinterpret_cast<const float*>(ws + p.e2_off); vb.idx[i] = idx_out[i];
    }
    if (n == 0) return DVQ_OK;
    hipStream_t st = as_stream(stream);
    const bool rot = dtype == DVQ_BF16 && D == 256 && p.n_tiles >= kVqRotMinTiles && p.n_tiles <= kVqRotMaxTiles && n >= kVqRotMinRows;
    if ((D == 256 || D == 128 || D == 64) && !rot)
        return dtype == DVQ_BF16 ? launch_fast<DVQ_BF16>(vb, n_batch, p, n, D, st) : launch_fast<DVQ_F32>(vb, n_batch, p, n, D, st);
    for (int i = 0; i < n_batch; ++i) {   // shapes without a batched kernel: one launch each
        const int rc = dtype == DVQ_BF16 ? launch_search<DVQ_BF16>(z[i], vb.packed[i], p, n, D, idx_out[i], st)
                                         : launch_search<DVQ_F32>(z[i], vb.packed[i], p, n, D, idx_out[i], st);
        if (rc) return rc;
    }
    return DVQ_OK;
}

extern "C" int dvq_vq_argmin(const void* z, const void* E, int64_t n, int K, int D, int dtype, int64_t* idx_out, void* zq_out,
                             void* workspace, void* stream) {
    VqPlan p;
    int rc = vq_check(z, E, n, K, D, dtype, idx_out, workspace, &p);
    if (rc) return rc;
    rc = vq_prepare_impl(E, K, D, dtype, workspace, p, as_stream(stream));
    if (rc) return rc;
    return dvq_vq_argmin_prepared(z, E, n, K, D, dtype, idx_out, zq_out, workspace, stream);
}

extern "C" int dvq_vq_argmin_static(const void* z, const void* E, int64_t n, int K, int D, int dtype, int64_t* idx_out,
                                    void* zq_out, void* stream) {
    VqPlan p;
    if (!vq_plan(K, D, dtype, &p))
        return fail(DVQ_EINVAL, "vq: unsupported shape K=%d D=%d dtype=%d (need 1<=K<=65536, 16<=D<=4096, D%%16==0)", K, D, dtype);
    if (p.total > (size_t)DVQ_VQ_STATIC_WORKSPACE_BYTES)
        return fail(DVQ_EINVAL, "vq_argmin_static: codebook K=%d D=%d dtype=%d needs a workspace of %zu bytes, the built-in one "
                    "holds %zu; use dvq_vq_argmin with a caller-owned workspace", K, D, dtype, p.total,
                    (size_t)DVQ_VQ_STATIC_WORKSPACE_BYTES);
    void* ws = nullptr;
    if (int rc = vq_static_scratch(&ws)) return rc;
    // every byte the search reads (packed tiles incl. padding codes, |E|^2 of all Kpad codes) is rewritten by the
    // prepare step below, so nothing of the previous call's contents has to be cleared
    return dvq_vq_argmin(z, E, n, K, D, dtype, idx_out, zq_out, ws, stream);
}

extern "C" int dvq_vq_gather(const int64_t* idx, const float* E, int64_t n, int K, int D, float* out, int64_t ld_out, void* stream) {
    if (n < 0 || K < 1 || D < 4 || (D % 4) != 0 || ld_out < D || (ld_out % 4) != 0)
        return fail(DVQ_EINVAL, "vq_gather: bad shape n=%lld K=%d D=%d ld=%lld", (long long)n, K, D, (long long)ld_out);
    if (n > 0 && (!idx || !E || !out)) return fail(DVQ_EINVAL, "vq_gather: null pointer");
    if (((uintptr_t)E & 15) || ((uintptr_t)out & 15)) return fail(DVQ_EINVAL, "vq_gather: E and out must be 16-byte aligned");
    return vq_gather_impl(idx, E, n, K, D, DVQ_F32, out, ld_out, as_stream(stream));
}
