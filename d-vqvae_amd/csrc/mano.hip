// MANO hand layer: PCA pose -> Rodrigues -> kinematic chain -> pose blend shapes ->
// linear blend skinning, one launch (gfx950).
//
// Replaces MANO.forward (DVQ-VAE-2/MANO-master/mano/model.py:311-359) and lbs with its
// helpers (mano/lbs.py:33-265) for use_pca=True, pose2rot=True.  The reference always
// passes its own all-zero betas to lbs (model.py:335), so v_shaped = v_template and the
// rest joints J = J_regressor * v_shaped are constants supplied by the host.
//
// Per hand: 244 B in, 9.3 KB out and ~0.5 M multiply-adds, of which the pose blend-shape product
// [135] x [135, 3V] is the bulk.  A workgroup takes 16 hands:
//   A  Rodrigues, one thread per (hand, joint);
//   B  the kinematic chain, one thread per (hand, matrix row): row a of T_i = T_parent(i) [R_i | t_i] depends on
//      row a of the parent alone, so the three rows of a hand advance side by side;
//   C  per group of 16 vertices the blend shapes as a 16x16x4 fp32 MFMA product -- posedirs^T (16 vertices of one
//      coordinate axis x 4 pose features) times the pose features (4 x 16 hands); the accumulator then holds, per lane,
//      ONE hand and four whole vertices (the x, y and z tiles line up), so skinning continues in registers: the
//      transform of a vertex is gathered from its few non-zero skinning weights (1-6 of 16; table of (joint, weight)
//      pairs in ascending joint order, so the sum equals the reference's dense one bit for bit: the skipped terms are
//      exact zeros), and the lane writes its four vertices as 48 contiguous bytes.
// posedirs (1.26 MB, L2-resident) is read once per workgroup, 12 B per lane.
#include "common.h"

namespace dvq {

constexpr int kMH = 16;       // hands per workgroup
constexpr int kJ = 16;        // joints
constexpr int kPF = 135;      // pose-feature length = 15 joints * 9
constexpr int kKS = 34;       // MFMA k-steps of 4 over the pose features (135 + one zero)
constexpr int kKSP = 36;      // k-steps of the lane-ordered copy of posedirs (the last two are zeros): three turns of a ring of 12
constexpr int kManoThreads = 256;    // dense form: phase A uses all of them (hand x joint), phase B the first 48, phase C four waves
constexpr int kSkinW = 8;     // slots per vertex in the skinning table
constexpr int kSlotGroups = 128;   // vertex groups whose slot count is kept in LDS (2048 vertices; later groups take all slots)

struct f32x3 { float x, y, z; };
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef f32x4 f32x4u __attribute__((aligned(4)));   // four floats at any 4-byte boundary (a hand's rows start at multiples of 8 bytes)

// TABLE: skinning weights come as kSkinW (joint, weight) slots per vertex; otherwise as the dense [V,16] matrix.
// WAVES = 8 (table form): two waves per SIMD, one's skinning under the other's MFMAs; the group's operand then comes in
// six chunks of six k-steps through two register buffers (102 registers for all of it would not leave room for two
// waves; chunks of nine leave the kernel a few registers short, and a spilled register here costs more than the chunking gains)
// and the pose features are read from LDS at each step instead of living in 34 registers.
template <bool TABLE, int WAVES, bool LANES>
__global__ __launch_bounds__(WAVES * 64) void mano_lbs_kernel(
    const float* __restrict__ global_orient, const float* __restrict__ hand_pose, const float* __restrict__ transl,
    int ld_go, int ld_hp, int ld_tr, int B, int V, int ncomp, const float* __restrict__ v_shaped, const float* __restrict__ joints,
    const float* __restrict__ posedirs, const float* __restrict__ weights, const float* __restrict__ comps,
    const float* __restrict__ pose_mean, const int* __restrict__ parents, const unsigned char* __restrict__ skin_joints,
    const float* __restrict__ skin_weights, const float* __restrict__ pd_lanes, float* __restrict__ verts_out,
    float* __restrict__ joints_out, float* __restrict__ full_pose_out, float* __restrict__ v_posed_out) {
    __shared__ float sR[kMH][kJ][9];     // rotation matrices
    // relative transforms, upper 3x4; 4 floats of padding per hand: the skinning gather reads sA[h][j] with h = lane & 15,
    // and an unpadded hand stride (192 words) would put all sixteen hands on the same LDS banks
    __shared__ __attribute__((aligned(16))) float sAp[kMH][kJ * 12 + 4];
    auto sA = [&](int hh, int jj) -> float* { return &sAp[hh][jj * 12]; };
    __shared__ float sT[kMH][kJ][12];    // world transforms, upper 3x4
    __shared__ float sPF[kMH][kKSP * 4]; // pose feature (R[1:] - I), zero-padded
    __shared__ float sTr[kMH][4];        // translation (0 when absent)
    __shared__ float sHP[kMH][48];       // PCA coefficients of the block's hands
    __shared__ float sComps[45 * 45];    // PCA basis
    __shared__ float sJ[kJ * 3];         // rest joints
    __shared__ int sPar[kJ];
    __shared__ int sSlots[kSlotGroups];  // table form: slots in use by any vertex of a group of 16
    // per wave: a group's 16 hands x 16 vertices x 3 results on their way out (row padded from 48 so that the hands'
    // rows start on different banks); [..][0] takes the vertices, [..][1] the pre-skinning vertices
    __shared__ __attribute__((aligned(16))) float sOut[WAVES][2][kMH][52];

    constexpr int THREADS = WAVES * 64;
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * kMH;
    const int nh = (B - b0) < kMH ? (B - b0) : kMH;

    // Everything the two serial phases read goes to LDS in one round of coalesced loads: fetched where it is used
    // (45 dependent steps of the PCA product, 16 of the chain) each load would cost a trip to L2.
    for (int i = tid; i < nh * ncomp; i += THREADS) sHP[i / ncomp][i % ncomp] = hand_pose[(int64_t)(b0 + i / ncomp) * ld_hp + i % ncomp];
    for (int i = tid; i < ncomp * 45; i += THREADS) sComps[i] = comps[i];
    if (tid < kJ * 3) sJ[tid] = joints[tid];
    if (tid < kJ) sPar[tid] = parents[tid];
    if (TABLE && tid < kSlotGroups) sSlots[tid] = 0;
    __syncthreads();
    if (TABLE) {   // (joined with the other phases' results by the barrier after the Rodrigues phase)
        for (int v = tid; v < V && v < 16 * kSlotGroups; v += THREADS) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(skin_weights + (int64_t)v * kSkinW);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(skin_weights + (int64_t)v * kSkinW + 4);
            int cnt = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (w0[e] != 0.f) cnt = e + 1;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (w1[e] != 0.f) cnt = e + 5;
            }
            atomicMax(&sSlots[v >> 4], cnt);
        }
    }

    // ---- A: full pose and Rodrigues, one thread per (hand, joint)   (lbs.py:167-198)
    if (tid < kMH * kJ) {
        const int hnd = tid / kJ, j = tid % kJ;
        if (j < kKSP * 4 - kPF) sPF[hnd][kPF + j] = 0.f;
        if (hnd < nh) {
            const int b = b0 + hnd;
            float rv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v;
                if (j == 0) {
                    v = global_orient ? global_orient[(int64_t)b * ld_go + c] : 0.f;
                } else {
                    const int col = 3 * (j - 1) + c;
                    v = 0.f;
                    for (int i = 0; i < ncomp; ++i) v = __builtin_fmaf(sHP[hnd][i], sComps[i * 45 + col], v);
                }
                v += pose_mean[3 * j + c];
                rv[c] = v;
                if (full_pose_out) full_pose_out[b * 48 + 3 * j + c] = v;
            }
            const float ax = rv[0] + 1e-8f, ay = rv[1] + 1e-8f, az = rv[2] + 1e-8f;
            const float angle = sqrtf(ax * ax + ay * ay + az * az);
            const float rx = rv[0] / angle, ry = rv[1] / angle, rz = rv[2] / angle;
            const float cs = cosf(angle), sn = sinf(angle);
            const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
            const float omc = 1.f - cs;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float kk = K[a * 3 + 0] * K[0 * 3 + c];
                    kk = __builtin_fmaf(K[a * 3 + 1], K[1 * 3 + c], kk);
                    kk = __builtin_fmaf(K[a * 3 + 2], K[2 * 3 + c], kk);
                    const float id = (a == c) ? 1.f : 0.f;
                    const float Rv = (id + sn * K[a * 3 + c]) + omc * kk;
                    sR[hnd][j][a * 3 + c] = Rv;
                    if (j > 0) sPF[hnd][(j - 1) * 9 + a * 3 + c] = Rv - id;
                }
            if (j == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) sTr[hnd][c] = transl ? transl[(int64_t)b * ld_tr + c] : 0.f;
            }
        } else if (j > 0) {   // hands past the end of the batch: finite operands for the matrix product
#pragma unroll
            for (int e = 0; e < 9; ++e) sPF[hnd][(j - 1) * 9 + e] = 0.f;
        }
    }
    __syncthreads();

    // ---- B: kinematic chain, one thread per (hand, row of the 3x4 transform)   (lbs.py:214-265)
    if (tid < 3 * nh) {
        const int hnd = tid / 3, a = tid % 3;
        for (int i = 0; i < kJ; ++i) {
            const int par = sPar[i];
            float t[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) t[c] = (i == 0) ? sJ[c] : sJ[i * 3 + c] - sJ[par * 3 + c];
            float T[4];
            if (i == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) T[c] = sR[hnd][0][a * 3 + c];
                T[3] = t[a];
            } else {
                const float* P = &sT[hnd][par][a * 4];   // written by this very thread in an earlier step
                const float p0 = P[0], p1 = P[1], p2 = P[2], p3 = P[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v = p0 * sR[hnd][i][0 * 3 + c];
                    v = __builtin_fmaf(p1, sR[hnd][i][1 * 3 + c], v);
                    v = __builtin_fmaf(p2, sR[hnd][i][2 * 3 + c], v);
                    T[c] = v;
                }
                float v = p0 * t[0];
                v = __builtin_fmaf(p1, t[1], v);
                v = __builtin_fmaf(p2, t[2], v);
                T[3] = v + p3;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) sT[hnd][i][a * 4 + e] = T[e];
            // A = T - pad(T * [J;0])   (lbs.py:260-263)
            float v = T[0] * sJ[i * 3 + 0];
            v = __builtin_fmaf(T[1], sJ[i * 3 + 1], v);
            v = __builtin_fmaf(T[2], sJ[i * 3 + 2], v);
            sA(hnd, i)[a * 4 + 0] = T[0];
            sA(hnd, i)[a * 4 + 1] = T[1];
            sA(hnd, i)[a * 4 + 2] = T[2];
            sA(hnd, i)[a * 4 + 3] = T[3] - v;
            if (joints_out) joints_out[((b0 + hnd) * kJ + i) * 3 + a] = T[3] + sTr[hnd][a];
        }
    }
    __syncthreads();

    // ---- C: per group of 16 vertices: pose blend shapes on the matrix cores, then skinning   (lbs.py:90-118)
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane & 15, q = lane >> 4;
    constexpr bool CHUNKED = WAVES == 8;
    float bfrag[CHUNKED ? 1 : kKS];                     // B operand: pose feature 4s + q of hand h
    if (!CHUNKED) {
#pragma unroll
        for (int s = 0; s < kKS; ++s) bfrag[s] = sPF[h][4 * s + q];
    }
    const int n_groups = (V + 15) / 16;
    const int64_t prow = 3 * (int64_t)V;
    // A operand of group g: vertex 16g + (lane & 15), pose feature 4s + q; tile t = coordinate axis t.  All 34 k-steps
    // of a group are fetched at once (12 B per lane each) and the next group's are requested before this group's
    // skinning, so the L2 latency is paid under vector work.  Feature 135 does not exist: its B operand is zero, the
    // load is clamped to feature 134; vertices past the end are clamped too and never stored.
    constexpr int CH = 6, NCH = (kKS + CH - 1) / CH;     // chunked form: k-steps per chunk, chunks per group
    static_assert(NCH % 2 == 0, "the last chunk must use the second buffer: the next group's first chunk is loaded into the first meanwhile");
    f32x3 afrag[CHUNKED ? 1 : kKS];
    f32x3 abuf[2][CHUNKED ? CH : 1];
    auto load_group = [&](int g) {
        int vr = 16 * g + h;
        if (vr >= V) vr = V - 1;
        const float* pd = posedirs + 3 * (int64_t)vr;
#pragma unroll
        for (int s = 0; s < kKS; ++s) {
            const int k = (4 * s + q) < kPF ? (4 * s + q) : kPF - 1;
            afrag[CHUNKED ? 0 : s] = *reinterpret_cast<const f32x3*>(pd + k * prow);
        }
    };
    auto load_chunk = [&](int g, int ci, f32x3* dst) {
        int vr = 16 * g + h;
        if (vr >= V) vr = V - 1;
        const float* pd = posedirs + 3 * (int64_t)vr;
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int ks = ci * CH + s;
            if (ks < kKS) {
                const int k = (4 * ks + q) < kPF ? (4 * ks + q) : kPF - 1;
                dst[s] = *reinterpret_cast<const f32x3*>(pd + k * prow);
            }
        }
    };
    // the skinning slots of this lane's four vertices, requested before the group's MFMAs and used after them
    constexpr int NSLOT = TABLE ? kSkinW : kJ;
    f32x4 sw[4][NSLOT / 4];
    uint2 sj[4];                                        // eight joint ids of a vertex, one byte each
    f32x3 vs[4];                                        // rest position of the four vertices
    auto load_skin = [&](int g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int v = 16 * g + 4 * q + i;
            if (v >= V) v = V - 1;
#pragma unroll
            for (int c = 0; c < NSLOT / 4; ++c)
                sw[i][c] = *reinterpret_cast<const f32x4*>((TABLE ? skin_weights : weights) + (int64_t)v * NSLOT + 4 * c);
            if (TABLE) sj[i] = *reinterpret_cast<const uint2*>(skin_joints + (int64_t)v * kSkinW);
            vs[i] = *reinterpret_cast<const f32x3*>(v_shaped + (int64_t)v * 3);
        }
    };
    // Lane-ordered posedirs: a ring of 12 k-steps in registers, refilled one k-step at a time right behind the products that
    // free a slot, so that 11 loads (8.4 KB per wave) are in flight all the time -- during the skinning too.
    constexpr bool RINGED = CHUNKED && LANES;
    constexpr int RING = 12;
    static_assert(kKSP % RING == 0, "a k-step keeps its ring slot from group to group");
    f32x3 ring[RINGED ? RING : 1];
    const float* pdl = LANES ? pd_lanes + lane * 3 : nullptr;
    auto ring_load = [&](int g, int ks) { return *reinterpret_cast<const f32x3*>(pdl + ((int64_t)g * kKSP + ks) * 192); };
    if (wave < n_groups) {
        if constexpr (RINGED) {
#pragma unroll
            for (int s = 0; s < RING - 1; ++s) ring[s] = ring_load(wave, s);
        } else if (CHUNKED) {
            load_chunk(wave, 0, abuf[0]);
        } else {
            load_group(wave);
        }
    }
    for (int g = wave; g < n_groups; g += WAVES) {
        const int v0 = 16 * g;
        if (TABLE) load_skin(g);
        f32x4 acc[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (RINGED) {
            // (the next group's loads are unconditional -- the last group re-reads itself -- and a scheduling barrier after
            // every step keeps each load where it is written: left to itself the compiler sinks the loads next to their use)
            const int gn = g + WAVES < n_groups ? g + WAVES : g;
            float bfr[3];
            bfr[0] = sPF[h][q];
            bfr[1] = sPF[h][4 + q];
#pragma unroll
            for (int ks = 0; ks < kKSP; ++ks) {
                const int nk = ks + RING - 1;          // goes into the slot the previous step freed
                ring[nk % RING] = nk < kKSP ? ring_load(g, nk) : ring_load(gn, nk - kKSP);
                if (ks + 2 < kKSP) bfr[(ks + 2) % 3] = sPF[h][4 * (ks + 2) + q];
                const float bf = bfr[ks % 3];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[ks % RING].x, bf, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[ks % RING].y, bf, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[ks % RING].z, bf, acc[2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (CHUNKED) {
#pragma unroll
            for (int ci = 0; ci < NCH; ++ci) {
                // the next chunk (or the next group's first) is requested into the other buffer before this one is used
                if (ci + 1 < NCH) load_chunk(g, ci + 1, abuf[(ci + 1) & 1]);
                else if (g + WAVES < n_groups) load_chunk(g + WAVES, 0, abuf[0]);
#pragma unroll
                for (int s = 0; s < CH; ++s) {
                    const int ks = ci * CH + s;
                    if (ks < kKS) {
                        const float bf = sPF[h][4 * ks + q];
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[ci & 1][s].x, bf, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[ci & 1][s].y, bf, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[ci & 1][s].z, bf, acc[2], 0, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < kKS; ++s) {
                const float bf = bfrag[s];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[s].x, bf, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[s].y, bf, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[s].z, bf, acc[2], 0, 0, 0);
            }
            if (g + WAVES < n_groups) load_group(g + WAVES);
        }
        if (!TABLE) load_skin(g);
        // this lane: hand h, vertices v0 + 4q + i (register i of the three tiles)
        const int64_t hb = (int64_t)(b0 + (h < nh ? h : 0)) * V;
        // Table slots are non-zero first: the number of slots any vertex of the group needs bounds the loop.
        const int n_slots = TABLE ? (g < kSlotGroups ? __builtin_amdgcn_readfirstlane(sSlots[g]) : kSkinW) : NSLOT;
        if (h >= nh) continue;
        const bool full = v0 + 16 <= V;
        // T_i = sum_j w_j A_j over the slots (joints ascending), VP of the lane's four vertices side by side: the LDS reads
        // of a slot step (3 per vertex) are in flight together, where one vertex at a time waits out the LDS latency per slot.
        constexpr int VP = TABLE ? 2 : 1;
#pragma unroll
        for (int i0 = 0; i0 < 4; i0 += VP) {
        // (kept as pairs of neighbours within a row, the order the LDS reads deliver: one packed multiply-add per pair)
        f32x2 T2[VP][6];
#pragma unroll
        for (int u = 0; u < VP; ++u)
#pragma unroll
            for (int e = 0; e < 6; ++e) T2[u][e] = f32x2{0.f, 0.f};
#pragma unroll
        for (int e = 0; e < NSLOT; ++e) {
            if (e >= n_slots) break;
            f32x4 A0[VP], A1[VP], A2[VP];
#pragma unroll
            for (int u = 0; u < VP; ++u) {
                const int i = i0 + u;
                const int j = TABLE ? (int)(((e < 4 ? sj[i].x : sj[i].y) >> (8 * (e & 3))) & 15u) : e;
                A0[u] = *reinterpret_cast<const f32x4*>(sA(h, j));
                A1[u] = *reinterpret_cast<const f32x4*>(sA(h, j) + 4);
                A2[u] = *reinterpret_cast<const f32x4*>(sA(h, j) + 8);
            }
#pragma unroll
            for (int u = 0; u < VP; ++u) {
                const float w = sw[i0 + u][e / 4][e % 4];
                const f32x2 ww = {w, w};
                T2[u][0] = __builtin_elementwise_fma(ww, f32x2{A0[u][0], A0[u][1]}, T2[u][0]);
                T2[u][1] = __builtin_elementwise_fma(ww, f32x2{A0[u][2], A0[u][3]}, T2[u][1]);
                T2[u][2] = __builtin_elementwise_fma(ww, f32x2{A1[u][0], A1[u][1]}, T2[u][2]);
                T2[u][3] = __builtin_elementwise_fma(ww, f32x2{A1[u][2], A1[u][3]}, T2[u][3]);
                T2[u][4] = __builtin_elementwise_fma(ww, f32x2{A2[u][0], A2[u][1]}, T2[u][4]);
                T2[u][5] = __builtin_elementwise_fma(ww, f32x2{A2[u][2], A2[u][3]}, T2[u][5]);
            }
        }
        float T[VP][12];
#pragma unroll
        for (int u = 0; u < VP; ++u)
#pragma unroll
            for (int e = 0; e < 12; ++e) T[u][e] = T2[u][e / 2][e % 2];
#pragma unroll
        for (int u = 0; u < VP; ++u) {
            const int i = i0 + u;
            const int v = v0 + 4 * q + i;
            if (v >= V) break;
            // v_posed = pose_offsets + v_shaped  (lbs.py:101)
            const float px = acc[0][i] + vs[i].x, py = acc[1][i] + vs[i].y, pz = acc[2][i] + vs[i].z;
            if (v_posed_out) {   // kept for dvq_mano_lbs_backward
                if (full) {
                    float* st = &sOut[wave][1][h][(4 * q + i) * 3];
                    st[0] = px, st[1] = py, st[2] = pz;
                } else {
                    f32x3 o = {px, py, pz};
                    *reinterpret_cast<f32x3*>(v_posed_out + (hb + v) * 3) = o;
                }
            }
            f32x3 o;
            float* dst = &o.x;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float r = T[u][a * 4 + 0] * px;
                r = __builtin_fmaf(T[u][a * 4 + 1], py, r);
                r = __builtin_fmaf(T[u][a * 4 + 2], pz, r);
                r = r + T[u][a * 4 + 3];
                dst[a] = r + sTr[h][a];
            }
            if (full) {
                float* st = &sOut[wave][0][h][(4 * q + i) * 3];
                st[0] = o.x, st[1] = o.y, st[2] = o.z;
            } else {
                *reinterpret_cast<f32x3*>(verts_out + (hb + v) * 3) = o;
            }
        }
        }
        // A hand's 16 vertices of the group are 192 contiguous bytes of the output, and a lane holds 48 of them: written
        // from the registers, each store instruction would touch every line of every hand.  Through LDS the four lanes
        // of a hand write 64 contiguous bytes per instruction (LDS operations of one wave execute in order, and a hand's
        // row is written and read by its own four lanes only).  The last, partial group stores from the registers.
        if (full) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&sOut[wave][0][h][16 * k + 4 * q]);
                *reinterpret_cast<f32x4u*>(verts_out + (hb + v0) * 3 + 16 * k + 4 * q) = t;
            }
            if (v_posed_out) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(&sOut[wave][1][h][16 * k + 4 * q]);
                    *reinterpret_cast<f32x4u*>(v_posed_out + (hb + v0) * 3 + 16 * k + 4 * q) = t;
                }
            }
        }
    }
}

}  // namespace dvq

using namespace dvq;

extern "C" int dvq_mano_lbs_rows(const float* global_orient, int64_t ld_go, const float* hand_pose, int64_t ld_hp, const float* transl,
                                 int64_t ld_tr, int B, const DvqManoConsts* m, float* verts_out, float* joints_out,
                                 float* full_pose_out, float* v_posed_out, void* stream) {
    if (B < 0) return fail(DVQ_EINVAL, "mano: negative batch");
    if (B == 0) return DVQ_OK;
    if (!m || !hand_pose || !verts_out) return fail(DVQ_EINVAL, "mano: null pointer");
    if (m->n_verts < 1 || m->n_comps < 1 || m->n_comps > 45) return fail(DVQ_EINVAL, "mano: bad n_verts/n_comps (%d, %d)", m->n_verts, m->n_comps);
    if (ld_hp < m->n_comps || (global_orient && ld_go < 3) || (transl && ld_tr < 3) || ld_go > INT32_MAX || ld_hp > INT32_MAX || ld_tr > INT32_MAX)
        return fail(DVQ_EINVAL, "mano: a row stride is shorter than its row (or beyond 2^31)");
    if (!m->v_shaped || !m->joints || !m->posedirs || !m->lbs_weights || !m->components || !m->pose_mean || !m->parents)
        return fail(DVQ_EINVAL, "mano: null constant table");
    const bool table = m->skin_joints != nullptr;
    if (table && (!m->skin_weights || m->skin_width != kSkinW || ((uintptr_t)m->skin_joints & 7) || ((uintptr_t)m->skin_weights & 15)))
        return fail(DVQ_EINVAL, "mano: the skinning table needs skin_weights, skin_width == %d and aligned arrays (joints 8, weights 16 bytes)", kSkinW);
    if (((uintptr_t)m->lbs_weights & 15) != 0) return fail(DVQ_EINVAL, "mano: lbs_weights must be 16-byte aligned");
    if (((uintptr_t)m->posedirs_lanes & 3) != 0) return fail(DVQ_EINVAL, "mano: posedirs_lanes must be 4-byte aligned");
    const int grid = (B + kMH - 1) / kMH;
    const int lg = (int)ld_go, lh = (int)ld_hp, lt = (int)ld_tr;
    if (table && m->posedirs_lanes)
        hipLaunchKernelGGL((mano_lbs_kernel<true, 8, true>), dim3(grid), dim3(512), 0, as_stream(stream), global_orient, hand_pose, transl,
                           lg, lh, lt, B, m->n_verts, m->n_comps, m->v_shaped, m->joints, m->posedirs, m->lbs_weights, m->components,
                           m->pose_mean, m->parents, m->skin_joints, m->skin_weights, m->posedirs_lanes, verts_out, joints_out,
                           full_pose_out, v_posed_out);
    else if (table)
        hipLaunchKernelGGL((mano_lbs_kernel<true, 8, false>), dim3(grid), dim3(512), 0, as_stream(stream), global_orient, hand_pose, transl,
                           lg, lh, lt, B, m->n_verts, m->n_comps, m->v_shaped, m->joints, m->posedirs, m->lbs_weights, m->components,
                           m->pose_mean, m->parents, m->skin_joints, m->skin_weights, nullptr, verts_out, joints_out, full_pose_out,
                           v_posed_out);
    else
        hipLaunchKernelGGL((mano_lbs_kernel<false, 4, false>), dim3(grid), dim3(kManoThreads), 0, as_stream(stream), global_orient, hand_pose,
                           transl, lg, lh, lt, B, m->n_verts, m->n_comps, m->v_shaped, m->joints, m->posedirs, m->lbs_weights,
                           m->components, m->pose_mean, m->parents, nullptr, nullptr, nullptr, verts_out, joints_out, full_pose_out,
                           v_posed_out);
    return check_launch("mano_lbs");
}

extern "C" int dvq_mano_lbs(const float* global_orient, const float* hand_pose, const float* transl, int B,
                            const DvqManoConsts* m, float* verts_out, float* joints_out, float* full_pose_out, float* v_posed_out,
                            void* stream) {
    if (B > 0 && !global_orient) return fail(DVQ_EINVAL, "mano: null pointer");
    return dvq_mano_lbs_rows(global_orient, 3, hand_pose, m ? m->n_comps : 0, transl, 3, B, m, verts_out, joints_out, full_pose_out,
                             v_posed_out, stream);
}
