// Test-time adaptation of generated grasps (gfx950): the two device pieces of the refinement
// loop in DVQ-VAE/gen_HDMO_TTA.py:133-152 that the MANO forward does not cover.
//
//  * tta_contact_grad_kernel -- nearest hand vertex of every object point, interior test against
//    the vertex normals, penetration and contact terms (DVQ-VAE/utils/loss.py:72-127,
//    utils/utils_loss.py:7-45) and d loss / d vertices, one workgroup per grasp.  The hand (778
//    vertices + normals, 25 KB) lives in LDS and is read as wave-wide broadcasts; the 3000 x 778
//    distance table exists only in registers.  Because every object point pulls on exactly one
//    vertex, the gradient is  sum_i 2 (v - o_i)  over the points attached to v: it is gathered as
//    a count and a fixed-point sum of o_i per vertex with integer LDS atomics, which makes the
//    result independent of the order the points arrive in.
//  * mano_lbs_backward_kernel -- the vector-Jacobian product of the MANO layer
//    (DVQ-VAE-2/MANO-master/mano/lbs.py:33-265 differentiated by hand): skinning -> pose blend
//    shapes -> kinematic chain -> Rodrigues -> PCA, sixteen hands per workgroup.
#include "common.h"

namespace dvq {

constexpr int kTtaThreads = 256;
constexpr int kTtaWaves = kTtaThreads / 64;
constexpr float kFix = 4294967296.f;       // 2^32: metres -> fixed point
constexpr float kUnfix = 1.f / 4294967296.f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kTtaPts = 4;   // object points per thread and pass (two packed-fp32 pairs)
// 768 threads x 4 points cover a 3000-point cloud in one pass, and twelve waves per workgroup (two workgroups
// per CU) give each SIMD six waves to hide the LDS latency of the vertex reads behind other waves' arithmetic.
constexpr int kCgThreads = 768;
constexpr int kCgWaves = kCgThreads / 64;
constexpr int kCgGroup = 8;    // hand vertices per scan step (one compare-select of the running minimum per step and point)

// Vertices are searched in `order`: the n_contact vertices of the contact set first, then the others.
// The nearest contact-set vertex is then simply the running minimum at the end of the first group, and
// every (point, vertex) pair costs three packed subtract/multiply-add pairs and one compare-select, not two.
// Exact distance ties resolve to the lowest vertex id (both parts of `order` must list their vertices ascending).
__global__ __launch_bounds__(kCgThreads, 6) void tta_contact_grad_kernel(
    const float* __restrict__ obj, int64_t sb, int64_t sc, int64_t sp, int N, const float* __restrict__ verts, int V, int nC,
    const int* __restrict__ faces, const int* __restrict__ vf_ptr, const int* __restrict__ vf,
    const int* __restrict__ order, float thresh, float penetr_scale, float contact_scale, float w_penetr,
    float w_contact, float* __restrict__ grad_verts, float* __restrict__ losses, int* __restrict__ nn_idx) {
    extern __shared__ __attribute__((aligned(16))) char tta_smem[];
    // Slots: the contact set in slots [0, nC), the other vertices from V1 = nC rounded up to a whole group of kCgGroup, the whole
    // rounded up again to Vp; the spare slots hold a point at infinity, so the scan reads a group's slots from one
    // address without clamping and a spare slot never wins.
    const int V1 = (nC + kCgGroup - 1) / kCgGroup * kCgGroup, Vp = V1 + (V - nC + kCgGroup - 1) / kCgGroup * kCgGroup;
    f32x4* hv = reinterpret_cast<f32x4*>(tta_smem);                       // [slot] x, y, z of its vertex; w = vertex id as bits
    f32x4* hn = hv + Vp;                                                 // [slot] vertex normal (unnormalised)
    long long* sumP = reinterpret_cast<long long*>(hn + Vp);             // [slot][3] fixed-point sum of o_i, penetrating points
    long long* sumC = sumP + 3 * Vp;                                     // [slot][3] same, contact points
    int* cntP = reinterpret_cast<int*>(sumC + 3 * Vp);
    int* cntC = cntP + Vp;
    int* rank = cntC + Vp;                                               // vertex id -> slot
    // slot -> position in `order`, or -1 for a spare slot
    auto order_pos = [&](int slot) { return slot < nC ? slot : (slot >= V1 && slot - V1 < V - nC) ? nC + slot - V1 : -1; };
    __shared__ float sPen[kCgWaves], sCon[kCgWaves];
    __shared__ int sN[kCgWaves];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float* vb = verts + (int64_t)b * V * 3;
    const float inf = __builtin_inff();

    for (int k = tid; k < Vp; k += kCgThreads) {
        const int op = order_pos(k);
        f32x4 h;
        if (op >= 0) {
            const int v = order[op];
            h[0] = vb[3 * v]; h[1] = vb[3 * v + 1]; h[2] = vb[3 * v + 2];
            h[3] = __int_as_float(v);
            rank[v] = k;
        } else {
            h[0] = h[1] = h[2] = inf;
            h[3] = __int_as_float(0x7fffffff);
        }
        hv[k] = h;
        cntP[k] = 0; cntC[k] = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) { sumP[3 * k + c] = 0; sumC[3 * k + c] = 0; }
    }
    __syncthreads();
    // vertex normals: sum of (p1 - p0) x (p2 - p0) over the incident faces
    for (int k = tid; k < Vp; k += kCgThreads) {
        const int op = order_pos(k);
        if (op < 0) continue;
        const int v = order[op];
        float nx = 0.f, ny = 0.f, nz = 0.f;
        for (int q = vf_ptr[v]; q < vf_ptr[v + 1]; ++q) {
            const int f = vf[q];
            const f32x4 p0 = hv[rank[faces[3 * f]]], p1 = hv[rank[faces[3 * f + 1]]], p2 = hv[rank[faces[3 * f + 2]]];
            const float ux = p1[0] - p0[0], uy = p1[1] - p0[1], uz = p1[2] - p0[2];
            const float wx = p2[0] - p0[0], wy = p2[1] - p0[1], wz = p2[2] - p0[2];
            nx += uy * wz - uz * wy;
            ny += uz * wx - ux * wz;
            nz += ux * wy - uy * wx;
        }
        f32x4 n; n[0] = nx; n[1] = ny; n[2] = nz; n[3] = 0.f;
        hn[k] = n;
    }
    __syncthreads();

    float pen = 0.f, con = 0.f;
    int ncon = 0;
    const float* ob = obj + (int64_t)b * sb;
    for (int i0 = tid; i0 < N; i0 += kTtaPts * kCgThreads) {
        f32x2 px[2], py[2], pz[2], best[2];
        int bi[kTtaPts];
#pragma unroll
        for (int q = 0; q < kTtaPts; ++q) {
            int i = i0 + q * kCgThreads;
            if (i >= N) i = i0;
            px[q >> 1][q & 1] = ob[i * sp]; py[q >> 1][q & 1] = ob[sc + i * sp]; pz[q >> 1][q & 1] = ob[2 * sc + i * sp];
            best[q >> 1][q & 1] = inf;
            bi[q] = 0;
        }
        // kCgGroup vertices at a time, four and four: only the group's minimum is compared with the running best and
        // only the group's first slot is remembered; the winner inside the group is resolved after the scan.
        auto scan = [&](int k0, int k1) {
            for (int kg = k0; kg < k1; kg += kCgGroup) {
                float m[kTtaPts];
#pragma unroll
                for (int half = 0; half < kCgGroup / 4; ++half) {
                    f32x2 d[4][2];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const f32x4 h = hv[kg + 4 * half + u];
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const f32x2 dx = px[g] - h[0], dy = py[g] - h[1], dz = pz[g] - h[2];
                            d[u][g] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float mh = fminf(fminf(d[0][g][e], d[1][g][e]), fminf(d[2][g][e], d[3][g][e]));
                            m[2 * g + e] = half == 0 ? mh : fminf(m[2 * g + e], mh);
                        }
                }
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        if (m[2 * g + e] < best[g][e]) { best[g][e] = m[2 * g + e]; bi[2 * g + e] = kg; }
            }
        };
        // first slot of the group [kg, kg + kCgGroup) at the group's minimum distance
        auto resolve = [&](int q, int kg, int k1) {
            const float ox = px[q >> 1][q & 1], oy = py[q >> 1][q & 1], oz = pz[q >> 1][q & 1];
            float bd = inf;
            int bk = kg;
            for (int k = kg; k < kg + kCgGroup && k < k1; ++k) {
                const f32x4 h = hv[k];
                const float dx = ox - h[0], dy = oy - h[1], dz = oz - h[2];
                const float dd = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                if (dd < bd) { bd = dd; bk = k; }
            }
            return bk;
        };
        scan(0, V1);
        float cbest[kTtaPts];
        int ci[kTtaPts];
#pragma unroll
        for (int q = 0; q < kTtaPts; ++q) { cbest[q] = best[q >> 1][q & 1]; ci[q] = resolve(q, bi[q], V1); bi[q] = ci[q]; }
        // the other vertices are scanned from scratch; the nearer of the two winners is the nearest vertex, and at
        // exactly equal distances the one with the lower vertex id (inside either part the order is ascending, so
        // the first met is the lowest): ties go to the lowest index, as a plain scan over all vertices would give
#pragma unroll
        for (int q = 0; q < kTtaPts; ++q) { best[q >> 1][q & 1] = inf; bi[q] = V1; }
        scan(V1, Vp);
        float bdist[kTtaPts];
#pragma unroll
        for (int q = 0; q < kTtaPts; ++q) {
            bdist[q] = cbest[q];
            int win = ci[q];
            if (V > nC) {
                const int other = resolve(q, bi[q], Vp);
                const float d2 = best[q >> 1][q & 1];
                const bool lower = __float_as_int(hv[other][3]) < __float_as_int(hv[ci[q]][3]);
                if (d2 < cbest[q] || (d2 == cbest[q] && lower)) { win = other; bdist[q] = d2; }
            }
            bi[q] = win;
        }
#pragma unroll
        for (int q = 0; q < kTtaPts; ++q) {
            const int i = i0 + q * kCgThreads;
            if (i >= N) continue;
            const float ox = px[q >> 1][q & 1], oy = py[q >> 1][q & 1], oz = pz[q >> 1][q & 1], bd = bdist[q];
            const f32x4 h = hv[bi[q]], n = hn[bi[q]];
            const float dot = (h[0] - ox) * n[0] + (h[1] - oy) * n[1] + (h[2] - oz) * n[2];
            const long long fx = __float2ll_rn(ox * kFix), fy = __float2ll_rn(oy * kFix), fz = __float2ll_rn(oz * kFix);
            if (dot > 0.f) {
                pen += bd;
                atomicAdd(&cntP[bi[q]], 1);
                atomicAdd(reinterpret_cast<unsigned long long*>(&sumP[3 * bi[q] + 0]), (unsigned long long)fx);
                atomicAdd(reinterpret_cast<unsigned long long*>(&sumP[3 * bi[q] + 1]), (unsigned long long)fy);
                atomicAdd(reinterpret_cast<unsigned long long*>(&sumP[3 * bi[q] + 2]), (unsigned long long)fz);
            }
            if (bd < thresh) {
                ++ncon;
                con += cbest[q];
                atomicAdd(&cntC[ci[q]], 1);
                atomicAdd(reinterpret_cast<unsigned long long*>(&sumC[3 * ci[q] + 0]), (unsigned long long)fx);
                atomicAdd(reinterpret_cast<unsigned long long*>(&sumC[3 * ci[q] + 1]), (unsigned long long)fy);
                atomicAdd(reinterpret_cast<unsigned long long*>(&sumC[3 * ci[q] + 2]), (unsigned long long)fz);
            }
            if (nn_idx) nn_idx[(int64_t)b * N + i] = __float_as_int(h[3]);
        }
    }
    pen = wave_sum(pen); con = wave_sum(con); ncon = wave_sum_i(ncon);
    if (lane == 0) { sPen[wave] = pen; sCon[wave] = con; sN[wave] = ncon; }
    __syncthreads();
    float penT = 0.f, conT = 0.f;
    int nT = 0;
#pragma unroll
    for (int w = 0; w < kCgWaves; ++w) { penT += sPen[w]; conT += sCon[w]; nT += sN[w]; }
    const float cP = 2.f * w_penetr * penetr_scale;
    const float cC = nT > 0 ? 2.f * w_contact * contact_scale / (float)nT : 0.f;
    if (tid == 0) {
        losses[2 * b] = penetr_scale * penT;
        losses[2 * b + 1] = nT > 0 ? contact_scale * conT / (float)nT : 0.f;
    }
    float* gb = grad_verts + (int64_t)b * V * 3;
    for (int k = tid; k < Vp; k += kCgThreads) {
        if (order_pos(k) < 0) continue;
        const f32x4 h = hv[k];
        const int v = __float_as_int(h[3]);
        const int nP = cntP[k], nC2 = cntC[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const long long fv = __float2ll_rn(h[c] * kFix);
            const float dP = (float)(fv * nP - sumP[3 * k + c]) * kUnfix;     // sum of (v - o_i), penetrating points
            const float dC = (float)(fv * nC2 - sumC[3 * k + c]) * kUnfix;    // same, contact points
            gb[3 * v + c] = cP * dP + cC * dC;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Vector-Jacobian product of the MANO layer, sixteen hands per workgroup, laid out like the forward kernel
// (csrc/mano.hip): a lane is (hand = lane & 15, q = lane >> 4) and owns vertices v0 + 4q + i of a group of 16.
//   1  full pose and Rodrigues, one thread per (hand, joint); 2  kinematic chain, one thread per (hand, matrix row);
//   3  per vertex group: T_v from the vertex's few non-zero skinning weights, dL/dv_posed = T_v^T g_v in registers,
//      and at once the B operand of dL/dpose_feature = posedirs . dL/dv_posed on the 16x16x4 fp32 MFMA (nine tiles of 16
//      features x 16 hands; k = the four vertices v0 + 4q + i of the lanes' q, one coordinate per instruction) --
//      posedirs is read once per workgroup, 12 B per lane, a group ahead; the four waves' partial tiles meet in LDS and
//      are summed in wave order;
//   4  dL/dA_j = sum_v w_vj g_v [v_posed_v; 1]^T, one thread per (hand, joint), vertices in ascending order;
//   5  chain backward: the part that climbs the tree works on rows (row a of dG_parent needs row a of dG_child only),
//      one thread per (hand, row); then per (hand, joint) dL/dR_j, Rodrigues backward and the PCA projection.
// Every sum has a fixed order: the result does not depend on launch geometry or timing.
constexpr int kBwJ = 16;
constexpr int kBwPF = 135;
constexpr int kBwHB = 16;                 // hands per workgroup
constexpr int kBwPT = 9;                  // tiles of 16 pose features (135 -> 144)
constexpr int kBwSkinW = 8;               // slots per vertex in the skinning table (as in csrc/mano.hip)

struct bw3 { float x, y, z; };

struct BwShared {
    float R[kBwHB][kBwJ][9];
    float Ap[kBwHB][kBwJ * 12 + 4];      // skinning transforms; padded against bank conflicts of the per-hand gather
    float T[kBwHB][kBwJ][12];            // world transforms
    float Rv[kBwHB][kBwJ][3];            // full pose (axis-angle per joint)
    float dA[kBwHB][kBwJ][12];
    float dGR[kBwHB][kBwJ][9];
    float dGt[kBwHB][kBwJ][3];
    float dPF[kBwHB][kBwPT * 16];
    float dPose[kBwHB][48];
    float part[kTtaWaves][kBwPT][4][64]; // per-wave partial tiles of dL/dpose_feature
    float Tr[kTtaWaves][kBwHB][3];       // per-wave partial sums of g (= dL/dtransl)
    float HP[kBwHB][48];
    float Comps[45 * 45];
    float J[kBwJ * 3];
    int Par[kBwJ];
};

template <bool TABLE>
__global__ __launch_bounds__(kTtaThreads) void mano_lbs_backward_kernel(
    const float* __restrict__ global_orient, const float* __restrict__ hand_pose, int B, int V, int ncomp,
    const float* __restrict__ joints, const float* __restrict__ posedirs, const float* __restrict__ weights,
    const float* __restrict__ comps, const float* __restrict__ pose_mean, const int* __restrict__ parents,
    const unsigned char* __restrict__ skin_joints, const float* __restrict__ skin_weights,
    const float* __restrict__ v_posed, const float* __restrict__ grad_verts, float* __restrict__ g_orient,
    float* __restrict__ g_pose, float* __restrict__ g_transl) {
    extern __shared__ __attribute__((aligned(16))) char bw_smem[];
    BwShared& sh = *reinterpret_cast<BwShared*>(bw_smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * kBwHB;
    const int nh = (B - b0) < kBwHB ? (B - b0) : kBwHB;
    const int64_t V3 = 3 * (int64_t)V;

    for (int i = tid; i < nh * ncomp; i += kTtaThreads) sh.HP[i / ncomp][i % ncomp] = hand_pose[(int64_t)b0 * ncomp + i];
    for (int i = tid; i < ncomp * 45; i += kTtaThreads) sh.Comps[i] = comps[i];
    if (tid < kBwJ * 3) sh.J[tid] = joints[tid];
    if (tid < kBwJ) sh.Par[tid] = parents[tid];
    __syncthreads();

    // ---- 1: full pose, rotations   (lbs.py:167-198)
    {
        const int h = tid / kBwJ, j = tid % kBwJ;
        if (h < nh) {
            const int b = b0 + h;
            float rv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v;
                if (j == 0) {
                    v = global_orient[b * 3 + c];
                } else {
                    const int col = 3 * (j - 1) + c;
                    v = 0.f;
                    for (int i = 0; i < ncomp; ++i) v = __builtin_fmaf(sh.HP[h][i], sh.Comps[i * 45 + col], v);
                }
                v += pose_mean[3 * j + c];
                rv[c] = v;
                sh.Rv[h][j][c] = v;
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
                    sh.R[h][j][a * 3 + c] = ((a == c ? 1.f : 0.f) + sn * K[a * 3 + c]) + omc * kk;
                }
        } else {   // hands past the end of the batch: finite transforms for the gathers of phase 3
#pragma unroll
            for (int e = 0; e < 12; ++e) sh.Ap[h][j * 12 + e] = 0.f;
        }
    }
    __syncthreads();
    // ---- 2: kinematic chain, one thread per (hand, row of the 3x4 transform)   (lbs.py:214-265)
    if (tid < 3 * nh) {
        const int h = tid / 3, a = tid % 3;
        for (int i = 0; i < kBwJ; ++i) {
            const int par = sh.Par[i];
            float t[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) t[c] = (i == 0) ? sh.J[c] : sh.J[i * 3 + c] - sh.J[par * 3 + c];
            float T[4];
            if (i == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) T[c] = sh.R[h][0][a * 3 + c];
                T[3] = t[a];
            } else {
                const float* P = &sh.T[h][par][a * 4];   // written by this very thread in an earlier step
                const float p0 = P[0], p1 = P[1], p2 = P[2], p3 = P[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v = p0 * sh.R[h][i][0 * 3 + c];
                    v = __builtin_fmaf(p1, sh.R[h][i][1 * 3 + c], v);
                    v = __builtin_fmaf(p2, sh.R[h][i][2 * 3 + c], v);
                    T[c] = v;
                }
                float v = p0 * t[0];
                v = __builtin_fmaf(p1, t[1], v);
                v = __builtin_fmaf(p2, t[2], v);
                T[3] = v + p3;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) sh.T[h][i][a * 4 + e] = T[e];
            float v = T[0] * sh.J[i * 3 + 0];
            v = __builtin_fmaf(T[1], sh.J[i * 3 + 1], v);
            v = __builtin_fmaf(T[2], sh.J[i * 3 + 2], v);
            float* A = &sh.Ap[h][i * 12 + a * 4];
            A[0] = T[0]; A[1] = T[1]; A[2] = T[2];
            A[3] = T[3] - v;
        }
    }
    __syncthreads();

    // ---- 3: skinning backward and the pose blend-shape product, per group of 16 vertices
    {
        const int h = lane & 15, q = lane >> 4;
        const int hc = h < nh ? h : 0;                         // surplus lanes repeat hand 0 with zero gradient
        const int n_groups = (V + 15) / 16;
        constexpr int NSLOT = TABLE ? kBwSkinW : kBwJ;
        f32x4 acc[kBwPT];
#pragma unroll
        for (int t = 0; t < kBwPT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float tr0 = 0.f, tr1 = 0.f, tr2 = 0.f;
        // A operand of group g: pose feature 16t + (lane & 15), vertex 16g + 4q + i: the three coordinates as one 12-byte
        // load.  Features 135..143 do not exist: their rows are clamped to 134 and never read back.
        bw3 afrag[kBwPT][4];
        auto load_group_slot = [&](int g, int i) {
            int v = 16 * g + 4 * q + i;
            if (v >= V) v = V - 1;
#pragma unroll
            for (int t = 0; t < kBwPT; ++t) {
                const int pf = (16 * t + h) < kBwPF ? (16 * t + h) : kBwPF - 1;
                afrag[t][i] = *reinterpret_cast<const bw3*>(posedirs + pf * V3 + 3 * (int64_t)v);
            }
        };
        auto load_group = [&](int g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) load_group_slot(g, i);
        };
        // gradients and skinning slots of this lane's four vertices of a group
        struct Small {
            bw3 gv[4];
            f32x4 sw[4][NSLOT / 4];
            uint2 sj[4];
        };
        auto load_small = [&](int g, Small& m) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int v = 16 * g + 4 * q + i;
                if (v >= V) v = V - 1;
                m.gv[i] = *reinterpret_cast<const bw3*>(grad_verts + ((int64_t)(b0 + hc) * V + v) * 3);
#pragma unroll
                for (int c = 0; c < NSLOT / 4; ++c)
                    m.sw[i][c] = *reinterpret_cast<const f32x4*>((TABLE ? skin_weights : weights) + (int64_t)v * NSLOT + 4 * c);
                if (TABLE) m.sj[i] = *reinterpret_cast<const uint2*>(skin_joints + (int64_t)v * kBwSkinW);
            }
        };
        // One wave per SIMD: nothing else hides a load, so everything a group needs is requested a group ahead -- the
        // small per-vertex data before the previous group's matrix instructions, the A fragments of vertex slot i as
        // soon as the previous group's instructions on slot i are through.
        Small cur, nxt;
        if (wave < n_groups) {
            load_group(wave);
            if (TABLE) load_small(wave, cur);
        }
        for (int g = wave; g < n_groups; g += kTtaWaves) {
            const int v0 = 16 * g;
            const bool more = g + kTtaWaves < n_groups;
            if (!TABLE) load_small(g, cur);   // (the dense matrix: 64 weights per lane, no room to hold two groups)
            int n_slots = NSLOT;
            if (TABLE) {   // table slots are non-zero first: what any lane of the wave still needs bounds the loop
                n_slots = 0;
#pragma unroll
                for (int e = 0; e < kBwSkinW; ++e) {
                    const bool any = cur.sw[0][e / 4][e % 4] != 0.f || cur.sw[1][e / 4][e % 4] != 0.f || cur.sw[2][e / 4][e % 4] != 0.f ||
                                     cur.sw[3][e / 4][e % 4] != 0.f;
                    if (__ballot(any) != 0) n_slots = e + 1;
                }
            }
            float dvp[4][3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool live = (v0 + 4 * q + i) < V && h < nh;
                const float gx = live ? cur.gv[i].x : 0.f, gy = live ? cur.gv[i].y : 0.f, gz = live ? cur.gv[i].z : 0.f;
                float TR[9];
#pragma unroll
                for (int e = 0; e < 9; ++e) TR[e] = 0.f;
#pragma unroll
                for (int e = 0; e < NSLOT; ++e) {
                    if (e >= n_slots) break;
                    const int j = TABLE ? (int)(((e < 4 ? cur.sj[i].x : cur.sj[i].y) >> (8 * (e & 3))) & 15u) : e;
                    const float w = cur.sw[i][e / 4][e % 4];
                    const float* A = &sh.Ap[h][j * 12];
                    const f32x4 A0 = *reinterpret_cast<const f32x4*>(A);
                    const f32x4 A1 = *reinterpret_cast<const f32x4*>(A + 4);
                    const f32x4 A2 = *reinterpret_cast<const f32x4*>(A + 8);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        TR[c] = __builtin_fmaf(w, A0[c], TR[c]);
                        TR[3 + c] = __builtin_fmaf(w, A1[c], TR[3 + c]);
                        TR[6 + c] = __builtin_fmaf(w, A2[c], TR[6 + c]);
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) dvp[i][c] = TR[0 * 3 + c] * gx + TR[1 * 3 + c] * gy + TR[2 * 3 + c] * gz;
                tr0 += gx; tr1 += gy; tr2 += gz;
            }
            if (TABLE && more) load_small(g + kTtaWaves, nxt);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int t = 0; t < kBwPT; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[t][i].x, dvp[i][0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[t][i].y, dvp[i][1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[t][i].z, dvp[i][2], acc[t], 0, 0, 0);
                }
                if (more) load_group_slot(g + kTtaWaves, i);
            }
            if (TABLE) cur = nxt;
        }
        // accumulator: lane (hand h, row group q), register i <-> pose feature 16t + 4q + i
#pragma unroll
        for (int t = 0; t < kBwPT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) sh.part[wave][t][i][lane] = acc[t][i];
        tr0 += __shfl_xor(tr0, 16, 64); tr0 += __shfl_xor(tr0, 32, 64);
        tr1 += __shfl_xor(tr1, 16, 64); tr1 += __shfl_xor(tr1, 32, 64);
        tr2 += __shfl_xor(tr2, 16, 64); tr2 += __shfl_xor(tr2, 32, 64);
        if (q == 0) { sh.Tr[wave][h][0] = tr0; sh.Tr[wave][h][1] = tr1; sh.Tr[wave][h][2] = tr2; }
    }
    __syncthreads();
    for (int o = tid; o < kBwPT * 4 * 64; o += kTtaThreads) {
        const int l = o & 63, i = (o >> 6) & 3, t = o >> 8;
        float x = sh.part[0][t][i][l];
#pragma unroll
        for (int w = 1; w < kTtaWaves; ++w) x += sh.part[w][t][i][l];
        sh.dPF[l & 15][16 * t + 4 * (l >> 4) + i] = x;
    }

    // ---- 4: dL/dA_j[a][k] = sum_v w_vj g_v[a] [v_posed_v; 1][k]: per hand a [16 joints x V] x [V x 12] product on the
    // 16x16x4 MFMA, a wave per hand.  A = weights^T (4 vertices x 16 joints: 256 contiguous bytes of the dense matrix),
    // B = the outer product g_v (x) [v_posed_v; 1] formed on the lane: column c <-> (a, k) = (c / 4, c % 4), 12 of 16 used.
    {
        const int c = lane & 15, kq = lane >> 4;
        const int ca = c >> 2, ck = c & 3;
        for (int hh = wave; hh < nh; hh += kTtaWaves) {
            const float* g = grad_verts + (int64_t)(b0 + hh) * V3;
            const float* vp = v_posed + (int64_t)(b0 + hh) * V3;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            constexpr int U = 8;                                     // k-steps whose operands are requested together
            float w[U], ga[U], pk[U], wn[U], gan[U], pkn[U];
            auto fetch = [&](int v0, float (&ww)[U], float (&gg)[U], float (&pp)[U]) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    int v = v0 + 4 * u + kq;
                    if (v >= V) v = V - 1;
                    ww[u] = weights[v * kBwJ + c];                   // A: joint c of vertex v
                    gg[u] = g[3 * v + (ca < 3 ? ca : 0)];
                    pp[u] = ck < 3 ? vp[3 * v + ck] : 1.f;
                }
            };
            fetch(0, w, ga, pk);
            for (int v0 = 0; v0 < V; v0 += 4 * U) {
                if (v0 + 4 * U < V) fetch(v0 + 4 * U, wn, gan, pkn);   // the next batch, under this batch's products
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool in = (v0 + 4 * u + kq) < V;
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(in ? w[u] : 0.f, (in && c < 12) ? ga[u] * pk[u] : 0.f, acc, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) { w[u] = wn[u]; ga[u] = gan[u]; pk[u] = pkn[u]; }
            }
            // lane (column c, row group kq), register i <-> joint 4 kq + i
            if (c < 12) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sh.dA[hh][4 * kq + i][c] = acc[i];
            }
        }
    }
    __syncthreads();

    // ---- 5a: chain backward, the climb, one thread per (hand, row)   A = [G_R | G_t - G_R J]
    if (tid < 3 * nh) {
        const int h = tid / 3, a = tid % 3;
        for (int j = 0; j < kBwJ; ++j) {
            const float dt = sh.dA[h][j][a * 4 + 3];
            sh.dGR[h][j][a * 3 + 0] = sh.dA[h][j][a * 4 + 0] - dt * sh.J[j * 3 + 0];
            sh.dGR[h][j][a * 3 + 1] = sh.dA[h][j][a * 4 + 1] - dt * sh.J[j * 3 + 1];
            sh.dGR[h][j][a * 3 + 2] = sh.dA[h][j][a * 4 + 2] - dt * sh.J[j * 3 + 2];
            sh.dGt[h][j][a] = dt;
        }
        // G_R,i = G_R,par R_i ; G_t,i = G_R,par rel_i + G_t,par   (children have larger indices than their parents)
        for (int i = kBwJ - 1; i >= 1; --i) {
            const int par = sh.Par[i];
            const float r0 = sh.dGR[h][i][a * 3 + 0], r1 = sh.dGR[h][i][a * 3 + 1], r2 = sh.dGR[h][i][a * 3 + 2];
            const float dgt = sh.dGt[h][i][a];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float rel = sh.J[i * 3 + c] - sh.J[par * 3 + c];
                sh.dGR[h][par][a * 3 + c] += r0 * sh.R[h][i][c * 3 + 0] + r1 * sh.R[h][i][c * 3 + 1] + r2 * sh.R[h][i][c * 3 + 2] + dgt * rel;
            }
            sh.dGt[h][par][a] += dgt;
        }
    }
    __syncthreads();

    // ---- 5b: dL/dR_j and Rodrigues backward, one thread per (hand, joint)
    {
        const int h = tid / kBwJ, j = tid % kBwJ;
        if (h < nh) {
            float dR[9];
            if (j == 0) {
#pragma unroll
                for (int e = 0; e < 9; ++e) dR[e] = sh.dGR[h][0][e];
            } else {
                const float* P = sh.T[h][sh.Par[j]];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        dR[a * 3 + c] = (P[0 * 4 + a] * sh.dGR[h][j][0 * 3 + c] + P[1 * 4 + a] * sh.dGR[h][j][1 * 3 + c] +
                                         P[2 * 4 + a] * sh.dGR[h][j][2 * 3 + c]) + sh.dPF[h][(j - 1) * 9 + a * 3 + c];
            }
            const float vx = sh.Rv[h][j][0], vy = sh.Rv[h][j][1], vz = sh.Rv[h][j][2];
            const float ex = vx + 1e-8f, ey = vy + 1e-8f, ez = vz + 1e-8f;
            const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
            const float inv = 1.f / ang;
            const float nx = vx * inv, ny = vy * inv, nz = vz * inv;
            const float cs = cosf(ang), sn = sinf(ang), omc = 1.f - cs;
            const float K[9] = {0.f, -nz, ny, nz, 0.f, -nx, -ny, nx, 0.f};
            float KK[9];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    KK[a * 3 + c] = K[a * 3 + 0] * K[0 * 3 + c] + K[a * 3 + 1] * K[1 * 3 + c] + K[a * 3 + 2] * K[2 * 3 + c];
            float dS = 0.f, dO = 0.f;
#pragma unroll
            for (int e = 0; e < 9; ++e) { dS += dR[e] * K[e]; dO += dR[e] * KK[e]; }
            float dK[9];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) {
                    float t = 0.f;
#pragma unroll
                    for (int m = 0; m < 3; ++m) t += dR[p * 3 + m] * K[qq * 3 + m] + K[m * 3 + p] * dR[m * 3 + qq];
                    dK[p * 3 + qq] = sn * dR[p * 3 + qq] + omc * t;
                }
            const float dnx = dK[2 * 3 + 1] - dK[1 * 3 + 2];
            const float dny = dK[0 * 3 + 2] - dK[2 * 3 + 0];
            const float dnz = dK[1 * 3 + 0] - dK[0 * 3 + 1];
            float da = cs * dS + sn * dO;
            da -= (dnx * vx + dny * vy + dnz * vz) * inv * inv;
            sh.dPose[h][3 * j + 0] = dnx * inv + da * ex * inv;
            sh.dPose[h][3 * j + 1] = dny * inv + da * ey * inv;
            sh.dPose[h][3 * j + 2] = dnz * inv + da * ez * inv;
        }
    }
    __syncthreads();

    if (tid < nh * 3) {
        const int h = tid / 3, c = tid % 3, b = b0 + h;
        if (g_orient) g_orient[b * 3 + c] = sh.dPose[h][c];
        if (g_transl) g_transl[b * 3 + c] = ((sh.Tr[0][h][c] + sh.Tr[1][h][c]) + sh.Tr[2][h][c]) + sh.Tr[3][h][c];
    }
    if (g_pose) {   // pose = coefficients x components (model.py:330-331)
        for (int o = tid; o < nh * ncomp; o += kTtaThreads) {
            const int h = o / ncomp, c = o % ncomp;
            float acc = 0.f;
            for (int m = 0; m < 45; ++m) acc = __builtin_fmaf(sh.Comps[c * 45 + m], sh.dPose[h][3 + m], acc);
            g_pose[(int64_t)(b0 + h) * ncomp + c] = acc;
        }
    }
}

}  // namespace dvq

using namespace dvq;

extern "C" int dvq_mano_lbs_backward(const float* global_orient, const float* hand_pose, int B, const DvqManoConsts* m,
                                     const float* v_posed, const float* grad_verts, float* grad_global_orient,
                                     float* grad_hand_pose, float* grad_transl, void* stream) {
    if (B < 0) return fail(DVQ_EINVAL, "mano_backward: negative batch");
    if (B == 0) return DVQ_OK;
    if (!m || !global_orient || !hand_pose || !v_posed || !grad_verts) return fail(DVQ_EINVAL, "mano_backward: null pointer");
    if (m->n_verts < 1 || m->n_verts > 2500 || m->n_comps < 1 || m->n_comps > 45)
        return fail(DVQ_EINVAL, "mano_backward: bad n_verts/n_comps (%d, %d)", m->n_verts, m->n_comps);
    if (!m->joints || !m->posedirs || !m->lbs_weights || !m->components || !m->pose_mean || !m->parents)
        return fail(DVQ_EINVAL, "mano_backward: null constant table");
    if (((uintptr_t)m->lbs_weights & 15) != 0) return fail(DVQ_EINVAL, "mano_backward: lbs_weights must be 16-byte aligned");
    const bool table = m->skin_joints != nullptr;
    if (table && (!m->skin_weights || m->skin_width != kBwSkinW || ((uintptr_t)m->skin_joints & 7) || ((uintptr_t)m->skin_weights & 15)))
        return fail(DVQ_EINVAL, "mano_backward: the skinning table needs skin_weights, skin_width == %d and aligned arrays", kBwSkinW);
    const size_t lds = sizeof(BwShared);
    const dim3 grid((B + kBwHB - 1) / kBwHB), blk(kTtaThreads);
    if (table) {
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(mano_lbs_backward_kernel<true>), lds, "mano_lbs_backward")) return rc;
        hipLaunchKernelGGL(mano_lbs_backward_kernel<true>, grid, blk, lds, as_stream(stream), global_orient, hand_pose, B, m->n_verts,
                           m->n_comps, m->joints, m->posedirs, m->lbs_weights, m->components, m->pose_mean, m->parents, m->skin_joints,
                           m->skin_weights, v_posed, grad_verts, grad_global_orient, grad_hand_pose, grad_transl);
    } else {
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(mano_lbs_backward_kernel<false>), lds, "mano_lbs_backward")) return rc;
        hipLaunchKernelGGL(mano_lbs_backward_kernel<false>, grid, blk, lds, as_stream(stream), global_orient, hand_pose, B, m->n_verts,
                           m->n_comps, m->joints, m->posedirs, m->lbs_weights, m->components, m->pose_mean, m->parents, nullptr,
                           nullptr, v_posed, grad_verts, grad_global_orient, grad_hand_pose, grad_transl);
    }
    return check_launch("mano_lbs_backward");
}

extern "C" int dvq_tta_contact_grad(const float* obj, int64_t obj_batch_stride, int64_t obj_chan_stride, int64_t obj_point_stride,
                                    int n_points, const float* verts, int B, const DvqHandMesh* mesh, float contact_dist_sq,
                                    float penetr_scale, float contact_scale, float w_penetr, float w_contact, float* grad_verts,
                                    float* losses, int* nn_idx, void* stream) {
    if (B < 0) return fail(DVQ_EINVAL, "tta_contact_grad: negative batch");
    if (B == 0) return DVQ_OK;
    if (!obj || !verts || !mesh || !grad_verts || !losses) return fail(DVQ_EINVAL, "tta_contact_grad: null pointer");
    if (n_points < 1 || n_points > (1 << 20)) return fail(DVQ_EINVAL, "tta_contact_grad: n_points out of range (%d)", n_points);
    if (mesh->n_verts < 1 || mesh->n_verts > 1500) return fail(DVQ_EINVAL, "tta_contact_grad: 1..1500 hand vertices (got %d)", mesh->n_verts);
    if (mesh->n_faces < 0 || !mesh->faces || !mesh->vert_face_ptr || !mesh->vert_face || !mesh->search_order)
        return fail(DVQ_EINVAL, "tta_contact_grad: incomplete hand mesh");
    if (mesh->n_contact < 1 || mesh->n_contact > mesh->n_verts) return fail(DVQ_EINVAL, "tta_contact_grad: empty contact set");
    // slots: contact set and the other vertices, each rounded up to whole scan groups (see the kernel)
    const int slots = (mesh->n_contact + kCgGroup - 1) / kCgGroup * kCgGroup + (mesh->n_verts - mesh->n_contact + kCgGroup - 1) / kCgGroup * kCgGroup;
    const size_t lds = (size_t)slots * (16 + 16 + 24 + 24 + 4 + 4) + (size_t)mesh->n_verts * 4;
    if (lds > 150 * 1024) return fail(DVQ_EINVAL, "tta_contact_grad: hand mesh too large for LDS");
    // the kernel also holds a few hundred bytes of static LDS, so ask for what is needed, not for the whole array
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(tta_contact_grad_kernel), lds, "tta_contact_grad")) return rc;
    hipLaunchKernelGGL(tta_contact_grad_kernel, dim3(B), dim3(kCgThreads), lds, as_stream(stream), obj, obj_batch_stride,
                       obj_chan_stride, obj_point_stride, n_points, verts, mesh->n_verts, mesh->n_contact, mesh->faces, mesh->vert_face_ptr,
                       mesh->vert_face, mesh->search_order, contact_dist_sq, penetr_scale, contact_scale, w_penetr, w_contact,
                       grad_verts, losses, nn_idx);
    return check_launch("tta_contact_grad");
}
