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
//    shapes -> kinematic chain -> Rodrigues -> PCA, four hands per workgroup.
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
    f32x4* hv = reinterpret_cast<f32x4*>(tta_smem);                       // [k] x, y, z of vertex order[k]; w = order[k] as bits
    f32x4* hn = hv + V;                                                  // [k] vertex normal (unnormalised)
    long long* sumP = reinterpret_cast<long long*>(hn + V);              // [k][3] fixed-point sum of o_i, penetrating points
    long long* sumC = sumP + 3 * V;                                      // [k][3] same, contact points
    int* cntP = reinterpret_cast<int*>(sumC + 3 * V);
    int* cntC = cntP + V;
    int* rank = cntC + V;                                                // vertex id -> k
    __shared__ float sPen[kCgWaves], sCon[kCgWaves];
    __shared__ int sN[kCgWaves];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float* vb = verts + (int64_t)b * V * 3;
    const float inf = __builtin_inff();

    for (int k = tid; k < V; k += kCgThreads) {
        const int v = order[k];
        f32x4 h;
        h[0] = vb[3 * v]; h[1] = vb[3 * v + 1]; h[2] = vb[3 * v + 2];
        h[3] = __int_as_float(v);
        hv[k] = h;
        rank[v] = k;
        cntP[k] = 0; cntC[k] = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) { sumP[3 * k + c] = 0; sumC[3 * k + c] = 0; }
    }
    __syncthreads();
    // vertex normals: sum of (p1 - p0) x (p2 - p0) over the incident faces
    for (int k = tid; k < V; k += kCgThreads) {
        const int v = order[k];
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
        // Four vertices at a time: only the group's minimum is compared with the running best and only
        // the group's first vertex is remembered; the winner inside the group is resolved after the scan.
        auto scan = [&](int k0, int k1) {
            for (int kg = k0; kg < k1; kg += 4) {
                f32x2 d[4][2];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kg + u < k1 ? kg + u : k1 - 1;
                    const f32x4 h = hv[k];
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
                        const float m = fminf(fminf(d[0][g][e], d[1][g][e]), fminf(d[2][g][e], d[3][g][e]));
                        if (m < best[g][e]) { best[g][e] = m; bi[2 * g + e] = kg; }
                    }
            }
        };
        // first vertex of the group [kg, min(kg+4, k1)) at the group's minimum distance
        auto resolve = [&](int q, int kg, int k1) {
            const float ox = px[q >> 1][q & 1], oy = py[q >> 1][q & 1], oz = pz[q >> 1][q & 1];
            float bd = inf;
            int bk = kg;
            for (int k = kg; k < kg + 4 && k < k1; ++k) {
                const f32x4 h = hv[k];
                const float dx = ox - h[0], dy = oy - h[1], dz = oz - h[2];
                const float dd = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                if (dd < bd) { bd = dd; bk = k; }
            }
            return bk;
        };
        scan(0, nC);
        float cbest[kTtaPts];
        int ci[kTtaPts];
#pragma unroll
        for (int q = 0; q < kTtaPts; ++q) { cbest[q] = best[q >> 1][q & 1]; ci[q] = resolve(q, bi[q], nC); bi[q] = ci[q]; }
        // the other vertices are scanned from scratch; the nearer of the two winners is the nearest vertex, and at
        // exactly equal distances the one with the lower vertex id (inside either part the order is ascending, so
        // the first met is the lowest): ties go to the lowest index, as a plain scan over all vertices would give
#pragma unroll
        for (int q = 0; q < kTtaPts; ++q) { best[q >> 1][q & 1] = inf; bi[q] = nC; }
        scan(nC, V);
        float bdist[kTtaPts];
#pragma unroll
        for (int q = 0; q < kTtaPts; ++q) {
            bdist[q] = cbest[q];
            int win = ci[q];
            if (V > nC) {
                const int other = resolve(q, bi[q], V);
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
    for (int k = tid; k < V; k += kCgThreads) {
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
constexpr int kBwJ = 16;
constexpr int kBwPF = 135;
constexpr int kBwHB = 4;        // hands per workgroup: every posedirs element fetched feeds kBwHB multiply-adds
constexpr int kBwSlices = kTtaThreads / kBwJ;   // vertex slices of the dL/dA reduction

struct BwShared {
    float Rv[kBwHB][kBwJ][3];      // full pose (axis-angle per joint)
    float R[kBwHB][kBwJ][9];
    float T[kBwHB][kBwJ][12];      // world transforms
    float A[kBwHB][kBwJ][12];      // skinning transforms
    float dA[kBwHB][kBwJ][12];
    float dPF[kBwHB][kBwPF + 1];
    float dGR[kBwHB][kBwJ][9];
    float dGt[kBwHB][kBwJ][3];
    float dR[kBwHB][kBwJ][9];
    float dPose[kBwHB][48];
    float Tr[kBwHB][kTtaWaves][3];
};

__global__ __launch_bounds__(kTtaThreads, 2) void mano_lbs_backward_kernel(
    const float* __restrict__ global_orient, const float* __restrict__ hand_pose, int B, int V, int ncomp,
    const float* __restrict__ joints, const float* __restrict__ posedirs, const float* __restrict__ weights,
    const float* __restrict__ comps, const float* __restrict__ pose_mean, const int* __restrict__ parents,
    const float* __restrict__ v_posed, const float* __restrict__ grad_verts, float* __restrict__ g_orient,
    float* __restrict__ g_pose, float* __restrict__ g_transl) {
    extern __shared__ __attribute__((aligned(16))) char bw_smem[];
    // [kBwHB][3V] dL/dv_posed; before it is filled the same memory holds the per-wave partial sums of dL/dA
    float* sDVP = reinterpret_cast<float*>(bw_smem);
    float* sPart = sDVP;                                  // [kBwHB][waves][16][12]
    __shared__ BwShared sh;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * kBwHB;
    const int nh = (B - b0) < kBwHB ? (B - b0) : kBwHB;
    const int V3 = 3 * V;

    // ---- forward pieces needed again: full pose, rotations   (lbs.py:167-198)
    if (tid < kBwHB * kBwJ && tid / kBwJ < nh) {
        const int h = tid / kBwJ, j = tid % kBwJ, b = b0 + h;
        float rv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v;
            if (j == 0) {
                v = global_orient[b * 3 + c];
            } else {
                const int col = 3 * (j - 1) + c;
                v = 0.f;
                for (int i = 0; i < ncomp; ++i) v = __builtin_fmaf(hand_pose[b * ncomp + i], comps[i * 45 + col], v);
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
    }
    __syncthreads();
    // ---- kinematic chain, one thread per hand   (lbs.py:214-265)
    if (tid < nh) {
        const int h = tid;
        for (int i = 0; i < kBwJ; ++i) {
            const int par = parents[i];
            float t[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) t[c] = (i == 0) ? joints[c] : joints[i * 3 + c] - joints[par * 3 + c];
            float T[12];
            if (i == 0) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) T[a * 4 + c] = sh.R[h][0][a * 3 + c];
                    T[a * 4 + 3] = t[a];
                }
            } else {
                const float* P = sh.T[h][par];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float v = P[a * 4 + 0] * sh.R[h][i][0 * 3 + c];
                        v = __builtin_fmaf(P[a * 4 + 1], sh.R[h][i][1 * 3 + c], v);
                        v = __builtin_fmaf(P[a * 4 + 2], sh.R[h][i][2 * 3 + c], v);
                        T[a * 4 + c] = v;
                    }
                    float v = P[a * 4 + 0] * t[0];
                    v = __builtin_fmaf(P[a * 4 + 1], t[1], v);
                    v = __builtin_fmaf(P[a * 4 + 2], t[2], v);
                    T[a * 4 + 3] = v + P[a * 4 + 3];
                }
            }
            const float jx = joints[i * 3 + 0], jy = joints[i * 3 + 1], jz = joints[i * 3 + 2];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float v = T[a * 4 + 0] * jx;
                v = __builtin_fmaf(T[a * 4 + 1], jy, v);
                v = __builtin_fmaf(T[a * 4 + 2], jz, v);
#pragma unroll
                for (int c = 0; c < 4; ++c) sh.T[h][i][a * 4 + c] = T[a * 4 + c];
                sh.A[h][i][a * 4 + 0] = T[a * 4 + 0]; sh.A[h][i][a * 4 + 1] = T[a * 4 + 1]; sh.A[h][i][a * 4 + 2] = T[a * 4 + 2];
                sh.A[h][i][a * 4 + 3] = T[a * 4 + 3] - v;
            }
        }
    }

    // ---- dL/dA_j[a][k] = sum_v w_vj g_v[a] [v_posed_v; 1][k]: thread (joint, vertex slice), then slices are summed
    {
        const int j = tid & (kBwJ - 1), slice = tid >> 4;
        const int per = (V + kBwSlices - 1) / kBwSlices;
        const int v0 = slice * per, v1 = (v0 + per) < V ? (v0 + per) : V;
#pragma unroll 1
        for (int h = 0; h < nh; ++h) {
            const float* g = grad_verts + (int64_t)(b0 + h) * V3;
            const float* vp = v_posed + (int64_t)(b0 + h) * V3;
            float acc[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) acc[e] = 0.f;
            for (int v = v0; v < v1; ++v) {
                const float w = weights[v * kBwJ + j];
                const float p0 = vp[3 * v], p1 = vp[3 * v + 1], p2 = vp[3 * v + 2];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float wg = w * g[3 * v + a];
                    acc[a * 4 + 0] = __builtin_fmaf(wg, p0, acc[a * 4 + 0]);
                    acc[a * 4 + 1] = __builtin_fmaf(wg, p1, acc[a * 4 + 1]);
                    acc[a * 4 + 2] = __builtin_fmaf(wg, p2, acc[a * 4 + 2]);
                    acc[a * 4 + 3] += wg;
                }
            }
#pragma unroll
            for (int e = 0; e < 12; ++e) {       // the four slices that share a wave
                float x = acc[e];
                x += __shfl_xor(x, 16, 64);
                x += __shfl_xor(x, 32, 64);
                if (lane < kBwJ) sPart[((h * kTtaWaves + wave) * kBwJ + j) * 12 + e] = x;
            }
        }
    }
    __syncthreads();
    for (int o = tid; o < nh * kBwJ * 12; o += kTtaThreads) {
        const int h = o / (kBwJ * 12), r = o % (kBwJ * 12);
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < kTtaWaves; ++w) x += sPart[(h * kTtaWaves + w) * kBwJ * 12 + r];
        sh.dA[h][r / 12][r % 12] = x;
    }
    __syncthreads();

    // ---- skinning backward, per vertex: verts = T_v [v_posed; 1] + transl, T_v = sum_j w_vj A_j
#pragma unroll 1
    for (int h = 0; h < nh; ++h) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        for (int v = tid; v < V; v += kTtaThreads) {
            float TR[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) TR[e] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(weights + v * kBwJ + 4 * q);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            TR[a * 3 + c] = __builtin_fmaf(w[jj], sh.A[h][4 * q + jj][a * 4 + c], TR[a * 3 + c]);
            }
            const float* g = grad_verts + ((int64_t)(b0 + h) * V + v) * 3;
            const float g0 = g[0], g1 = g[1], g2 = g[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) sDVP[h * V3 + 3 * v + c] = TR[0 * 3 + c] * g0 + TR[1 * 3 + c] * g1 + TR[2 * 3 + c] * g2;
            t0 += g0; t1 += g1; t2 += g2;
        }
        t0 = wave_sum(t0); t1 = wave_sum(t1); t2 = wave_sum(t2);
        if (lane == 0) { sh.Tr[h][wave][0] = t0; sh.Tr[h][wave][1] = t1; sh.Tr[h][wave][2] = t2; }
    }
    __syncthreads();

    // ---- dL/dpose_feature[p] = sum_e posedirs[p][e] dv_posed[e]   (lbs.py:99-101), four p per wave and pass
    for (int p0 = 4 * wave; p0 < kBwPF; p0 += 4 * kTtaWaves) {
        float acc[4][kBwHB];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < kBwHB; ++h) acc[u][h] = 0.f;
        const float* pd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pd[u] = posedirs + (int64_t)((p0 + u) < kBwPF ? (p0 + u) : (kBwPF - 1)) * V3;
        for (int e = lane; e < V3; e += 64) {
            float d[kBwHB];
#pragma unroll
            for (int h = 0; h < kBwHB; ++h) d[h] = h < nh ? sDVP[h * V3 + e] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float x = pd[u][e];
#pragma unroll
                for (int h = 0; h < kBwHB; ++h) acc[u][h] = __builtin_fmaf(x, d[h], acc[u][h]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < kBwHB; ++h) {
                const float x = wave_sum(acc[u][h]);
                if (lane == 0 && p0 + u < kBwPF) sh.dPF[h][p0 + u] = x;
            }
    }
    __syncthreads();

    // ---- chain backward, one thread per hand (children have larger indices than their parents)
    if (tid < nh) {
        const int h = tid;
        for (int j = 0; j < kBwJ; ++j) {
            const float jx = joints[j * 3], jy = joints[j * 3 + 1], jz = joints[j * 3 + 2];
#pragma unroll
            for (int a = 0; a < 3; ++a) {   // A = [G_R | G_t - G_R J]
                const float dt = sh.dA[h][j][a * 4 + 3];
                sh.dGR[h][j][a * 3 + 0] = sh.dA[h][j][a * 4 + 0] - dt * jx;
                sh.dGR[h][j][a * 3 + 1] = sh.dA[h][j][a * 4 + 1] - dt * jy;
                sh.dGR[h][j][a * 3 + 2] = sh.dA[h][j][a * 4 + 2] - dt * jz;
                sh.dGt[h][j][a] = dt;
            }
        }
        for (int i = kBwJ - 1; i >= 1; --i) {
            const int par = parents[i];
            const float* P = sh.T[h][par];
            float rel[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) rel[c] = joints[i * 3 + c] - joints[par * 3 + c];
            float dGR[9], dGt[3];
#pragma unroll
            for (int e = 0; e < 9; ++e) dGR[e] = sh.dGR[h][i][e];
#pragma unroll
            for (int e = 0; e < 3; ++e) dGt[e] = sh.dGt[h][i][e];
            // G_R,i = G_R,par R_i ; G_t,i = G_R,par rel_i + G_t,par
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float dR = P[0 * 4 + a] * dGR[0 * 3 + c] + P[1 * 4 + a] * dGR[1 * 3 + c] + P[2 * 4 + a] * dGR[2 * 3 + c];
                    sh.dR[h][i][a * 3 + c] = dR + sh.dPF[h][(i - 1) * 9 + a * 3 + c];
                    sh.dGR[h][par][a * 3 + c] += dGR[a * 3 + 0] * sh.R[h][i][c * 3 + 0] + dGR[a * 3 + 1] * sh.R[h][i][c * 3 + 1] +
                                                 dGR[a * 3 + 2] * sh.R[h][i][c * 3 + 2] + dGt[a] * rel[c];
                }
#pragma unroll
            for (int a = 0; a < 3; ++a) sh.dGt[h][par][a] += dGt[a];
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) sh.dR[h][0][e] = sh.dGR[h][0][e];
    }
    __syncthreads();

    // ---- Rodrigues backward, one thread per (hand, joint)
    if (tid < kBwHB * kBwJ && tid / kBwJ < nh) {
        const int h = tid / kBwJ, j = tid % kBwJ;
        const float vx = sh.Rv[h][j][0], vy = sh.Rv[h][j][1], vz = sh.Rv[h][j][2];
        const float ex = vx + 1e-8f, ey = vy + 1e-8f, ez = vz + 1e-8f;
        const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
        const float inv = 1.f / ang;
        const float nx = vx * inv, ny = vy * inv, nz = vz * inv;
        const float cs = cosf(ang), sn = sinf(ang), omc = 1.f - cs;
        const float K[9] = {0.f, -nz, ny, nz, 0.f, -nx, -ny, nx, 0.f};
        float KK[9], dR[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                KK[a * 3 + c] = K[a * 3 + 0] * K[0 * 3 + c] + K[a * 3 + 1] * K[1 * 3 + c] + K[a * 3 + 2] * K[2 * 3 + c];
                dR[a * 3 + c] = sh.dR[h][j][a * 3 + c];
            }
        float dS = 0.f, dO = 0.f;
#pragma unroll
        for (int e = 0; e < 9; ++e) { dS += dR[e] * K[e]; dO += dR[e] * KK[e]; }
        float dK[9];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float t = 0.f;
#pragma unroll
                for (int m = 0; m < 3; ++m) t += dR[p * 3 + m] * K[q * 3 + m] + K[m * 3 + p] * dR[m * 3 + q];
                dK[p * 3 + q] = sn * dR[p * 3 + q] + omc * t;
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
    __syncthreads();

    if (tid < nh * 3) {
        const int h = tid / 3, c = tid % 3, b = b0 + h;
        if (g_orient) g_orient[b * 3 + c] = sh.dPose[h][c];
        if (g_transl) g_transl[b * 3 + c] = ((sh.Tr[h][0][c] + sh.Tr[h][1][c]) + sh.Tr[h][2][c]) + sh.Tr[h][3][c];
    }
    if (g_pose) {   // pose = coefficients x components (model.py:330-331)
        for (int o = tid; o < nh * ncomp; o += kTtaThreads) {
            const int h = o / ncomp, c = o % ncomp;
            float acc = 0.f;
            for (int m = 0; m < 45; ++m) acc = __builtin_fmaf(comps[c * 45 + m], sh.dPose[h][3 + m], acc);
            g_pose[(b0 + h) * ncomp + c] = acc;
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
    size_t lds = (size_t)kBwHB * 3 * m->n_verts * sizeof(float);
    const size_t part = (size_t)kBwHB * kTtaWaves * kBwJ * 12 * sizeof(float);
    if (lds < part) lds = part;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(mano_lbs_backward_kernel), lds, "mano_lbs_backward")) return rc;
    hipLaunchKernelGGL(mano_lbs_backward_kernel, dim3((B + kBwHB - 1) / kBwHB), dim3(kTtaThreads), lds, as_stream(stream), global_orient, hand_pose, B,
                       m->n_verts, m->n_comps, m->joints, m->posedirs, m->lbs_weights, m->components, m->pose_mean, m->parents,
                       v_posed, grad_verts, grad_global_orient, grad_hand_pose, grad_transl);
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
    const size_t lds = (size_t)mesh->n_verts * (16 + 16 + 24 + 24 + 4 + 4 + 4);
    if (lds > 150 * 1024) return fail(DVQ_EINVAL, "tta_contact_grad: hand mesh too large for LDS");
    // the kernel also holds a few hundred bytes of static LDS, so ask for what is needed, not for the whole array
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(tta_contact_grad_kernel), lds, "tta_contact_grad")) return rc;
    hipLaunchKernelGGL(tta_contact_grad_kernel, dim3(B), dim3(kCgThreads), lds, as_stream(stream), obj, obj_batch_stride,
                       obj_chan_stride, obj_point_stride, n_points, verts, mesh->n_verts, mesh->n_contact, mesh->faces, mesh->vert_face_ptr,
                       mesh->vert_face, mesh->search_order, contact_dist_sq, penetr_scale, contact_scale, w_penetr, w_contact,
                       grad_verts, losses, nn_idx);
    return check_launch("tta_contact_grad");
}
