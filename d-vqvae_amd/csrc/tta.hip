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
//    shapes -> kinematic chain -> Rodrigues -> PCA, one workgroup per hand.
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

// Vertices are searched in `order`: the n_contact vertices of the contact set first, then the others.
// The nearest contact-set vertex is then simply the running minimum at the end of the first group, and
// every (point, vertex) pair costs three packed subtract/multiply-add pairs and one compare-select, not two.
// Exact distance ties resolve to the vertex met first in that order.
__global__ __launch_bounds__(kTtaThreads) void tta_contact_grad_kernel(
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
    __shared__ float sPen[kTtaWaves], sCon[kTtaWaves];
    __shared__ int sN[kTtaWaves];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float* vb = verts + (int64_t)b * V * 3;
    const float inf = __builtin_inff();

    for (int k = tid; k < V; k += kTtaThreads) {
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
    for (int k = tid; k < V; k += kTtaThreads) {
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
    for (int i0 = tid; i0 < N; i0 += kTtaPts * kTtaThreads) {
        f32x2 px[2], py[2], pz[2], best[2];
        int bi[kTtaPts];
#pragma unroll
        for (int q = 0; q < kTtaPts; ++q) {
            int i = i0 + q * kTtaThreads;
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
        // the running best now names an exact vertex; mark it so that a later group start is distinguishable
        int exact[kTtaPts];
#pragma unroll
        for (int q = 0; q < kTtaPts; ++q) { exact[q] = bi[q]; bi[q] = -1; }
        scan(nC, V);
#pragma unroll
        for (int q = 0; q < kTtaPts; ++q) bi[q] = bi[q] < 0 ? exact[q] : resolve(q, bi[q], V);
#pragma unroll
        for (int q = 0; q < kTtaPts; ++q) {
            const int i = i0 + q * kTtaThreads;
            if (i >= N) continue;
            const float ox = px[q >> 1][q & 1], oy = py[q >> 1][q & 1], oz = pz[q >> 1][q & 1], bd = best[q >> 1][q & 1];
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
    for (int w = 0; w < kTtaWaves; ++w) { penT += sPen[w]; conT += sCon[w]; nT += sN[w]; }
    const float cP = 2.f * w_penetr * penetr_scale;
    const float cC = nT > 0 ? 2.f * w_contact * contact_scale / (float)nT : 0.f;
    if (tid == 0) {
        losses[2 * b] = penetr_scale * penT;
        losses[2 * b + 1] = nT > 0 ? contact_scale * conT / (float)nT : 0.f;
    }
    float* gb = grad_verts + (int64_t)b * V * 3;
    for (int k = tid; k < V; k += kTtaThreads) {
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

__global__ __launch_bounds__(kTtaThreads) void mano_lbs_backward_kernel(
    const float* __restrict__ global_orient, const float* __restrict__ hand_pose, int B, int V, int ncomp,
    const float* __restrict__ joints, const float* __restrict__ posedirs, const float* __restrict__ weights,
    const float* __restrict__ comps, const float* __restrict__ pose_mean, const int* __restrict__ parents,
    const float* __restrict__ v_posed, const float* __restrict__ grad_verts, float* __restrict__ g_orient,
    float* __restrict__ g_pose, float* __restrict__ g_transl) {
    extern __shared__ __attribute__((aligned(16))) char bw_smem[];
    float* sG = reinterpret_cast<float*>(bw_smem);     // [3V] dL/dverts of this hand
    float* sVP = sG + 3 * V;                           // [3V] v_posed
    float* sDVP = sVP + 3 * V;                         // [3V] dL/dv_posed
    __shared__ float sRv[kBwJ][3];      // full pose (axis-angle per joint)
    __shared__ float sR[kBwJ][9];
    __shared__ float sT[kBwJ][12];      // world transforms
    __shared__ float sA[kBwJ][12];      // skinning transforms
    __shared__ float sdA[kBwJ][12];
    __shared__ float sdPF[kBwPF + 1];
    __shared__ float sdGR[kBwJ][9];
    __shared__ float sdGt[kBwJ][3];
    __shared__ float sdR[kBwJ][9];
    __shared__ float sdPose[48];
    __shared__ float sTr[kTtaWaves][3];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;

    // forward pieces needed again: full pose, rotations   (lbs.py:167-198)
    if (tid < kBwJ) {
        const int j = tid;
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
            sRv[j][c] = v;
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
                sR[j][a * 3 + c] = ((a == c ? 1.f : 0.f) + sn * K[a * 3 + c]) + omc * kk;
            }
    }
    for (int e = tid; e < 3 * V; e += kTtaThreads) {
        sG[e] = grad_verts[(int64_t)b * 3 * V + e];
        sVP[e] = v_posed[(int64_t)b * 3 * V + e];
    }
    __syncthreads();
    // kinematic chain   (lbs.py:214-265)
    if (tid == 0) {
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
                    for (int c = 0; c < 3; ++c) T[a * 4 + c] = sR[0][a * 3 + c];
                    T[a * 4 + 3] = t[a];
                }
            } else {
                const float* P = sT[par];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float v = P[a * 4 + 0] * sR[i][0 * 3 + c];
                        v = __builtin_fmaf(P[a * 4 + 1], sR[i][1 * 3 + c], v);
                        v = __builtin_fmaf(P[a * 4 + 2], sR[i][2 * 3 + c], v);
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
                for (int c = 0; c < 4; ++c) sT[i][a * 4 + c] = T[a * 4 + c];
                sA[i][a * 4 + 0] = T[a * 4 + 0]; sA[i][a * 4 + 1] = T[a * 4 + 1]; sA[i][a * 4 + 2] = T[a * 4 + 2];
                sA[i][a * 4 + 3] = T[a * 4 + 3] - v;
            }
        }
    }
    __syncthreads();

    // skinning backward, per vertex: verts = T_v [v_posed; 1] + transl, T_v = sum_j w_vj A_j
    float tr[3] = {0.f, 0.f, 0.f};
    for (int v = tid; v < V; v += kTtaThreads) {
        float TR[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) TR[e] = 0.f;
#pragma unroll
        for (int j = 0; j < kBwJ; ++j) {
            const float w = weights[v * kBwJ + j];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) TR[a * 3 + c] = __builtin_fmaf(w, sA[j][a * 4 + c], TR[a * 3 + c]);
        }
        const float g0 = sG[3 * v], g1 = sG[3 * v + 1], g2 = sG[3 * v + 2];
#pragma unroll
        for (int c = 0; c < 3; ++c) sDVP[3 * v + c] = TR[0 * 3 + c] * g0 + TR[1 * 3 + c] * g1 + TR[2 * 3 + c] * g2;
        tr[0] += g0; tr[1] += g1; tr[2] += g2;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float s = wave_sum(tr[c]);
        if (lane == 0) sTr[wave][c] = s;
    }
    __syncthreads();

    // dL/dA_j[a][k] = sum_v w_vj g_v[a] [v_posed_v; 1][k]
    if (tid < kBwJ * 12) {
        const int j = tid / 12, a = (tid % 12) / 4, k = tid % 4;
        float acc = 0.f;
        if (k < 3) {
            for (int v = 0; v < V; ++v) acc = __builtin_fmaf(weights[v * kBwJ + j] * sG[3 * v + a], sVP[3 * v + k], acc);
        } else {
            for (int v = 0; v < V; ++v) acc = __builtin_fmaf(weights[v * kBwJ + j], sG[3 * v + a], acc);
        }
        sdA[j][a * 4 + k] = acc;
    }
    // dL/dpose_feature[p] = sum_e posedirs[p][e] dv_posed[e]   (lbs.py:99-101)
    for (int p = wave; p < kBwPF; p += kTtaWaves) {
        const float* pd = posedirs + (int64_t)p * 3 * V;
        float acc = 0.f;
        for (int e = lane; e < 3 * V; e += 64) acc = __builtin_fmaf(pd[e], sDVP[e], acc);
        acc = wave_sum(acc);
        if (lane == 0) sdPF[p] = acc;
    }
    __syncthreads();

    // chain backward (children have larger indices than their parents)
    if (tid == 0) {
        for (int j = 0; j < kBwJ; ++j) {
            const float jx = joints[j * 3], jy = joints[j * 3 + 1], jz = joints[j * 3 + 2];
#pragma unroll
            for (int a = 0; a < 3; ++a) {   // A = [G_R | G_t - G_R J]
                const float dt = sdA[j][a * 4 + 3];
                sdGR[j][a * 3 + 0] = sdA[j][a * 4 + 0] - dt * jx;
                sdGR[j][a * 3 + 1] = sdA[j][a * 4 + 1] - dt * jy;
                sdGR[j][a * 3 + 2] = sdA[j][a * 4 + 2] - dt * jz;
                sdGt[j][a] = dt;
            }
        }
        for (int i = kBwJ - 1; i >= 1; --i) {
            const int par = parents[i];
            const float* P = sT[par];
            float rel[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) rel[c] = joints[i * 3 + c] - joints[par * 3 + c];
            float dGR[9], dGt[3];
#pragma unroll
            for (int e = 0; e < 9; ++e) dGR[e] = sdGR[i][e];
#pragma unroll
            for (int e = 0; e < 3; ++e) dGt[e] = sdGt[i][e];
            // G_R,i = G_R,par R_i ; G_t,i = G_R,par rel_i + G_t,par
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float dR = P[0 * 4 + a] * dGR[0 * 3 + c] + P[1 * 4 + a] * dGR[1 * 3 + c] + P[2 * 4 + a] * dGR[2 * 3 + c];
                    sdR[i][a * 3 + c] = dR + sdPF[(i - 1) * 9 + a * 3 + c];
                    sdGR[par][a * 3 + c] += dGR[a * 3 + 0] * sR[i][c * 3 + 0] + dGR[a * 3 + 1] * sR[i][c * 3 + 1] +
                                            dGR[a * 3 + 2] * sR[i][c * 3 + 2] + dGt[a] * rel[c];
                }
#pragma unroll
            for (int a = 0; a < 3; ++a) sdGt[par][a] += dGt[a];
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) sdR[0][e] = sdGR[0][e];
    }
    __syncthreads();

    // Rodrigues backward, one thread per joint
    if (tid < kBwJ) {
        const int j = tid;
        const float vx = sRv[j][0], vy = sRv[j][1], vz = sRv[j][2];
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
                dR[a * 3 + c] = sdR[j][a * 3 + c];
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
        sdPose[3 * j + 0] = dnx * inv + da * ex * inv;
        sdPose[3 * j + 1] = dny * inv + da * ey * inv;
        sdPose[3 * j + 2] = dnz * inv + da * ez * inv;
    }
    __syncthreads();

    if (tid < 3) {
        if (g_orient) g_orient[b * 3 + tid] = sdPose[tid];
        if (g_transl) g_transl[b * 3 + tid] = ((sTr[0][tid] + sTr[1][tid]) + sTr[2][tid]) + sTr[3][tid];
    }
    if (g_pose && tid >= 64 && tid < 64 + ncomp) {   // pose = coefficients x components (model.py:330-331)
        const int c = tid - 64;
        float acc = 0.f;
        for (int m = 0; m < 45; ++m) acc = __builtin_fmaf(comps[c * 45 + m], sdPose[3 + m], acc);
        g_pose[b * ncomp + c] = acc;
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
    if (m->n_verts < 1 || m->n_verts > 4096 || m->n_comps < 1 || m->n_comps > 45)
        return fail(DVQ_EINVAL, "mano_backward: bad n_verts/n_comps (%d, %d)", m->n_verts, m->n_comps);
    if (!m->joints || !m->posedirs || !m->lbs_weights || !m->components || !m->pose_mean || !m->parents)
        return fail(DVQ_EINVAL, "mano_backward: null constant table");
    const size_t lds = (size_t)9 * m->n_verts * sizeof(float);
    hipLaunchKernelGGL(mano_lbs_backward_kernel, dim3(B), dim3(kTtaThreads), lds, as_stream(stream), global_orient, hand_pose, B,
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
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(tta_contact_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return fail(DVQ_ELAUNCH, "tta_contact_grad: cannot raise the dynamic LDS limit to %zu bytes", lds);
    hipLaunchKernelGGL(tta_contact_grad_kernel, dim3(B), dim3(kTtaThreads), lds, as_stream(stream), obj, obj_batch_stride,
                       obj_chan_stride, obj_point_stride, n_points, verts, mesh->n_verts, mesh->n_contact, mesh->faces, mesh->vert_face_ptr,
                       mesh->vert_face, mesh->search_order, contact_dist_sq, penetr_scale, contact_scale, w_penetr, w_contact,
                       grad_verts, losses, nn_idx);
    return check_launch("tta_contact_grad");
}
