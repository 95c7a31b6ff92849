// MANO hand layer: PCA pose -> Rodrigues -> kinematic chain -> pose blend shapes ->
// linear blend skinning, one launch (gfx950).
//
// Replaces MANO.forward (DVQ-VAE-2/MANO-master/mano/model.py:311-359) and lbs with its
// helpers (mano/lbs.py:33-265) for use_pca=True, pose2rot=True.  The reference always
// passes its own all-zero betas to lbs (model.py:335), so v_shaped = v_template and the
// rest joints J = J_regressor * v_shaped are constants supplied by the host.
//
// The work is bandwidth-shaped: per hand 244 B in, 9.3 KB out and ~0.5 M multiply-adds,
// of which the pose blend-shape product [135] x [135, 3V] is the bulk.  One workgroup
// therefore takes HB hands at once so that every posedirs element fetched (coalesced,
// 12 B per lane, L2-resident) feeds HB multiply-adds, and each lane owns whole vertices
// so the [B,V,3] result is written as contiguous 12-B-per-lane rows.
#include "common.h"

namespace dvq {

constexpr int kHB = 8;        // hands per workgroup
constexpr int kJ = 16;        // joints
constexpr int kPF = 135;      // pose-feature length = 15 joints * 9
constexpr int kManoThreads = 256;

__global__ __launch_bounds__(kManoThreads, 2) void mano_lbs_kernel(
    const float* __restrict__ global_orient, const float* __restrict__ hand_pose, const float* __restrict__ transl,
    int B, int V, int ncomp, const float* __restrict__ v_shaped, const float* __restrict__ joints,
    const float* __restrict__ posedirs, const float* __restrict__ weights, const float* __restrict__ comps,
    const float* __restrict__ pose_mean, const int* __restrict__ parents, float* __restrict__ verts_out,
    float* __restrict__ joints_out, float* __restrict__ full_pose_out, float* __restrict__ v_posed_out) {
    __shared__ float sR[kHB][kJ][9];     // rotation matrices
    __shared__ __attribute__((aligned(16))) float sA[kHB][kJ][12];    // relative transforms, upper 3x4
    __shared__ float sT[kHB][kJ][12];    // world transforms, upper 3x4
    __shared__ float sPF[kHB][kPF + 1];  // pose feature (R[1:] - I)
    __shared__ float sTr[kHB][4];        // translation (0 when absent)
    __shared__ float sVP[kHB][3][kManoThreads];  // posed vertex of each thread, per hand

    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * kHB;
    const int nh = (B - b0) < kHB ? (B - b0) : kHB;

    // ---- A: full pose and Rodrigues, one thread per (hand, joint)   (lbs.py:167-198)
    if (tid < kHB * kJ) {
        const int hnd = tid / kJ, j = tid % kJ;
        if (hnd < nh) {
            const int b = b0 + hnd;
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
                for (int c = 0; c < 3; ++c) sTr[hnd][c] = transl ? transl[b * 3 + c] : 0.f;
            }
        }
    }
    __syncthreads();

    // ---- B: kinematic chain, one thread per hand   (lbs.py:214-265)
    if (tid < nh) {
        const int hnd = tid;
        for (int i = 0; i < kJ; ++i) {
            const int par = parents[i];
            float t[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) t[c] = (i == 0) ? joints[c] : joints[i * 3 + c] - joints[par * 3 + c];
            float T[12];
            if (i == 0) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) T[a * 4 + c] = sR[hnd][0][a * 3 + c];
                    T[a * 4 + 3] = t[a];
                }
            } else {
                const float* P = sT[hnd][par];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float v = P[a * 4 + 0] * sR[hnd][i][0 * 3 + c];
                        v = __builtin_fmaf(P[a * 4 + 1], sR[hnd][i][1 * 3 + c], v);
                        v = __builtin_fmaf(P[a * 4 + 2], sR[hnd][i][2 * 3 + c], v);
                        T[a * 4 + c] = v;
                    }
                    float v = P[a * 4 + 0] * t[0];
                    v = __builtin_fmaf(P[a * 4 + 1], t[1], v);
                    v = __builtin_fmaf(P[a * 4 + 2], t[2], v);
                    T[a * 4 + 3] = v + P[a * 4 + 3];
                }
            }
#pragma unroll
            for (int e = 0; e < 12; ++e) sT[hnd][i][e] = T[e];
            // A = T - pad(T * [J;0])   (lbs.py:260-263)
            const float jx = joints[i * 3 + 0], jy = joints[i * 3 + 1], jz = joints[i * 3 + 2];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float v = T[a * 4 + 0] * jx;
                v = __builtin_fmaf(T[a * 4 + 1], jy, v);
                v = __builtin_fmaf(T[a * 4 + 2], jz, v);
                sA[hnd][i][a * 4 + 0] = T[a * 4 + 0];
                sA[hnd][i][a * 4 + 1] = T[a * 4 + 1];
                sA[hnd][i][a * 4 + 2] = T[a * 4 + 2];
                sA[hnd][i][a * 4 + 3] = T[a * 4 + 3] - v;
            }
            if (joints_out) {
                const int b = b0 + hnd;
#pragma unroll
                for (int c = 0; c < 3; ++c) joints_out[(b * kJ + i) * 3 + c] = T[c * 4 + 3] + sTr[hnd][c];
            }
        }
    }
    __syncthreads();

    // ---- C: per vertex: pose blend shapes, then skinning   (lbs.py:90-118)
    for (int v = tid; v < V; v += kManoThreads) {
        {
            float vp[kHB][3];
#pragma unroll
            for (int hnd = 0; hnd < kHB; ++hnd) { vp[hnd][0] = 0.f; vp[hnd][1] = 0.f; vp[hnd][2] = 0.f; }
            const float* pd = posedirs + 3 * v;
            const int stride = 3 * V;
#pragma unroll 3
            for (int p = 0; p < kPF; ++p) {
                const float d0 = pd[(int64_t)p * stride + 0], d1 = pd[(int64_t)p * stride + 1], d2 = pd[(int64_t)p * stride + 2];
#pragma unroll
                for (int hnd = 0; hnd < kHB; ++hnd) {
                    const float f = sPF[hnd][p];
                    vp[hnd][0] = __builtin_fmaf(f, d0, vp[hnd][0]);
                    vp[hnd][1] = __builtin_fmaf(f, d1, vp[hnd][1]);
                    vp[hnd][2] = __builtin_fmaf(f, d2, vp[hnd][2]);
                }
            }
            const float vx = v_shaped[v * 3 + 0], vy = v_shaped[v * 3 + 1], vz = v_shaped[v * 3 + 2];
#pragma unroll
            for (int hnd = 0; hnd < kHB; ++hnd) {  // v_posed = pose_offsets + v_shaped  (lbs.py:101)
                sVP[hnd][0][tid] = vp[hnd][0] + vx;
                sVP[hnd][1][tid] = vp[hnd][1] + vy;
                sVP[hnd][2][tid] = vp[hnd][2] + vz;
            }
        }
        float w[kJ];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(weights + v * kJ + 4 * q);
            w[4 * q + 0] = t[0]; w[4 * q + 1] = t[1]; w[4 * q + 2] = t[2]; w[4 * q + 3] = t[3];
        }
#pragma unroll 1
        for (int hnd = 0; hnd < nh; ++hnd) {
            const float px = sVP[hnd][0][tid], py = sVP[hnd][1][tid], pz = sVP[hnd][2][tid];
            float T[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = 0.f;
            // T = sum_j w_j A_j: zero weights (13 of 16 per vertex on average) contribute exact
            // zeros, as in the reference's dense product
#pragma unroll
            for (int j = 0; j < kJ; ++j) {
                const f32x4 A0 = *reinterpret_cast<const f32x4*>(&sA[hnd][j][0]);
                const f32x4 A1 = *reinterpret_cast<const f32x4*>(&sA[hnd][j][4]);
                const f32x4 A2 = *reinterpret_cast<const f32x4*>(&sA[hnd][j][8]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    T[e] = __builtin_fmaf(w[j], A0[e], T[e]);
                    T[4 + e] = __builtin_fmaf(w[j], A1[e], T[4 + e]);
                    T[8 + e] = __builtin_fmaf(w[j], A2[e], T[8 + e]);
                }
            }
            float* dst = verts_out + ((int64_t)(b0 + hnd) * V + v) * 3;
            if (v_posed_out) {   // kept for dvq_mano_lbs_backward
                float* vpd = v_posed_out + ((int64_t)(b0 + hnd) * V + v) * 3;
                vpd[0] = px; vpd[1] = py; vpd[2] = pz;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float r = T[a * 4 + 0] * px;
                r = __builtin_fmaf(T[a * 4 + 1], py, r);
                r = __builtin_fmaf(T[a * 4 + 2], pz, r);
                r = r + T[a * 4 + 3];
                dst[a] = r + sTr[hnd][a];
            }
        }
    }
}

}  // namespace dvq

using namespace dvq;

extern "C" int dvq_mano_lbs(const float* global_orient, const float* hand_pose, const float* transl, int B,
                            const DvqManoConsts* m, float* verts_out, float* joints_out, float* full_pose_out, float* v_posed_out,
                            void* stream) {
    if (B < 0) return fail(DVQ_EINVAL, "mano: negative batch");
    if (B == 0) return DVQ_OK;
    if (!m || !global_orient || !hand_pose || !verts_out) return fail(DVQ_EINVAL, "mano: null pointer");
    if (m->n_verts < 1 || m->n_comps < 1 || m->n_comps > 45) return fail(DVQ_EINVAL, "mano: bad n_verts/n_comps (%d, %d)", m->n_verts, m->n_comps);
    if (!m->v_shaped || !m->joints || !m->posedirs || !m->lbs_weights || !m->components || !m->pose_mean || !m->parents)
        return fail(DVQ_EINVAL, "mano: null constant table");
    if (((uintptr_t)m->lbs_weights & 15) != 0) return fail(DVQ_EINVAL, "mano: lbs_weights must be 16-byte aligned");
    const int grid = (B + kHB - 1) / kHB;
    hipLaunchKernelGGL(mano_lbs_kernel, dim3(grid), dim3(kManoThreads), 0, as_stream(stream), global_orient, hand_pose, transl, B,
                       m->n_verts, m->n_comps, m->v_shaped, m->joints, m->posedirs, m->lbs_weights, m->components, m->pose_mean,
                       m->parents, verts_out, joints_out, full_pose_out, v_posed_out);
    return check_launch("mano_lbs");
}
