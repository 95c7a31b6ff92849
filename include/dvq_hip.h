/*
 * dvq_hip.h -- C ABI of libdvq_hip.so: MI355X (gfx950) kernels for the D-VQVAE
 * grasp-generation inference path.
 *
 * The reference (MICLAB-BUPT/D-VQVAE) has no FFI of its own: its boundary is the
 * Python nn.Module surface.  Each entry point below replaces the body of one
 * reference forward(); the Python modules under d-vqvae_amd/network and
 * d-vqvae_amd/mano keep the reference's class names, signatures and state_dict
 * keys and call these through ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to a contiguous row-major buffer owned by
 *    the caller; nothing here allocates, frees or synchronises;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *  - every function returns DVQ_OK (0) or a negative DVQ_E* code and never aborts;
 *    dvq_last_error() then describes the failure (per calling thread);
 *  - dtype: DVQ_F32 = exact fp32 arithmetic, DVQ_BF16 = bf16 operands with fp32
 *    accumulation.
 */
#ifndef DVQ_HIP_H
#define DVQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVQ_OK 0
#define DVQ_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define DVQ_ELAUNCH (-2)  /* the HIP runtime rejected a launch               */
#define DVQ_ENODEV (-3)   /* no usable gfx950 device                         */

#define DVQ_F32 0
#define DVQ_BF16 1

#define DVQ_VERSION 113   /* major*10000 + minor*100 + patch */

int dvq_version(void);
const char* dvq_last_error(void);

/* ---------------------------------------------------------------- VQ argmin
 * Replaces VectorQuantizer.forward(z, istrain=False)
 * (reference: DVQ-VAE/network/vqvae/quantizer.py:44-54):
 *   d[i,k] = fl( fl(|z_i|^2 + |E_k|^2) - 2 * <z_i, E_k> ),  idx[i] = first argmin_k d[i,k]
 *   z_q[i] = E[idx[i]]
 * The [n,K] distance matrix stays in registers.  z: [n,D], E: [K,D], both of `dtype`
 * (float for DVQ_F32, bfloat16 bits for DVQ_BF16).  idx_out: int64 [n].  zq_out
 * (nullable): [n,D] of `dtype`, an exact copy of the chosen rows.
 * Limits: D % 16 == 0, 16 <= D <= 4096, 1 <= K <= 65536, n >= 0.
 *
 * `workspace` holds the codebook re-laid in MFMA fragment order plus |E_k|^2:
 * dvq_vq_workspace_bytes(K, D, dtype) bytes, 256-byte aligned.
 *   dvq_vq_prepare : fills the workspace from E (once per codebook version).
 *   dvq_vq_argmin  : prepare + search in one call (workspace is scratch).
 *   dvq_vq_argmin_prepared : search only, on a workspace filled by dvq_vq_prepare
 *                    with the same (E, K, D, dtype).
 */
size_t dvq_vq_workspace_bytes(int K, int D, int dtype);
int dvq_vq_prepare(const void* E, int K, int D, int dtype, void* workspace, void* stream);
int dvq_vq_argmin(const void* z, const void* E, int64_t n, int K, int D, int dtype,
                  int64_t* idx_out, void* zq_out, void* workspace, void* stream);
int dvq_vq_argmin_prepared(const void* z, const void* E, int64_t n, int K, int D, int dtype,
                           int64_t* idx_out, void* zq_out, const void* workspace, void* stream);

/* dvq_vq_argmin without the workspace argument, for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The workspace is
 * a DVQ_VQ_STATIC_WORKSPACE_BYTES array inside the library, one copy per GPU, held until the process ends; the call
 * allocates and frees nothing.  Codebooks with dvq_vq_workspace_bytes(K, D, dtype) above that size are rejected with
 * DVQ_EINVAL (K=8192 x D=256 fits in both dtypes).  Same results as dvq_vq_argmin.  All calls on one GPU share the
 * array: calls that overlap on one GPU (different streams or threads) race; calls on one stream are safe.
 * Under stream capture: the first call on a device looks the array's address up (hipGetSymbolAddress) and the first
 * call that reaches a kernel variant raises its dynamic-LDS limit (hipFuncSetAttribute); neither belongs in a capture,
 * so make one call with the same (K, D, dtype) and a row count of the same order on that device before capturing. */
#define DVQ_VQ_STATIC_WORKSPACE_BYTES (9u << 20)
int dvq_vq_argmin_static(const void* z, const void* E, int64_t n, int K, int D, int dtype,
                         int64_t* idx_out, void* zq_out, void* stream);

/* Up to 8 searches of ONE shape (n, K, D, dtype), each with its own latents, codebook and prepared workspace, as one
 * launch: the six part codebooks of DVQVAE.forward (DVQ-VAE/network/DVQVAE.py:130-136, vqvae0..5.inference).  z, E,
 * idx_out, workspace: host arrays of n_batch device pointers.  Indices only (no z_q). */
int dvq_vq_argmin_prepared_batched(int n_batch, const void* const* z, const void* const* E, int64_t n, int K, int D,
                                   int dtype, int64_t* const* idx_out, const void* const* workspace, void* stream);

/* Replaces VectorQuantizer.get_emb (quantizer.py:68-75), batched: out[b] = E[idx[b]].
 * idx: int64 [n]; E, out: float.  `ld_out` is the row stride of `out` in elements
 * (>= D) so that rows can be written straight into a wider concatenated buffer. */
int dvq_vq_gather(const int64_t* idx, const float* E, int64_t n, int K, int D,
                  float* out, int64_t ld_out, void* stream);

/* ------------------------------------------------------- PointNet encoder
 * Replaces the per-point shared MLP + global max-pool of STN3d.forward and
 * PointNetEncoder.forward (reference: DVQ-VAE/network/pointnet_encoder.py:27-33 and
 * :146-164):  per point  h1 = relu(A1 x + c1) [64],  h2 = relu(A2 h1 + c2) [128],
 * h3 = A3 h2 + c3 [1024] (optionally relu), then max over the N points.
 * A*, c* are the Conv1d(k=1) weights with the eval-mode BatchNorm folded in
 * (dvq_fold_bn).  When `trans` is given the xyz channels of every point are first
 * multiplied by the cloud's 3x3 matrix (row vector times matrix, :146).
 *
 * x: float [B,C,N] (channel-major, as the reference's datasets produce), C = 3 or 4; with point_major != 0 the
 * same clouds laid out [B,N,C] (what a vertex array is: the permute + copy of gen_net.py:120 is then not needed).
 * trans: float [B,3,3] or NULL.  feat_out: float [B,1024].
 * w3p: A3 packed by dvq_pointnet_pack_w3 for `dtype`.
 */
typedef struct {
    const float* a1;   /* [64, C]   */
    const float* c1;   /* [64]      */
    const float* a2;   /* [128, 64] */
    const float* c2;   /* [128]     */
    const void*  w3p;  /* packed [1024,128], see dvq_pointnet_pack_w3 */
    const float* c3;   /* [1024]    */
} DvqPointMlp;

size_t dvq_pointnet_w3_bytes(int dtype);
int dvq_pointnet_pack_w3(const float* a3 /* [1024,128] */, int dtype, void* w3p, void* stream);
int dvq_pointnet_mlp_max(const float* x, const float* trans, int B, int C, int N, int point_major,
                         const DvqPointMlp* w, int relu3, int dtype, float* feat_out, void* stream);

/* Up to 8 independent dvq_pointnet_mlp_max calls (own clouds, point count, transform and weights each; one C and
 * dtype for all) as ONE launch: the six hand-part encoders of DVQVAE.forward (DVQ-VAE/network/DVQVAE.py:20-22,
 * 88-96 -- `hand_encoder(hand_xyz[:, :, idx])` for six vertex sets) and their six transform nets. */
typedef struct {
    const float* x;        /* [B, C, N] */
    const float* trans;    /* [B, 3, 3] or NULL */
    int B, N;
    DvqPointMlp w;
    int relu3;
    float* feat_out;       /* [B, 1024] */
} DvqPointGroup;
/* The vertex subsets the part encoders read (DVQ-VAE/network/DVQVAE.py:49-51, 90-95: centre the hand, then
 * `hand_xyz[:, :, idx]` for six index lists) in one launch.  pts [B,3,V]; idx: the lists back to back (device,
 * values in [0,V), NOT checked); part_ptr [n_parts+1]: offsets into idx (device); out: part k is a contiguous
 * [B,3,n_k] block starting at float offset 3*B*part_ptr[k].  center != 0 subtracts the per-axis mean over all V. */
int dvq_gather_parts(const float* pts, int B, int V, const int* idx, const int* part_ptr, int n_parts, int center,
                     float* out, void* stream);
int dvq_pointnet_mlp_max_grouped(const DvqPointGroup* groups, int n_groups, int C, int dtype, void* stream);

/* The same product with bf16 matrix-core operands and fp32 accumulation (the throughput configuration of the
 * heads: DVQ-VAE/network/DVQVAE.py:161-185, pointnet_encoder.py:35-36): x fp32 [M,K] (row stride ldx) is rounded to
 * bf16 as it is read, Wb is the weight as bf16 [N,K] row-major (dvq_to_bf16 makes it), y = act(x Wb^T + b) fp32.
 * K % 32 == 0. */
int dvq_to_bf16(const float* in, void* out_bf16, int64_t n, void* stream);
/* Up to 8 products of ONE shape (own x, W, b, y each) as one launch: the six part heads of DVQVAE.forward
 * (DVQ-VAE/network/DVQVAE.py:15-17, 97 -- `handembnn(feature)` for six parts) and the fully connected layers of the
 * six part transform nets (pointnet_encoder.py:35-43).  x, W, b, y: host arrays of n_batch device pointers (b may be
 * NULL); w_bf16 != 0: the W[i] are bf16 copies and the bf16 kernel runs (K % 32 == 0, no identity). */
int dvq_linear_batched(int n_batch, const float* const* x, int64_t ldx, const void* const* W, const float* const* b,
                       int M, int K, int N, int relu, int add_identity3, float* const* y, int64_t ldy, int w_bf16,
                       void* stream);
int dvq_linear_bf16(const float* x, int64_t ldx, const void* Wb, const float* b, int M, int K, int N,
                    int relu, float* y, int64_t ldy, void* stream);

/* Folds an eval-mode BatchNorm1d into the preceding affine layer:
 *   s = gamma / sqrt(var + eps);  A = s * W (row-wise);  c = s * (b - mean) + beta.
 * W: [out,in] -> A: [out,in];  b, gamma, beta, mean, var, c: [out]. */
int dvq_fold_bn(const float* W, const float* b, const float* gamma, const float* beta,
                const float* mean, const float* var, float eps, int out_f, int in_f,
                float* A, float* c, void* stream);

/* ----------------------------------------------------------------- linear
 * y = act(x W^T + b): the nn.Linear(+ReLU) layers of the STN head
 * (pointnet_encoder.py:35-37) and of Encoder/Decoder (DVQ-VAE/network/DVQVAE.py:161-185).
 * x: [M, K] with row stride ldx; W: [N, K]; b: [N] or NULL; y: [M, N] with row stride ldy.
 * add_identity3 != 0 adds the flattened 3x3 identity to a 9-wide output (pointnet_encoder.py:39-43).
 * fp32 arithmetic. */
int dvq_linear(const float* x, int64_t ldx, const float* W, const float* b, int M, int K, int N,
               int relu, int add_identity3, float* y, int64_t ldy, void* stream);

/* ------------------------------------------------------------------- MANO
 * Replaces MANO.forward + lbs (reference: DVQ-VAE-2/MANO-master/mano/model.py:311-359,
 * mano/lbs.py:33-265) for use_pca=True, pose2rot=True.  As in the reference the shape
 * coefficients passed by the caller are ignored (model.py:335 hands lbs the module's own
 * zero betas), so v_shaped and the rest joints are constants precomputed on the host.
 *
 * global_orient: [B,3]; hand_pose: [B,ncomps] PCA coefficients; transl: [B,3] or NULL.
 * verts_out: [B,V,3]; joints_out: [B,16,3] or NULL; full_pose_out: [B,48] or NULL;
 * v_posed_out: [B,V,3] or NULL -- the vertices after the pose blend shapes and before skinning
 * (lbs.py:101), which dvq_mano_lbs_backward needs.
 */
typedef struct {
    int n_verts;              /* V (778)                                  */
    int n_comps;              /* PCA components in use (<= 45)            */
    const float* v_shaped;    /* [V,3]  v_template + betas*shapedirs      */
    const float* joints;      /* [16,3] J_regressor * v_shaped            */
    const float* posedirs;    /* [135, V*3]                               */
    const float* lbs_weights; /* [V,16]                                   */
    const float* components;  /* [n_comps,45]                             */
    const float* pose_mean;   /* [48]                                     */
    const int*   parents;     /* [16], parents[0] = -1                    */
    /* optional table of the non-zero skinning weights (lbs_weights has 1-6 per vertex): vertex v, slot e ->
     * joint skin_joints[v*skin_width + e] with weight skin_weights[v*skin_width + e], the vertex's non-zero joints in
     * ascending order, remaining slots weight 0 (any valid joint).  NULL: the forward pass reads all 16 dense weights. */
    const unsigned char* skin_joints;
    const float* skin_weights;
    int skin_width;           /* 8 (vertices with more non-zero weights: leave the table NULL) */
    /* optional copy of posedirs in the order the forward kernel's lanes read it, [ceil(V/16), 36, 64, 3]: entry
     * (g, s, lane) holds the three coordinates of vertex min(16g + lane%16, V-1) for pose feature 4s + lane/16, zeros for
     * features >= 135 (the last two steps s are all zero), so that one wave load is 768 contiguous bytes and a step keeps
     * its place in a ring of 12.  Used with the skinning table; NULL: the kernel reads posedirs. */
    const float* posedirs_lanes;
} DvqManoConsts;

int dvq_mano_lbs(const float* global_orient, const float* hand_pose, const float* transl, int B,
                 const DvqManoConsts* m, float* verts_out, float* joints_out, float* full_pose_out,
                 float* v_posed_out, void* stream);
/* The same with a row stride (in floats) per input, so that the three inputs can be column ranges of wider rows -- the
 * call sites slice them out of one [B,61] parameter vector (reference: DVQ-VAE/gen_diverse_grasp_obman.py:243-253) and
 * gen() out of the decoder's [B,55] output (DVQ-VAE/network/gen_net.py:114-118) -- and a NULL global_orient for an
 * all-zero one (gen() passes zeros there). */
int dvq_mano_lbs_rows(const float* global_orient, int64_t ld_go, const float* hand_pose, int64_t ld_hp,
                      const float* transl, int64_t ld_tr, int B, const DvqManoConsts* m, float* verts_out,
                      float* joints_out, float* full_pose_out, float* v_posed_out, void* stream);

/* Vector-Jacobian product of dvq_mano_lbs's vertices: what autograd computes for the reference's
 * MANO layer when a loss on the vertices is back-propagated to the pose parameters, as the
 * test-time adaptation does (reference: DVQ-VAE/gen_HDMO_TTA.py:136-151).  Same inputs as the
 * forward call plus its v_posed output and grad_verts = dL/dvertices [B,V,3].
 * Outputs (each nullable): dL/dglobal_orient [B,3], dL/dhand_pose [B,ncomps], dL/dtransl [B,3].
 * The shape coefficients have no gradient (the forward ignores them). */
int dvq_mano_lbs_backward(const float* global_orient, const float* hand_pose, int B, const DvqManoConsts* m,
                          const float* v_posed, const float* grad_verts, float* grad_global_orient,
                          float* grad_hand_pose, float* grad_transl, void* stream);

/* ------------------------------------------------- test-time adaptation: contact / penetration
 * Loss and its gradient w.r.t. the hand vertices for the refinement loop of
 * DVQ-VAE/gen_HDMO_TTA.py:141-151, per grasp (the reference's batch of one, row by row):
 *   nn(i)      nearest hand vertex of object point i, squared distance d_i   (utils_loss.py:7-24)
 *   interior_i (v_nn(i) - o_i) . normal(v_nn(i)) > 0                          (utils_loss.py:27-45)
 *   penetr  = penetr_scale * sum_{interior} d_i                               (loss.py:120)
 *   contact = contact_scale * sum_{d_i < contact_dist_sq} d'_i / #{d_i < contact_dist_sq},
 *             d'_i = squared distance to the nearest vertex of the contact set (loss.py:93-101,126)
 *   loss    = w_contact * contact + w_penetr * penetr                         (gen_HDMO_TTA.py:150)
 * Vertex normals are the sums of the incident faces' area-weighted normals (loss.py:116-117; only
 * their direction is used).  A grasp with no point closer than the threshold has contact = 0.
 * The nearest-neighbour masks are constants of the gradient, as in the reference.
 *
 * obj: coordinate c of point i of grasp b at obj[b*obj_batch_stride + c*obj_chan_stride + i*obj_point_stride]
 *      ([B,4,N] clouds: 4N, N, 1).  verts: [B,V,3].
 * grad_verts: [B,V,3] = dloss/dverts.  losses: [B,2] = (penetr, contact).  nn_idx: int [B,N] or NULL.
 */
typedef struct {
    int n_verts;
    int n_faces;
    int n_contact;                     /* size of the contact set (>= 1)                                  */
    const int* faces;                  /* [F,3]                                                           */
    const int* vert_face_ptr;          /* [V+1] offsets into vert_face                                    */
    const int* vert_face;              /* faces incident to each vertex, grouped by vertex                */
    const int* search_order;           /* [V] all vertex ids: the n_contact contact-set vertices, ascending, */
                                       /* then the others, ascending; equal distances go to the lowest id    */
} DvqHandMesh;

int dvq_tta_contact_grad(const float* obj, int64_t obj_batch_stride, int64_t obj_chan_stride, int64_t obj_point_stride,
                         int n_points, const float* verts, int B, const DvqHandMesh* mesh, float contact_dist_sq,
                         float penetr_scale, float contact_scale, float w_penetr, float w_contact,
                         float* grad_verts, float* losses, int* nn_idx, void* stream);

/* ------------------------------------------------------------ prior (PixelCNN)
 * Matrix products of the GatedPixelCNN prior evaluated cell by cell (reference:
 * DVQ-VAE/network/pixelcnn/models.py:65-88 gated layer, :161-174 forward, :176-198 generate),
 * with the gate and residual fused.  `dtype` DVQ_BF16: bf16 operands / fp32 accumulation (throughput configuration);
 * DVQ_F32: exact fp32 on the 32x32x2 matrix instruction (the reference's arithmetic; parity configuration).
 * Activations are kept between calls in "packed" form of that dtype (dvq_prior_packed_bytes(rows, features, dtype)
 * bytes; layout in csrc/prior_gemm.hip), weights are packed once by dvq_prior_pack_weight; every buffer a call
 * touches must have been packed for the same dtype.
 *
 * dvq_prior_gemm runs 1..3 independent jobs of the same shape in one launch.  A job computes
 *   y[m, :] = bias + sum over taps  W_tap * x_tap[m, :]        (N outputs, M rows)
 * and finishes with one of
 *   DVQ_PRIOR_PACK  out_packed = y (+ resid) (relu if `relu`)                      [N features]
 *   DVQ_PRIOR_GATE  out_packed = tanh(u[:N/2]) * sigmoid(u[N/2:]), u = y + cond[label[m]]
 *                   (models.py:25-27, 70-75); out_pre (optional) = y              [N/2], [N]
 *   DVQ_PRIOR_ROWS  out_rows[m, n] = y (relu if `relu`)   (fp32 row-major: the logits, or the output of an MLP head)
 */
#define DVQ_PRIOR_PACK 0
#define DVQ_PRIOR_GATE 1
#define DVQ_PRIOR_ROWS 2
#define DVQ_PRIOR_MAX_TAPS 10

typedef struct {
    const void* x;      /* packed activations, 32*k_blocks features */
    const void* w;      /* packed weights [N, 32*k_blocks]           */
    int k_blocks;
} DvqPriorTap;

typedef struct {
    DvqPriorTap taps[DVQ_PRIOR_MAX_TAPS];
    int n_taps;
    const float* bias;      /* [N]                                         */
    const float* cond;      /* [n_classes, N] class embedding (GATE)       */
    const int64_t* label;   /* [M] class of each row (GATE)                */
    const void* resid;      /* packed [N features] added before relu (PACK) or NULL */
    void* out_packed;
    void* out_pre;
    float* out_rows;
    const float* pre_add;   /* GATE, optional: [n_classes, N] added to y by label BEFORE the gate (part of out_pre too) */
} DvqPriorJob;

size_t dvq_prior_packed_bytes(int rows, int features, int dtype);
/* W element (n,k) at W[n*ldn + k*ldk] (a kernel tap of a conv weight is a strided view) */
int dvq_prior_pack_weight(const float* W, int64_t ldn, int64_t ldk, int N, int K, int dtype, void* out, void* stream);
/* out(packed)[m] = table[idx[m*idx_stride]]: embedding rows of sampled indices, d features */
int dvq_prior_embed(const int64_t* idx, int64_t idx_stride, const float* table, int V, int M, int d, int dtype, void* out,
                    void* stream);
int dvq_prior_gemm(const DvqPriorJob* jobs, int n_jobs, int M, int N, int epilogue, int relu, int dtype, void* stream);
/* out_b[m] = table_b[idx[m]] for b = 0..n-1 (n <= 96), tables and outputs packed activations of `dtype`: tables hold V
 * rows, outputs M rows, `features` each.  For what the sampling loop computes from the class label alone
 * (models.py:176-198 with nothing sampled yet: the vertical stack above the first grid row and the whole first cell),
 * evaluated once per class and handed out to the rows of the batch.  idx is clamped to [0, V). */
typedef struct {
    const void* table;
    void* out;
    int features;
} DvqPriorGather;
int dvq_prior_gather_rows(const DvqPriorGather* list, int n, const int64_t* idx, int M, int V, int dtype, void* stream);
/* Vertical stack of one grid row of a kernel-3 layer (models.py:72-73 for the three cells of a row at once):
 *   y_j = bias + sum over input rows r and columns c with |c - j| <= 1 of  w[r][c - j + 1] * x[r][c],   j = 0, 1, 2
 * followed by the GATE epilogue of dvq_prior_gemm per cell (out_packed[j]: N/2 features; out_pre[j], optional: N).
 * Same numbers as three dvq_prior_gemm jobs; each activation and weight unit is fetched once for all three cells.
 * bf16 buffers only. */
typedef struct {
    const void* x[2][3];     /* packed activations of input row r, grid column c (32*k_blocks features) */
    const void* w[2][3];     /* packed weights [N, 32*k_blocks] of tap (row r, column offset -1, 0, +1) */
    int n_rows;              /* 1 or 2 input rows                                                       */
    int k_blocks;
    const float* bias;       /* [N]            */
    const float* cond;       /* [n_classes, N] */
    const int64_t* label;    /* [M]            */
    void* out_packed[3];
    void* out_pre[3];
    const float* pre_add[3]; /* optional, per cell: [n_classes, N] added to y_j by label before the gate (part of out_pre) */
} DvqPriorVRow;
int dvq_prior_vrow(const DvqPriorVRow* v, int M, int N, void* stream);

/* Class ordering of a batch for the sampling loop (no counterpart in the reference, which samples one grid per call): a
 * stable counting sort of the rows by label in one launch.  label: int64 [B] (values clamped to 0..n_classes-1,
 * n_classes <= 1024); u (nullable): float64 [B, n_cols], the rows' uniforms.  Writes label_sorted int64 [B], u_sorted
 * [B, n_cols] and perm int32 [B] (sorted position -> original row).  dvq_unpermute_rows puts int64 rows computed in
 * sorted order back: out[perm[r], :] = in[r, :]. */
int dvq_prior_class_order(const int64_t* label, const double* u, int64_t B, int n_cols, int n_classes,
                          int64_t* label_sorted, double* u_sorted, int* perm, void* stream);
int dvq_unpermute_rows(const int64_t* in, const int* perm, int64_t B, int n_cols, int64_t* out, void* stream);

/* The call sites' parameter vector (reference: DVQ-VAE/gen_diverse_grasp_obman.py:243-247, same in the ho3d / FHAB /
 * grab / HDMO scripts): param[:,0:10] = recon[:,0:10]; [10:13] = pos[:,0:3]; [13:58] = recon[:,10:55]; [58:61] =
 * pos[:,3:6].  recon: [B,55] with row stride ld_recon, pos: [B,6] with row stride ld_pos, param: [B,61] contiguous. */
int dvq_assemble_param(const float* recon, int64_t ld_recon, const float* pos, int64_t ld_pos, int64_t B, float* param,
                       void* stream);

/* One cell of the sampling loop (models.py:189-196): p = softmax(logits[m, :]) / sum, x = inverse CDF of p at the
 * uniform u[m*u_stride] in [0,1) (fp64 running sum; the number of cdf entries <= u, clamped to n_classes-1), written to
 * x_out[m*x_stride]; if emb_out is given, row x of `table` [n_classes, d] is also written there as packed activations
 * of `emb_dtype` (the next cell's input).  logits: fp32 [M, n_classes], n_classes <= 1024. */
int dvq_prior_sample(const float* logits, int n_classes, int M, const double* u, int64_t u_stride, int64_t* x_out,
                     int64_t x_stride, const float* table, int d, int emb_dtype, void* emb_out, void* stream);

/* The uniforms dvq_prior_sample consumes, out[i*n_cells + c] in [0,1) for object first_id + i and grid cell c: the top 53
 * bits of splitmix64 applied twice to the counter (first_id + i)*1024 + c + seed*0x9E3779B97F4A7C15 (with
 * 0x632BE59BD9B4E019 added between the two rounds).  No counterpart in the reference (which draws from torch's global
 * generator, models.py:193); a pure function of (seed, object id, cell), so any sharding of a batch draws the same grids. */
int dvq_prior_uniforms(int64_t first_id, int64_t n, int n_cells, uint64_t seed, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DVQ_HIP_H */
