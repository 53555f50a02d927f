/*
 * s2c_fused.h -- C ABI of the MI355X-first, POINT-MAJOR set-abstraction kernels
 * in libs2c_hip.so (scan2cap_amd/csrc/s2c_sa.hip).
 *
 * These entry points have no one-to-one counterpart in the reference's FFI (the
 * nine `pointnet2._ext` functions are in s2c_ops.h); they implement, for the
 * build's own Python layers, the work the reference spreads over
 * QueryAndGroup.forward (lib/pointnet2/pointnet2_utils.py:347-359),
 * SharedMLP's Conv2d/BatchNorm2d/ReLU (lib/pointnet2/pytorch_utils.py:11-120),
 * F.max_pool2d (lib/pointnet2/pointnet2_modules.py:255-257) and their autograd
 * backward, on row-major (rows x channels) matrices:
 *      rows = (scene, centre, sample), channels contiguous.
 * Same conventions as s2c_ops.h: device pointers, fp32 / int32, asynchronous on
 * `stream` (hipStream_t as void*), 0 on success, s2c_fused_last_error_string().
 * All BN kernels require C % 4 == 0 (every BN width of the model is 64..256).
 */
#ifndef S2C_FUSED_H
#define S2C_FUSED_H
#ifdef __cplusplus
extern "C" {
#endif

const char *s2c_fused_last_error_string(void);

/* X[(b,j,k), 0:3]   = (xyz[b, idx[b,j,k]] - new_xyz[b,j]) [/ radius if normalize]
 * X[(b,j,k), 3:3+C] = feats[b, idx[b,j,k], 0:C]
 * feats is point-major with arbitrary row / batch strides (in floats), so the raw
 * (B,N,3+C) input cloud can be read in place (feats = cloud + 3, row stride 3+C). */
int s2c_sa_gather_rows(int b, int n, int m, int ns, int C,
                       long long feat_row_stride, long long feat_batch_stride,
                       float radius, int normalize, const float *xyz,
                       const float *new_xyz, const float *feats, const int *idx,
                       float *X, void *stream);

/* backward of s2c_sa_gather_rows: d_feats (b,n,C) dense point-major, d_xyz
 * (b,n,3), d_new_xyz (b,m,3); any of d_feats / d_xyz may be NULL (skipped).
 * Outputs are zeroed by the callee. */
int s2c_sa_scatter_rows(int b, int n, int m, int ns, int C, float radius,
                        int normalize, const float *dX, const int *idx,
                        float *d_feats, float *d_xyz, float *d_new_xyz,
                        void *stream);

/* First layer of a set-abstraction stack in point space (replaces s2c_sa_gather_gemm on the
 * training path; QueryAndGroup + first Conv2d, pointnet2_utils.py:347-359, pytorch_utils.py:67-120):
 *   Y[(b,j,s), :] = P[b, idx[b,j,s], :] + W[:, 0:3] ((xyz[b,idx] - new_xyz[b,j]) (/ radius))
 * with P (b, n, N) = feats W[:, 3:]^T computed per POINT by s2c_rows_gemm (NULL: no features).
 * N % 4 == 0, N <= 256; W (N x ldw) row-major, its first three columns are used.  partial (may be
 * NULL): s2c_sa_gather_add_blocks(rows) x 2N column sums / sums of squares for
 * s2c_bn_finalize_partials. */
int s2c_sa_gather_add_blocks(long long rows);
int s2c_sa_gather_add(int b, int n, int m, int ns, int N, float radius, int normalize,
                      const float *xyz, const float *new_xyz, const float *P, const int *idx,
                      const float *W, int ldw, float *Y, float *partial, void *stream);
/* Inference: the same pass with the frozen BatchNorm (+ ReLU) behind the layer applied before the store:
 * Y = relu?(y * sc[c] + sh[c]), sc = gamma / sqrt(var + eps), sh = beta - mean * sc (gamma / beta may be
 * NULL: 1 / 0) -- the evaluation path's first set-abstraction layer in point space. */
int s2c_sa_gather_add_eval(int b, int n, int m, int ns, int N, float radius, int normalize,
                           const float *xyz, const float *new_xyz, const float *P, const int *idx,
                           const float *W, int ldw, const float *gamma, const float *beta,
                           const float *mean, const float *var, float eps, int relu, float *Y,
                           void *stream);

/* P (M x N, row stride ldp) = A (M x K, row stride lda; rows at any 4-byte address) W^T (W: N x K,
 * row stride ldw) on the exact fp32 matrix instruction (an fp32 FMA chain over k): the per-point
 * product in front of s2c_sa_gather_add (csrc/s2c_pgemm.hip). */
int s2c_point_gemm(long long M, int N, int K, const float *A, long long lda, const float *W,
                   int ldw, float *P, int ldp, void *stream);
/* The same product on the streaming kernel of csrc/s2c_gemm2.hip (bf16x3 split products, LDS-DMA ring;
 * fp32-accurate like every rows GEMM) for TALL inputs (SA1 of the BASELINE workloads: 320 000 points):
 * A rows at any 4-byte address and stride.  Returns -2 when the shape is not taken (K % 4, fewer than
 * 131072 rows, N > 128): the caller runs s2c_point_gemm. */
int s2c_point_gemm_stream(long long M, int N, int K, const float *A, long long lda, const float *W,
                          int ldw, float *P, int ldp, void *stream);

/* For the weight gradient of a gather-fused layer whose inputs need no gradient:
 * Z (b,n,C) = sum of the dY rows (b*m*ns x C) that gathered each point (zeroed by the
 * callee), S (b,m,C) = sum over the ns rows of each centre.  Then
 * dW = [ (Z^T xyz - S^T new_xyz)(/r) | Z^T feats ]  -- no (rows x (3+C)) operand. */
int s2c_sa_scatter_sum(int b, int n, int m, int ns, int C, const float *dY, const int *idx,
                       float *Z, float *S, void *stream);
/* the same with dY = BatchNorm(+ReLU) backward of (dA, Y) formed on the fly (coef from
 * s2c_bn_relu_bwd_stats; the arithmetic of s2c_bn_relu_bwd): the first layer's dY is never
 * written or re-read */
int s2c_sa_scatter_sum_bn_bwd(int b, int n, int m, int ns, int C, const float *dA,
                              const float *Y, const float *scale, const float *shift,
                              const float *mean, const float *invstd, const float *coef,
                              int relu, const int *idx, float *Z, float *S, void *stream);

/* Feature propagation on point-major rows (pointnet2_modules.py:398-410):
 * out (b*n, C2+C1) = [ three_interpolate(known (b,m,C2), idx (b,n,3), weight (b,n,3)) |
 * skip (b,n,C1) with the given row / batch strides in floats ].  skip may be NULL with
 * C1 == 0. */
int s2c_fp_interp_rows(int b, int n, int m, int C2, int C1, const float *known,
                       const int *idx, const float *weight, const float *skip,
                       long long skip_row_stride, long long skip_batch_stride, float *out,
                       void *stream);
/* d_known (b,m,C2) = scatter of w * dOut[:, :C2] (dOut row stride ld); zeroed by the
 * callee.  The skip gradient is dOut[:, C2:] itself. */
int s2c_fp_interp_rows_grad(int b, int n, int m, int C2, int ld, const float *dOut,
                            const int *idx, const float *weight, float *d_known,
                            void *stream);

/* number of row slabs (partial-sum blocks) the statistics kernels use for M rows;
 * `partial` buffers must hold s2c_bn_stat_blocks(M) * 2 * C floats. */
int s2c_bn_stat_blocks(long long M);

/* training-mode BatchNorm statistics of Y (M x C): writes scale = gamma*invstd,
 * shift = beta - mean*scale, save_mean, save_invstd and updates running_mean /
 * running_var (momentum, unbiased variance) like torch.nn.BatchNorm in train(). */
int s2c_bn_train_stats(long long M, int C, const float *Y, float *partial,
                       float eps, float momentum, const float *gamma,
                       const float *beta, float *running_mean, float *running_var,
                       float *scale, float *shift, float *save_mean,
                       float *save_invstd, long long *num_batches_tracked /* += 1, may be NULL */,
                       void *stream);

/* eval-mode coefficients from the running statistics */
int s2c_bn_eval_coeffs(int C, float eps, const float *gamma, const float *beta,
                       const float *running_mean, const float *running_var,
                       float *scale, float *shift, float *save_mean,
                       float *save_invstd, void *stream);

/* A = Y*scale + shift, followed by ReLU when relu != 0 */
int s2c_bn_relu(long long M, int C, const float *Y, const float *scale,
                const float *shift, float *A, int relu, void *stream);

/* out[j,c] = max_k relu(Y[(j,k),c]*scale + shift); arg = first maximising k;
 * ymax (optional) = Y at that k (raw, pre-BN) for the backward statistics */
int s2c_bn_relu_max(long long J, int ns, int C, const float *Y, const float *scale,
                    const float *shift, float *out, int *arg, float *ymax,
                    void *stream);

/* backward of BN(+ReLU) given dA (M x C); coef: 3*C floats scratch;
 * frozen != 0: statistics are constants (eval mode). */
int s2c_bn_relu_bwd(long long M, int C, const float *dA, const float *Y,
                    const float *scale, const float *shift, const float *mean,
                    const float *invstd, const float *gamma, int relu, int frozen,
                    float *partial, float *coef, float *dgamma, float *dbeta,
                    float *dY, void *stream);

/* The same split in two: (1) the statistics half -- dgamma, dbeta and the three per-channel
 * coefficient rows `coef` (3*C floats) of the apply formula; (2) the apply half fused into
 * the operand load of the layer's input-gradient GEMM (csrc/s2c_gemm.hip, PRO_BNBWD):
 *   dY = bn_relu_backward(dA, Y)  (side output, bit-identical to s2c_bn_relu_bwd's dY)
 *   dX = dY Wt^T,  Wt = W^T stored (N x C) row-major with row stride ldw
 * -- one pass over (dA, Y) instead of an apply pass plus a GEMM that re-reads dY (reference:
 * autograd of Conv -> BatchNorm -> ReLU, lib/pointnet2/pytorch_utils.py:67-120).
 * s2c_bn_bwd_gemm returns -2 when the bf16x3 GEMM is switched off. */
/* ... and with the statistics half formed elsewhere: s2c_bn_bwd_gemm_next_stats leaves the column
 * sums of the PREVIOUS layer's BatchNorm backward (its upstream gradient is that GEMM's output) in
 * `npartial` (s2c_rows_gemm_blocks(M, N) rows of [s1 | s2]); s2c_bn_bwd_finalize_partials turns
 * them into coef / dgamma / dbeta, s2c_bn_relu_bwd_apply is the apply half on its own. */
int s2c_bn_bwd_gemm_next_stats(long long M, int C, int N, const float *dA, const float *Y,
                               const float *scale, const float *shift, const float *mean,
                               const float *invstd, const float *coef, int relu, const float *Wt,
                               int ldw, float *dY, float *dX, int ldx, const float *nY,
                               const float *nscale, const float *nshift, const float *nmean,
                               const float *ninvstd, int nrelu, float *npartial, void *stream);
/* the plain input-gradient GEMM dX = dY W (as s2c_rows_gemm with Wt) with the same epilogue, for
 * N > 64 (-2 otherwise): the layer in front of a pooled / BN-free layer */
int s2c_pool_bwd_input_grad_next_stats(long long M, int N, int KA, int C3, int ns, const float *A,
                                       int lda, const short *arg, const float *dk,
                                       const float *Wcat, int ldw, const float *cvec, float *dA,
                                       int ldd, const float *nY, const float *nscale,
                                       const float *nshift, const float *nmean,
                                       const float *ninvstd, int nrelu, float *npartial,
                                       const float *pscale, const float *pshift, int prelu,
                                       void *stream);
int s2c_rows_gemm_next_stats(long long M, int N, int K, const float *A, int lda, const float *W,
                             int ldw, float *Y, const float *nY, const float *nscale,
                             const float *nshift, const float *nmean, const float *ninvstd,
                             int nrelu, float *npartial, void *stream);
int s2c_bn_bwd_finalize_partials(int nblk, long long M, int C, const float *partial, int frozen,
                                 const float *gamma, const float *invstd, float *coef,
                                 float *dgamma, float *dbeta, void *stream);
int s2c_bn_relu_bwd_apply(long long M, int C, const float *dA, const float *Y, const float *scale,
                          const float *shift, const float *mean, const float *invstd,
                          const float *coef, int relu, float *dY, void *stream);
int s2c_bn_relu_bwd_stats(long long M, int C, const float *dA, const float *Y,
                          const float *scale, const float *shift, const float *mean,
                          const float *invstd, const float *gamma, int relu, int frozen,
                          float *partial, float *coef, float *dgamma, float *dbeta,
                          void *stream);
int s2c_bn_bwd_gemm(long long M, int C, int N, const float *dA, const float *Y,
                    const float *scale, const float *shift, const float *mean,
                    const float *invstd, const float *coef, int relu, const float *Wt,
                    int ldw, float *dY, float *dX, int ldx, void *stream);

/* backward of BN+ReLU+max-pool given dOut (J x C) and arg (J x C): dY (J*ns x C) */
int s2c_bn_relu_max_bwd(long long J, int ns, int C, const float *dOut,
                        const int *arg, const float *ymax, const float *Y,
                        const float *scale,
                        const float *shift, const float *mean, const float *invstd,
                        const float *gamma, int frozen, float *partial, float *coef,
                        float *dgamma, float *dbeta, float *dY, void *stream);

/* ---- hand-written fp32 MFMA GEMM for the shared-MLP layers (csrc/s2c_gemm.hip)
 * Y[M x N] = pro(A)[M x K] * W^T, W (N x K) row-major.  pro = identity when
 * pscale == NULL, else relu(A*pscale[k] + pshift[k]) (previous layer's BN+ReLU
 * fused into the tile staging).  partial != NULL: per-row-block column
 * [sum | sumsq] of Y, s2c_rows_gemm_blocks(M,N) * 2N floats, to be reduced by
 * s2c_bn_finalize_partials (BN batch statistics without another pass over Y). */
int s2c_rows_gemm_blocks(long long M, int N);
/* 1 when s2c_rows_gemm (gather = 0) / s2c_sa_gather_gemm (gather = 1, K = 3 + C) run this
 * shape on the streaming kernel of csrc/s2c_gemm2.hip (persistent waves, LDS-DMA ring; tall
 * operands with N <= 128 whose W planes leave room for the rings in LDS), 0 when on the
 * tiled kernel.  Same results contract either way.  S2C_GEMM_STREAM=0 switches it off. */
int s2c_rows_stream_supported(long long M, int N, int K, int gather);
/* Y = relu?(A * scale[k] + shift[k]) W^T (+ partials as s2c_rows_gemm) with the activated
 * operand also written to side (M x K, row stride ld_side; may be NULL): the previous layer's
 * s2c_bn_relu pass folded into this layer's streaming GEMM (pytorch_utils.py:100-120 between
 * two convs).  Returns -2 when the shape is not one the streaming kernel takes. */
int s2c_rows_gemm_side_supported(long long M, int N, int K);   /* 1: the call below takes it (relu = 1) */
int s2c_rows_gemm_bn_relu_side(long long M, int N, int K, const float *A, int lda,
                               const float *scale, const float *shift, int relu, float *side,
                               int ld_side, const float *W, int ldw, float *Y, int ldy,
                               float *partial, void *stream);
/* Pooled last layer of a training stack on the streaming kernel: the products and statistics
 * partials of s2c_rows_gemm_bn_relu_side (scale == NULL: plain operand), and per centre
 * (pool_ns = 16 / 32 / 64 consecutive rows) and column the extremum of Y that BatchNorm + ReLU +
 * max-pool (pointnet2_modules.py:255-257) will select -- the maximum where gamma[col] >= 0 (or
 * gamma == NULL), the minimum otherwise: relu(y * gamma * invstd + shift) is monotone in y -- with
 * its first row index (ext, aext: J x N, J = M / pool_ns): all the pooled layer needs once the
 * statistics are known (s2c_pool_select).  Y may be NULL (not written).  -2: shape not taken. */
int s2c_rows_gemm_pool_raw(long long M, int N, int K, const float *A, int lda, const float *scale,
                           const float *shift, int relu, float *side, int ld_side,
                           const float *W, int ldw, int pool_ns, const float *gamma, float *ext,
                           int *aext, float *Y, int ldy, float *partial, void *stream);
/* out (J x C) = relu(ext * scale + shift): with ext (= ymax) and aext (= arg) of the call above
 * the three outputs of s2c_bn_relu_max */
int s2c_pool_select(long long J, int C, const float *ext, const float *scale, const float *shift,
                    float *out, void *stream);
/* Backward of a max-pooled BatchNorm + ReLU layer WITHOUT its (M x C3) tensors Y3 / dY3
 * (DESIGN 4.3): with Y3 = A W3^T, dY3 = dkrow - g (.) Y3 + e per channel, hence
 *   dA  = dkrow W3 - A (W3^T diag(g) W3) + e W3           (s2c_pool_bwd_input_grad)
 *   dW3 = SP - diag(g) W3 (A^T A) + e (x) colsum(A)        (SP, colsum: s2c_pool_bwd_sp, one pass over A;
 *                                                            partial = blocks x (C3 K + K) floats)
 * with dk = k0 * routed gradient (s2c_pool_bwd_dk), coef from s2c_bn_relu_max_bwd_stats. */
int s2c_bn_relu_max_bwd_stats(long long J, int ns, int C, const float *dOut, const float *ymax,
                              const float *scale, const float *shift, const float *mean,
                              const float *invstd, const float *gamma, int frozen,
                              float *partial, float *coef, float *dgamma, float *dbeta,
                              void *stream);
/* dk and (arg16 != NULL) the arg-max rows as int16 for s2c_pool_bwd_input_grad */
int s2c_pool_bwd_dk(long long J, int C, const float *dOut, const float *ymax, const float *scale,
                    const float *shift, const float *coef, const int *arg, float *dk,
                    short *arg16, void *stream);
int s2c_pool_bwd_sp_blocks(long long J);
/* the small matrices: Wcat (K x (K+C3)) = [-G^T | W^T], cvec (K), ge (2 C3) = g | e from coef;
 * dW (C3 x K) from the SP / colsum partials summed over the workgroups (C3 K + K floats) and
 * Gram = A^T A (K x K) */
int s2c_pool_bwd_prep(int C3, int K, const float *coef, const float *mean, const float *invstd,
                      const float *W, float *Wcat, float *cvec, float *ge, void *stream);
int s2c_pool_bwd_final(int C3, int K, const float *partial_sum, const float *gram,
                       const float *W, const float *coef, const float *mean, const float *invstd,
                       float *dW, void *stream);
/* pscale / pshift (both or NULL) in the three calls below: A is the PREVIOUS layer's pre-activation and the
 * layer's input relu?(A pscale + pshift) is formed on the way (the forward kept no copy of it);
 * s2c_pool_bwd_input_grad*: KA <= 64 and ns >= 32 then (-2 otherwise). */
int s2c_pool_bwd_sp(long long J, int ns, int C3, int K, const float *A, const int *arg,
                    const float *dk, float *partial, const float *pscale, const float *pshift,
                    int prelu, void *stream);
int s2c_pool_bwd_supported(long long M, int N, int KA, int C3);   /* 1: the call below takes it */
int s2c_pool_bwd_input_grad(long long M, int N, int KA, int C3, int ns, const float *A, int lda,
                            const short *arg16, const float *dk, const float *Wcat, int ldw,
                            const float *cvec, float *dA, int ldd, const float *pscale,
                            const float *pshift, int prelu, void *stream);
/* workgroups of the streaming kernel's persistent grid (default 240): leave out the CUs held
 * by kernels that run beside it on other streams (one FPS workgroup per scene).  Returns the
 * previous value; workgroups <= 0 only queries. */
int s2c_gemm_set_stream_grid(int workgroups);
/* switch the streaming kernel on / off at run time; returns the previous setting */
int s2c_gemm_set_stream(int on);
/* Inference layers (frozen BatchNorm): out = [max over groups of pool_ns rows of]
 * relu?(Y * scale + shift), Y = A W^T, scale = gamma / sqrt(var + eps), shift = beta -
 * mean * scale -- BN, ReLU and the set-abstraction max-pool (pointnet2_modules.py:
 * 251-257) in the GEMM epilogue.  pool_ns in {0, 16, 32, 64}; out is (M x N) or
 * (M/pool_ns x N) with row stride ldo.  gamma / beta may be NULL. */
int s2c_rows_gemm_bn_eval(long long M, int N, int K, const float *A, int lda, const float *W,
                          int ldw, const float *gamma, const float *beta, const float *mean,
                          const float *var, float eps, int relu, int pool_ns, float *out,
                          int ldo, void *stream);
int s2c_sa_gather_gemm_bn_eval(int b, int n, int m, int ns, int C, long long feat_row_stride,
                               long long feat_batch_stride, float radius, int normalize,
                               const float *xyz, const float *new_xyz, const float *feats,
                               const int *idx, int N, const float *W, int ldw,
                               const float *gamma, const float *beta, const float *mean,
                               const float *var, float eps, int relu, int pool_ns, float *out,
                               int ldo, void *stream);

/* A WHOLE set-abstraction stage of the inference path in one launch (csrc/s2c_sa_fused.hip):
 * ball-query rows -> gather -> 3 x [1x1 conv, frozen BatchNorm, ReLU] -> max over the ns rows
 * of a centre (pointnet2_modules.py:226-257); out (b*m x layers[2].N), row stride ldo.  Nothing
 * between the cloud and the pooled features is written.  Applies when the three weight
 * matrices fit LDS as bf16x3 planes: C <= 13 feature channels, widths <= 64 / 64 / 128, ns in
 * {16, 32, 64}, m*ns % 32 == 0 (s2c_sa_fused_eval_supported); returns -2 otherwise (caller
 * runs the per-layer entry points above). */
typedef struct s2c_eval_layer {
  const float *W;           /* (N x K) row-major, row stride ldw */
  const float *gamma, *beta, *mean, *var;   /* gamma / beta may be NULL */
  int N, ldw;
  float eps;
} s2c_eval_layer;
int s2c_sa_fused_eval_supported(int ns, int C, int N1, int N2, int N3);
int s2c_sa_fused_eval(int b, int n, int m, int ns, int C, long long feat_row_stride,
                      long long feat_batch_stride, float radius, int normalize,
                      const float *xyz, const float *new_xyz, const float *feats, const int *idx,
                      const s2c_eval_layer *layers, float *out, int ldo, void *stream);

/* products of s2c_rows_gemm / s2c_sa_gather_gemm: 1 = bf16x3 split on the bf16 matrix
 * pipe (fp32-accurate, ~1e-7 relative; default), 0 = exact fp32 MFMA chain.  Returns
 * the previous setting. */
int s2c_gemm_set_split(int on);

/* problems with N > 64 that the streaming kernel does not take: 1 (default)
 * = rows_gemm_c64_kernel (K in 64-chunks of fp32 in LDS, next chunk in flight, two
 * workgroups per CU), 0 = the 32-k-slice kernel (which keeps the BatchNorm-backward prologue of
 * s2c_bn_bwd_gemm either way).  Identical results.  Returns the previous setting. */
int s2c_gemm_set_c64(int on);
/* 1 (default): that kernel on 128 x 32 workgroup tiles when 128 x 128 ones would
 * cover the chip once or less (bit-identical values; not for the pooled epilogues); returns the
 * previous setting */
int s2c_gemm_set_c64_narrow(int on);
int s2c_rows_gemm(long long M, int N, int K, const float *A, int lda, const float *W,
                  int ldw, const float *pscale, const float *pshift, float *Y,
                  int ldy, float *partial, void *stream);
/* first SA layer with the grouping fused into the operand load:
 * Y[(b,j,s),:] = [ (xyz[b,idx]-new_xyz[b,j]) (/radius) | feats[b,idx,:] ] W^T,
 * K = 3 + C; same arguments as s2c_sa_gather_rows, same partials as s2c_rows_gemm. */
int s2c_sa_gather_gemm(int b, int n, int m, int ns, int C, long long feat_row_stride,
                       long long feat_batch_stride, float radius, int normalize,
                       const float *xyz, const float *new_xyz, const float *feats,
                       const int *idx, int N, const float *W, int ldw, float *Y,
                       int ldy, float *partial, void *stream);
int s2c_bn_finalize_partials(int nblk, long long M, int C, const float *partial,
                             float eps, float momentum, const float *gamma,
                             const float *beta, float *running_mean,
                             float *running_var, float *scale, float *shift,
                             float *save_mean, float *save_invstd,
                             long long *num_batches_tracked, void *stream);
/* ... and Wt (Cin x C, contiguous) = the transpose of the layer's weight W (C x Cin, row stride ldw) out
 * of the same launch: the operand of the backward's input-gradient GEMMs of the tall layers (a copy
 * kernel per layer and step otherwise). */
int s2c_bn_finalize_partials_wt(int nblk, long long M, int C, const float *partial, float eps,
                                float momentum, const float *gamma, const float *beta,
                                float *running_mean, float *running_var, float *scale, float *shift,
                                float *save_mean, float *save_invstd, long long *num_batches_tracked,
                                const float *W, int ldw, int Cin, float *Wt, void *stream);

/* ---- teacher-forced top-down caption decoder (csrc/s2c_decoder.hip) -------
 * Small-batch (R <= a few dozen rows) building blocks of one recurrent step of
 * TopDownSceneCaptionModule._step (models/caption_module.py:250-292) and of its
 * back-propagation through time.  All matrices row-major fp32; feature counts
 * and leading dimensions multiples of 4. */

/* out[r,o] = epi( W[o,:I] . x[r,:I] + bias[o] + add1[r,o] + add2[r,o] );
 * epi: 0 none, 1 ReLU, 2 multiply by (gate[r,o] > 0).  Null pointers skip a term. */
int s2c_small_linear(int R, int O, int I, const float *W, int ldw, const float *x,
                     int ldx, const float *bias, const float *add1, int ld1,
                     const float *add2, int ld2, const float *gate, int ldg, int epi,
                     float *out, int ldo, void *stream);

/* the same product as a descriptor */
typedef struct s2c_lin_desc {
  const float *W, *x, *bias, *add1, *add2, *gate;
  float *out; /* may be NULL for problem 1 when a GRU epilogue consumes the value */
  int O, I, ldw, ldx, ld1, ld2, ldg, ldo, epi;
} s2c_lin_desc;

/* gate part of a GRUCell backward (see s2c_gru_gates_bwd), H == O of the product */
typedef struct s2c_gru_bwd_desc {
  const float *sr, *sz, *sn, *sghn, *hprev;
  float *dgi, *dgh, *dh_direct;
} s2c_gru_bwd_desc;

/* ONE launch for two independent products p1, p2 (p2 may be NULL); g1 (may be NULL)
 * treats p1's values as dh' of a GRUCell and emits dgi, dgh, dh_direct for it.  The
 * back-propagation-through-time chain of models/caption_module.py:250-292 is 6 such
 * launches per step. */
int s2c_small_linear_pair(int R, const s2c_lin_desc *p1, const s2c_lin_desc *p2,
                          const s2c_gru_bwd_desc *g1, void *stream);

/* torch.nn.GRUCell forward; saves r, z, n and gh_n (R x H each) for backward */
int s2c_gru_fwd(int R, int H, int I, const float *Wih, const float *Whh,
                const float *bih, const float *bhh, const float *x, int ldx,
                const float *h, float *hnew, float *sr, float *sz, float *sn,
                float *sghn, void *stream);

/* gate part of GRUCell backward: dh' = dh1 (+ dh2) -> dgi, dgh (R x 3H),
 * dh_direct = dh' * z */
int s2c_gru_gates_bwd(int R, int H, const float *dh1, const float *dh2,
                      const float *sr, const float *sz, const float *sn,
                      const float *sghn, const float *hprev, float *dgi, float *dgh,
                      float *dh_direct, void *stream);

/* additive attention: scores -> masked softmax alpha (R x K) -> att (R x F) */
int s2c_attn_fwd(int R, int K, int H, int F, const float *M, const float *q, int ldq,
                 const float *wa, const float *mask, const float *O, float *scores,
                 float *alpha, float *att, int lda, void *stream);

/* Few keys (K <= 32: the num_locals gather): the attention above AND the layer that consumes it
 * in one launch: x2 (R x E) = relu(Wl[:, 0:F] att + bias + add), Wl (E x ldw) row-major; alpha and
 * att are stored for the backward pass.  (caption_module.py:274-287) */
int s2c_attn_x2_fwd(int R, int K, int H, int F, int E, const float *M, const float *q, int ldq,
                    const float *wa, const float *mask, const float *O, const float *Wl, int ldw,
                    const float *bias, const float *add, int ld_add, float *alpha, float *att,
                    int lda, float *x2, int ldx2, void *stream);
/* The decoder's whole forward recurrence (T steps of caption_module.py:250-292) as ONE persistent
 * kernel: 256 co-resident workgroups exchange the five per-step vectors as tagged 8-byte values
 * (csrc/s2c_decoder_persist.hip) instead of meeting at 5 T launch boundaries.  Same operands and
 * saved tensors as the s2c_small_linear / s2c_gru_fwd / s2c_attn_x2_fwd chain; R <= 8 rows,
 * K <= 32 keys, H, E <= 512, F <= 256 (all multiples of 4), T <= 62.
 *   xbuf    s2c_decoder_fwd_persist_xbuf_pairs(H, E) 8-byte words, zeroed ONCE at allocation
 *   nonce   one uint32 (zeroed once), advanced by every launch;  started, fail: one uint32, zero
 * Returns -2 (nothing launched) when s2c_decoder_fwd_persist_supported says no: shapes outside the
 * limits, S2C_DECODER_PERSIST=0, or a device on which the grid would not be co-resident. */
typedef struct s2c_dec_fwd_args {
  int R, K, H, E, F, T, ldtd, ldlang;
  int backoff, pad_;              /* x 64 cycles between two polls of the same data (0: measured best) */
  const float *W_td_h2;           /* (E, H) column block of map_topdown, row stride ldtd */
  const float *Pw, *Ptf;          /* (R, T, E) word projections, (R, E) target projection + bias */
  const float *W_ih1, *W_hh1, *b_ih1, *b_hh1;
  const float *Wqh;               /* (H + E, H): [map_hidd ; map_lang[:, F:]] */
  const float *M, *wa, *mask, *O; /* (R, K, H), (H), (R, K), (R, K, F) */
  const float *W_lang, *b_lang;   /* (E, ldlang): the first F columns are used */
  const float *W_ih2, *W_hh2, *b_ih2, *b_hh2;
  float *H1, *H2;                 /* (T + 1, R, H), slice 0 = initial state (zeros) */
  float *X1, *X2;                 /* (T, R, E) */
  float *S;                       /* (2 cells, 4, T, R, H): r, z, n, gh_n of GRU cell 1, then of cell 2 */
  float *C;                       /* NULL or (2, 4, T, R, H): cr, cz, cn, cnr with d(gi) = dh' [cr | cz | cn],
                                     d(gh) = dh' [cr | cz | cnr] -- what s2c_decoder_bwd_persist reads */
  float *QL;                      /* (T, R, H + E) */
  float *ALPHA, *ATT;             /* (T, R, K), (T, R, F) */
  unsigned long long *xbuf;
  unsigned long long *prof;       /* NULL, or 8 x T x 16 words: phase stamps of workgroup 0 */
  unsigned int *nonce, *started, *fail;   /* fail: raised (1) when a poll gave up -- results invalid (and
                                             H2[T, 0, 0] is set to NaN so that the loss shows it) */
} s2c_dec_fwd_args;
int s2c_decoder_fwd_persist(const s2c_dec_fwd_args *a, void *stream);
int s2c_decoder_fwd_persist_supported(int R, int K, int H, int E, int F, int T);
long long s2c_decoder_fwd_persist_xbuf_pairs(int H, int E);
void s2c_decoder_persist_set(int on);   /* 0: always refuse (the launch chain runs) */
long long s2c_decoder_persist_args_sizeof(int which);   /* 0: s2c_dec_fwd_args, 1: s2c_dec_bwd_args */

/* ... and its back-propagation through time (the 5-launches-per-step chain of decoder_fused.py's
 * backward) as one persistent kernel.  Needs the coefficient arrays C1 / C2 written by
 * s2c_decoder_fwd_persist, and two products formed before the loop:
 *   P (R, K, E) = O W_lang[:, :F]^T,   Latt (T, R, E) = ATT W_lang[:, :F]^T
 * (the attention backward needs datt = W_lang[:, :F]^T da2 only inside <datt, O_k> = <da2, P_k> and
 * <datt, att_t> = <da2, Latt_t>).  Transposed weights as s2c_batch_prep leaves them; WT_hl (H, H + E) =
 * [W_h^T | W_lang[:, F:]^T].  Outputs: DA1 (T, R, E), DQA (T, R, H + E) = [dq | da2], DG = DGI1, DGH1,
 * DGI2, DGH2 (4, T, R, 3H), dM (R, K, H) and dwa_rows (R, H) -- the last two written once, complete.
 * xbuf: s2c_decoder_bwd_persist_xbuf_pairs(H, E) 8-byte words zeroed once; nonce / started / fail as
 * in the forward kernel (its own three words). */
typedef struct s2c_dec_bwd_args {
  int R, K, H, E, T, pad_;
  const float *dH2;                 /* (T, R, H) gradient of the classifier input */
  const float *C, *S;               /* (2, 4, T, R, H) each, as s2c_dec_fwd_args left them */
  const float *X1, *X2;             /* (T, R, E) */
  const float *QL, *ALPHA, *M, *wa; /* (T, R, H + E), (T, R, K), (R, K, H), (H) */
  const float *P, *Latt;            /* (R, K, E), (T, R, E) */
  const float *WT_ih2, *WT_hh2, *WT_hl, *WT_ih1, *WT_hh1, *WT_td;
  float *DA1, *DQA, *DG, *dM, *dwa_rows;   /* DG (4, T, R, 3H): DGI1, DGH1, DGI2, DGH2 */
  unsigned long long *xbuf;
  unsigned long long *prof;         /* NULL, or 8 x T x 16 words: phase stamps of workgroup 0 */
  unsigned int *nonce, *started, *fail;
} s2c_dec_bwd_args;
int s2c_decoder_bwd_persist(const s2c_dec_bwd_args *a, void *stream);
int s2c_decoder_bwd_persist_supported(int R, int K, int H, int E, int T);
long long s2c_decoder_bwd_persist_xbuf_pairs(int H, int E);

/* backward mirror for K <= 32 keys: datt = W_lang^T[:F] da2 formed inside the attention backward
 * (one launch instead of s2c_small_linear_pair + s2c_attn_bwd); dM / dwa_rows accumulate, dq (row
 * stride lddq) is overwritten.  F a power of two in 32..256, E <= 512. */
int s2c_attn_bwd_x2(int R, int K, int H, int F, int E, const float *da2, int ldda,
                    const float *WlT, int ldw, const float *att, int lda, const float *alpha,
                    const float *O, const float *M, const float *q, int ldq, const float *wa,
                    float *dM, float *dq, int lddq, float *dwa_rows, void *stream);

/* its backward from datt (R x F) and the saved forward output att (R x F):
 * dM (R x K x H) and dwa_rows (R x H; dwa = its sum over rows) ACCUMULATE (caller
 * zeroes them once); dq (R x H) is overwritten.  No atomics: deterministic.
 * dO = sum_t alpha_t (x) datt_t has no recurrence: one batched GEMM in the caller. */
int s2c_attn_bwd(int R, int K, int H, int F, const float *datt, int ldd,
                 const float *att, int lda, const float *alpha, const float *O,
                 const float *M, const float *q, int ldq, const float *wa, float *dM,
                 float *dq, float *dwa_rows, void *stream);


/* local attention of the greedy decode (caption_module.py:502-592 with num_locals):
 * mapped (R,L,H) = map_feat of the L gathered objects of each row, q (R,H) = map_hidd(h1),
 * wa (H) / ba = the `attend` layer, valid (R,L) 0/1 or NULL, feats (R,L,F):
 * alpha (R,L) = softmax_l(wa . tanh(mapped + q) + ba), att (R,F) = sum_l alpha feats.
 * L <= 32.  Forward only (evaluation). */
int s2c_attn_local_fwd(int R, int L, int H, int F, const float *mapped, const float *q,
                       int ldq, const float *wa, float ba, const float *valid,
                       const float *feats, float *alpha, float *att, int lda, void *stream);
/* the same, additionally (att may be NULL) writing att as bf16x3 planes (3 x R x ldp, plane stride
 * pstride elements; tiled: the TILED layout, rows allocated up to a multiple of 32): the operand
 * format of s2c_planes_gemm below */
int s2c_attn_local_fwd_planes(int R, int L, int H, int F, const float *mapped, const float *q,
                              int ldq, const float *wa, float ba, const float *valid,
                              const float *feats, float *alpha, float *att, int lda,
                              unsigned short *planes, long long pstride, int ldp, int tiled,
                              void *stream);

/* Scene-shared attention of the greedy decoder with num_locals = -1 (the reference's default command line;
 * caption_module.py:270-285 with valid_prop_masks = object_masks, :536): R = B * rows_per_scene query rows,
 * row r of scene b attends over that scene's K keys.  M (B K, H) = map_feat of the scene's objects,
 * valid (B, K) 0/1 or NULL, O (B K, F), q (R, H; row stride ldq), wa (H), ba:
 * alpha (R, K) = softmax_j(wa . tanh(M[b, j] + q[r]) + ba), att (R, F; may be NULL) = sum_j alpha O,
 * optionally also as bf16x3 planes (as s2c_attn_local_fwd_planes).  K <= 512.  Forward only. */
int s2c_attn_scene_fwd(int R, int rows_per_scene, int K, int H, int F, const float *M,
                       const float *valid, const float *O, const float *q, int ldq, const float *wa,
                       float ba, float *alpha, float *att, int lda, unsigned short *planes,
                       long long pstride, int ldp, int tiled, void *stream);

/* ---------------------------------------------------------------------------
 * Detection loss of get_scene_cap_loss (lib/loss_helper.py:24-187, :381-491;
 * utils/nn_distance.py:13-59) as 2 forward + 1 backward launches.
 * All arrays contiguous; labels int64 as lib/dataset.py produces them; seed_inds
 * int32.  B scenes, S seeds x VF votes, N points, K proposals, G (<=256) padded GT
 * boxes, NH heading bins, NS size clusters, NC classes (each <= 64), K <= 1024. */
typedef struct s2c_detloss_args {
  int B, S, VF, N, K, G, NH, NS, NC, ld_center_label;
  int ld_scores;   /* > 0: objectness / heading / size / semantic score arrays (and their
                      gradients, and d center) are columns of one (B,K,ld_scores) matrix */
  float near_threshold, far_threshold, obj_w0, obj_w1;
  const float *seed_xyz;              /* (B,S,3) */
  const float *vote_xyz;              /* (B,S*VF,3) */
  const int *seed_inds;               /* (B,S) */
  const float *vote_label;            /* (B,N,9) */
  const long long *vote_label_mask;   /* (B,N) */
  const float *agg_xyz;               /* (B,K,3) aggregated_vote_xyz */
  const float *center_label;          /* (B,G,ld_center_label), first 3 used */
  const float *objectness_scores;     /* (B,K,2) */
  const float *center;                /* (B,K,3) */
  const float *box_label_mask;        /* (B,G) */
  const long long *heading_class_label; /* (B,G) */
  const float *heading_residual_label;  /* (B,G) */
  const long long *size_class_label;    /* (B,G) */
  const float *size_residual_label;     /* (B,G,3) */
  const long long *sem_cls_label;       /* (B,G) */
  const float *heading_scores;        /* (B,K,NH) */
  const float *heading_res_norm;      /* (B,K,NH) */
  const float *size_scores;           /* (B,K,NS) */
  const float *size_res_norm;         /* (B,K,NS,3) */
  const float *sem_cls_scores;        /* (B,K,NC) */
  const float *mean_size_arr;         /* (NS,3) */
  /* outputs of the forward (saved for the backward) */
  long long *objectness_label;        /* (B,K) */
  float *objectness_mask;             /* (B,K) */
  long long *object_assignment;       /* (B,K) */
  int *vote_arg;                      /* (B,S) */
  int *center_g1;                     /* (B,K) */
  int *center_k2;                     /* (B,G) */
  float *partial;                     /* (B, s2c_detection_loss_partial_floats()) */
  float *stats;                       /* 20 floats: [0..8] vote, objectness, center,
    heading_cls, heading_reg, size_cls, size_reg, sem_cls, box; [9] = 10*(vote +
    0.5 objectness + box + 0.1 sem_cls); [10] pos_ratio; [11] neg_ratio; [12] obj_acc;
    [13..16] denominators */
} s2c_detloss_args;

typedef struct s2c_detloss_grads {    /* d stats[9] / d input, same shapes */
  float *vote_xyz, *objectness_scores, *center, *heading_scores, *heading_res_norm,
      *size_scores, *size_res_norm, *sem_cls_scores;
} s2c_detloss_grads;

int s2c_detection_loss_partial_floats(void);
int s2c_detection_loss_fwd(const s2c_detloss_args *a, void *stream);
/* gup: device pointer to the upstream gradient of stats[9] (one float) */
int s2c_detection_loss_bwd(const s2c_detloss_args *a, const s2c_detloss_grads *d,
                           const float *gup, void *stream);

/* ---------------------------------------------------------------------------
 * parse_predictions post-processing (lib/ap_helper.py:40-178), float64 like numpy.
 * counts[b,k] = number of points of scene b inside box k (cuboid rotated by `angle`
 * about the Y axis, utils/box_util.py:340-358; closed intervals).  pts: float32,
 * row stride pt_stride (>= 3) and batch stride in floats -- the (B,N,3+C) point
 * cloud is read in place.  Replaces the per-box scipy Delaunay hull test
 * (model_util_scannet.py:13-22). */
int s2c_boxes_count_points(int b, int n, int K, const float *pts, long long pt_stride,
                           long long pt_batch_stride, const double *center,
                           const double *size, const double *angle, int *counts,
                           void *stream);

/* greedy NMS per scene (utils/nms.py:13-151): boxes (b,K,6) = [x1,y1,z1,x2,y2,z2],
 * descending score, suppress IoU > thresh (old_type: inter / area_j); cls != NULL:
 * only boxes of the same class suppress each other (nms_3d_faster_samecls, which
 * also adds 1e-8 to the union: add_eps).  valid (b,K) 0/1 in, keep (b,K) 0/1 out.
 * K <= 1024. */
int s2c_nms(int b, int K, const double *boxes, const double *score, const long long *cls,
            const unsigned char *valid, double thresh, int old_type, int add_eps,
            unsigned char *keep, void *stream);

/* `_query_locals` (models/graph_module.py:182-222, models/caption_module.py:322-381)
 * for T targets per scene at once: corners (B,K,8,3) float64, object_masks (B,K) and
 * target_ids (B,T) int64 -> local_masks (B,T,K) float 0/1 and ids (B,T,L) int64,
 * ascending.  corner_mode 1: min over the target's 8 corners, 0: centre distance. */
int s2c_query_locals(int B, int K, int T, int L, const double *corners,
                     const long long *object_masks, const long long *target_ids,
                     int corner_mode, int include_self, double overlay_threshold,
                     float *local_masks, long long *ids_out, void *stream);

/* EdgeConv message passing (models/graph_module.py:74-115): edge e = (b,i,l) from row i
 * to column j = nbr[b,i,l] (int64, (B,K,L)); x (B,K,F).
 *   s2c_edge_rows:    rows (B*K*L, 2F) = [ x[b,j] | x[b,i] - x[b,j] ]
 *   s2c_edge_scatter: out (B,K,F) = sum over edges into j of msg[e]*slot[e] (zeroed by the
 *                     callee), msgm (B*K*L,F) = msg*slot;  slot (B,K,L) uint8
 * and their gradients (dx zeroed by the callee; d_msgm may be NULL). */
int s2c_edge_rows(int B, int K, int L, int F, const float *x, const long long *nbr,
                  float *rows, void *stream);
int s2c_edge_rows_grad(int B, int K, int L, int F, const float *d_rows, const long long *nbr,
                       float *dx, void *stream);
int s2c_edge_scatter(int B, int K, int L, int F, const float *msg, const long long *nbr,
                     const unsigned char *slot, float *out, float *msgm, void *stream);
int s2c_edge_scatter_grad(int B, int K, int L, int F, const float *d_out, const float *d_msgm,
                          const long long *nbr, const unsigned char *slot, float *d_msg,
                          void *stream);

/* The teacher-forced decoder's inputs in one launch (models/caption_module.py:250-292 with
 * _add_relation_feat :394-414 restricted to the rows the decoder reads): target_feats (B x F) =
 * obj[b, tgt[b]], local (B x L x F): local[b, l] = obj[b, id] + sum_t [nbr[b, tgt[b], t] == id]
 * rel[b, tgt[b], t], id = local_ids[b, l].  obj (B x K x F), rel (B x K x LR x F) or NULL, nbr
 * (B x K x LR), ids int64 (clamped to 0 .. K-1).  _grad: d_obj (B x K x F) and d_rel (B x K x LR x F,
 * or NULL) written in full (no zero fill, no atomics: deterministic); d_target may be NULL. */
int s2c_local_feats(int B, int K, int L, int LR, int F, const float *obj, const float *rel,
                    const long long *nbr, const long long *tgt, const long long *local_ids,
                    float *target_feats, float *local, void *stream);
int s2c_local_feats_grad(int B, int K, int L, int LR, int F, const float *d_target,
                         const float *d_local, const long long *nbr, const long long *tgt,
                         const long long *local_ids, float *d_obj, float *d_rel, void *stream);

/* Weight gradient of a rows x channels layer (autograd of the reference's 1x1 convolutions /
 * linears, lib/pointnet2/pytorch_utils.py:11-120): dW (Cout x Cin, row stride lddw) =
 * dY^T A with dY (M x Cout, row stride ldy) and A (M x Cin, row stride lda), fp32-accurate
 * bf16x3 MFMA products, deterministic (fixed-order) reduction over the rows.
 * workspace: s2c_weight_grad_workspace_bytes() bytes (scratch, any content);
 * counters:  s2c_weight_grad_counter_bytes() bytes that must be ZERO on entry and are
 * zero again on exit (allocate once, zero once).  Both may be NULL when the plan has a
 * single slab of rows. */
long long s2c_weight_grad_workspace_bytes(long long M, int Cout, int Cin);
long long s2c_weight_grad_counter_bytes(long long M, int Cout, int Cin);
/* counters == NULL: the kernel only writes one partial (Cout x Cin) tile per row slab into
 * `workspace` (s2c_weight_grad_slabs(M, Cout, Cin) of them; one slab: straight into dW) and the
 * caller adds them up behind a kernel boundary (s2c_multi_colsum). */
int s2c_weight_grad_slabs(long long M, int Cout, int Cin);
int s2c_weight_grad(long long M, int Cout, int Cin, const float *dY, long long ldy,
                    const float *A, long long lda, float *dW, int lddw, void *workspace,
                    void *counters, void *stream);
/* Several independent weight gradients in ONE launch (the layers of a stack).  Job j: partial tiles
 * into `part` (s2c_weight_grad_slabs(M, Cout, Cin) of them, Cout x Cin each, row stride Cin; one slab:
 * straight into dW, part may be NULL); the caller adds the slabs up (s2c_multi_colsum). */
#define S2C_DW_MAX_JOBS 16
typedef struct s2c_dw_job {
  const float *dY, *A;
  float *dW, *part;
  long long M, ldy, lda;
  int Cout, Cin, lddw, pad_;
} s2c_dw_job;
typedef struct s2c_dw_jobs {
  int n_jobs, pad_;
  s2c_dw_job job[S2C_DW_MAX_JOBS];
} s2c_dw_jobs;
int s2c_weight_grad_multi(const s2c_dw_jobs *jobs, void *stream);

/* Box bookkeeping of the proposal stage (models/proposal_module.py:80-144,
 * model_util_scannet.py:165-172, utils/box_util.py:360-383) from the head output net
 * (B,K,nout) f32 [2 objectness | 3 centre offset | NH heading scores | NH heading residuals |
 * NS size scores | NS*3 size residuals (normalised) | num_class semantic scores] and the
 * decoded centres (B,K,3) f32: bbox_corner (B,K,8,3) f64 for heading 0 (ScanNet),
 * bbox_mask / sem_cls (B,K) i64 = arg-max of the objectness / semantic scores (first
 * maximum), size_class (B,K) i64 (may be NULL). */
int s2c_proposal_decode(int B, int K, int nout, int num_heading_bin, int num_size_cluster,
                        int num_class, const float *net, const float *center,
                        const float *mean_size_f32, const double *mean_size_f64,
                        double *bbox_corner, long long *bbox_mask, long long *sem_cls,
                        long long *size_class, void *stream);

/* target_ids (B) i64 / target_ious (B) f32: the proposal with the largest axis-aligned IoU
 * (utils/box_util.py:183-209, float64) against each sample's ground-truth box
 * (models/caption_module.py:16-38); first maximum. */
int s2c_select_target(int B, int K, const double *bbox_corner, const double *ref_box_corner,
                      long long *target_ids, float *target_ious, void *stream);

/* good[b] (bool) = target_ious[b] > min_iou; *mean = mean IoU of the good samples (0 if none):
 * `good_bbox_masks` / `pred_ious` of caption_module.py:494-498 in one launch. */
int s2c_good_bbox_stats(int B, const float *target_ious, float min_iou, unsigned char *good,
                        float *mean, void *stream);

/* Caption loss (lib/loss_helper.py:189-230): masked cross-entropy (ignore_index 0, rows of
 * samples with good[b] == 0 excluded) and word accuracy of the teacher-forced logits pred
 * (B,T,V) f32 contiguous against target (B,T) i64 (row stride target_stride elements).
 * fwd: row_lse (B*T), row_stats (B*T,4) scratch, out[0] = cap_loss, out[1] = cap_acc,
 * out[2] = 1 / (sum good + 1e-6).  bwd: dpred (B,T,V) = d cap_loss / d pred * gup[0]. */
int s2c_caption_loss_fwd(int B, int T, int V, const float *pred, const long long *target,
                         long long target_stride, const unsigned char *good, float *row_lse,
                         float *row_stats, float *out, void *stream);
int s2c_caption_loss_bwd(int B, int T, int V, const float *pred, const long long *target,
                         long long target_stride, const unsigned char *good,
                         const float *row_lse, const float *fwd_out, const float *gup,
                         float *dpred, void *stream);

/* Vote head (models/voting_module.py:49-58, models/capnet.py:97-98) on rows: net (M,3+C) =
 * [xyz offset | feature residual] of the vote MLP, seed_xyz (M,3), seed_feat (M,C) with row
 * stride seed_ld ->  vote_xyz (M,3) = seed_xyz + offset,  y (M,C) = f / ||f||_2 with
 * f = seed_feat + residual,  norm (M).  bwd: d_net (M,3+C) = [g_xyz | (g_y - y <g_y,y>) / norm]
 * (g_xyz may be NULL; g_y addressed as g_y[row*gy_row_stride + c*gy_col_stride]);
 * d_seed (M,C) = d_net[:, 3:] as a dense copy (may be NULL); d seed_xyz = g_xyz. */
int s2c_vote_head_fwd(int M, int C, const float *net, const float *seed_xyz,
                      const float *seed_feat, long long seed_ld, float *vote_xyz, float *y,
                      float *norm, void *stream);
int s2c_vote_head_bwd(int M, int C, const float *g_xyz, const float *g_y,
                      long long gy_row_stride, long long gy_col_stride, const float *y,
                      const float *norm, float *d_net, float *d_seed, void *stream);

/* One launch for up to 8 transposes dst[j] (cols x rows, dense) = src[j]^T (rows x cols, row
 * stride lds[j]) and up to 8 float buffers to zero (the set-up of the decoder's backward). */
typedef struct s2c_prep_args {
  int n_transpose, n_zero;
  const float *src[8];
  float *dst[8];
  int rows[8], cols[8];
  long long lds[8];
  float *zero[8];
  long long zero_count[8];
} s2c_prep_args;
int s2c_batch_prep(const s2c_prep_args *a, void *stream);

/* Up to S2C_COLSUM_MAX_JOBS partial-sum jobs in one launch: out[j] (n[j]) = sum over the S[j]
 * slabs of part[j] (S[j] x n[j], dense), fixed order.  part[j] may be out[j] itself (S = 1).
 * sub[j] != NULL: out[j] is rows of ncol[j] elements, and the first sub_cols[j] columns of every row r
 * become (sum - sum over the sub_S[j] slabs of sub[j][s][r][c]) / sub_div[j] (sub_div 0: no division) --
 * the coordinate columns of a grouped first layer's weight gradient minus the centres' share, over the
 * radius (pointnet2_utils.py:317-376: grouped_xyz -= new_xyz; /= radius) without launches of its own. */
#define S2C_COLSUM_MAX_JOBS 32
typedef struct s2c_colsum_args {
  int n_jobs;
  int S[S2C_COLSUM_MAX_JOBS];
  long long n[S2C_COLSUM_MAX_JOBS];
  const float *part[S2C_COLSUM_MAX_JOBS];
  float *out[S2C_COLSUM_MAX_JOBS];
  const float *sub[S2C_COLSUM_MAX_JOBS];
  int sub_S[S2C_COLSUM_MAX_JOBS], ncol[S2C_COLSUM_MAX_JOBS], sub_cols[S2C_COLSUM_MAX_JOBS];
  float sub_div[S2C_COLSUM_MAX_JOBS];
} s2c_colsum_args;
int s2c_multi_colsum(const s2c_colsum_args *a, void *stream);

/* Up to 16 row-sum jobs in one launch: out[j] (C[j]) = sum over the M[j] rows of X[j]
 * (M[j] x C[j], row stride ld[j]) -- bias gradients.  Fixed summation order.
 * chunk_rows > 0: slabs of chunk_rows rows are summed separately, slab q of job j into
 * out[j] + q*C[j] (ceil(M/chunk_rows) x C floats; add them up with s2c_multi_colsum). */
typedef struct s2c_rowsum_args {
  int n_jobs;
  int chunk_rows;
  int C[16];
  long long M[16], ld[16];
  const float *X[16];
  float *out[16];
} s2c_rowsum_args;
int s2c_multi_rowsum(const s2c_rowsum_args *a, void *stream);


/* ---- fp32-accurate GEMMs on PRE-SPLIT bf16x3 planes (csrc/s2c_planes.hip) -------------------
 * The greedy caption decoder (models/caption_module.py:502-592: R = B*K rows x 29 tokens; the
 * products of `_step` :250-292, both GRU cells and the classifier :553) as hand-written MFMA
 * kernels.  A matrix X (rows x K) is held as three bf16 planes hi / mid / lo (x = hi + mid + lo),
 * row-major with row stride ld (a multiple of 32 >= K, zero beyond K), plane j at p + j * pstride;
 * Y = A W^T is formed from the six plane products with i + j <= 2 in an fp32 accumulator
 * (the arithmetic of the rows GEMMs in s2c_gemm.hip).  A = up to two K segments side by side
 * ([x | h]: nothing is concatenated); segment 0 may be gathered by a row map.
 * TILED planes (what the kernel streams fastest; W always, activations where no row map applies): the
 * matrix (rows a multiple of 32 -- allocate up to the next multiple of 128 for an A operand --, ld a
 * multiple of 16) as blocks of 32 rows x 16 columns = 1 KB ordered [row block][column block]; inside a
 * block row r32 sits at 32 r32 bytes with its two 16-byte halves swapped where (r32 >> 3) & 1 -- the image
 * one LDS-DMA instruction lands in LDS: element (r, k) at
 *   ((r >> 5) (ld >> 4) + (k >> 4)) 512 + ((r & 31) 2 + (((k >> 3) & 1) ^ (((r & 31) >> 3) & 1))) 8 + (k & 7). */
typedef struct s2c_planes_seg {
  const unsigned short *p;      /* plane 0 */
  long long pstride;            /* elements from one plane to the next */
  int ld;                       /* row stride in elements, a multiple of 8 */
  int kc;                       /* number of 32-column chunks of this segment */
  const int *rowmap;            /* NULL, or: output row r reads source row rowmap[r] */
  int rowdiv;                   /* > 0 (and rowmap NULL): output row r reads source row r / rowdiv */
  int tiled;                    /* 0: row-major planes; 1: TILED planes (below) -- no row map then */
} s2c_planes_seg;

typedef struct s2c_planes_gemm_args {
  int M, N;                     /* output rows; output columns (gru: hidden units, a multiple of 32) */
  int gru, relu, nseg;
  int dbg;                      /* 0 (timing experiments: 1 = no products, 2 = no DMA) */
  s2c_planes_seg seg[2];
  /* greedy feedback: NULL, or (M x ntokkeys) arg-max keys as the `amax` output below leaves them;
   * segment 0 then reads source row = the column of the row's largest key (first maximum) */
  const unsigned long long *tokkeys;
  int ntokkeys, ldw;
  /* W planes, TILED: ceil(N / 128) * 128 rows (gru: 4 N rows) x ldw, ldw = 32 * (seg[0].kc + seg[1].kc);
   * gru: rows 128 c .. 128 c + 127 = [r | z | n_i | n_h] of units 32 c .. 32 c + 31, the n_i rows
   * zero in segment 1's columns and the n_h rows zero in segment 0's (those products are skipped) */
  const unsigned short *W;
  long long wpstride;
  const float *bias;            /* NULL or (N); gru: (4, N) = b_ir + b_hr, b_iz + b_hz, b_in, b_hn */
  const float *add;             /* NULL or (M x N) row stride ldadd: added before the ReLU */
  float *C;                     /* NULL or fp32 output (M x N), row stride ldc */
  unsigned short *P;            /* NULL or plane output: row stride ldp (multiple of 32 >= N), zero beyond N */
  long long ppstride;
  const float *hprev;           /* gru: previous hidden state (M x N) fp32, row stride ldh;
                                   out = n + z (hprev - n) (ATen's fused GRU cell) */
  unsigned long long *amax;     /* NULL or (M x namax): per 128-column tile the key of the row's maximum:
                                   (order-preserving bits of the value) << 32 | (2^32 - 1 - column) */
  int ldadd, ldc, ldp, ldh, namax;
  int ptiled;                   /* P is written TILED */
  int big_ok;                   /* every tiled operand and W are allocated to multiples of 256 rows (W: 256
                                   rows per 64 gru units): the 256 x 256 tile kernel may be used */
  /* two outputs side by side (one pass over A for two consumers).  nsplit > 0 (a multiple of 128; not
   * gru): W rows [0, nsplit) produce C -- n1 <= nsplit valid columns, `bias`, `amax`, `P` -- and W rows
   * [nsplit, N) produce C2 (N - nsplit columns, row stride ldc2), to which `add` (not to C) is added */
  int nsplit, n1, ldc2;
  float *C2;
} s2c_planes_gemm_args;
int s2c_planes_gemm(const s2c_planes_gemm_args *a, void *stream);
void s2c_planes_set_big(int mode);   /* 256 x 256 tiles: -1 by grid size (default), 0 never, 1 whenever big_ok */
long long s2c_planes_args_sizeof(int which);   /* 0: s2c_planes_gemm_args, 1: s2c_planes_seg */
/* planes of an fp32 matrix: X (rows_in x K, row stride ldx) -> P (3 x rows_out x ldp) bf16, zero in
 * rows >= rows_in and columns >= K; ldp a multiple of 8 */
int s2c_planes_split(long long rows_in, int K, const float *X, long long ldx, long long rows_out,
                     int ldp, unsigned short *P, long long pstride, int tiled, void *stream);


/* ---- many SMALL fp32 GEMMs in one launch (csrc/s2c_mgemm.hip) ---------------------------------
 * The hoisted, recurrence-free products of the teacher-forced caption decoder
 * (models/caption_module.py:252, 275, 472 and their backward): C_j = A_j B_j (+ bias_j) (+ C_j),
 * M_j x K_j by K_j x N_j, exact fp32 FMA chains in k order.  Operands are addressed through index maps
 * (element offsets, < 2^31): ix(i, {div, hi, lo}) = div > 0 ? (i / div) hi + (i % div) lo : i lo;
 * A(m, k) = A[ix(m, am) + ix(k, ak)], B(k, n) = B[ix(k, bk) + ix(n, bn)], C(m, n) = C[ix(m, cm) + n]. */
typedef struct s2c_mgemm_axis { int div, hi, lo; } s2c_mgemm_axis;
typedef struct s2c_mgemm_job {
  const float *A, *B;
  float *C;
  const float *bias;            /* NULL or (N): added to every row */
  int M, N, K;
  s2c_mgemm_axis am, ak, bk, bn, cm;
  int accumulate;               /* 1: C += A B (+ bias) */
  int ksplit;                   /* > 1: the reduction in ksplit ranges, added into C with float atomics
                                   (C zeroed by the caller, or holding what is to be accumulated to) */
  int tile0, pad_;              /* tile0: filled in by s2c_mgemm */
} s2c_mgemm_job;
#define S2C_MGEMM_MAX_JOBS 32
typedef struct s2c_mgemm_args {
  int n_jobs, pad_;
  s2c_mgemm_job job[S2C_MGEMM_MAX_JOBS];
} s2c_mgemm_args;
/* jobs of one call must not depend on each other's outputs */
int s2c_mgemm(const s2c_mgemm_args *a, void *stream);
long long s2c_mgemm_args_sizeof(void);


/* ---- tall weight gradients as a streaming kernel (csrc/s2c_dwstream.hip, round 5) ---------------
 * part (parts x C x N) = per-workgroup partial sums of dW[c, n] = sum_m dY[m, c] A[m, n] (add them up
 * with s2c_multi_colsum); parts = s2c_weight_grad_stream_parts(...) (0: shape not taken -- C % 64,
 * ldy % 4, 16-byte aligned dY, more than 8 tiles of 64 x 64, fewer than 1024 rows).  A: any N, rows
 * dword-aligned (row stride lda >= N: e.g. the (B,N,3+C) cloud's rows or its feature columns, read in
 * place); A == dY (same strides): the Gram matrix, loaded once.  fp32-accurate bf16x3 MFMA products,
 * deterministic.  Autograd of lib/pointnet2/pytorch_utils.py:67-120 (the shared MLPs' 1x1 convs). */
int s2c_weight_grad_stream_parts(long long M, int C, int N, const float *dY, long long ldy,
                                 const float *A, long long lda);
int s2c_weight_grad_stream(long long M, int C, int N, const float *dY, long long ldy,
                           const float *A, long long lda, float *part, void *stream);
int s2c_weight_grad_stream_set_grid(int workgroups);
/* ... with the operand relu?(A pscale[n] + pshift[n]) formed on the way: a layer's input activation
 * recomputed from the previous layer's pre-activation A (the arithmetic of the forward GEMM's
 * BatchNorm + ReLU prologue), so that the forward writes no activation side output for this product.
 * Shapes as s2c_weight_grad_stream_parts, except single tiles with N <= 16 (-2); A == dY: the Gram matrix
 * of the activation (both operands transformed). */
int s2c_weight_grad_stream_act(long long M, int C, int N, const float *dY, long long ldy,
                               const float *A, long long lda, const float *pscale,
                               const float *pshift, int prelu, float *part, void *stream);

/* ---- a 64 -> 64 BatchNorm(+ReLU) layer's backward in one pass (csrc/s2c_bnbwd_fused.hip, round 5) ---
 * dY = BatchNorm(+ReLU)-backward(dA, Y) (scale .. coef as for s2c_bn_bwd_gemm) is never written:
 *   dX (M x 64) = dY W                     W (64 x 64) AS STORED (row = output channel, row stride ldw);
 *   dWpart (parts x 64 x 64): partial sums of dW = dY^T relu?(nY nscale + nshift) -- the layer's input is
 *                     the previous layer's activation, recomputed from its pre-activation nY (M x 64);
 *   npartial (parts x 128): [s1 | s2] rows of the previous layer's BatchNorm-backward column sums over
 *                     (dX, nY), as s2c_bn_bwd_gemm_next_stats leaves them.
 * parts = s2c_bn_bwd_dx_dw64_parts(M) (0: not taken -- M % 16 != 0 or M < 4096); all row tensors
 * contiguous (row stride 64), 16-byte aligned.  Replaces s2c_bn_bwd_gemm_next_stats + the dY tensor +
 * s2c_weight_grad_stream + the forward's activation side output for that layer (SA1's second layer).
 * Autograd of lib/pointnet2/pytorch_utils.py:67-120 inside pointnet2_modules.py:251-257. */
int s2c_bn_bwd_dx_dw64_parts(long long M);
int s2c_bn_bwd_dx_dw64(long long M, const float *dA, const float *Y, const float *scale,
                       const float *shift, const float *mean, const float *invstd, const float *coef,
                       int relu, const float *W, int ldw, float *dX, const float *nY,
                       const float *nscale, const float *nshift, const float *nmean,
                       const float *ninvstd, int nrelu, float *dWpart, float *npartial, void *stream);

/* ---- the small products of the layer stacks (csrc/s2c_sgemm.hip, round 5) --------------------------
 * Y (M x N, row stride ldy) = A (M x K, row stride lda) B (+ bias[n]):
 *   b_transposed = 0: B is (K x N) row-major, row stride ldb -- the input gradient dX = dY W with W as
 *                     stored (no transposed copy);
 *   b_transposed = 1: B is W (N x K) row-major, Y = A W^T -- the forward of a Conv1d / Linear layer.
 * 64 x 64 tiles, the four waves of a workgroup split K; fp32-accurate bf16x3 MFMA products.
 * s2c_small_gemm_supported: 4 <= K <= 1024, (K x N) form: N % 4 == 0, ldb % 4 == 0, B 16-byte aligned.
 * Autograd / forward of lib/pointnet2/pytorch_utils.py:67-120 at 2048 .. 32768 rows. */
int s2c_small_gemm_supported(long long M, int N, int K, long long lda, long long ldb, int b_transposed);
int s2c_small_gemm(long long M, int N, int K, const float *A, long long lda, const float *B,
                   long long ldb, int b_transposed, const float *bias, float *Y, long long ldy,
                   void *stream);
/* The general form: operand rows through index maps (index i -> element offset (i / div) * hi + (i % div) * lo;
 * div == 0: i * lo -- e.g. row (r, t) of a (T, R, .) tensor), three layouts, optional split of K over
 * workgroups with float atomics into a ZEROED Y (a long reduction behind few tiles; not deterministic):
 *   form 0: Y = A B,    A (M x K) rows by arow(m), k contiguous;  B (K x N) rows by brow(k), N % 4 == 0
 *   form 1: Y = A W^T,  A as above;                                W (N x K) rows by brow(n)
 *   form 2: Y = A^T B,  A (K x M) rows by arow(k), M % 4 == 0;     B (K x N) rows by brow(k)
 * Y rows by crow(m).  The teacher-forced decoder's classifier products (models/decoder_fused.py). */
typedef struct s2c_sgemm_map { int div, hi, lo; } s2c_sgemm_map;
typedef struct s2c_sgemm_args {
  const float *A, *B, *bias;
  float *Y;
  long long M;
  int N, K, form, ksplit;
  s2c_sgemm_map arow, brow, crow;
  int pad_;
} s2c_sgemm_args;
int s2c_small_gemm_ex(const s2c_sgemm_args *g, void *stream);

/* ---- the optimizer step (csrc/s2c_optim.hip) ------------------------------------------------
 * torch.optim.Adam's update (scripts/train.py:138, lib/solver.py:293-302: `optim.Adam(params, lr,
 * weight_decay)`, one step per batch; amsgrad / maximize off, L2 weight decay added to the gradient)
 * of up to S2C_ADAM_MAX_TENSORS parameter tensors in ONE launch.  Tensor i: parameter and gradient
 * (numel fp32 each; grad == NULL: skipped, as torch skips a parameter without a gradient), its first /
 * second moments at exp_avg + offset / exp_avg_sq + offset (flat buffers of the caller; offset in
 * elements, a multiple of 4 keeps the 16-byte path).  first_block[i] = sum over j < i of
 * ceil(numel_j / s2c_adam_chunk()), first_block[n_tensors] = the grid.  step[i] (fp32, device) holds the
 * number of updates tensor i has had; the kernel uses step[i] + 1 and stores it back for the tensors
 * with a gradient (the last workgroup to finish does, through *counter, which must be 0 at the first
 * launch and is left 0). */
#define S2C_ADAM_MAX_TENSORS 128
typedef struct s2c_adam_tensor {
  float *param;
  const float *grad;
  int numel, offset;
} s2c_adam_tensor;
typedef struct s2c_adam_args {
  int n_tensors, pad_;
  double lr, beta1, beta2, eps, weight_decay;   /* torch passes them to its kernels as doubles too */
  float *exp_avg, *exp_avg_sq, *step;
  unsigned *counter;
  int first_block[S2C_ADAM_MAX_TENSORS + 1];
  int pad3_;
  s2c_adam_tensor t[S2C_ADAM_MAX_TENSORS];
} s2c_adam_args;
int s2c_adam_chunk(void);
int s2c_adam_multi(const s2c_adam_args *a, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* S2C_FUSED_H */
