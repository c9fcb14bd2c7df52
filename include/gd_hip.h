/* gd_hip.h — C ABI of the MI355X (gfx950) geometric-distillation kernels.
 *
 * Drop-in boundary for the hot path of kaist-cvml/3d-vlm-gd.  The reference's
 * own native-op convention is dust3r/croco/models/curope/curope.cpp:49-69 (plain functions on
 * caller-owned tensors, argument checks that raise, current stream, no global state); every entry
 * point here follows it with plain pointers and sizes:
 *   - returns 0 on success, negative on error; gd_last_error() holds the message (thread-local);
 *   - never allocates: outputs and workspaces are caller-owned (…_workspace_bytes() queries);
 *   - launches on the hipStream_t passed as `stream` (void*); stateless and thread-safe: the only process-wide state is a
 *     table of A/B options read ONCE from GD_* environment variables at first use (csrc/gd_knobs.h; thread-safe static
 *     initialisation) and the dlopen'ed RCCL binding of gd_comm_*; the two debug hooks below are the documented exceptions;
 *   - dtype codes: 0 = float32, 1 = bfloat16; row-major, contiguous last dimension.
 * Each declaration cites the reference interface it replaces (paths relative to the reference root).
 */
#ifndef GD_HIP_H
#define GD_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GD_F32 0
#define GD_BF16 1
#define GD_F32X3 2   /* gd_attention_{fwd,bwd}: fp32 tensors, every product as three bf16 MFMAs of (hi, lo) splits (TF32-class; see gd_split3); gd_gemm_nt c_dtype: split output */
#define GD_F16 3     /* IEEE half: the operand format of the tf32h engine (gd_cast_f16 -> gd_gemm_nt_scaled, gd_attention_*): 11-bit significands = TF32's */

const char* gd_last_error(void);
/* Bumped whenever the exported surface changes (round 4: gd_amax_scale and gd_lora_bwd_fused_scaled took one more argument -> 2; round 6: the
 * entry points of five shelved experiments — gd_gemm_nt_lnfold_emit / _apply, gd_ln_fold_stats, gd_stream_create_cu_mask, gd_stream_destroy — left the
 * library, and round 5's gd_adapter_fused_h_ln / gd_kp_gather_fwd_ln are counted -> 3); the Python loader refuses a library whose version differs
 * from the one its signature table was written for. */
#define GD_ABI_VERSION 3
int gd_abi_version(void);

/* Debug hooks (no reference counterpart; NOT for production use, process-wide, not thread-safe).
 * gd_debug_set / gd_debug_get: override / read one option of the table above by its name ("cv_grid", "cv_dbg", "gemm_persist",
 *   ... — csrc/gd_knobs.h) — how tests force few persistent blocks and how the anatomy tools switch kernel parts off.
 * gd_gemm_phase_probe: per-phase shader-clock cycles of the persistent gemm_nt kernel.  Works ONLY in a library built with
 *   -DGD_GEMM_STAGE_PROBE (the probe's s_memtime reads cost 20 % even unarmed, so the shipped build has no probe code and this call
 *   fails with a message when asked to arm).  enable=1 arms it for subsequent gd_gemm_nt calls, enable=0 disarms; out6 (host, SIX
 *   values, may be NULL) receives the sums over blocks since arming: [0] waiting for a tile's first K stage, [1] main loop,
 *   [2] epilogue items, [3] tiles, [4] epilogue set-up + next-tile DMA issue, [5] the part of [1] spent in the per-K-step wait + barrier. */
int gd_debug_set(const char* name, int value);
int gd_debug_get(const char* name);
int gd_gemm_phase_probe(int enable, unsigned long long* out6);

/* C[M,N] = epilogue(alpha * A[M,K] . W[N,K]^T); replaces torch nn.Linear / torch.bmm on the student path
 * (timm VisionTransformer qkv/proj/fc1/fc2, utils/model.py:57-71 LoRA, :7-25 Adapter; src/finetune_timm_vggt.py:516-517).
 * Epilogue order: +bias[N](f32) -> +lora_t[M,rt].lora_b[rt,N](f32, rt<=8) -> store preact -> act(1 GELU erf,2 ReLU,
 * 3 GELU with preact receiving GELU'(v) instead of v) -> *act'(dact_src) (1 dGELU(src), 2 src>0, 3 v*=src: the stored
 * derivative of act 3) -> +residual -> +C (accumulate).  batch>1: grid over batch strides.
 * c_dtype GD_F32X3 (tf32x engine): the f32 result leaves as its bf16 operand split C[M, 3N] = [hi | lo | hi] (gd_split3 which 0,
 * ldc >= 3N bf16 elements) — the left operand of the next split GEMM — with preact / dact_src f32; served for bf16 (split) operands,
 * M >= 1024, N >= 256, K % 64 == 0 and the act 1|3 (+preact) or dact 3 epilogues; anything else is an error, not a reroute. */
int gd_gemm_nt(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
               int batch, long sA, long sW, long sC, int ab_dtype, int c_dtype, float alpha,
               const float* bias, const float* lora_t, const float* lora_b, int lora_rt, void* preact, long ldp,
               int act, const void* dact_src, long ldd, int dact, const void* residual, long ldr, int accumulate,
               void* stream);

/* G[N,K](f32) += alpha * Y[M,N]^T . X[M,K]; weight gradients of the trainable LoRA / Adapter / refine_conv
 * tensors (autograd of utils/model.py:7-71, src/finetune_timm_vggt.py:146 refine_conv). */
int gd_gemm_tn(const void* Y, const void* X, float* G, int M, int N, int K, long ldy, long ldx, long ldg,
               int batch, long sY, long sX, long sG, int y_dtype, int x_dtype, float alpha, void* stream);

/* Multi-head self-attention, head_dim 64, flash-style (no N x N matrix): replaces F.scaled_dot_product_attention in
 * timm Attention.forward (same arithmetic as vggt/layers/attention.py:51-71).  qkv [B,N,3,H,64] packed
 * QKV-GEMM output, o [B,N,H*64], lse [B,H,N] f32 (natural log); backward writes dqkv [B,N,3,H,64] as (dq, dk, dv)
 * (grad_order 0) or (dq, dv, dk) (grad_order 1: the q and v gradients, the only ones the q/v LoRA factors contract, become
 * one contiguous 2*H*64-column block); grad_order bit 1 (values 2, 3): dK is not needed (a block whose input receives no
 * gradient: only the q / v LoRA factors learn there) — its columns are left unwritten and the dK/dV kernel runs the dV half only;
 * delta_ws: f32 scratch of 2 * B * H * Npad floats, Npad = N rounded up to a multiple of 64 (ABI version 3; it was B * H * N): the dQ kernel leaves, per
 * (image, head) row of Npad queries, [-delta | -lse in log2 units] for the dK/dV kernel (pads: 0 / -1e30). */
int gd_attention_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int head_dim, float scale, int dtype,
                     void* stream);
int gd_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, float* delta_ws,
                     int B, int N, int H, int head_dim, float scale, int dtype, int grad_order, void* stream);

/* LoRA backward of one block in one pass (utils/model.py:57-71): dt [M,8] f32 = dqv [M,K] . bt^T and gbt [8,K] f32 += t^T . dqv for
 * the bf16 (dq, dv) gradient block dqv (row stride ldx elements), t [M,8] f32 the saved rank projections, bt [8,K] bf16 the B factors
 * transposed.  bt == NULL: only the second product (dt may be NULL too) — any [8, K] += t^T . X with an [M, 8] f32 left operand, e.g. the
 * LoRA-A gradient dt^T . LN(x).  K % 256 == 0, K / 256 in {1, 2, 3, 4, 6, 8}. */
int gd_lora_bwd_fused(const void* dqv, long ldx, const float* t, const void* bt, float* dt, float* gbt, int M, int K, void* stream);
/* The same pass (autograd of _LoRA_qkv, utils/model.py:57-71) on either 16-bit operand type (dtype GD_BF16 | GD_F16) with the tf32h engine's device-side scales: t is multiplied by *t_mul_dev
 * before it is split into its high and low 16-bit parts (a GRADIENT in the t role goes in under the step's power-of-two scale), dt and the gbt
 * partial by *out_mul_dev on the way out (1 / s when dqv or t carried s).  NULL = 1.  dt_scaled = 1: dt leaves WITHOUT *out_mul_dev — still
 * in dqv's scaled domain, for consumers that take it under s anyway (the LoRA rank update of a scaled-domain dX GEMM, the LoRA-A gradient). */
int gd_lora_bwd_fused_scaled(const void* dqv, long ldx, const float* t, const void* bt, float* dt, float* gbt, int M, int K, int dtype,
                             const float* t_mul_dev, const float* out_mul_dev, int dt_scaled, void* stream);

/* Dense cost-volume KL for P pairs, fused: calculate_cost_loss (src/finetune_timm_vggt.py:488-533 variant 0,
 * src/finetune_timm_mast3r.py:504-540 variant 1) = F.normalize + bmm x2 + softmax + get_masked_patch_cost
 * (utils/functions.py:402-422) + kl_divergence_map (utils/losses.py:5-15).
 * f1,f2 [P,hw,C] raw features; t1,t2 [P,hw,ldt] f32 teacher maps with row stride ldt >= hw elements (ldt % 4 == 0 gives
 * 16-byte aligned rows: the fast path; pad entries are never read as data); m1,m2 [P,hw] uint8 row masks;
 * loss [P] f32; stats [P,2,hw,4] f32 (saved for the backward).
 * tstats [P,2,hw,4] f32 = per teacher row {max(rowsum, 1e-8), W = sum_j t, A = sum_j t log t, 0} with
 * t = max(T / rowsum, 1e-8): it depends on the teacher maps only, so a caller that caches a pair's targets computes it once
 * with gd_cost_volume_teacher_stats and every step then reads each map a single time; tstats == NULL: computed inside. */
size_t gd_cost_volume_kl_workspace_bytes(int P, int hw, int C, int dtype, int backward);
int gd_cost_volume_teacher_stats(const float* t1, const float* t2, int P, int hw, int ldt, float* tstats, void* stream);
int gd_cost_volume_kl_fwd(const void* f1, const void* f2, const float* t1, const float* t2, int ldt, const float* tstats,
                          const unsigned char* m1, const unsigned char* m2, int P, int hw, int C, int variant, int dtype,
                          float* loss, float* stats, void* workspace, void* stream);
/* The same forward when the caller already holds the inverse L2 norms of the feature rows, inv_norm_k [P, hw] = 1 / max(||f_k row||, 1e-12)
 * of the rows as stored (gd_tap_mean_norm_fwd hands them out while it writes the features): the kernel's own pass over the
 * features (135 MB at the step's size) is replaced by a 2 P hw-thread copy. */
int gd_cost_volume_kl_fwd_prenorm(const void* f1, const void* f2, const float* inv_norm1, const float* inv_norm2, const float* t1,
                                  const float* t2, int ldt, const float* tstats, const unsigned char* m1, const unsigned char* m2,
                                  int P, int hw, int C, int variant, int dtype, float* loss, float* stats, void* workspace,
                                  void* stream);
int gd_cost_volume_kl_bwd(const void* f1, const void* f2, const float* t1, const float* t2, int ldt,
                          const unsigned char* m1, const unsigned char* m2, int P, int hw, int C, int dtype,
                          const float* gloss, const float* stats, void* df1, void* df2, void* workspace, void* stream);

/* nn.LayerNorm forward / backward-to-input (timm Block.norm1/norm2, model.norm; frozen affine).  dres and dres2
 * (optional, dtype of x, row stride ldx) are added to dx: the residual-stream gradient, and — for a tapped block output,
 * which feeds the next block, the tap's norm and the un-normed tap mean — the third consumer's gradient, so that the
 * sum autograd would form with two extra passes comes out of this one kernel.  dy may be f32 while x is bf16 (dy_dtype). */
int gd_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M,
                     int D, long ldx, long ldy, float eps, int dtype, int y_dtype, void* stream);
int gd_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                     const void* dres, const void* dres2, void* dx, int M, int D, long ldd, long ldx, float dyscale,
                     int dtype, int dy_dtype, void* stream);
/* Adapter / BlockWithAdapter (utils/model.py:7-25): out = x + up(relu(down(x))), D -> 64 -> D, no bias — one fused pass
 * (x read once, out written once; the 64-wide hidden tile never leaves the CU except as the saved copy `hidden`).
 *   forward : gate_src = NULL (ReLU), w1 = down.weight [64, D], w2 = up.weight [D, 64]
 *   backward: x = dOut, gate_src = saved forward hidden [M, 64] (keep where > 0), w1 = up.weight^T [64, D],
 *             w2 = down.weight^T [D, 64]; `hidden` then receives d(hidden) for the weight gradients (gd_gemm_tn).
 * bf16, bottleneck 64, D in {256, 512, 768, 1024} (gd_adapter_fused_supported); other shapes: two gd_gemm_nt calls. */
int gd_adapter_fused_supported(int D, int bottleneck, int dtype);
int gd_adapter_fused(const void* x, const void* w1, const void* w2, const void* gate_src, void* hidden, void* out, int M,
                     int D, int bottleneck, int dtype, void* stream);
/* F.normalize(p=2, dim=-1) on fp32 rows (src/finetune_timm_vggt.py:328) and its backward. */
int gd_l2norm_fwd(const float* x, float* y, float* inv, int M, int D, float eps, void* stream);
int gd_l2norm_bwd(const float* y, const float* dy, const float* inv, float* dx, int M, int D, void* stream);

/* a0 + patch-embed prologue: torchvision bilinear resize (h,w)->(H,W) (src/finetune_timm_vggt.py:270,340), timm
 * Normalize (:153), im2col for the PxP/stride-P conv of timm PatchEmbed; col [B*(H/P)*(W/P), Kp] zero-padded, in `dtype`
 * (GD_F32 | GD_BF16 | GD_F16: the tf32h engine takes the projection's fp16 operand straight from here — the patch conv is frozen). */
int gd_patch_im2col(const float* img, void* col, int B, int h, int w, int H, int W, int P, int Kp,
                    const float* mean3, const float* std3, int dtype, void* stream);
/* The same with the conv stride decoupled from the patch size: src/evaluate_timm.py:262-266 sets
 * `model.model.patch_embed.proj.stride = (s, s)` with s = patch/2 for dense tracking features (overlapping patches);
 * col [B*(1+(H-P)/stride_y)*(1+(W-P)/stride_x), Kp], row (gy, gx) = the PxP window at (gy*stride_y, gx*stride_x). */
int gd_patch_im2col_strided(const float* img, void* col, int B, int h, int w, int H, int W, int P, int stride_y,
                            int stride_x, int Kp, const float* mean3, const float* std3, int dtype, void* stream);
/* timm _pos_embed: cls + pos[0] | patch + pos[1:]  ->  tokens [B, Np+1, D]. */
int gd_assemble_tokens(const void* patch, const float* cls, const float* pos, void* out, int B, int Np, int D,
                       int dtype, void* stream);
/* refine_conv 3x3/pad 1 (src/finetune_timm_vggt.py:146,325) WITHOUT im2col: gd_stack3_rows writes, for every row r of a grid
 * whose lines carry one zero separator column (pitch = gw + 1), the three vertically adjacent feature vectors side by side
 * ([B*gh*pitch + 2, 3D], one zero guard row at each end); the conv, its transpose and its weight gradient are then single
 * gd_gemm_nt / gd_gemm_tn calls on the OVERLAPPING-row view A[r][k] = buf[r*3D + k], K = 9D, k = (dx+1)*3D + (dy+1)*D + c.
 * src: token layout (src_pitch = gw, grid starts src_row0 elements into each image of src_bstride elements) or a pitched grid
 * (src_pitch = gw + 1).  gd_unpitch_tokens: pitched [B, gh, gw+1, D] -> token layout [B, prefix + gh*gw, D], prefix rows zero.
 * (gd_im2col3x3 / gd_col2im3x3: the materialising form, kept for shapes the view cannot serve.) */
int gd_stack3_rows(const void* src, void* dst, int B, int gh, int gw, int D, long src_bstride, long src_row0, int src_pitch,
                   int src_dtype, int dst_dtype, void* stream);
int gd_unpitch_tokens(const void* src, void* dst, int B, int gh, int gw, int D, int prefix, int dtype, void* stream);
int gd_im2col3x3(const void* x, long bstride, void* col, int B, int gh, int gw, int D, int dtype, void* stream);
int gd_col2im3x3(const void* dcol, void* dx, long bstride, int B, int gh, int gw, int D, int dtype, void* stream);
/* interpolate_features (utils/functions.py:55-76) on 1..4 token-major grids, averaged; backward scatters into fp32
 * gradient grids (batch stride bstride elements, pre-zeroed).  pitch = tokens per grid line in memory (gw for a dense grid,
 * gw + 1 for the separator-column layout of gd_stack3_rows' GEMM output). */
int gd_kp_gather_fwd(const void* const* grids, int ngrid, long bstride, int grid_dtype, const float* kp, float* out,
                     int B, int Nk, int gh, int gw, int D, float sx, float sy, int img_h, int img_w, int patch,
                     int stride, int pitch, void* stream);
/* The same gather on RAW block outputs with `model.norm` (the final LayerNorm: src/finetune_timm_vggt.py:262-276, get_intermediate_feature's
 * `self.model.norm(feat)`) applied where the grids are sampled: means[t] / rstds[t] = per-token LayerNorm statistics of grid t ([B][sstride], addressed
 * from the same token as grids[t]), ln_w / ln_b the shared affine [D].  A tapped block's output is the next block's input, whose LayerNorm forward
 * already took the statistics: the taps' normed copies are never materialised.  f32 / bf16 grids, 16-byte rows. */
int gd_kp_gather_fwd_ln(const void* const* grids, const float* const* means, const float* const* rstds, int ngrid, long bstride, long sstride,
                        int grid_dtype, const float* ln_w, const float* ln_b, const float* kp, float* out, int B, int Nk, int gh, int gw, int D,
                        float sx, float sy, int img_h, int img_w, int patch, int stride, int pitch, void* stream);
int gd_kp_gather_bwd(float* const* dgrids, int ngrid, long bstride, const float* kp, const float* dout, int B, int Nk,
                     int gh, int gw, int D, float sx, float sy, int img_h, int img_w, int patch, int stride, int pitch,
                     void* stream);
/* Input gradient of gd_kp_patch_gather without atomics: U [B*Nk, 9*D] = dfeat . W (one GEMM against gd_conv_weight_pack's wu) ->
 * dtok [B][prefix_rows + gh*gw][D] in `dtype`, every element written once, contributions summed in keypoint order.  Nk <= 1024. */
int gd_kp_patch_bwd_det(const void* U, void* dtok, int dtype, long bstride, int prefix_rows, const float* kp, int B, int Nk,
                        int gh, int gw, int D, float sx, float sy, int img_h, int img_w, int patch, int stride, void* stream);
/* refine_conv (nn.Conv2d weight [D][D][3][3], src/finetune_timm_vggt.py:146) packed for the step in one pass:
 * wk[n][(ky,kx,c)] (forward / weight gradient at the keypoints) and wt[c][(kx',ky',n)] = W[n][c][2-ky'][2-kx'] (the flipped kernel
 * of the transposed convolution on the stacked-row view) and, when wu != NULL, wu[(ky,kx,c)][n] (the [9D, D] operand of
 * dcol = dfeat . wu^T for gd_kp_patch_bwd_det), all in `dtype`. */
int gd_conv_weight_pack(const float* weight, void* wk, void* wt, void* wu, int D, int dtype, void* stream);
/* interpolate_features backward without atomics (deterministic): writes EVERY element of dgrid [B][prefix_rows + gh*pitch][D] in
 * out_dtype (prefix rows and separator columns zero), each position = the fixed-order sum of scale * w * dout[kp] over the
 * keypoints that touch it.  Nk <= 1024, D % 8 == 0. */
int gd_kp_gather_bwd_det(void* dgrid, int out_dtype, long bstride, int prefix_rows, const float* kp, const float* dout,
                         float scale, int B, int Nk, int gh, int gw, int D, float sx, float sy, int img_h, int img_w,
                         int patch, int stride, int pitch, void* stream);
/* get_feature's refine_conv + interpolate_features (src/finetune_timm_vggt.py:319-325) with the two linear maps swapped:
 * out[b*Nk + k][(ky, kx, c)] = bilinear mix (the same four neighbours and weights as gd_kp_gather_fwd) of the 3x3 input patches
 * (zero padding 1) of ONE token grid [gh x pitch, D] -> [B*Nk, 9*D] in the grid's dtype; the conv is then a GEMM over B*Nk rows
 * against the weight packed as [D_out, (ky, kx, c)].  D*elsize must be a multiple of 16 bytes. */
int gd_kp_patch_gather(const void* grid, long bstride, int grid_dtype, const float* kp, void* out, int B, int Nk, int gh,
                       int gw, int D, float sx, float sy, int img_h, int img_w, int patch, int stride, int pitch,
                       void* stream);
/* the same gather (src/finetune_timm_vggt.py:319-325) from an fp32 grid with the taps written as fp16 (tf32h engine: the K = 9D GEMM's operand and the weight gradient's, no fp32 block
 * and no cast pass); D % 4 == 0 */
int gd_kp_patch_gather_h(const float* grid, long bstride, const float* kp, void* out16, int B, int Nk, int gh, int gw, int D, float sx, float sy,
                         int img_h, int img_w, int patch, int stride, int pitch, void* stream);
/* extract_kp_depth (utils/functions.py:348-372) and get_patch_mask_from_kp_tensor (:375-399; mask pre-zeroed). */
int gd_kp_depth(const float* depth, const float* kp, float* out, int B, int Nk, int H, int W, void* stream);
int gd_patch_mask(const float* kp, unsigned char* mask, int B, int Nk, int H, int W, int P, void* stream);

/* get_feature_cost tail (src/finetune_timm_mast3r.py:321-337, src/finetune_timm_vggt.py:342-353): mean of 1..4 tap
 * outputs [B, prefix+hw, D] with the prefix token dropped -> [B,hw,D]; backward fills each of the ngrid buffers
 * dgrids[t] [B, prefix+hw, D] with scale * dout (prefix rows zero) — scale = 1 / (number of taps averaged); the taps all
 * receive the same gradient, so one buffer (ngrid = 1) can be handed to every tap.  D must be a multiple of 8. */
int gd_tap_mean_fwd(const void* const* grids, int ngrid, long bstride, int prefix, void* out, int B, int hw, int D,
                    int dtype, void* stream);
/* gd_tap_mean_fwd that also hands out inv_norm [B, hw] = 1 / max(||out row||, 1e-12) of the rows as stored (what F.normalize divides
 * by, src/finetune_timm_vggt.py:514-515; feeds gd_cost_volume_kl_fwd_prenorm). */
int gd_tap_mean_norm_fwd(const void* const* grids, int ngrid, long bstride, int prefix, void* out, float* inv_norm, int B, int hw,
                         int D, int dtype, void* stream);
int gd_tap_mean_bwd(void* const* dgrids, int ngrid, int prefix, const void* dout, int B, int hw, int D, float scale,
                    int dtype, void* stream);

/* calculate_matching_loss after the descriptor GEMM (src/finetune_timm_vggt.py:543-572 variant 0,
 * src/finetune_timm_mast3r.py:560-589 variant 1): sim [P,Nmax,Nmax] -> loss [P] and dsim = dloss/dsim (fused). */
int gd_smooth_ap(const float* sim, const float* pts3d_1, const float* pts3d_2, const int* counts, int P, int Nmax,
                 int variant, float thres3d_neg, float temp, float* loss, float* dsim, float* row_ws, void* stream);

/* "ME" variant of the matching loss (src/finetune_timm_me.py:191-220): positives = every (i,j) with |p1_i - p2_j| <
 * thres3d_pos (dynamic count), negatives = distance > thres3d_neg; loss[p] = mean over the pair's positives of
 * 1 - (ap1 + ap2)/2 (0 when there is none); dsim = d loss[p] / d sim.  row_ws: P*Nmax*2 floats.  Nmax <= 4096. */
int gd_smooth_ap_me(const float* sim, const float* pts3d_1, const float* pts3d_2, const int* counts, int P, int Nmax,
                    float thres3d_pos, float thres3d_neg, float temp, float* loss, float* dsim, float* row_ws,
                    void* stream);
/* pairwise_logistic_ranking_loss (utils/losses.py:18-41) with DepthAwareFeatureFusion (utils/model.py:100-127) on
 * pre-projected u = W1 f [S,Nmax,128]: loss [S], du (scaled by gscale/count), head_grad[516] += {b1,ln_w,ln_b,w2,b2} (nullable),
 * head_grad_sets [S,516] = the same per set (nullable). */
size_t gd_pair_rank_workspace_bytes(int S);
int gd_pair_rank(const float* u, const float* depth, const int* counts, const float* gscale, int S, int Nmax,
                 float depth_threshold, const float* b1, const float* ln_w, const float* ln_b, const float* w2,
                 const float* b2, float* loss, float* du, float* head_grad, float* head_grad_sets, void* workspace,
                 void* stream);
/* The step's host-side glue as kernels (round 6; each replaces a handful of torch elementwise / reduce launches on [P]-vectors and small tensors):
 *   gd_loss_combine_fwd   : loss = mean_p keep_p (w[0] t0[p] + w[1] t1[p] + w[2] t2[p] + w[3] t3[p]) — the weighted loss sum and `.mean()` of
 *                           training_step (src/finetune_timm_vggt.py:599-616; src/finetune_timm_mast3r.py:650-653); keep_p = counts == NULL || counts[p] > 0
 *                           (a pair without keypoints: constant zero loss, src/finetune_timm_mast3r.py:604-607); terms_out [4][P] (nullable) = keep_p t_i[p];
 *                           w4: four HOST floats;
 *   gd_loss_combine_bwd   : grads [4][P] = g[0] w[i] keep_p / P (the chain rule through it);
 *   gd_depth_bwd_combine  : the backward of calculate_depth_loss's two terms from the unit-gradient outputs of gd_pair_rank / gd_depth_l1:
 *                           du_out = 0.5 g_intra[p] du_rank + g_l1[p] du_l1 ([P][2][N][128]; rows written view-major [2][P][N] when view_major), hg_out [516]
 *                           = sum_p 0.5 g_intra[p] (hg_rank[2p] + hg_rank[2p+1]) + g_l1[p] hg_l1[p];
 *   gd_scale_and_transpose: out [B][R][C] = in * g[b] and out_t [B][C][R] = its transpose (smooth-AP backward: dsim g contracted from both sides).
 * gd_pair_rank / gd_depth_l1: gscale may be NULL (= 1 per set). */
int gd_loss_combine_fwd(const float* t0, const float* t1, const float* t2, const float* t3, const float* w4, const int* counts, int P, float* loss,
                        float* terms_out, void* stream);
int gd_loss_combine_bwd(const float* g, const float* w4, const int* counts, int P, float* grads, void* stream);
int gd_depth_bwd_combine(const float* du_rank, const float* du_l1, const float* hg_rank, const float* hg_l1, const float* g_l1, const float* g_intra, int P,
                         int N, int view_major, float* du_out, float* hg_out, void* stream);
int gd_scale_and_transpose(const float* in, const float* g, float* out, float* out_t, int B, int R, int C, void* stream);
/* F.l1_loss(head(f1 - f2), tanh(d1 - d2)) (src/finetune_timm_vggt.py:475-479); u [P,2,Nmax,128]. */
int gd_depth_l1(const float* u, const float* d1, const float* d2, const int* counts, const float* gscale, int P,
                int Nmax, const float* b1, const float* ln_w, const float* ln_b, const float* w2, const float* b2,
                float* loss, float* du, float* head_grad, float* head_grad_sets, void* workspace, void* stream);

/* Eager DepthAwareFeatureFusion.forward (utils/model.py:101-127, `depths=None` branch: fusion_layer + tanh) on rows already
 * projected by its first Linear, u = W1 f [M,128] (gd_gemm_nt): out[m] = tanh(w2 . GELU(LayerNorm_128(u[m] + b1)) + b2).
 * Backward: du [M,128] and head_grad[516] += {b1, ln_w, ln_b, w2, b2} gradients for the upstream dout [M]. */
int gd_depth_head_fwd(const float* u, int M, const float* b1, const float* ln_w, const float* ln_b, const float* w2,
                      const float* b2, float* out, void* stream);
int gd_depth_head_bwd(const float* u, const float* dout, int M, const float* b1, const float* ln_w, const float* ln_b,
                      const float* w2, const float* b2, float* du, float* head_grad, void* stream);

/* Stand-alone forms of the reference's loss helpers, for callers that bind the reference's function names
 * (gd_amd/compat.py); the training step uses the fused kernels above instead and never materialises these maps.
 * gd_sigmoid_temp: utils/functions.py:24-33 sigmoid(tensor, temp): exponent = clamp(-x / temp, -50, 50), y = 1 / (1 + exp(.));
 *   dy == NULL: out = y; dy != NULL: out = dy * dy/dx (zero where the clamp is active).
 * gd_masked_patch_cost_{fwd,bwd}: utils/functions.py:402-422 get_masked_patch_cost(cost [B,rows,cols], mask_patch_1 [rows],
 *   mask_patch_2 [cols] or NULL, eps, use_softmax, temperature); bwd takes the forward's output y.
 * gd_kl_divergence_map_{fwd,bwd}: utils/losses.py:5-15 kl_divergence_map(t, p, eps) over `rows` rows of `cols` entries ->
 *   loss[1]; row_ws: rows floats; bwd writes dt and / or dp (nullable) for the upstream scalar gloss[1]. */
int gd_sigmoid_temp(const float* x, const float* dy, float* out, long n, float temp, void* stream);
int gd_masked_patch_cost_fwd(const float* cost, const unsigned char* m1, const unsigned char* m2, float* out, int B,
                             int rows, int cols, float eps, int use_softmax, float temperature, void* stream);
int gd_masked_patch_cost_bwd(const float* cost, const float* y, const float* dy, const unsigned char* m1,
                             const unsigned char* m2, float* dcost, int B, int rows, int cols, float eps,
                             int use_softmax, float temperature, void* stream);
int gd_kl_divergence_map_fwd(const float* t, const float* p, long rows, int cols, float eps, float* loss,
                             float* row_ws, void* stream);
int gd_kl_divergence_map_bwd(const float* t, const float* p, const float* gloss, long rows, int cols, float eps,
                             float* dt, float* dp, void* stream);

/* Lightning gradient_clip_val=1.0 (src/main.py:153) + torch.optim.AdamW (src/finetune_timm_vggt.py:642-648) on the
 * flat fp32 trainable buffer; grads are multiplied by grad_scale first. */
size_t gd_adamw_workspace_bytes(void);
int gd_clip_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, int step,
                       float lr, float weight_decay, float beta1, float beta2, float eps, float max_norm,
                       float grad_scale, float* grad_norm_out, void* workspace, void* stream);
/* The same step restricted to n_ranges element ranges [ranges[2r], ranges[2r+1]) of the flat buffer (host array, range starts
 * multiples of 4): elements outside them are left untouched — torch.optim.AdamW skips parameters whose .grad is None
 * (depth_attention, utils/model.py:92-97: no moments, no weight decay); the clip norm is taken over the whole buffer. */
int gd_clip_adamw_ranges(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, int step, float lr,
                         float weight_decay, float beta1, float beta2, float eps, float max_norm, float grad_scale,
                         float* grad_norm_out, void* workspace, const long* ranges, int n_ranges, void* stream);
int gd_cast(const void* in, void* out, long n, float scale, int in_dtype, int out_dtype, void* stream);
/* TF32-class products on the bf16 matrix cores (gfx950 has no TF32 / xf32 MFMA; the reference's MASt3R path runs its matmuls in TF32:
 * dust3r/croco/models/croco.py:12 `torch.backends.cuda.matmul.allow_tf32 = True`): an f32 operand is written as three bf16 planes per
 * row — which = 0, left operand: [hi | lo | hi]; which = 1, right operand: [hi | hi | lo]; hi = bf16(x), lo = bf16(x - hi) — and
 * gd_gemm_nt on the two 3K-wide bf16 operands (fp32 accumulation, f32 output and epilogue tensors) evaluates
 * hi_a hi_w + lo_a hi_w + hi_a lo_w: relative error ~4e-6 of the product sum, against ~3e-4 for TF32 and ~2e-3 for plain bf16
 * (tests/test_gpu_gemm.py::test_split3_product_accuracy).  in [rows, K] f32 (row stride ld_in), out [rows, 3K] bf16. */
int gd_split3(const float* in, void* out, long rows, int K, long ld_in, int which, void* stream);

/* TF32-class products from ONE fp16 MFMA per term (the tf32h engine): fp16 carries TF32's 11-bit significand, the matrix cores take it at the
 * bf16 rate, and the missing exponent range is covered by scaling — forward activations and frozen weights go in as they are (saturated at
 * +-65504, far above anything a ViT produces), a gradient tensor is multiplied by a power of two taken from its own maximum on the device.
 * gd_cast_f16: out [rows, K] fp16 = sat(in * scale * (scale_dev ? *scale_dev : 1)), in f32 with row stride ld_in.
 * gd_amax_scale: scale3 (device, 3 floats) <- {s, 1/s, unused} with s the power of two that puts max|in| into (target/2, target]; a non-finite
 *   element makes s (and 1/s) NaN, so every consumer's result is poisoned as the f32 engine's arithmetic would be.  amax_slots: 256 zeroed words
 *   owned by the caller (zeroed again on return).  gd_scale_from_amax: the same {s, 1/s} from slots another kernel filled (gd_layernorm_bwd_ex).
 * gd_cast_f16_ex: gd_cast_f16 that also counts, in range_counters (128 words of partial sums, accumulated; nullable), the results that saturated at
 *   +-65504 (sum of words 0-63) and the non-zero inputs that fell below fp16's normal range 2^-14 (sum of words 64-127) — the engine's run-time view
 *   of its range contract (partial sums: one atomic per block, spread so that they do not queue on one line).
 * gd_gemm_nt_scaled: gd_gemm_nt with alpha multiplied by the device scalar *alpha_dev (the 1/s of a scaled operand) — no host round trip.
 * gd_gemm_nt itself takes ab_dtype GD_F16 (f32 results with f32 epilogue tensors, or c_dtype GD_F16: fp16 C, preact and dact_src). */
int gd_cast_f16(const float* in, void* out, long rows, int K, long ld_in, float scale, const float* scale_dev, void* stream);
/* Cost-volume KL of the tf32h engine: gd_cost_volume_kl_fwd_prenorm takes dtype GD_F16 (fp16 copies of the fp32 features, the fp32 rows' norms);
 * the backward recomputes S from the same fp16 copies, writes G = dloss/dS as fp16 under a power-of-two scale taken from `gloss` on the device,
 * contracts it on the fp16 MFMA kernels and takes the gradient through the L2 normalisation in fp32 on the fp32 features (df1, df2 fp32). */
/* Kept-row forward for SPARSE row masks (the MASt3R trainer's keypoint-patch masks, src/finetune_timm_mast3r.py:515-519 -> get_masked_patch_cost,
 * utils/functions.py:402-422, keep at most N_kp of the hw rows; same loss as src/finetune_timm_mast3r.py:522-540): both directions as compacted row
 * problems — kept rows of one view (gathered) against all rows of the other — instead of one hw x hw sweep; same loss and saved statistics as
 * gd_cost_volume_kl_fwd_prenorm.  kcap: a multiple of 128, >= the number of kept rows of any (pair, view). */
size_t gd_cost_volume_kl_rows_workspace_bytes(int P, int hw, int kcap);
int gd_cost_volume_kl_fwd_rows(const void* f1, const void* f2, const float* inv_norm1, const float* inv_norm2, const float* t1, const float* t2, int ldt,
                               const float* tstats, const unsigned char* m1, const unsigned char* m2, int P, int hw, int C, int kcap, int variant,
                               int dtype, float* loss, float* stats, void* workspace, void* stream);
/* Kept-row backward (autograd of src/finetune_timm_mast3r.py:522-540 / utils/losses.py:5-15 under those masks), the counterpart of
 * gd_cost_volume_kl_fwd_rows (same masks, same kcap; `stats` as saved by either forward): G = dloss/dS only for
 * the kept rows of each direction ([kcap, hw] and its transpose instead of two [hw, hw] matrices), four batched contractions (kept rows' gradients,
 * scattered back to their rows; the other view's gradient, dense), then the gradient through the L2 normalisation.
 * dtype GD_F32 | GD_BF16: f1, f2, df1, df2 in that type, f1h = f2h = NULL.  dtype GD_F16 (tf32h engine): f1, f2, df1, df2 fp32 and f1h, f2h the fp16
 * copies the forward ran on (G under the device-side power-of-two scale of gd_cost_volume_kl_bwd_h). */
size_t gd_cost_volume_kl_bwd_rows_workspace_bytes(int P, int hw, int C, int kcap, int dtype);
int gd_cost_volume_kl_bwd_rows(const void* f1, const void* f2, const void* f1h, const void* f2h, const float* t1, const float* t2, int ldt,
                               const unsigned char* m1, const unsigned char* m2, int P, int hw, int C, int kcap, int dtype, const float* gloss,
                               const float* stats, void* df1, void* df2, void* workspace, void* stream);
size_t gd_cost_volume_kl_bwd_h_workspace_bytes(int P, int hw, int C);
int gd_cost_volume_kl_bwd_h(const float* f1, const float* f2, const void* f1h, const void* f2h, const float* t1, const float* t2, int ldt,
                            const unsigned char* m1, const unsigned char* m2, int P, int hw, int C, const float* gloss, const float* stats,
                            float* df1, float* df2, void* workspace, void* stream);
/* gd_gemm_nt_copy16: C[M,N] (f32) = alpha * (*alpha_dev) * A.W^T + bias + residual on fp16 operands, AND copy16[M,N] = fp16(sat(C * *copy_scale_dev)) —
 * the residual-stream result together with the fp16 operand the next product of the tf32h engine takes (persistent-kernel shapes only). */
int gd_gemm_nt_copy16(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc, float alpha,
                      const float* alpha_dev, const float* bias, const void* residual, long ldr, void* copy16, long ldc16,
                      const float* copy_scale_dev, void* stream);
/* gd_gemm_tn with alpha multiplied by the device scalar *alpha_dev (weight gradients contracted from SCALED fp16 gradient operands);
 * gd_gemm_tn takes fp16 Y and X on the MFMA kernel (N, K >= 64) and fp16 X on the N = 8 streaming kernel; with ONE of Y / X fp16 and the other fp32
 * (N, K >= 64) the fp32 one is rounded to fp16 inside the kernel — an fp32 Y times 1 / *alpha_dev first (a gradient under the scale its alpha undoes). */
int gd_gemm_tn_scaled(const void* Y, const void* X, float* G, int M, int N, int K, long ldy, long ldx, long ldg,
                      int batch, long sY, long sX, long sG, int y_dtype, int x_dtype, float alpha, const float* alpha_dev, void* stream);
/* gd_layernorm_fwd with y_dtype GD_F16 (f32 rows in): LN(x) written as the fp16 operand directly.  gd_layernorm_bwd_cast: the f32 backward
 * that also writes dx16 [M, D] = fp16(sat(dx * *scale_dev)) — backward + gd_cast_f16 in one pass. */
/* Bottleneck adapter of the tf32h engine, fused (utils/model.py:7-25 forward, and its backward-to-input): fp32 x / out, fp16 operands.
 * out32 = x32 + alpha * gate(fp16(x32 * in_scale) . w1^T) . w2^T; hidden [M,64] fp16 = the gated first product (relu when gate_src is null, else kept
 * where gate_src > 0); out16 (nullable) = fp16(out32 * copy_scale).  in_scale / alpha_dev / copy_scale: DEVICE scalars, null = 1. */
int gd_adapter_fused_h_supported(int D, int bottleneck, long M);
/* gd_adapter_fused_h (forward form: ReLU gate, no scales) that ALSO writes the LayerNorm of its result — the next block's `norm1` (timm Block.forward:
 * x = x + ...; the next block starts with norm1(x)) — from the rows the kernel still holds on chip: ln16 [M, D] = fp16(LayerNorm(out32; ln_w, ln_b, ln_eps)),
 * ln_mean / ln_rstd [M] = the row statistics gd_layernorm_bwd* takes.  Same shapes as gd_adapter_fused_h. */
int gd_adapter_fused_h_ln(const float* x32, const void* w1, const void* w2, void* hidden, float* out32, const float* ln_w, const float* ln_b,
                          float ln_eps, void* ln16, float* ln_mean, float* ln_rstd, int M, int D, int bottleneck, void* stream);
int gd_adapter_fused_h(const float* x32, const void* w1, const void* w2, const void* gate_src, void* hidden, float* out32, void* out16,
                       const float* in_scale, const float* alpha_dev, const float* copy_scale, int M, int D, int bottleneck, void* stream);
int gd_tap_mean_norm_fwd_h(const void* const* grids, int ngrid, long bstride, int prefix, float* out, void* out16, float* inv_norm,
                           int B, int hw, int D, void* stream);      /* gd_tap_mean_norm_fwd on fp32 taps + the fp16 copy of the rows it writes */
int gd_layernorm_bwd_cast(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                          const float* dres, const float* dres2, float* dx, void* dx16, const float* scale_dev, int M, int D,
                          long ldd, long ldx, float dyscale, void* stream);
int gd_amax_scale(const float* in, long rows, int K, long ld_in, float target, float* scale3, unsigned* amax_slots, void* stream);
int gd_scale_from_amax(unsigned* amax_slots, float target, float* scale2, void* stream);
int gd_cast_f16_ex(const float* in, void* out, long rows, int K, long ld_in, float scale, const float* scale_dev, unsigned* range_counters, void* stream);
/* The tf32h block backward's LayerNorm pass (autograd of nn.LayerNorm inside timm's Block, frozen affine: dX only) with the device-side extras:
 * dy f32 or fp16 (dy_dtype), multiplied by *dy_scale_dev when given (an fp16 dy under the block's scale s: 1/s); dx16 (nullable) = fp16(sat(dx *
 * *cast_scale_dev)); amax_slots (nullable, 256 words): max |dx| bit patterns, the NEXT block's gradient scale through gd_scale_from_amax without
 * a pass of its own; range_counters (nullable, 128 words): saturated / below-normal-range partial sums of dx16 as in gd_cast_f16_ex. */
int gd_layernorm_bwd_ex(const void* dy, int dy_dtype, const float* dy_scale_dev, const float* x, const float* gamma, const float* mean,
                        const float* rstd, const float* dres, const float* dres2, float* dx, void* dx16, const float* cast_scale_dev,
                        unsigned* amax_slots, unsigned* range_counters, int M, int D, long ldd, long ldx, float dyscale, void* stream);
int gd_gemm_nt_scaled(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
                      int batch, long sA, long sW, long sC, int ab_dtype, int c_dtype, float alpha, const float* alpha_dev,
                      const float* bias, const float* lora_t, const float* lora_b, int lora_rt, void* preact, long ldp,
                      int act, const void* dact_src, long ldd, int dact, const void* residual, long ldr, int accumulate,
                      void* stream);

/* flat_allreduce: the data-parallel step's one exchange — the sum over ranks of the flat fp32 gradient buffer — on RCCL over
 * xGMI (replaces Lightning DDP's bucketed all-reduce, src/main.py:147-151).  RCCL is bound at run time (dlopen, the copy the
 * process already carries first; 2.x ABI checked through ncclGetVersion): the calls fail with a message where it is absent.
 * gd_comm_rccl_version: the bound library's version code (or < 0).  gd_comm_unique_id: rank 0 draws the 128-byte id, the
 * caller hands it to every rank (any transport); gd_comm_init: collective over the nranks processes (one per GPU, device
 * already selected); gd_flat_allreduce: in place on buf[n], algo 0 = one ncclAllReduce, algo 1 = ncclReduceScatter +
 * ncclAllGather on the rank's n / nranks slice (n % nranks == 0).  RCCL picks the schedule in both forms. */
int gd_comm_rccl_version(void);
int gd_comm_unique_id(void* out128);
int gd_comm_init(void** comm, int nranks, int rank, const void* id128);
int gd_comm_destroy(void* comm);
int gd_flat_allreduce(void* comm, float* buf, long n, int nranks, int rank, int algo, void* stream);

/* Teacher -> target glue on the device.
 * gd_unproject_depth: vggt/utils/geometry.py:12-110 unproject_depth_map_to_point_map (depth [S,H,W], extrinsic [S,3,4],
 *   intrinsic [S,3,3] -> world points [S,H,W,3]).
 * gd_coview_masks: utils/functions.py:425-472 get_coview_masks, batched over P pairs (point maps [P,H,W,3]).
 * gd_nms_keypoints: utils/functions.py:475-507 sample_keypoints_nms up to the ordered candidate list: idx [P,cap]
 *   row-major linear indices of the local maxima, count [P]; keep_ws [P,H,W] scratch.
 * gd_nn_argmax: mast3r/fast_nn.py:11-62 bruteforce NN with dist='dot' (queries [Nq,D], database [Nb,D], D <= 32,
 *   optional active mask): idx [Nq] int32 (inactive entries untouched); key_ws [Nq] u64 scratch.
 * gd_point_cloud_to_depth: utils/functions.py:218-259 (points [P,Np,3], K [P,3,3] -> depth [P,h,w]); cnt_ws [P,h,w]. */
int gd_unproject_depth(const float* depth, const float* extrinsic, const float* intrinsic, float* out, int S, int H,
                       int W, void* stream);
int gd_coview_masks(const float* pm1, const float* pm2, const float* K1, const float* E1, const float* K2,
                    const float* E2, unsigned char* m1, unsigned char* m2, int P, int H, int W, void* stream);
int gd_nms_keypoints(const unsigned char* mask, const float* conf, int min_distance, unsigned char* keep_ws, int* idx,
                     int* count, int P, int H, int W, int cap, void* stream);
int gd_nn_argmax(const float* queries, const float* database, const unsigned char* active, int* idx,
                 unsigned long long* key_ws, int Nq, int Nb, int D, void* stream);
int gd_point_cloud_to_depth(const float* points, const float* K, float* depth, float* cnt_ws, int P, int Np, int w,
                            int h, void* stream);

/* post_process_depth (utils/functions.py:262-345) on P rasterised depth maps [P,H,W]: max-pool closing, two hole-filling
 * passes, median (kernel_size in {3, 5}), bilateral, guided filter (guidance = the bilateral map, input = the median map, as the
 * call site passes them), 3-sigma outlier replacement, joint bilateral.  The four kornia filters are restated from kornia's
 * published definitions (kornia is not vendored in the reference: parity unpinned).  workspace: 6 P H W floats. */
size_t gd_post_process_depth_workspace_bytes(int P, int H, int W);
int gd_post_process_depth(const float* depth, float* out, int P, int H, int W, int kernel_size, int bilateral_d,
                          float sigma_color, float sigma_space, int guided_r, float guided_eps, void* workspace, void* stream);

/* In-place 2-D RoPE of the frozen MASt3R teacher: replaces curope.rope_2d(tokens, positions, base, fwd)
 * (dust3r/croco/models/curope/curope.cpp:49-69, kernels.cu:17-82).  tokens [B,N,H,D] (token stride ld_tok elements),
 * positions int64 [B,N,2] (y,x); fwd = +F0 forward / -F0 backward. */
int gd_rope_2d(void* tokens, const long* positions, int B, int N, int H, int D, long ld_tok, float base, float fwd,
               int dtype, void* stream);

/* VGGT teacher -> distillation target: head- (and, through weight / accumulate, block-) averaged cross-view attention maps.
 * Replaces the `return_attn` branch of vggt/layers/attention.py:51-85 (two [B,H,n,n] softmax blocks per global block,
 * torch.cat on dim 0) + the head mean of src/finetune_timm_vggt.py:390-392 + the block mean of vggt/models/aggregator.py:273,
 * without ever materialising a per-head map.  q, k: [B, H, N, 64] (f32 | bf16, after q/k-norm and RoPE), N even;
 * n = N/2 - prefix; out [2B, n, n] fp32: rows 0..B-1 = softmax(q[prefix:N/2] k[N/2+prefix:]^T scale / temperature),
 * rows B..2B-1 the mirrored block; out = (accumulate ? out : 0) + weight * sum_h softmax_h.  workspace: row statistics. */
size_t gd_cross_view_attn_workspace_bytes(int B, int H, int N, int prefix);
int gd_cross_view_attn(const void* q, const void* k, float* out, int B, int H, int N, int prefix, int head_dim,
                       float scale, float temperature, float weight, int accumulate, int dtype, void* workspace,
                       void* stream);

/* MASt3R teacher -> distillation target `tgt_attn_map` (dust3r/dust3r/model.py:346-366) from the per-layer reciprocity
 * averages recip_scores[L][B][N1][N2] = (mean_h tgt_l + (mean_h src_l)^T) / 2 of the decoder's raw cross-attention scores:
 * softmax(. / temperature) over keys, column 0 := min of the layer's map, mean over layers -> out [B][N1][N2].
 * workspace: L*B*N1 floats.  N2 <= 2048. */
int gd_mast3r_attn_target(const float* recip_scores, int L, int B, int N1, int N2, float temperature, float* out,
                          float* workspace, void* stream);

/* ---- input pipeline on device ------------------------------------------------------------------------
 * vggt/utils/load_fn.py:12-146 `load_and_preprocess_images` after the file decode: PIL Image.resize(..., BICUBIC) on uint8
 * RGB (:87), ToTensor (:88), centre crop (:91-93) / white padding (:96-111, :121-139).  The resampler is Pillow's
 * (src/libImaging/Resample.c; third-party, restated): gd_pil_resample_ksize / gd_pil_resample_coeffs compute its
 * per-output-pixel window start, tap count and 22-bit fixed-point coefficients on the HOST (xmin, count: [out_size];
 * coeffs: [out_size * ksize]); gd_pil_resize_bicubic_u8 runs the horizontal then the vertical pass on the DEVICE
 * (src [H,W,C] uint8, tmp [H,new_w,C] uint8, dst [new_h,new_w,C] uint8; the coefficient arrays are device pointers);
 * bit-exact against Pillow (fixture G20). */
int gd_pil_resample_ksize(int in_size, int out_size);
int gd_pil_resample_coeffs(int in_size, int out_size, int* xmin_out, int* count_out, int* coeffs_out);
int gd_pil_resize_bicubic_u8(const unsigned char* src, unsigned char* tmp, unsigned char* dst, int H, int W, int C,
                             int new_h, int new_w, const int* xmin_h, const int* count_h, const int* coeffs_h,
                             int ksize_h, const int* xmin_v, const int* count_v, const int* coeffs_v, int ksize_v,
                             void* stream);
/* dst[c][y][x] (float32 CHW canvas H x W) = src[y - pad_top + crop_y0][x - pad_left][c] / 255 inside the crop_h x w window,
 * `fill` outside (load_fn.py:88-111,121-139). */
int gd_u8_to_chw_float(const unsigned char* src, float* dst, int h, int w, int C, int crop_y0, int crop_h, int pad_top,
                       int pad_left, int H, int W, float fill, void* stream);
/* data_utils/dataset_mast3r_scannetpp.py:185-207 colour augmentation on uint8 RGB [n,H,W,3]: albumentations ColorJitter
 * (factors [n][4] = brightness, contrast, saturation, hue; order [n][4] = permutation of 0..3, -1 = skip; gray_sum_ws: n
 * uint64) and GaussianBlur (ksize [n] in {0,3,5,7}; tmp: n*H*W*3 floats).  albumentations / OpenCV are absent: restated from
 * their published definitions, PARITY UNPINNED. */
int gd_color_jitter_u8(const unsigned char* src, unsigned char* dst, int n, int H, int W, const float* factors,
                       const int* order, unsigned long long* gray_sum_ws, void* stream);
int gd_gaussian_blur_u8(const unsigned char* src, float* tmp, unsigned char* dst, int n, int H, int W, const int* ksize,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif
