/* gd_hip.h — C ABI of the MI355X (gfx950) geometric-distillation kernels.
 *
 * Drop-in boundary for the hot path of kaist-cvml/3d-vlm-gd.  The reference's
 * own native-op convention is dust3r/croco/models/curope/curope.cpp:49-69 (plain functions on
 * caller-owned tensors, argument checks that raise, current stream, no global state); every entry
 * point here follows it with plain pointers and sizes:
 *   - returns 0 on success, negative on error; gd_last_error() holds the message (thread-local);
 *   - never allocates: outputs and workspaces are caller-owned (…_workspace_bytes() queries);
 *   - launches on the hipStream_t passed as `stream` (void*), stateless, thread-safe;
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

const char* gd_last_error(void);
int gd_abi_version(void);

/* C[M,N] = epilogue(alpha * A[M,K] . W[N,K]^T); replaces torch nn.Linear / torch.bmm on the student path
 * (timm VisionTransformer qkv/proj/fc1/fc2, utils/model.py:57-71 LoRA, :7-25 Adapter; src/finetune_timm_vggt.py:516-517).
 * Epilogue order: +bias[N](f32) -> +lora_t[M,rt].lora_b[rt,N](f32, rt<=8) -> store preact -> act(1 GELU erf,2 ReLU)
 * -> *act'(dact_src) (1 dGELU(src), 2 src>0) -> +residual -> +C (accumulate).  batch>1: grid over batch strides. */
int gd_gemm_nt(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
               int batch, long sA, long sW, long sC, int ab_dtype, int c_dtype, float alpha,
               const float* bias, const float* lora_t, const float* lora_b, int lora_rt, void* preact, long ldp,
               int act, const void* dact_src, long ldd, int dact, const void* residual, long ldr, int accumulate,
               void* stream);

/* G[N,K](f32) += alpha * Y[M,N]^T . X[M,K]; weight gradients of the trainable LoRA / Adapter / refine_conv
 * tensors (autograd of utils/model.py:7-71, src/finetune_timm_vggt.py:146 refine_conv). */
int gd_gemm_tn(const void* Y, const void* X, float* G, int M, int N, int K, long ldy, long ldx, long ldg,
               int y_dtype, int x_dtype, float alpha, void* stream);

/* Multi-head self-attention, head_dim 64, flash-style (no N x N matrix): replaces F.scaled_dot_product_attention in
 * timm Attention.forward (same arithmetic as vggt/layers/attention.py:51-71).  qkv [B,N,3,H,64] packed
 * QKV-GEMM output, o [B,N,H*64], lse [B,H,N] f32 (natural log); backward writes dqkv [B,N,3,H,64];
 * delta_ws [B,H,N] f32 scratch. */
int gd_attention_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int head_dim, float scale, int dtype,
                     void* stream);
int gd_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, float* delta_ws,
                     int B, int N, int H, int head_dim, float scale, int dtype, void* stream);

/* Dense cost-volume KL for P pairs, fused: calculate_cost_loss (src/finetune_timm_vggt.py:488-533 variant 0,
 * src/finetune_timm_mast3r.py:504-540 variant 1) = F.normalize + bmm x2 + softmax + get_masked_patch_cost
 * (utils/functions.py:402-422) + kl_divergence_map (utils/losses.py:5-15).
 * f1,f2 [P,hw,C] raw features; t1,t2 [P,hw,hw] f32 teacher maps; m1,m2 [P,hw] uint8 row masks;
 * loss [P] f32; stats [P,2,hw,4] f32 (saved for the backward). */
size_t gd_cost_volume_kl_workspace_bytes(int P, int hw, int C, int dtype, int backward);
int gd_cost_volume_kl_fwd(const void* f1, const void* f2, const float* t1, const float* t2, const unsigned char* m1,
                          const unsigned char* m2, int P, int hw, int C, int variant, int dtype, float* loss,
                          float* stats, void* workspace, void* stream);
int gd_cost_volume_kl_bwd(const void* f1, const void* f2, const float* t1, const float* t2, const unsigned char* m1,
                          const unsigned char* m2, int P, int hw, int C, int dtype, const float* gloss,
                          const float* stats, void* df1, void* df2, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif
