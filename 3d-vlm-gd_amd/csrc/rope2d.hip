// 2-D rotary position embedding, in place, on tokens[B,N,H,D] — the HIP twin of the reference's only native
// kernel (dust3r/croco/models/curope/kernels.cu:17-82, CPU form curope.cpp:11-47), used by the frozen MASt3R
// teacher's encoder/decoder attention.  D = 4 quarters [u_Y, v_Y, u_X, v_X], Q = D/4:
//     theta = pos[b,n,axis] * (fwd / base^(q/Q));  u' = u cos - v sin;  v' = v cos + u sin
// fwd = +F0 forward, -F0 backward (same kernel).  Memory-bound (2*B*N*H*D*elsize bytes): one wave per token; each lane
// owns (u,v) pairs across heads, so cos/sin are computed once per lane per token and reused for every head.
#include "gd_common.h"

template <typename T>
__global__ __launch_bounds__(256) void rope2d_kernel(T* tokens, const long* pos, int BN, int H, int D, long ld_tok,
                                                     float base, float fwd) {
    const int lane = threadIdx.x & 63, tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= BN) return;
    const int Q = D / 4, half = D / 2;
    T* t = tokens + (long)tok * ld_tok;
    for (int pr = lane; pr < half; pr += 64) {          // pair index inside a head: axis = pr / Q, q = pr % Q
        const int axis = pr / Q, qi = pr % Q;
        const float freq = (float)pos[(long)tok * 2 + axis] * (fwd / powf(base, (float)qi / (float)Q));
        float sn, cs;
        sincosf(freq, &sn, &cs);
        const int iu = axis * 2 * Q + qi, iv = iu + Q;
        for (int h = 0; h < H; ++h) {
            const float u = to_f32<T>(t[h * D + iu]), v = to_f32<T>(t[h * D + iv]);
            t[h * D + iu] = from_f32<T>(u * cs - v * sn);
            t[h * D + iv] = from_f32<T>(v * cs + u * sn);
        }
    }
}

// Mirrors rope_2d's checks (curope.cpp:54-59): tokens [B,N,H,D] (token stride ld_tok elements, H*D contiguous),
// positions int64 [B,N,2].
extern "C" int gd_rope_2d(void* tokens, const long* positions, int B, int N, int H, int D, long ld_tok, float base,
                          float fwd, int dtype, void* stream) {
    GD_REQUIRE(B > 0 && N > 0 && H > 0 && D > 0, "gd_rope_2d: bad shape B=%d N=%d H=%d D=%d", B, N, H, D);
    GD_REQUIRE(D % 4 == 0, "gd_rope_2d: tokens.shape[3] must be a multiple of 4 (got %d)", D);
    GD_REQUIRE(ld_tok >= (long)H * D, "gd_rope_2d: token stride smaller than H*D");
    GD_REQUIRE(dtype == GD_F32 || dtype == GD_BF16, "gd_rope_2d: bad dtype %d", dtype);
    const int BN = B * N;
    if (dtype == GD_BF16)
        hipLaunchKernelGGL(rope2d_kernel<bf16>, dim3(gd_cdiv(BN, 4)), dim3(256), 0, (hipStream_t)stream, (bf16*)tokens, positions, BN, H, D, ld_tok, base, fwd);
    else
        hipLaunchKernelGGL(rope2d_kernel<float>, dim3(gd_cdiv(BN, 4)), dim3(256), 0, (hipStream_t)stream, (float*)tokens, positions, BN, H, D, ld_tok, base, fwd);
    GD_LAUNCH_OK();
    return 0;
}
