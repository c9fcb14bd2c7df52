// Persistent 256 x 256 gemm_nt for gfx950: one 512-thread block per CU walks its XCD's chunk of the tile space.
//
// What it changes against gemm_nt_kernel<T,2,4,8> (same LDS-DMA ring, same swizzle, same wave tiling 2 x 4 of 128 x 64):
//   * PERMUTED W rows: the W rows of a wave's 64 columns are DMA'd into the LDS image in the order n = 4*fr + j
//     (nperm64), so after the MFMAs lane (g, fr) holds, for row 4g+r of m-tile i, the FOUR CONSECUTIVE columns
//     4fr..4fr+3 (one per n-tile j).  An output item (i, r) is then one `buffer_store_dwordx2` whose 16 consecutive
//     lanes cover one 128-byte row segment (4 rows per instruction): measured 40 B/clk per CU against 18 B/clk for
//     stores whose lanes land in 16 different rows (tools/micro/store_pattern.hip).  The whole epilogue (bias,
//     activation, gating, residual, stores) runs straight from the accumulators — no LDS staging, no barriers.
//   * the next tile's first two K stages, its LoRA tiles and its bias slice are DMA'd BEFORE (or in the middle of) the
//     epilogue of the current tile, so the pipeline-fill latency of a tile hides under the previous tile's epilogue,
//     and there is no block relaunch between tiles.
//   * the LoRA rank update is applied at the START of a tile (first MFMA chunk), bias comes from LDS.
#pragma once
#include "gemm_tile.h"
#include "gemm_frag.h"
#include <type_traits>

// Compile-time epilogue: SIDE 0 none | 1 v *= dGELU(dact_src) | 2 v += residual | 3 v *= dact_src (bf16 side tensor,
// prefetched); ACT 0 none | 1 GELU | 2 ReLU; PREACT 0 none | 1 store v before the activation | 3 store an fp16 copy of the FINAL v times *copy_scale (p.preact: the tf32h engine's next left operand) | 2 store GELU'(v) (the
// backward then only multiplies: SIDE 3 — recomputing dGELU from the stored pre-activation cost 20 k cycles per tile of
// the fc1 backward GEMM); CF32 C / preact are f32 (else bf16).
// (With these as run-time flags the 16-item unrolled epilogue was ~160 scalar branches per tile: 4 k cycles of a
// 45 k-cycle tile with nothing to do.)  Other combinations stay on gemm_nt_kernel.
// Two main-loop schedule experiments of round 3, measured against this kernel in one process (tools/bench_kernels.py gemm_ab,
// profiles/r03_gemm_variants.txt) and removed: (1) the eight DMA pieces of stage kt+2 issued one by one between the eight MFMAs
// that follow the stage barrier: -1 % on every shape; (2) waves 4-7 issuing their pieces half a chunk into K step kt+1 (so that the
// two waves of a SIMD alternate between vector-memory issue and MFMA work): -4...-12 %.  Both move DMA issue LATER: the loop is bound
// by the landing latency of the one stage in flight (64 KB / ~1.5 us = the 43 GB/s per CU it sustains), not by an issue bubble.
// ANAT (anatomy builds, tools/bench_kernels.py gemm_anat; never the product path): 1 = no operand DMA in the main loop (the LDS-read +
// MFMA + barrier loop alone), 2 = no MFMAs (DMA + LDS reads + barriers), 3 = neither LDS reads nor MFMAs (the DMA ring alone), 4 = no C stores (everything else of the epilogue stays).
// Instantiated only in -DGD_GEMM_ANATOMY builds (gemm.hip); round-3 results: profiles/r03_gemm_anatomy.txt.
// COUT (with CF32; 0 = C has the side tensors' type): 1 = C leaves as the three-plane bf16 operand split of the f32 result — [hi | lo | hi] over 3N columns of row stride
// ldc (gd_split3 'a' layout), the A operand of the next tf32x GEMM — instead of f32 followed by a gd_split3 pass; side / preact stay f32.
template <typename T, int SIDE, int ACT, int PREACT, bool CF32, int ANAT = 0, int COUT = 0>
__global__ __launch_bounds__(512) void gemm_nt_persist_kernel(GemmNtParams p) {
    constexpr int NWN = 4, WMT = 8, NW = 8, BM = 256, BN = 256;
    constexpr int KS = 128, RPP = 1024 / KS, NSLOT = 2;      // bytes of an operand row per ring stage; rows per 1-KB DMA piece; ring slots
    constexpr int ABYTES = BM * KS, STAGE = (BM + BN) * KS, APW = BM / RPP / NW, BPW = BN / RPP / NW;
    constexpr int SDEP = CF32 ? GD_SDEP32 : 16;   // side-input prefetch depth (items per lane in flight: 8 bytes each for bf16 side tensors, 16 for f32)
    constexpr int ssz = CF32 ? 4 : 2;     // element size of the side tensor (dact_src / residual): the dtype of C
    constexpr int LORA_OFF = NSLOT * STAGE, BIAS_OFF = LORA_OFF + (BM + BN) * 32;
    __shared__ __attribute__((aligned(16))) char smem[BIAS_OFF + 2 * BN * 4];
    typedef typename Mma<T>::Frag Frag;
    constexpr int KPL = sizeof(Frag) / sizeof(T);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int fr = lane & 15, g = lane >> 4;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    const long batch = blockIdx.y;
    const char* Ab = (const char*)p.A + batch * p.sA * (long)sizeof(T);
    const char* Wb = (const char*)p.W + batch * p.sW * (long)sizeof(T);
    const long lda_b = p.lda * (long)sizeof(T), ldw_b = p.ldw * (long)sizeof(T);
    const int nk = p.K * (int)sizeof(T) / KS;      // ring stages per tile
    static_assert(COUT == 0 || (COUT == 1 && CF32), "split output comes from f32 values");
    constexpr bool CSPLIT = COUT == 1;
    // fp16 results (tf32h engine) carry 11 significant bits (4.9e-4 relative): the sigmoid-form fit's 2.5e-5 absolute error on y Phi(y) and 1.1e-4 on the
    // derivative sit below that, and the erf form costs 60-70 us per fc1 launch — set to true for the erf form
    constexpr bool XGELU = false;
    constexpr int t16 = std::is_same<T, f16>::value ? GD_F16 : GD_BF16;      // dtype code of a 16-bit C
    constexpr int cdt = CF32 ? GD_F32 : t16, csz = CF32 ? 4 : 2;
    constexpr int ccsz = COUT ? 2 : csz;   // element size of C itself (preact / side tensors keep csz / ssz)
    char* Cb = (char*)p.C + batch * p.sC * (long)ccsz;
    const bool lora = p.lora_t != nullptr;
    const float alpha = p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha;      // gd_gemm_nt_scaled: the operands' power-of-two scale, undone here
    const float ia = 1.0f / alpha;
    const float cscale = (PREACT == 3 && p.copy_scale) ? *p.copy_scale : 1.0f;

    // DMA sources: wave-uniform tile base (SGPRs) + one 32-bit byte offset per lane and 1-KB piece.  LDS row rho of
    // piece i is (wave*APW + i)*8 + (lane>>3); rows past the matrix edge are clamped (read, never stored).
    const char* abase_t;
    const char* wbase_t;
    unsigned aoff[APW], woff[BPW];
    // K-steps are walked in a per-tile ROTATED order (p.k_rot): the tiles that share an A row panel (same tm) or a W panel
    // (same tn) run at the same time on one XCD; in lockstep they all miss the L2 on the same 128-byte K slice at the same
    // moment, rotated one tile fetches a slice and the others hit it.
    int krot = 0;
    auto set_tile = [&](int tm, int tn) {
        krot = p.k_rot ? (tn * p.k_rot + tm) % nk : 0;
        abase_t = Ab + (long)tm * BM * lda_b;
        wbase_t = Wb + (long)tn * BN * ldw_b;
        const int av = min(BM, p.M - tm * BM) - 1, wv = min(BN, p.N - tn * BN) - 1;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int row = (wave * APW + i) * RPP + (lane >> 3);
            aoff[i] = (unsigned)(min(row, av) * (int)lda_b + ((lane & 7) ^ swz(row)) * 16);
        }
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const int row = (wave * BPW + i) * RPP + (lane >> 3);
            woff[i] = (unsigned)(min(nperm64(row), wv) * (int)ldw_b + ((lane & 7) ^ swz(row)) * 16);
        }
    };
    auto issue = [&](int kt0, int buf) {
        if (ANAT == 1) return;
        const int kt = kt0 + krot >= nk ? kt0 + krot - nk : kt0 + krot;
        char* sA = smem + buf * STAGE;
        char* sB = sA + ABYTES;
#pragma unroll
        for (int i = 0; i < APW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(abase_t + kt * KS + aoff[i]),
                                             (__attribute__((address_space(3))) void*)(sA + (wave * APW + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase_t + kt * KS + woff[i]),
                                             (__attribute__((address_space(3))) void*)(sB + (wave * BPW + i) * 1024), 16, 0, 0);
    };
    // f32 LoRA tiles T[BM][8], B[8][BN] and the bias slice [BN] (slot = tile parity: the epilogue still reads the
    // current slice while the next one lands)
    auto issue_side = [&](int tm, int tn, int slot) {
        if (lora) {
            char* lT = smem + LORA_OFF;
            char* lB = lT + BM * 32;
            const int trow = min(tm * BM + (tid >> 1), p.M - 1);
            const int bk = tid / (BN / 4), bc = min(tn * BN + (tid % (BN / 4)) * 4, p.N - 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.lora_t + (long)trow * 8 + (tid & 1) * 4),
                                             (__attribute__((address_space(3))) void*)(lT + wave * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.lora_b + (long)bk * p.N + bc),
                                             (__attribute__((address_space(3))) void*)(lB + wave * 1024), 16, 0, 0);
        }
        if (p.bias && wave == 0) {
            const int bc = min(tn * BN + lane * 4, p.N - 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + bc),
                                             (__attribute__((address_space(3))) void*)(smem + BIAS_OFF + slot * BN * 4), 16, 0, 0);
        }
    };

    const int abase = (wm * WMT * 16 + fr) * KS, bbase = ABYTES + (wn * 64 + fr) * KS;
    const int sa = swz(fr);
    const unsigned lds0 = lds_off(smem);
    constexpr bool pre = SIDE != 0;
    const void* side_src = SIDE == 2 ? p.residual : p.dact_src;
    const long side_ld = SIDE == 2 ? p.ldr : p.ldd;

    // tile order inside an XCD's contiguous chunk: row-panel-major (group_m = 1: the tiles_n tiles of a row panel, then the next panel), or
    // GROUPED — group_m row panels per W panel before the next W panel (the 32 tiles an XCD works on at one time then share group_m A panels and
    // 32 / group_m W panels instead of ~3 A panels and the whole W: a smaller working set in the 4 MB L2)
    auto tile_of = [&](int wg_, int& tm_, int& tn_) {
        if (p.group_m <= 1) { tm_ = wg_ / tiles_n; tn_ = wg_ % tiles_n; return; }
        const int per = p.group_m * tiles_n, gid = wg_ / per, first = gid * p.group_m;
        const int gsz = min(tiles_m - first, p.group_m), r = wg_ - gid * per;
        tm_ = first + r % gsz; tn_ = r / gsz;
    };
    int t = blockIdx.x, slot = 0;
    if (t >= ntiles) return;
    int wg = xcd_remap(t, ntiles);
    int tm, tn;
    tile_of(wg, tm, tn);
    auto prologue = [&](int tm_, int tn_, int slot_) {
        set_tile(tm_, tn_);
        issue(0, 0);
        issue_side(tm_, tn_, slot_);
        if (nk > 1) issue(1, 1);
    };
    prologue(tm, tn, slot);
    // VMEM instructions every wave issues unconditionally in one epilogue (buffer ops, range-checked by the hardware)
    int after = 0;   // of those, how many were issued after this tile's stage-1 DMA (0 for the block's first tile)

    GD_PROBE_DECL(unsigned long long pc0 = 0; unsigned long long pw = 0; unsigned long long pm = 0; unsigned long long pe = 0; unsigned long long pn = 0; unsigned long long pd = 0; unsigned long long pb = 0;)
    for (;;) {
        GD_PROBE(pc0 = __builtin_amdgcn_s_memtime();)
        f32x4 acc[WMT][4];
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // stage 0, LoRA tiles and bias of this tile have landed
        wait_vm_le(after + (nk > 1 ? APW + BPW : 0));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        GD_PROBE({ const unsigned long long c = __builtin_amdgcn_s_memtime(); pw += c - pc0; pc0 = c; })
        if (lora) {
            // bf16 operands with f32 results (the tf32x engine): the chunk's 24 spare k slots carry the split-precision terms — lane groups
            // g = 0, 1, 2 all read the same eight k rows and contribute t_hi b_hi, t_lo b_hi, t_hi b_lo (as gd_split3 does for the main product)
            constexpr bool L3 = CF32 && std::is_same<T, bf16>::value && KPL == 8;
            const bool live = L3 ? g < 3 : KPL * g < 8;
            // B tile [8][BN] f32: this lane's k rows start at (KPL*g)&7; its columns for the n-tiles j = 0..3 are the four
            // consecutive floats 64*wn + 4*fr + j: one 16-byte read per k row
            const unsigned baddr = lds_off(smem + LORA_OFF + BM * 32) + 4 * (((KPL * g) & 7) * BN + wn * 64 + 4 * fr);
            const unsigned taddr = lds_off(smem + LORA_OFF) + 4 * ((wm * WMT * 16 + fr) * 8 + ((KPL * g) & 7));
            Frag bf[4];
            {
                f32x4 q0, q1, q2, q3, q4 = {0.f, 0.f, 0.f, 0.f}, q5 = q4, q6 = q4, q7 = q4;
                GD_DSR128(q0, baddr, 0); GD_DSR128(q1, baddr, 1024); GD_DSR128(q2, baddr, 2048); GD_DSR128(q3, baddr, 3072);
                if (KPL == 8) { GD_DSR128(q4, baddr, 4096); GD_DSR128(q5, baddr, 5120); GD_DSR128(q6, baddr, 6144); GD_DSR128(q7, baddr, 7168); }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
                const f32x4 qs[8] = {q0, q1, q2, q3, q4, q5, q6, q7};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < KPL; ++k) {
                        const float x = live ? qs[k][j] : 0.f;
                        bf[j][k] = (L3 && g == 2) ? (T)(x - (float)(T)x) : (T)x;
                    }
            }
#define GD_LT(i, IOFF)                                                                                             \
            {                                                                                                          \
                f32x4 t0, t1 = {0.f, 0.f, 0.f, 0.f};                                                                   \
                GD_DSR128(t0, taddr, IOFF);                                                                            \
                if (KPL == 8) GD_DSR128(t1, taddr, IOFF + 16);                                                         \
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0), "+v"(t1));                                             \
                const float ts[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};                          \
                Frag af;                                                                                               \
                _Pragma("unroll") for (int k = 0; k < KPL; ++k) {                                                      \
                    const float x = live ? ia * ts[k] : 0.f;                                                           \
                    af[k] = (L3 && g == 1) ? (T)(x - (float)(T)x) : (T)x;                                              \
                }                                                                                                      \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::mma(af, bf[j], acc[i][j]);           \
            }
            GD_LT(0, 0) GD_LT(1, 512) GD_LT(2, 1024) GD_LT(3, 1536) GD_LT(4, 2048) GD_LT(5, 2560) GD_LT(6, 3072) GD_LT(7, 3584)
#undef GD_LT
        }
        const int co0 = ((g ^ sa) * 16), co1 = (((4 + g) ^ sa) * 16);
        FragHead P, Q;
        FragTail tl;
        constexpr bool AN_MM = ANAT == 0 || ANAT == 1 || ANAT == 4, AN_RD = AN_MM || ANAT == 2;
        if (AN_RD) frag_head_issue(P, lds0 + abase + co0, lds0 + bbase + co0);
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned sbo = lds0 + (kt & 1) * STAGE, nsbo = lds0 + ((kt + 1) & 1) * STAGE;
            if (AN_MM) chunk_rows05<T>(P, tl, sbo + abase + co0, acc); else if (ANAT == 2) chunk_reads_only(P, tl, sbo + abase + co0);
            if (AN_RD) frag_head_issue(Q, sbo + abase + co1, sbo + bbase + co1);
            if (AN_MM) chunk_rows67<T>(P, tl, acc);
            if (AN_MM) chunk_rows05<T>(Q, tl, sbo + abase + co1, acc); else if (ANAT == 2) chunk_reads_only(Q, tl, sbo + abase + co1);
            // every LDS read of stage kt is done: stage barrier (stage kt+1 landed, slot kt&1 free), then refill the slot
            GD_PROBE_DECL(unsigned long long pb0 = 0;) GD_PROBE(pb0 = __builtin_amdgcn_s_memtime();)
            if (kt == 0) wait_vm_le(after);   // stage 1 is older than the previous epilogue's stores: those may still drain
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // -DGD_GEMM_STAGE_PROBE builds only (the two s_memtime reads in the K loop cost 20 % even unarmed): the time this wave
            // spends between its last LDS read of stage kt and the stage barrier = DMA landing wait + barrier skew.  Measured
            // 0.40 of the main loop on every shape (87 680 x 3072 x 768: 14.0 k of 35.5 k cycles per tile; 4096^3: 76.7 k of 191.6 k)
            GD_PROBE(pb += __builtin_amdgcn_s_memtime() - pb0;)
            if (kt + 2 < nk) issue(kt + 2, kt & 1);
            if (kt + 1 < nk && AN_RD) frag_head_issue(P, nsbo + abase + co0, nsbo + bbase + co0);
            if (AN_MM) chunk_rows67<T>(Q, tl, acc);
        }
        const int ctm = tm, ctn = tn, cslot = slot;
        GD_PROBE({ const unsigned long long c = __builtin_amdgcn_s_memtime(); pm += c - pc0; pc0 = c; })
        // ---- epilogue from the accumulators.  Item (i, r) = row 16i + 4g + r of the wave tile, this lane's four columns
        // 4fr..4fr+3: one 8-byte (bf16) / 16-byte (f32) store, 16 consecutive lanes = one contiguous row segment.
        const int vrows = min(BM, p.M - ctm * BM);
        auto mk = [&](const void* base, long ld, int es) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)((char*)base + (long)ctm * BM * ld * es), (short)0,
                                                     (int)min((long)0x7fffffff, (long)vrows * ld * es), 0x00020000);
        };
        const __amdgpu_buffer_rsrc_t crs = mk(Cb, p.ldc, ccsz);
        constexpr int psz = PREACT == 3 ? 2 : csz;
        const __amdgpu_buffer_rsrc_t prs = mk(PREACT ? p.preact : Cb, PREACT ? p.ldp : p.ldc, PREACT ? psz : ccsz);
        const __amdgpu_buffer_rsrc_t srs = mk(pre ? side_src : (const void*)Cb, pre ? side_ld : p.ldc, pre ? ssz : ccsz);
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            const unsigned biasaddr = lds_off(smem + BIAS_OFF) + cslot * BN * 4 + (wn * 64 + fr * 4) * 4;
            GD_DSR128(bv, biasaddr, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv));
        }
        int rloc = wm * WMT * 16 + g * 4, col0 = ctn * BN + wn * 64 + fr * 4;
        asm volatile("" : "+v"(rloc), "+v"(col0));   // keep the per-item byte offsets out of the main loop's live range (LICM)
        const int OOB = 0x7ffffff0;   // beyond every num_records: the hardware drops the access
        const int ldc_i = (int)p.ldc, ldp_i = (int)p.ldp, lds_i = (int)side_ld;   // (host: 256 * ld * size < 2^31)
        const bool cok = col0 < p.N;
        // byte offsets of item 0; item (i, r) adds the SCALAR (16 i + r) * ld * size — one v_add_u32 per access instead of a
        // per-item 32-bit multiply (quarter rate) + select.  A lane past N keeps OOB as its base: OOB + (< 2^31) stays >= every
        // num_records as an unsigned offset.
        const int cbase = cok ? (rloc * ldc_i + col0) * ccsz : OOB;
        const int pbase = cok ? (rloc * ldp_i + col0) * psz : OOB;
        const int sbase = cok ? (rloc * lds_i + col0) * ssz : OOB;
        constexpr int NITEM = 4 * WMT;   // idx = 4 i + r
        typedef typename std::conditional<CF32, gd_u32x4, gd_u32x2>::type SideReg;    // four side values of an item: f32 or bf16
        SideReg sd[SDEP] = {};
        auto side_load = [&](int idx) {
            const int off = sbase + ((idx >> 2) * 16 + (idx & 3)) * ssz * lds_i;
            if constexpr (CF32) sd[idx % SDEP] = __builtin_amdgcn_raw_buffer_load_b128(srs, off, 0, GD_PERSIST_SIDE_AUX);
            else sd[idx % SDEP] = __builtin_amdgcn_raw_buffer_load_b64(srs, off, 0, GD_PERSIST_SIDE_AUX);
        };
        auto side_vals = [&](int idx, float (&x)[4]) {
            if constexpr (CF32) {
                const f32x4 q = __builtin_bit_cast(f32x4, sd[idx % SDEP]);
                x[0] = q[0]; x[1] = q[1]; x[2] = q[2]; x[3] = q[3];
            } else {
                if constexpr (std::is_same<T, f16>::value) {
                    const f16x4 q = __builtin_bit_cast(f16x4, sd[idx % SDEP]);
                    x[0] = (float)q[0]; x[1] = (float)q[1]; x[2] = (float)q[2]; x[3] = (float)q[3];
                } else {
                    const bf16x4 q = __builtin_bit_cast(bf16x4, sd[idx % SDEP]);
                    x[0] = (float)q[0]; x[1] = (float)q[1]; x[2] = (float)q[2]; x[3] = (float)q[3];
                }
            }
        };
        if (pre) {
#pragma unroll
            for (int idx = 0; idx < SDEP; ++idx) side_load(idx);
        }
        // ---- the ring is free: the next tile's pipeline starts before (SIDE 0) or in the middle of (SIDE 1/2) this
        // tile's epilogue.  VMEM retires in issue order, so a side load younger than the DMA could only be consumed once the
        // whole prefetch has landed: with a side tensor ALL side loads go out first (SDEP up front, the rest while items
        // 0..DMA_AT-1 are processed) and the DMA follows at item DMA_AT.
        constexpr int DMA_AT = pre ? NITEM - SDEP : 0;
        t += gridDim.x;
        const bool more = t < ntiles;
        after = (NITEM - DMA_AT) * ((CSPLIT ? 3 : 1) + (PREACT ? 1 : 0));   // epilogue VMEM instructions younger than the stage-1 DMA
#pragma unroll
        for (int idx = 0; idx < NITEM; ++idx) {
            const int i = idx >> 2, r = idx & 3;
            if (idx == DMA_AT) {
                if (more) {
                    wg = xcd_remap(t, ntiles);
                    tile_of(wg, tm, tn); slot ^= 1;
                    prologue(tm, tn, slot);
                }
                asm volatile("" ::: "memory");   // nothing younger may be hoisted above the DMA: `after` counts on it
                GD_PROBE({ const unsigned long long c = __builtin_amdgcn_s_memtime(); pd += c - pc0; pc0 = c; })
            }
            const int coff = cbase + (i * 16 + r) * ccsz * ldc_i, poff = pbase + (i * 16 + r) * psz * ldp_i;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf(alpha, acc[i][j][r], bv[j]);
            if (PREACT == 1) bst4_aux<GD_PERSIST_STORE_AUX>(prs, poff, cdt, v);
            if (ACT == 1 && PREACT == 2) {   // GELU and its derivative from one shared exponential; the derivative is what is stored
                float dv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float Phi, dPhi;
                    if (CF32 || XGELU) { float ex; gelu_parts(v[j], Phi, ex); dPhi = ex * 0.39894228040143268f; }
                    else gelu_sig_parts(v[j], Phi, dPhi);       // bf16 outputs: the sigmoid-form fit (gd_common.h)
                    dv[j] = fmaf(v[j], dPhi, Phi);
                    v[j] *= Phi;
                }
                bst4_aux<GD_PERSIST_STORE_AUX>(prs, poff, cdt, dv);
            } else if (ACT == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (CF32 || XGELU) ? gelu_f(v[j]) : gelu_sig(v[j]);
            } else if (ACT == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (SIDE != 0) {
                float x[4];
                side_vals(idx, x);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (SIDE == 1) v[j] *= CF32 ? dgelu_f(x[j]) : dgelu_fast(x[j]);
                    if (SIDE == 2) v[j] += x[j];
                    if (SIDE == 3) v[j] *= x[j];
                }
            }
            if (PREACT == 3) {
                const float c16[4] = {v[0] * cscale, v[1] * cscale, v[2] * cscale, v[3] * cscale};
                bst4_aux<GD_PERSIST_STORE_AUX>(prs, poff, GD_F16, c16);
            }
            if (CSPLIT) {
                float lo[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) lo[j] = v[j] - (float)(bf16)v[j];
                const int plane = p.N * 2;
                bst4_aux<GD_PERSIST_STORE_AUX>(crs, coff, GD_BF16, v);
                bst4_aux<GD_PERSIST_STORE_AUX>(crs, coff + plane, GD_BF16, lo);
                bst4_aux<GD_PERSIST_STORE_AUX>(crs, coff + 2 * plane, GD_BF16, v);
            } else
            if (ANAT != 4) bst4_aux<GD_PERSIST_STORE_AUX>(crs, coff, cdt, v);
            else asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
            if (pre && idx + SDEP < NITEM) side_load(idx + SDEP);
        }
        GD_PROBE({ const unsigned long long c = __builtin_amdgcn_s_memtime(); pe += c - pc0; pn += 1; })
        if (!more) break;
    }
    GD_PROBE(if (tid == 0) {
        atomicAdd(p.probe + 0, pw); atomicAdd(p.probe + 1, pm); atomicAdd(p.probe + 2, pe); atomicAdd(p.probe + 3, pn); atomicAdd(p.probe + 4, pd); atomicAdd(p.probe + 5, pb);
    })
}
