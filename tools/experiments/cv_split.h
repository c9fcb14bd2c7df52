// SHELVED EXPERIMENT (round 6) — the THREE-ROLE form of the persistent cost-volume forward.  Compiles against cost_volume.hip (`make experiments`,
// compile_check_cv.hip); not part of the library.  It was wired in behind GD_CV_SPLIT for the measurements below and passed tests/test_gpu_cost_volume.py.
//
// Why.  The anatomy of cv_fwd_persist_kernel (bench.py `attainable.memory_streams_us`, tools/probe_cv_streams.py): the feature ring hides under the teacher
// stream, and the kernel without ANY feature DMA still takes 0.9 of the whole — the critical path is teacher burst -> land (12-13 us) -> epilogue (3.6 us) ->
// next burst, because the teacher tile lives in the accumulator layout in 64 registers per lane of the waves that also run the main loop.
//
// What.  The epilogue of tile t runs BESIDE the main loop of tile t + 1, on its own waves (640 threads):
//   * 2 LOADER waves own the LDS-DMA ring (three 32 KB slots, two K steps ahead) and the tile statistics;
//   * 4 MFMA waves (2 x 2, wave tile 64 x 64, one per SIMD) run the main loop and then only SCALE the accumulators to s' = S log2(e) and park them in LDS —
//     in two halves, because LDS has room for 32 KB of S next to the ring: row blocks 0-1 of every wave right after the main loop, row blocks 2-3 (held in 32
//     registers meanwhile) after K step 4 of the next tile, when the first half has been consumed;
//   * 4 EPILOGUE waves stream BOTH teacher tiles straight from HBM, row-contiguous for both directions (direction 1: a lane owns two columns of a row of S;
//     direction 2: a lane owns a ROW of S and walks the teacher row that is S's column, S read transposed from a padded image), with a rolling window of 48
//     loads per lane (64 KB per CU) re-issued slot by slot as it is consumed; they take exp2, the Z sums and the B sums — the 16 wave sums of a K step
//     TOGETHER (v_permlane32_swap / v_permlane16_swap halvings + one 16-lane DPP reduction per register: 40 instructions instead of 128) — and carry a
//     finished tile's partial sums to the slabs themselves.
// One s_barrier per K step joins all ten waves; per tile period of nk >= 12 steps: steps 0-3 epilogue half 0 of the previous tile, step 4 park half 1, steps
// 5-8 epilogue half 1, step 9 flush.  One drain period at the end.
//
// Measured (P = 32, 37 x 37, D = 768, bf16, whole op): 262 us against 258-265 for cv_fwd_persist_kernel on the same boxes — a draw.  Compile-time anatomy
// (ANAT): ring + fragment reads alone 137; + MFMAs 168 (one wave per SIMD: its LDS reads and its MFMAs do not overlap; the eight-wave main loop of the
// persistent kernel takes 123-130, the stream kernel's 104); + the epilogue waves' arithmetic without their loads 209; + the teacher loads 262 — 64 KB in
// flight per CU do not cover the latency at this rate (the epilogue waves stall ~50 us), and the reductions cost ~35 us.  A first form with one wave
// reduction + branch per sum took 425 us.  What it would need: the main loop on eight waves (14 waves: 128 registers, which the epilogue waves' windows do
// not fit), a deeper teacher window (T2 as 8-byte loads covering both halves with the halves split by row parity: 96 KB in flight).  Estimated 200-215 us
// with all of that: not pursued further in round 6.
// Compiler note: __builtin_amdgcn_permlane32_swap / permlane16_swap return the FIRST result register for both elements with this toolchain (ROCm 7.2):
// inline asm below.
//
// LDS: ring 3 x 32 KB | S half: 64 rows x 130 floats (row stride 520 bytes: the transposed column reads of direction 2 fall on 16 banks, 8-byte row
// reads stay aligned, and every access of a wave is ONE per-lane base address plus an immediate offset) | statistics 2 x 4 KB | partial sums 2 x 6 KB
#define CV3_S_OFF (3 * CVP_STAGE)
#define CV3_LDS 130
#define CV3_STAT_OFF (CV3_S_OFF + 64 * CV3_LDS * 4)
#define CV3_PART_OFF (CV3_STAT_OFF + 2 * 4096)
#define CV3_SMEM (CV3_PART_OFF + 2 * 6144)
static_assert(CV3_SMEM <= 160 * 1024, "cv_split: LDS layout");
// partial sums per tile parity (floats): rowsZ[128] rowsB[128] colsB[2][128] colsZ[8][128]   (6 KB, as CVP_PART_OFF reserves)
#define CV3_ROWZ 0
#define CV3_ROWB 128
#define CV3_COLB 256
#define CV3_COLZ 512

__device__ __forceinline__ int cv3_tile_row(int b, int h) { return (b >> 5) * 64 + h * 32 + (b & 31); }      // S-buffer row b of half h -> row of the 128-row tile

template <typename T, int ANAT = 0>      // ANAT (timing only): 1 = epilogue waves idle (barriers only), 2 = no teacher loads, 4 = no wave reductions / stores, 8 = no MFMAs
__global__ __launch_bounds__(640) void cv_fwd_split_kernel(CvTileParams q) {
    __shared__ __attribute__((aligned(16))) char smem[CV3_SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = q.hw, tiles = q.tiles, ldt = q.ldt;
    const long rowb = (long)q.C * sizeof(T);
    const int nk = (int)(rowb / 128);
    const int total = q.P * tiles * tiles, nbx = gridDim.x >> 3, xc = blockIdx.x & 7, kb = blockIdx.x >> 3;
    const int qT = total >> 3, rT = total & 7;
    const int beg = xc < rT ? xc * (qT + 1) : rT * (qT + 1) + (xc - rT) * qT;
    const int cnt = qT + (xc < rT ? 1 : 0);
    const int n_tiles = cnt > kb ? (cnt - kb + nbx - 1) / nbx : 0;
    if (n_tiles == 0) return;
    const int n_total = n_tiles * nk;
    const unsigned smem_base = (unsigned)(uintptr_t)smem;
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

    if (wave >= 8) {
        // ======================================= LOADER waves (two, three-slot ring) =======================================
        const int lw = wave - 8;
        const char* asrc[8];
        const char* wsrc[8];
        int it_i = 0, k_i = 0, slot_i = 0;
        auto issue = [&]() {
            char* sA = smem + slot_i * CVP_STAGE;
            char* sB = sA + 128 * 128;
            slot_i = slot_i == 2 ? 0 : slot_i + 1;
            if (k_i == 0) {
                const CvpTile t = cvp_tile(beg + kb + it_i * nbx, tiles);
                const char* Ab = (const char*)q.f1 + (long)t.p * hw * rowb;
                const char* Wb = (const char*)q.f2 + (long)t.p * hw * rowb;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = (lw * 8 + i) * 8 + (lane >> 3);
                    asrc[i] = Ab + (long)min(t.tm * 128 + row, hw - 1) * rowb + (((lane & 7) ^ swz(row)) * 16);
                    wsrc[i] = Wb + (long)min(t.tn * 128 + cv_nperm64(row), hw - 1) * rowb + (((lane & 7) ^ swz(row)) * 16);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {      // the tile's 256 row / column statistics: the step's oldest pieces, parity buffer of the tile
                    const int e = (lw * 2 + h) * 64 + lane, which = e >> 7;
                    const int idx = min((which ? t.tn : t.tm) * 128 + (e & 127), hw - 1);
                    const char* ssrc = (const char*)(q.stats + (((long)t.p * 2 + which) * hw + idx) * 4);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ssrc,
                                                     (__attribute__((address_space(3))) void*)(smem + CV3_STAT_OFF + (it_i & 1) * 4096 + (lw * 2 + h) * 1024),
                                                     16, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (long)k_i * 128),
                                                 (__attribute__((address_space(3))) void*)(sA + (lw * 8 + i) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + (long)k_i * 128),
                                                 (__attribute__((address_space(3))) void*)(sB + (lw * 8 + i) * 1024), 16, 0, 0);
            if (++k_i == nk) { k_i = 0; ++it_i; }
        };
        for (int n = 0; n < 2 && n < n_total; ++n) issue();
        for (int n = 0; n < n_total; ++n) {
            // step n is in LDS once at most the youngest step's 16 pieces are outstanding (loads retire in order)
            if (n_total - 1 - n >= 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (n + 2 < n_total) issue();
        }
        for (int k = 0; k < nk; ++k) cvp_barrier();      // drain period: the epilogue of the last tile
        cvp_barrier();
        return;
    }

    if (wave >= 4) {
        // ======================================= EPILOGUE waves =======================================
        const int e = wave - 4;
        f32x2 R1[16];      // direction 1: T1[tile row of S-buffer row e*16 + q][columns 2 lane, 2 lane + 1]
        float R2[32];      // direction 2: T2[tile column e*32 + q][tile row of S-buffer row `lane`]
        // Addresses (clamped into the pair's map: entries past the ragged edge are re-reads that meet s' = 0) as a wave-uniform row base (scalar registers) plus
        // ONE per-lane byte offset per direction, tile and half — the 48 loads of a half share two vector registers of address.
        auto v1 = [&](const CvpTile& t) __attribute__((always_inline)) -> unsigned { return (unsigned)min(t.tn * 128 + 2 * lane, ldt - 2) * 4u; };
        auto v2 = [&](const CvpTile& t, int h) __attribute__((always_inline)) -> unsigned { return (unsigned)min(t.tm * 128 + cv3_tile_row(lane, h), ldt - 1) * 4u; };
        // (the uniform part as a 32-bit byte offset pinned into a scalar register — the host only takes this kernel for maps below 4 GB)
        auto ld1 = [&](const CvpTile& t, int h, int qq, unsigned voff) __attribute__((always_inline)) -> f32x2 {
            const int row = min(t.tm * 128 + cv3_tile_row(e * 16 + qq, h), hw - 1);
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)t.p * (unsigned)hw + (unsigned)row) * (unsigned)ldt * 4u));
            return *(const f32x2*)((const char*)q.t1 + (size_t)so + voff);
        };
        auto ld2 = [&](const CvpTile& t, int qq, unsigned voff) __attribute__((always_inline)) -> float {
            const int col = min(t.tn * 128 + e * 32 + qq, hw - 1);
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)t.p * (unsigned)hw + (unsigned)col) * (unsigned)ldt * 4u));
            return *(const float*)((const char*)q.t2 + (size_t)so + voff);
        };
        CvpTile tc = cvp_tile(beg + kb, tiles);
        {
            const unsigned o1 = v1(tc), o2 = v2(tc, 0);
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) R1[qq] = ld1(tc, 0, qq, o1);
#pragma unroll
            for (int qq = 0; qq < 32; ++qq) R2[qq] = ld2(tc, qq, o2);
        }
        for (int k = 0; k < nk; ++k) cvp_barrier();      // period 0: nothing parked yet
        if (ANAT & 1) {
            for (int P = 1; P <= n_tiles; ++P)
                for (int k = 0; k < nk; ++k) cvp_barrier();
            cvp_barrier();
            return;
        }
        const char* sS1 = smem + CV3_S_OFF + (e * 16 * CV3_LDS + 2 * lane) * 4;      // direction 1: row e*16 + qq at + qq * 520 bytes
        const char* sS2 = smem + CV3_S_OFF + (lane * CV3_LDS + e * 32) * 4;          // direction 2: column e*32 + qq at + qq * 4 bytes
        // A K step's work ("quarter"): 4 rows of direction 1 (Z and B: 8 sums over the wave) + 8 teacher rows of direction 2 (8 sums) = 16 wave sums, taken
        // TOGETHER: two swap-and-add stages across the wave halves / 16-lane rows (v_permlane32_swap, v_permlane16_swap: 16 registers -> 4, each 16-lane row
        // of each register a different sum), one 16-lane DPP reduction per register — 40 instructions instead of 16 x 8 — then lane (row rho, position t < 4)
        // of register t holds sum 4t + {0, 2, 1, 3}[rho] and stores it itself: no scalar round trip, no branch per sum.
        const int c16 = lane & 15, rho = lane >> 4;
        const int kq = 4 * c16 + ((rho & 1) << 1 | (rho >> 1));      // which of the 16 sums this lane stores (c16 < 4)
        const int kind = kq >> 2;                                    // 0: Z of row u, 1: B of row u, 2 / 3: B of column u
        const int rl0 = (e >> 1) * 64 + (e & 1) * 16;                // tile row of S-buffer row e*16 (half 0, quarter 0)
        // per-lane constant parts of the destination (floats, in the parity's partial-sum block) and of the statistics entry whose row sum scales the B sums
        const int dstc = kind == 0 ? CV3_ROWZ + rl0 + (kq & 3) : kind == 1 ? CV3_ROWB + rl0 + (kq & 3) : CV3_COLB + e * 32 + (kq - 8);
        const int entc = kind < 2 ? rl0 + (kq & 3) : 128 + e * 32 + (kq - 8);
        const uintptr_t p1 = (uintptr_t)q.part1, p2 = (uintptr_t)q.part2;
        for (int P = 1; P <= n_tiles; ++P) {
            const CvpTile tn = cvp_tile(beg + kb + min(P, n_tiles - 1) * nbx, tiles);      // (past the end: the last tile's entries once more; nobody reads them)
            const char* sStb = smem + CV3_STAT_OFF + ((P - 1) & 1) * 4096;
            float* sP = (float*)(smem + CV3_PART_OFF + ((P - 1) & 1) * 6144);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 zc = {0.f, 0.f};
                const unsigned o1 = v1(h ? tn : tc), o2 = v2(h ? tn : tc, 1 - h);      // the entries this half's slots are re-issued for: the next half's
                float rs1 = 1.f, rs2 = 1.f;      // teacher row sums: lane q < 16 holds that of S-buffer row e*16 + q, lane q < 32 that of column e*32 + q
#pragma unroll
                for (int ws = 0; ws < 4; ++ws) {
                    cvp_barrier();
                    if (ws == 0) {
                        rs1 = *(const float*)(sStb + (rl0 + h * 32 + (lane & 15)) * 16 + 4);
                        rs2 = *(const float*)(sStb + (128 + e * 32 + (lane & 31)) * 16 + 4);
                    }
                    // ---- phase A: everything this quarter reads from LDS, back to back
                    f32x2 s1[4];
                    float s2[8];
#pragma unroll
                    for (int u = 0; u < 4; ++u) s1[u] = *(const f32x2*)(sS1 + (ws * 4 + u) * (CV3_LDS * 4));
#pragma unroll
                    for (int u = 0; u < 8; ++u) s2[u] = *(const float*)(sS2 + (ws * 8 + u) * 4);
                    const int ent = entc + (kind < 2 ? h * 32 + ws * 4 : ws * 8);
                    const float rsd = *(const float*)(sStb + ent * 16 + 4);
                    // ---- phase B: the teacher entries out of their registers (and the next half's in), exp2, products
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int qq = ws * 4 + u;
                        const f32x2 tv = R1[qq];
                        if (!(ANAT & 2)) R1[qq] = ld1(h ? tn : tc, 1 - h, qq, o1);
                        const float thr = CV_EPS * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rs1), qq));
                        const float e0 = __builtin_amdgcn_exp2f(s1[u][0]), e1 = __builtin_amdgcn_exp2f(s1[u][1]);
                        zc += f32x2{e0, e1};
                        v[u] = e0 + e1;
                        v[4 + u] = fmaf(fmaxf(tv[0], thr), s1[u][0], fmaxf(tv[1], thr) * s1[u][1]);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int qq = ws * 8 + u;
                        const float tv = R2[qq];
                        if (!(ANAT & 2)) R2[qq] = ld2(h ? tn : tc, qq, o2);
                        const float thr = CV_EPS * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rs2), qq));
                        v[8 + u] = fmaxf(tv, thr) * s2[u];
                    }
                    if (ANAT & 4) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) zc[0] += v[k];
                        continue;
                    }
                    // ---- phase C: the 16 wave sums together
                    float w[8], x[4];
#pragma unroll
                    // (inline asm: with this compiler both results of __builtin_amdgcn_permlane32_swap / permlane16_swap come back as the FIRST register)
                    for (int i = 0; i < 8; ++i) {
                        float a = v[2 * i], b = v[2 * i + 1];
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // a: {a.lo, b.lo}, b: {a.hi, b.hi}
                        w[i] = a + b;      // lanes 0-31: sum 2i over {l, l + 32}; lanes 32-63: sum 2i + 1
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float a = w[2 * t], b = w[2 * t + 1];
                        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // odd 16-lane rows of a <-> even rows of b
                        x[t] = row16_sum(a + b);      // rows 0..3: sums 4t, 4t + 2, 4t + 1, 4t + 3
                    }
                    // ---- phase D: lane (rho, t < 4) scales and stores its sum
                    float res = c16 == 0 ? x[0] : c16 == 1 ? x[1] : c16 == 2 ? x[2] : x[3];
                    if (kind) res *= LN2 * __builtin_amdgcn_rcpf(rsd);
                    if (c16 < 4) sP[dstc + (kind < 2 ? h * 32 + ws * 4 : h * 128 + ws * 8)] = res;
                }
                *(f32x2*)(sP + CV3_COLZ + (e * 2 + h) * 128 + 2 * lane) = zc;
                if (h == 0) cvp_barrier();      // step 4: the MFMA waves park half 1
            }
            for (int k = 9; k < nk; ++k) {
                cvp_barrier();
                if (k == 9) {
                    // the finished tile's partial sums: LDS -> slabs.  Wave e takes 64 entries: rows 0-63, rows 64-127, columns 0-63, columns 64-127.
                    // Entries past the ragged edge hold s' = 0, i.e. e = 1: the Z sums of the valid rows / columns collected one per invalid column / row.
                    const bool isrow = e < 2;
                    const int cc = (e & 1) * 64 + lane;
                    float Z, B;
                    if (isrow) {
                        Z = sP[CV3_ROWZ + cc] - (float)max(0, tc.tn * 128 + 128 - hw);
                        B = sP[CV3_ROWB + cc];
                    } else {
                        Z = -(float)max(0, tc.tm * 128 + 128 - hw);
#pragma unroll
                        for (int k8 = 0; k8 < 8; ++k8) Z += sP[CV3_COLZ + k8 * 128 + cc];      // fixed order: deterministic
                        B = sP[CV3_COLB + cc] + sP[CV3_COLB + 128 + cc];
                    }
                    const int idx = (isrow ? tc.tm : tc.tn) * 128 + cc, slab = isrow ? tc.tn : tc.tm;
                    const uintptr_t base = isrow ? p1 : p2;
                    if (idx < hw)
                        *(__attribute__((address_space(1))) f32x2*)(base + ((((long)tc.p * q.nslab + slab) * hw + idx) * 2) * sizeof(float)) = f32x2{Z, B};
                }
            }
            tc = tn;
        }
        cvp_barrier();
        return;
    }

    // ======================================= MFMA waves =======================================
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c = lane & 15;
    typedef typename Mma<T>::Frag Frag;
    const int sa = swz(c);
    const int abase = (wm * 64 + c) * 128, bbase = 128 * 128 + (wn * 64 + c) * 128;
    char* sSp = smem + CV3_S_OFF + ((wm * 32 + 4 * g) * CV3_LDS + wn * 64 + 4 * c) * 4;
    f32x4 held[2][4];      // s' of row blocks 2, 3 of the previous tile: [ib - 2][r], element jb
    // park row block ib (0..3) of this wave: S-buffer rows wm*32 + (ib & 1)*16 + 4g + r, columns wn*64 + 4c .. +3 (two 8-byte stores: rows are 8-byte aligned)
    auto park = [&](const f32x4 (&v)[4], int ib) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            char* d = sSp + ((ib & 1) * 16 + r) * (CV3_LDS * 4);
            *(f32x2*)d = f32x2{v[r][0], v[r][1]};
            *(f32x2*)(d + 8) = f32x2{v[r][2], v[r][3]};
        }
    };
    int slot = 0;
    for (int P = 0; P <= n_tiles; ++P) {
        const bool ml = P < n_tiles;
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nk; ++k) {
            cvp_barrier();
            if (k == 4 && P > 0) { park(held[0], 2); park(held[1], 3); }
            if (ml) {
                const char* sb = smem + slot * CVP_STAGE;
                slot = slot == 2 ? 0 : slot + 1;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const int co = (((kc * 4 + g) ^ sa) * 16);
                    Frag a[4], b[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) b[u] = *(const Frag*)(sb + bbase + u * 2048 + co);
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u] = *(const Frag*)(sb + abase + u * 2048 + co);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (ANAT & 8) acc[i][j][0] += (float)a[i][0] + (float)b[j][0];
                            else acc[i][j] = Mma<T>::mma(a[i], b[j], acc[i][j]);
                        }
                }
            }
        }
        if (ml) {
            // s' = acc * inv1 log2(e) * inv2; rows / columns past the ragged edge get a zeroed norm (s' = 0 there).  acc[ib][jb][r]: row ib*16 + 4g + r,
            // column 4c + jb of the wave tile (cv_nperm64 makes a lane's four n-tiles four consecutive columns)
            const CvpTile t = cvp_tile(beg + kb + P * nbx, tiles);
            const f32x4* sSt = (const f32x4*)(smem + CV3_STAT_OFF + (P & 1) * 4096);
            float inv2[4];
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const int cl = wn * 64 + 4 * c + jb;
                inv2[jb] = t.tn * 128 + cl < hw ? sSt[128 + cl][0] : 0.f;
            }
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                f32x4 v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = wm * 64 + ib * 16 + 4 * g + r;
                    const float i1 = t.tm * 128 + rl < hw ? sSt[rl][0] * LOG2E : 0.f;
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb) v[r][jb] = acc[ib][jb][r] * i1 * inv2[jb];
                }
                if (ib < 2) park(v, ib);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) held[ib - 2][r] = v[r];
                }
            }
        }
    }
    cvp_barrier();
}
