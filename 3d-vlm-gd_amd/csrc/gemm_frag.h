// Pieces of the persistent 256 x 256 tile kernels shared by gemm_persist.h (gemm_nt) and cv_persist256.h (cost-volume forward): the permuted
// W-row order, inline-asm LDS reads, the software-pipelined K-chunk of a 128 x 64 wave tile, counted vmcnt waits, auxiliary-policy stores.
#pragma once
#include "gemm_tile.h"
#include <type_traits>

__device__ __forceinline__ int nperm64(int rho) {   // LDS W-row (64w + 16j + fr) -> column 64w + 4fr + j of the 256-wide tile
    return (rho & ~63) | ((rho & 15) << 2) | ((rho >> 4) & 3);
}

// LDS reads of this kernel go through inline asm: for a ds_read the compiler can see, it inserts `s_waitcnt vmcnt`
// up to the most recent LDS-DMA (it cannot tell ring slots apart), which at the top of a tile means waiting for the
// previous tile's epilogue stores and the just-issued prefetch — exactly the overlap this kernel exists for.
// Results are "released" by a counted `s_waitcnt lgkmcnt(N)` that lists them as read-write operands (LDS returns in order).
#define GD_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define GD_DSR32(dst, addr, off) asm volatile("ds_read_b32 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
__device__ __forceinline__ unsigned lds_off(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

template <typename T>
__device__ __forceinline__ void mma_row(f32x4 (&acc)[4], f32x4 a, f32x4 b0, f32x4 b1, f32x4 b2, f32x4 b3) {
    typedef typename Mma<T>::Frag Frag;
    const Frag fa = __builtin_bit_cast(Frag, a);
    acc[0] = Mma<T>::mma(fa, __builtin_bit_cast(Frag, b0), acc[0]);
    acc[1] = Mma<T>::mma(fa, __builtin_bit_cast(Frag, b1), acc[1]);
    acc[2] = Mma<T>::mma(fa, __builtin_bit_cast(Frag, b2), acc[2]);
    acc[3] = Mma<T>::mma(fa, __builtin_bit_cast(Frag, b3), acc[3]);
}

// One 64-byte K chunk of the 128 x 64 wave tile = 12 fragment reads + 32 (transposed) MFMAs, software-pipelined:
// the FIRST six fragment reads of a chunk (b0..b3, a0, a1) are issued
// before the previous chunk's last two MFMA rows, and — across a K step — right after the stage barrier, which itself
// moves up to the point where the wave's LDS reads of the stage are complete (the eight MFMAs that follow need
// registers only).  The read latency at the head of every chunk and the barrier skew then sit under MFMA work instead
// of in front of it.  Register cost: none — the prefetched set reuses the registers of a0..a5, dead by then.
struct FragHead { f32x4 b0, b1, b2, b3, a0; };   // (five, not six: P and Q are both live at the stage barrier — 8 VGPRs decide spill / no spill)
__device__ __forceinline__ void frag_head_issue(FragHead& h, unsigned aaddr, unsigned baddr) {
    GD_DSR128(h.b0, baddr, 0); GD_DSR128(h.b1, baddr, 2048); GD_DSR128(h.b2, baddr, 4096); GD_DSR128(h.b3, baddr, 6144);
    GD_DSR128(h.a0, aaddr, 0);
}
struct FragTail { f32x4 a6, a7; };
// rows 0..5 of a chunk whose head is already in flight; returns with a6, a7 landed (every LDS read of the chunk done)
template <typename T>
__device__ __forceinline__ void chunk_rows05(FragHead& h, FragTail& t, unsigned aaddr, f32x4 (&acc)[8][4]) {
    f32x4 a1, a2, a3, a4, a5;
    GD_DSR128(a1, aaddr, 2048); GD_DSR128(a2, aaddr, 4096); GD_DSR128(a3, aaddr, 6144); GD_DSR128(a4, aaddr, 8192);
    GD_DSR128(a5, aaddr, 10240); GD_DSR128(t.a6, aaddr, 12288); GD_DSR128(t.a7, aaddr, 14336);
    asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(h.b0), "+v"(h.b1), "+v"(h.b2), "+v"(h.b3), "+v"(h.a0));
    mma_row<T>(acc[0], h.a0, h.b0, h.b1, h.b2, h.b3);
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a1));
    mma_row<T>(acc[1], a1, h.b0, h.b1, h.b2, h.b3);
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a2), "+v"(a3));
    mma_row<T>(acc[2], a2, h.b0, h.b1, h.b2, h.b3);
    mma_row<T>(acc[3], a3, h.b0, h.b1, h.b2, h.b3);
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a4), "+v"(a5));
    mma_row<T>(acc[4], a4, h.b0, h.b1, h.b2, h.b3);
    mma_row<T>(acc[5], a5, h.b0, h.b1, h.b2, h.b3);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t.a6), "+v"(t.a7));
}
// anatomy: the chunk's LDS reads without its MFMAs
__device__ __forceinline__ void chunk_reads_only(FragHead& h, FragTail& t, unsigned aaddr) {
    f32x4 a1, a2, a3, a4, a5;
    GD_DSR128(a1, aaddr, 2048); GD_DSR128(a2, aaddr, 4096); GD_DSR128(a3, aaddr, 6144); GD_DSR128(a4, aaddr, 8192);
    GD_DSR128(a5, aaddr, 10240); GD_DSR128(t.a6, aaddr, 12288); GD_DSR128(t.a7, aaddr, 14336);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h.b0), "+v"(h.b1), "+v"(h.b2), "+v"(h.b3), "+v"(h.a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(t.a6), "+v"(t.a7));
}
template <typename T>
__device__ __forceinline__ void chunk_rows67(const FragHead& h, const FragTail& t, f32x4 (&acc)[8][4]) {
    mma_row<T>(acc[6], t.a6, h.b0, h.b1, h.b2, h.b3);
    mma_row<T>(acc[7], t.a7, h.b0, h.b1, h.b2, h.b3);
}


// wait until at most n vector-memory operations of this wave are outstanding (n rounded DOWN to a multiple of 8:
// conservative).  VMEM operations retire in issue order on gfx9-family parts, so "the S youngest may stay in flight"
// is how a wave lets its epilogue stores drain under the next tile's main loop while still seeing its DMA land.
__device__ __forceinline__ void wait_vm_le(int n) {
    if (n >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if (n >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    else if (n >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
typedef __attribute__((ext_vector_type(4))) unsigned gd_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned gd_u32x2;
template <int AUX>
__device__ __forceinline__ void bst4_aux(__amdgpu_buffer_rsrc_t rs, int off, int dt, const float (&v)[4]) {
    if (dt == GD_BF16) {
        const bf16x4 b = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(gd_u32x2, b), rs, off, 0, AUX);
    } else if (dt == GD_F16) {
        const f16x4 b = f16_sat4(v[0], v[1], v[2], v[3]);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(gd_u32x2, b), rs, off, 0, AUX);
    } else {
        const f32x4 a = {v[0], v[1], v[2], v[3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gd_u32x4, a), rs, off, 0, AUX);
    }
}
// Store cache policy (aux bits: 1 sc0, 2 nt, 16 sc1).  C leaves with nt|sc1: measured on MI355X (87680 x 3072 x 768)
// plain / sc1 stores stretch the NEXT tile's main loop from 31.7 k to 46 k cycles (the written lines fight the operand
// panels for the L2 and the drain blocks the vmcnt-ordered DMA); with the non-temporal hint the main loop is unaffected.
#ifndef GD_PERSIST_STORE_AUX
#define GD_PERSIST_STORE_AUX 18
#endif
// phase probe (gd_gemm_phase_probe): present in -DGD_GEMM_STAGE_PROBE builds only, armed when p.probe != null
#ifdef GD_GEMM_STAGE_PROBE
#define GD_PROBE(...) if (p.probe) { __VA_ARGS__ }
#define GD_PROBE_DECL(...) __VA_ARGS__
#else
#define GD_PROBE(...)
#define GD_PROBE_DECL(...)
#endif
#ifndef GD_SDEP32
#define GD_SDEP32 8      // f32 side tensors: items in flight per lane (16 bytes each)
#endif
#ifndef GD_PERSIST_SIDE_AUX
#define GD_PERSIST_SIDE_AUX 2     // side tensors are read once: stream them past the L2's operand panels
#endif

