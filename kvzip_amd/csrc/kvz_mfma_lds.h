// kvz_mfma_lds.h - pieces shared by the MFMA kernels that stream tiles through LDS (kvz_score.hip, kvz_flash2.hip):
// 32x32x16 MFMA wrappers, LDS-DMA (global_load_lds_dwordx4 issued from assembly), the counted wait and the fence-free barrier.
#pragma once
#include "kvz_common.h"

namespace kvz {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma32;
template <> struct Mfma32<_Float16> {
    typedef h8 v8;
    __device__ static inline f16v mfma(v8 a, v8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    // first MFMA of a chain (C = 0) written INTO the registers of `acc`: the tied operand keeps an accumulator in one physical
    // register tuple for the whole kernel (left to the allocator every chain is a fresh 16-tuple, and under pressure the hunt
    // for free aligned tuples spills the fragment registers).  No software wait states are needed after it: the next MFMA of
    // the chain accumulates into exactly the same registers (back-to-back SrcC = vDst is interlocked).
    __device__ static inline void mfma_first(f16v& acc, v8 a, v8 b) {
        asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "+v"(acc) : "v"(a), "v"(b));
    }
};
template <> struct Mfma32<__bf16> {
    typedef b8 v8;
    __device__ static inline f16v mfma(v8 a, v8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    __device__ static inline void mfma_first(f16v& acc, v8 a, v8 b) {
        asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "+v"(acc) : "v"(a), "v"(b));
    }
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---- staging: one 128-row tile, HBM/L2 -> LDS directly (global_load_lds_dwordx4, no VGPR round trip) ---------
// One wave-instruction writes 1 KiB = 64 lanes x 16 B LINEARLY (wave-uniform base + lane*16).  The XOR swizzle of
// the tile is therefore applied on the SOURCE side: LDS position p of a row receives global chunk p ^ f(row), and
// fragment reads use lds_off(row, chunk) = position chunk ^ f(row) (same involution on both sides).
// rowptr clamps out-of-range rows to a valid row: loads are UNCONDITIONAL; out-of-range rows are neutralised
// downstream (causal limit in pass A, m = +inf statistics in pass B).
//
// The LDS-DMA instruction is issued from inline assembly.  Through the builtin, the compiler cannot prove that a
// ds_read of one tile buffer does not alias the DMA write that is in flight into the OTHER buffer and puts
// s_waitcnt vmcnt(0) in front of the first fragment read after every staging call - the full global latency, once
// per tile (20 % of pass A in an in-kernel trace).  Ordering between the DMA and the fragment reads is by
// stage_wait() + the block barrier, exactly as designed.  (The opposite choice - builtin DMA, assembly ds_reads -
// is NOT safe: register copies the compiler inserts between an assembly read and its s_waitcnt read stale data.)
// M0 (LDS base of the DMA) is not used by anything else in these kernels.
__device__ static inline void lds_dma16(const void* gsrc, const char* lds_dst /* wave-uniform */) {
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)(lds_dst));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(la) : "memory");
}
// the same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset (no 64-bit VALU address arithmetic,
// half the address registers); the LDS destination is given as a BYTE ADDRESS in LDS (an integer: a generic pointer costs a
// null-checked address-space cast per piece)
__device__ static inline uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(lptr_t)(p); }
__device__ static inline void lds_dma16a(const char* sbase /* wave-uniform */, uint32_t voff, uint32_t lds_byte /* wave-uniform */) {
    const uint32_t la = __builtin_amdgcn_readfirstlane(lds_byte);
    const uint64_t b = (uint64_t)(uintptr_t)sbase;
    // (readfirstlane returns a signed int: widen through uint32_t, a sign-extended low word would wipe out the high word)
    const uint64_t bs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(bs), "s"(la) : "memory");
}
__device__ static inline void lds_dma16s(const char* sbase /* wave-uniform */, uint32_t voff, const char* lds_dst /* wave-uniform */) {
    lds_dma16a(sbase, voff, lds_addr(lds_dst));
}
// all LDS-DMA of this wave has landed (the compiler does not count the assembly loads: its own vmcnt waits can only
// become more conservative, never weaker, because the counter retires in order)
__device__ static inline void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// block barrier without the release/acquire fences of __syncthreads(): the fence makes the compiler wait for vmcnt(0)
// whenever it has a global load, store or atomic of its own in flight, which (the hardware counter being shared) drains
// every tile staged ahead.  LDS traffic of this wave is complete (lgkmcnt 0) before the barrier.
__device__ static inline void block_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// reference rounding chain (attention/score.py:57): half(matmul) / sqrt(D) -> half.
// The division is an IEEE fp32 division whose result is immediately rounded to 16 bits.  Because the dividend is a
// 16-bit value there are only 65536 cases, and the host verifies exhaustively (find_exact_reciprocal) that one
// fp32 multiply by `rcp` gives the identical 16-bit result for all of them; if no such constant exists the kernel
// falls back to the true division.
template <typename T, bool FAST>
__device__ static inline float round_chain(float acc, float c, float rcp) {
    const T h1 = (T)acc;
    const float d = FAST ? (float)h1 * rcp : (float)h1 / c;
    const T h2 = (T)d;
    return (float)h2;
}


}  // namespace kvz
