// kvz_score.hip — KV importance scoring on the gfx950 matrix cores.
//
// Replaces KVScore._get_score (+_mask_causal) of the reference (attention/score.py:36-85), which
// materialises five [1,Hkv,G,q,k] tensors per layer and chunk (cat, matmul, div, softmax, slice).  Here
// nothing of size q*k ever leaves the chip: two tiled passes over Q.K^T on v_mfma_f32_32x32x16.
//
//   pass A (row statistics)   for every query row r=(g,i):  m_r = max_j x[r,j],  l_r = sum_j exp(x[r,j]-m_r)
//                             over keys j in  sink ++ ctx chunk ++ repeat chunk (causal inside the repeat chunk)
//   pass B (column maximum)   for every ctx key j:  t_j = max_r ( (x[r,j]-m_r) - log l_r )
//   finalise                  score_j = half( exp(t_j) )        (exp is monotone: max_r softmax = exp(max_r log-softmax))
//
// x follows the reference's rounding chain exactly:  x = half( float(half(q.k [fp32 accumulate])) / float(sqrt(D)) ).
//
// Tiling (both passes): one operand is STREAMED through LDS in 128-row tiles, the other is STATIONARY in
// registers (32 rows per wave).  In both passes the QUERY ROW is the column of the 32x32 MFMA result, i.e.
// one query row per lane: its softmax statistics (pass A: running max / sum, pass B: m_r and log l_r) are
// per-lane scalars and the 16 accumulator registers of a lane are 16 different keys.  Pass A streams keys
// (reduction over keys = over registers, lane-local); pass B streams query rows and keeps 16 running
// per-key maxima per lane that are reduced across lanes once at the end.  LDS tiles are XOR-swizzled on
// 16-byte chunks so that ds_read_b128 fragment reads are bank-conflict free.
// Launches per (layer, chunk): pass A, pass B (which merges pass A's partial statistics in its prologue); the deferred path
// turns the log buffer of a whole context into 16-bit scores with ONE finalize launch when the scores are read.
// Bound: SIMD issue - the VALU rounding chain (24 cycles per logit) and the matrix pipe share the issue port of a SIMD
// (timelines and counters: profiles/r3_passA_timeline.txt, r3_passB_timeline.txt, r3_pmc_traffic.json).
#include "kvz_common.h"
#include "kvz_mfma_lds.h"

#include <math.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include <type_traits>

namespace kvz {

constexpr int SC_TILE = 128;     // streamed rows per LDS tile (32 KiB at D = 128)
constexpr int PB_WAVES = 8;              // waves per block of pass B: one 8-wave block per CU (three 32-KiB row-tile buffers; 4 LDS-DMA pieces per wave and tile)
constexpr int PB_OCC = PB_WAVES / 4;     // waves per SIMD the register budget is sized for
struct ScoreArgs {
    const void* q;       // [Hkv*G, q_len, D]
    const void* k;       // [Hkv, klen, D]
    int64_t q_head_stride, k_head_stride;  // elements
    int klen, sink, start, m, q_len, G;
    float2* stats;       // [max_seg, Hkv, stats_stride]  partial (m_r, l'_r) of each segment (l' relative to fl(m*log2e))
    int stats_stride;    // G*q_len rounded up to a multiple of SC_TILE; the padding rows hold (+inf, 0) after the merge
    float* colpart;      // [row_splits, Hkv, m]  per-slice column maxima of the log-softmax
    void* out;           // [Hkv, m] half
    int64_t out_head_stride;
    int row_splits;      // pass B
    int unit_rows;       // rows per unit of the pass-A partition (PA_ROWS)
    int* unit_nseg;      // [units] partial statistics per unit (written by the pass-A block that finishes the unit, read by pass B)
    int max_seg;         // most partials any unit has (PaPlan::max_seg)
    int n_kv_heads;
    float c;             // float32(sqrt(D))
    float rcp;           // reciprocal constant r such that half(x*r) == half(x/c) for EVERY 16-bit x (0 = none found)
    FastDiv dq, dh;      // q_len, Hkv
    uint32_t* log_out;   // non-NULL: pass B merges its row slices by atomic unsigned-min on the bit patterns of the (non-positive)
    int64_t log_head_stride;  // fp32 log-scores into [Hkv, log_head_stride] instead of writing colpart; no finalize launch
    // ---- exact pruning of pass B (round 5: score_rowstatT2_kernel, score_merge_kernel, score_bounds2_kernel, score_colmax_sparse_kernel)
    uint16_t* colu;      // [Hkv, nkb, n_groups, 2, 32] (+ 128 spare bytes)  16-bit patterns of max over the 16 rows of each HALF of group g of the logit x_rj
    float2* gbound;      // [Hkv, n_groups]     (max, min) over the rows of group g of  n_r = -(m_r + log l_r);  (NaN, NaN): a row with NaN statistics
    float* nrow;         // [Hkv, 32 n_groups]  n_r per query row (-inf for the padding rows of the last group)
    uint32_t* entries;   // [Hkv, nkb * n_groups] compacted candidate pairs per KV head: g | kb << 11 | h << 25, the pairs of one (h, kb) contiguous
    uint32_t* counter;   // [Hkv] number of entries per head (zeroed by score_merge_kernel, filled by score_bounds2_kernel)
    uint32_t* redo;      // [PLAN_MAX_BLOCKS] per block of the key-per-lane pass: items to be redone by its slow loop (self-clearing)
    int n_groups, nkb;
    int all_pairs;       // (debug knob: every pair is a candidate)
    // ---- candidates at KEY granularity (round 6: score_bounds3_kernel, score_colmax_keys_kernel)
    uint32_t* gcount;    // [Hkv, n_groups]            candidate keys of row group g (zeroed by score_merge_kernel, filled by score_bounds3_kernel)
    uint32_t* klist;     // [Hkv, n_groups, kcap]      their ctx indices j (any order)
    int kcap;            // nkb * 32
};

// the same chain, result kept as the 16-bit value (maxima are taken on 16-bit values, the exp2 / subtraction
// arguments read it through the mixed-precision fma: no separate conversion back to fp32)
template <typename T, bool FAST>
__device__ static inline T round_chain_h(float acc, float c, float rcp) {
    const T h1 = (T)acc;
    const float d = FAST ? (float)h1 * rcp : (float)h1 / c;
    return (T)d;
}
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
template <typename T> __device__ static inline uint32_t bits16(T v) {
    uint16_t b;
    __builtin_memcpy(&b, &v, 2);
    return b;
}
template <typename T> __device__ static inline float pair_lo(uint32_t p) {
    if constexpr (std::is_same<T, _Float16>::value) return (float)__builtin_bit_cast(h2v, p)[0];
    else return __builtin_bit_cast(float, p << 16);
}
template <typename T> __device__ static inline float pair_hi(uint32_t p) {
    if constexpr (std::is_same<T, _Float16>::value) return (float)__builtin_bit_cast(h2v, p)[1];
    else return __builtin_bit_cast(float, p & 0xffff0000u);
}

template <int D> struct ScoreCfg {
    static constexpr int ROW_BYTES = D * 2;
    static constexpr int CPR = ROW_BYTES / 16;                       // 16-byte chunks per row
    static constexpr int KK = D / 16;                                // MFMA k-steps
    static constexpr int TILE_BYTES = SC_TILE * ROW_BYTES;
    // swizzled byte offset of 16-byte chunk `chunk` of tile row `row`
    __device__ static inline int lds_off(int row, int chunk) {
        if (D == 128) return row * ROW_BYTES + ((chunk ^ (row & 15)) << 4);
        return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4);
    }
};

// ---- fragment reads: 8 (D=128) / 4 (D=64) ds_read_b128 per 32-row block ------------------------------------------
template <int D> struct FragAddr {
    static constexpr int KK = D / 16;
    uint32_t a[KK];  // LDS byte address of this lane's chunk kk of row (lane & 31) in a tile at LDS offset 0
    __device__ inline void init(const char* lds_base, int l31, int half) {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            a[kk] = (uint32_t)(uintptr_t)(lptr_t)(lds_base) + (uint32_t)ScoreCfg<D>::lds_off(l31, kk * 2 + half);
    }
};
template <int D>
__device__ static inline void frag_load(u32x4 (&fr)[D / 16], const FragAddr<D>& fa, int byte_off /* compile-time after unrolling */) {
    typedef const __attribute__((address_space(3))) u32x4* lp;
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) fr[kk] = *(lp)(uintptr_t)(fa.a[kk] + byte_off);
}

template <int D, int NW, typename RowPtr>
__device__ static inline void stage_tile(char* buf, int row0, RowPtr rowptr, int wave, int lane) {
    typedef ScoreCfg<D> C;
    constexpr int ROWS_PER_INSTR = 1024 / C::ROW_BYTES;  // 4 (D = 128) or 8 (D = 64)
    constexpr int INSTR = C::TILE_BYTES / 1024;
    constexpr int PER_WAVE = INSTR / NW;
    static_assert(PER_WAVE >= 1 && INSTR % NW == 0, "tile / wave layout");
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int ci = i * NW + wave;  // wave-uniform 1-KiB piece of the tile
        const int row = ci * ROWS_PER_INSTR + lane / C::CPR;
        const int p = lane % C::CPR;
        const int chunk = (D == 128) ? (p ^ (row & 15)) : (p ^ ((row >> 1) & 7));
        const char* src = rowptr(row0 + row) + chunk * 16;
        lds_dma16(src, buf + ci * 1024);
    }
}

// Fast variant for a tile whose 128 rows are CONSECUTIVE in memory (the common case): the address is a wave-uniform
// base (SGPRs) plus ONE per-lane byte offset that is the same for every piece, because the swizzle term of a row only
// depends on (wave, lane) when the piece stride NW*ROWS_PER_INSTR is a multiple of 16 rows.  This removes the per-lane
// clamp / segment select / 64-bit multiply-add (pass A) and the integer division (pass B) from every staged load.
template <int D, int NW>
__device__ static inline uint32_t stage_lane_offset(int wave, int lane) {
    typedef ScoreCfg<D> C;
    constexpr int ROWS_PER_INSTR = 1024 / C::ROW_BYTES;
    static_assert((NW * ROWS_PER_INSTR) % 16 == 0, "swizzle must not depend on the piece index");
    const int row = wave * ROWS_PER_INSTR + lane / C::CPR;
    const int p = lane % C::CPR;
    const int chunk = (D == 128) ? (p ^ (row & 15)) : (p ^ ((row >> 1) & 7));
    return (uint32_t)(row * C::ROW_BYTES + chunk * 16);
}
template <int D, int NW>
__device__ static inline void stage_tile_linear_a(uint32_t lds_byte /* of the buffer */, const char* base, uint32_t lane_off, int wave) {
    typedef ScoreCfg<D> C;
    constexpr int INSTR = C::TILE_BYTES / 1024;
    constexpr int PER_WAVE = INSTR / NW;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) lds_dma16a(base + (size_t)i * NW * 1024, lane_off, lds_byte + (uint32_t)((i * NW + wave) * 1024));
}
template <int D, int NW>
__device__ static inline void stage_tile_linear(char* buf, const char* base, uint32_t lane_off, int wave) {
    stage_tile_linear_a<D, NW>(lds_addr(buf), base, lane_off, wave);
}

// ---- pass A: per-query-row softmax statistics ------------------------------------------------------
// Online softmax in the exp2 domain: the running pseudo-maximum ml2 = fl(m * log2e) is an fp32 number, every term
// is exp2(fma(x, log2e, -ml2)) (one rounding), and the common factor 2^(m*log2e - ml2) that this introduces
// into l is removed exactly at the end (delta = fma(m, log2e, -ml2)).
//
// Execution shape (from in-kernel s_memtime traces, per-block Gantt charts and ablations: DESIGN.md 3.1, profiles/r*_timeline.txt):
//  * one 8-wave block per CU (two waves per SIMD), 32 query rows per wave, 256 rows per key tile: half the L2->LDS
//    traffic per logit of a 128-row block.  (The one-wave-per-SIMD variant with 64 rows per wave measured 27 % slower in round 3
//    even with the query rows in the accumulator file - a lone wave cannot fill its own dependency stalls,
//    profiles/r3_ab_one_wave_per_simd.txt.)
//  * No global load with a register destination inside the kernel: key tiles AND the query rows of the next item come
//    in by LDS-DMA issued from assembly.  A load the compiler knows about makes it place s_waitcnt vmcnt(n) wherever
//    one of the affected registers is touched, and because the hardware counter also holds the DMA, every such wait
//    drains the tiles staged ahead.  For the same reason the barrier is a bare s_barrier (no fence) and the work
//    list is STATIC (no atomic queue): an exactly balanced partition of the (row tile, head, key tile) space, cut on the host
//    (PaPlan below) - one-block-per-item launches lost 27 % to packing and 2-4 us per block waiting for the first tile.
//  * Latencies are hidden by distance: the DMA of tile p+2 is issued when tile p is handed back, fragment reads run
//    one 32-key block ahead into a second register set (a prefetch into the registers the chain in flight still
//    reads stalls the in-order issue), and the hand-over barrier sits right after the LAST matrix chain of a tile
//    has been issued, not after its epilogue.  The tile stream runs across item boundaries.
#ifndef KVZ_PA_FAR_BASE
#define KVZ_PA_FAR_BASE 1
#endif
#ifndef KVZ_T2_BLOCKMASK
#define KVZ_T2_BLOCKMASK 0
#endif
constexpr int PA_WAVES = 8;
constexpr int PA_RG = 1;                        // 32-row groups per wave
constexpr int PA_ROWS = PA_WAVES * PA_RG * 32;  // query rows per work item

// ---- the instruction stream of a wave (round 2: software pipeline inside the wave) --------------------------------------
// The matrix chain of 32-key block b+1 is issued INSIDE the epilogue of block b, one MFMA per group of
// ~9-18 VALU instructions (groups are pinned with sched_barrier): a wave no longer alternates 256 cycles of matrix pipe
// with ~500 cycles of VALU, it keeps both busy, and its partner on the SIMD fills the dependency stalls of the chain.
// The epilogue itself is cut from ~136 to ~80 VALU instructions per 32x32 block:
//   * no running-maximum tree: the exponentials are taken against a REFERENCE m_ref (a 16-bit value, so x - m_ref stays
//     exact) that is only moved when a block's sum of exponentials exceeds 2^16 (cold path: maximum tree, rescale,
//     redo of the block); the first block of every item takes that path once and sets m_ref to its own maximum.
//     Softmax is shift invariant: pass B uses (x - m_ref) - log(sum exp(x - m_ref)), m_ref need not be the maximum;
//   * accumulators are not cleared (the first MFMA of a chain takes the constant 0 as C);
//   * fp16: rounding chain and exponent argument of FOUR logits are one assembly block of 8 instructions ordered so that
//     every consumer of a 16-bit (dst_sel) write is at least one instruction behind it (no s_nop).
// Placement of the KK MFMAs of a chain over the 8 half-groups of a pipeline step (half-group = 2 x quad index + half).
// The chain ends one half-group early - the next step starts by reading this accumulator, and a chain that ends with the step
// makes it wait for the last MFMA (the compiler pads 12 wait states) - by issuing the first two MFMAs back to back (dependent
// MFMAs issued back to back take the accumulator forwarding path).
template <int KK> struct MfmaSched {
    // first k-step and number of k-steps issued in half-group hg
    __device__ static constexpr int first(int hg) {
        if (KK == 8) return hg == 0 ? 0 : hg + 1;
        return hg / 2;  // KK == 4: even half-groups only
    }
    __device__ static constexpr int count(int hg) {
        if (KK == 8) return hg == 0 ? 2 : (hg == 7 ? 0 : 1);
        return (hg & 1) ? 0 : 1;
    }
};
template <typename T, bool FAST>
__device__ static inline void quad_args(float a0, float a1, float a2, float a3, uint32_t& xa, uint32_t& xb, float (&arg)[4],
                                        float c, float rcp, float L2E /* multiplier of x */, float neg_ml2 /* addend */) {
    if constexpr (std::is_same<T, _Float16>::value && FAST) {
        // the first rounding inside the block as well: left outside, the compiler converts all 16 accumulators at the top of the
        // step and keeps the 8 packed results (and a copy per quad) alive - registers the 64-row kernel does not have
        // (round 4: the second rounding IN PLACE by v_fma_mixlo_f16 / v_fma_mixhi_f16 - 10 instead of 12 instructions per four logits,
        // bit-identical on all 65 536 inputs and on 400 random shapes - is 1.7 % SLOWER in the scoring loop: the 16-bit writes
        // preserve the other half of their destination, i.e. read it, and chain the two halves of a pair)
        asm("v_cvt_pk_f16_f32 %[xa], %[a0], %[a1]\n\t"
            "v_cvt_pk_f16_f32 %[xb], %[a2], %[a3]\n\t"
            "v_fma_mix_f32 %[g0], %[xa], %[r], 0 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g1], %[xa], %[r], 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g2], %[xb], %[r], 0 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g3], %[xb], %[r], 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_cvt_pk_f16_f32 %[xa], %[g0], %[g1]\n\t"
            "v_cvt_pk_f16_f32 %[xb], %[g2], %[g3]\n\t"
            "v_fma_mix_f32 %[g0], %[xa], %[l2e], %[nm] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g2], %[xb], %[l2e], %[nm] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g1], %[xa], %[l2e], %[nm] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g3], %[xb], %[l2e], %[nm] op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : [xa] "=&v"(xa), [xb] "=&v"(xb), [g0] "=&v"(arg[0]), [g1] "=&v"(arg[1]), [g2] "=&v"(arg[2]), [g3] "=&v"(arg[3])
            : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [r] "s"(rcp), [l2e] "v"(L2E), [nm] "v"(neg_ml2));
    } else if constexpr (std::is_same<T, __bf16>::value && FAST) {
        // bf16: there is no mixed-precision fma that reads a bf16 half, so every value goes f32 -> bf16 -> f32 twice:
        // v_cvt_pk_bf16_f32 rounds two values per instruction, the way back is a shift / a mask, the two fp32 multiplications (by
        // rcp, then by log2e with the addend) are PLAIN v_mul_f32 / v_fma_f32.  Round 2 had them packed (v_pk_mul_f32 /
        // v_pk_fma_f32 on pairs, 16 instead of these 20 instructions per four logits): bit-identical, but packed fp32 arithmetic
        // issues slower than two plain instructions on this part - the plain form takes 2 us off pass B (45.1 -> 43.1 us) and
        // +1.2 % in the bf16 scoring loop (round 4, same box, three interleaved rounds).  Assembly, because the compiler packs
        // adjacent fp32 operations again.
        asm("v_cvt_pk_bf16_f32 %[xa], %[a0], %[a1]\n\t"
            "v_cvt_pk_bf16_f32 %[xb], %[a2], %[a3]\n\t"
            "v_lshlrev_b32 %[g0], 16, %[xa]\n\t"
            "v_and_b32 %[g1], 0xffff0000, %[xa]\n\t"
            "v_lshlrev_b32 %[g2], 16, %[xb]\n\t"
            "v_and_b32 %[g3], 0xffff0000, %[xb]\n\t"
            "v_mul_f32 %[g0], %[r], %[g0]\n\t"
            "v_mul_f32 %[g1], %[r], %[g1]\n\t"
            "v_mul_f32 %[g2], %[r], %[g2]\n\t"
            "v_mul_f32 %[g3], %[r], %[g3]\n\t"
            "v_cvt_pk_bf16_f32 %[xa], %[g0], %[g1]\n\t"
            "v_cvt_pk_bf16_f32 %[xb], %[g2], %[g3]\n\t"
            "v_lshlrev_b32 %[g0], 16, %[xa]\n\t"
            "v_and_b32 %[g1], 0xffff0000, %[xa]\n\t"
            "v_lshlrev_b32 %[g2], 16, %[xb]\n\t"
            "v_and_b32 %[g3], 0xffff0000, %[xb]\n\t"
            "v_fma_f32 %[g0], %[g0], %[l2e], %[nm]\n\t"
            "v_fma_f32 %[g1], %[g1], %[l2e], %[nm]\n\t"
            "v_fma_f32 %[g2], %[g2], %[l2e], %[nm]\n\t"
            "v_fma_f32 %[g3], %[g3], %[l2e], %[nm]"
            : [xa] "=&v"(xa), [xb] "=&v"(xb), [g0] "=&v"(arg[0]), [g1] "=&v"(arg[1]), [g2] "=&v"(arg[2]), [g3] "=&v"(arg[3])
            : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [r] "s"(rcp), [l2e] "v"(L2E), [nm] "v"(neg_ml2));
    } else {
        const T x0 = round_chain_h<T, FAST>(a0, c, rcp), x1 = round_chain_h<T, FAST>(a1, c, rcp);
        const T x2 = round_chain_h<T, FAST>(a2, c, rcp), x3 = round_chain_h<T, FAST>(a3, c, rcp);
        xa = bits16(x0) | (bits16(x1) << 16);
        xb = bits16(x2) | (bits16(x3) << 16);
        arg[0] = __builtin_fmaf((float)x0, L2E, neg_ml2);
        arg[1] = __builtin_fmaf((float)x1, L2E, neg_ml2);
        arg[2] = __builtin_fmaf((float)x2, L2E, neg_ml2);
        arg[3] = __builtin_fmaf((float)x3, L2E, neg_ml2);
    }
}
// the four exponentials of a quad and their sums into the two partial sums of the row (one assembly block: left to the
// compiler the additions are SLP-packed into v_pk_add_f32 with s_nop hazards and all exponentials sink to the end of the
// step, away from the MFMAs they are meant to cover).  Every v_add is 4 instructions behind its v_exp (transcendental
// forwarding hazard: 1 wait state).  Hand-placed v_pk_add_f32 on a register pair (two instead of four additions, 78 instead
// of 89 VALU instructions per block, bit-identical sums) is 1.5 % SLOWER in the scoring loop (round 4, same box): a plain fp32
// VOP2 issues in 2.5 cycles on this part, the packed form does not, and the pair adds a dependent chain.
// first quad of a step: the partial sums are WRITTEN (e0 + e2, e1 + e3) - the same values as ((0 + e0) + e2), ((0 + e1) + e3), two v_add
// and two v_mov fewer per 32x32 block
__device__ static inline void quad_sum_first(const float (&arg)[4], float& ps0, float& ps1) {
    float e0, e1, e2, e3;
    asm("v_exp_f32 %[e0], %[a0]\n\t"
        "v_exp_f32 %[e1], %[a1]\n\t"
        "v_exp_f32 %[e2], %[a2]\n\t"
        "v_exp_f32 %[e3], %[a3]\n\t"
        "s_nop 0\n\t"
        "v_add_f32 %[p0], %[e0], %[e2]\n\t"
        "v_add_f32 %[p1], %[e1], %[e3]"
        : [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "=&v"(e3), [p0] "=&v"(ps0), [p1] "=&v"(ps1)
        : [a0] "v"(arg[0]), [a1] "v"(arg[1]), [a2] "v"(arg[2]), [a3] "v"(arg[3]));
}
__device__ static inline void quad_sum(const float (&arg)[4], float& ps0, float& ps1) {
    float e0, e1, e2, e3;
    asm("v_exp_f32 %[e0], %[a0]\n\t"
        "v_exp_f32 %[e1], %[a1]\n\t"
        "v_exp_f32 %[e2], %[a2]\n\t"
        "v_exp_f32 %[e3], %[a3]\n\t"
        "v_add_f32 %[p0], %[p0], %[e0]\n\t"
        "v_add_f32 %[p1], %[p1], %[e1]\n\t"
        "v_add_f32 %[p0], %[p0], %[e2]\n\t"
        "v_add_f32 %[p1], %[p1], %[e3]"
        : [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "=&v"(e3), [p0] "+v"(ps0), [p1] "+v"(ps1)
        : [a0] "v"(arg[0]), [a1] "v"(arg[1]), [a2] "v"(arg[2]), [a3] "v"(arg[3]));
}
// maximum of the 16 chain results held as 8 packed registers
template <typename T> __device__ static inline float max_packed16(const uint32_t (&xp)[8]) {
    if constexpr (std::is_same<T, _Float16>::value) {
        h2v m = __builtin_bit_cast(h2v, xp[0]);
#pragma unroll
        for (int p = 1; p < 8; ++p) m = __builtin_elementwise_max(m, __builtin_bit_cast(h2v, xp[p]));
        return fmaxf((float)m[0], (float)m[1]);
    } else {
        float m = -INFINITY;
#pragma unroll
        for (int p = 0; p < 8; ++p) m = fmaxf(m, fmaxf(pair_lo<T>(xp[p]), pair_hi<T>(xp[p])));
        return m;
    }
}

constexpr float PA2_SUM_LIMIT = 65536.f;
constexpr float PA2_SUM_LOW = 9.5367431640625e-07f;  // 2^-20: lower bound for the FIRST block of an item (reference still 0)

// ---- static, exactly balanced partition of pass A (round 2) --------------------------------------------------------------
// The (row tile, head, key tile) space is laid out as ONE sequence of key tiles - units u = rt * Hkv + h in order, each with the
// key tiles up to the causal limit of its last row - and cut into equal ranges, one per persistent block.  A block walks its
// range as segments (= consecutive key tiles of one unit); a unit's statistics come out as one partial per block that touched
// it (slot = ordinal of the block inside the unit).  Built on the host per launch (a few hundred integer operations) and passed
// BY VALUE: no index arithmetic with divisions in the kernel, no tail of unevenly loaded blocks.
constexpr int PLAN_MAX_BLOCKS = 256;
struct PaPlan {
    uint16_t unit[PLAN_MAX_BLOCKS + 1];  // block b starts at key tile tile[b] of unit unit[b] and ends where block b+1 starts
    uint16_t tile[PLAN_MAX_BLOCKS + 1];
    uint16_t perm[PLAN_MAX_BLOCKS];      // hardware block -> range of the partition (round 6, key-per-lane kernel): workgroup b runs on XCD b % 8 and every
                                         // XCD has its own L2, so the ranges are dealt head by head - the ~32 blocks of an XCD then stream the key tiles
                                         // of ONE KV head (two XCDs per head at Hkv = 4) instead of every head's, and the fabric carries each head's
                                         // keys to two L2s, not eight
    int nb;                              // blocks
    int max_seg;                         // most blocks that touch one unit (= partial statistics per row)
};
// key tiles of row tile rt (rows rt*rows .. +rows): up to the causal limit of its last row (virtual keys sink ++ ctx ++ repeat)
__host__ __device__ static inline int plan_ntiles(int rt, int rows, int R, int q_len, int sink, int m) {
    const int r0 = rt * rows, r1 = (R - 1 < r0 + rows - 1) ? R - 1 : r0 + rows - 1;
    const int qmax = (r0 / q_len == r1 / q_len) ? (r1 % q_len) : (q_len - 1);
    return (sink + m + qmax + 1 + SC_TILE - 1) / SC_TILE;
}
// cost model of the partition, in quarter key tiles: entering a unit (query rows, first chain, bookkeeping) and a key tile that
// reaches into the causal zone of the unit's rows (the block waits for the waves that run the masked epilogue: 1.45 x); a plain
// tile costs 4  (4 / 6 / 8 and a switch cost of 0 / 4 / 6 / 12 measured: within 1 % of each other, 6 / 0 best)
constexpr int PLAN_SWITCH_Q = 0, PLAN_MASKED_Q = 6;
// does key tile t hold the causal limit of some row of row tile rt?  (those rows' wave runs the masked epilogue there and the
// other waves of the block wait for it at the hand-over)  The rows of a row tile are one or two runs of consecutive positions.
static inline bool plan_tile_masked(int rt, int t, int rows, int R, int q_len, int sink, int m) {
    const int r0 = rt * rows, r1 = (R - 1 < r0 + rows - 1) ? R - 1 : r0 + rows - 1;
    const int lo = t * SC_TILE, hi = lo + SC_TILE - 1;  // a row with limit L has tile t partially hidden iff lo <= L < hi
    auto hit = [&](int qa, int qb) { const int La = sink + m + qa, Lb = sink + m + qb; return La < hi && Lb >= lo; };
    if (r0 / q_len == r1 / q_len) return hit(r0 % q_len, r1 % q_len);
    return hit(r0 % q_len, q_len - 1) || hit(0, r1 % q_len) || (r1 / q_len - r0 / q_len > 1);
}
static bool make_plan(PaPlan& p, int rows, int sink, int m, int q_len, int G, int Hkv) {
    const int R = G * q_len, RT = (R + rows - 1) / rows;
    const int64_t U = (int64_t)RT * Hkv;
    if (U > 65535 || (sink + m + q_len) / SC_TILE + 1 > 65535) return false;
    constexpr int SW = PLAN_SWITCH_Q, MQ = PLAN_MASKED_Q;
    // cost of the first t tiles of row tile rt (entering included)
    auto cost = [&](int rt, int t) -> int64_t {
        int64_t c = SW;
        for (int i = 0; i < t; ++i) c += plan_tile_masked(rt, i, rows, R, q_len, sink, m) ? MQ : 4;
        return c;
    };
    // inverse: the smallest t with cost(rt, t) >= c
    auto tile_at = [&](int rt, int64_t c) -> int {
        int64_t acc = SW;
        int t = 0;
        while (acc < c) { acc += plan_tile_masked(rt, t, rows, R, q_len, sink, m) ? MQ : 4; ++t; }
        return t;
    };
    int64_t W = 0, tiles = 0;
    for (int rt = 0; rt < RT; ++rt) {
        const int nt = plan_ntiles(rt, rows, R, q_len, sink, m);
        W += cost(rt, nt) * Hkv;
        tiles += (int64_t)nt * Hkv;
    }
    p.nb = (int)(tiles < PLAN_MAX_BLOCKS ? tiles : PLAN_MAX_BLOCKS);
    // cut b in absolute tile index: from the cost axis, then clamped so that every block keeps at least one tile
    int u = 0;
    int64_t cbefore = 0, tbefore = 0;  // cost / tiles of the units before u
    int nt = plan_ntiles(0, rows, R, q_len, sink, m);
    int64_t cu = cost(0, nt);
    int64_t prev = -1;
    int wu = 0;           // unit that holds absolute tile index `g` (second cursor, for the conversion back)
    int64_t wbefore = 0;
    int wnt = nt;
    for (int b = 0; b <= p.nb; ++b) {
        const int64_t target = W * b / p.nb;
        while (u < U && cbefore + cu <= target) {
            cbefore += cu;
            tbefore += nt;
            ++u;
            if (u < U) { nt = plan_ntiles(u / Hkv, rows, R, q_len, sink, m); cu = cost(u / Hkv, nt); }
        }
        int64_t g = tbefore + ((u < U) ? tile_at(u / Hkv, target - cbefore) : 0);
        if (g <= prev) g = prev + 1;
        if (g < b) g = b;
        if (g > tiles - (p.nb - b)) g = tiles - (p.nb - b);
        if (b == 0) g = 0;
        prev = g;
        while (wu < U && wbefore + wnt <= g) {
            wbefore += wnt;
            ++wu;
            if (wu < U) wnt = plan_ntiles(wu / Hkv, rows, R, q_len, sink, m);
        }
        p.unit[b] = (uint16_t)wu;
        p.tile[b] = (uint16_t)(g - wbefore);
    }
    for (int b = p.nb + 1; b <= PLAN_MAX_BLOCKS; ++b) { p.unit[b] = p.unit[p.nb]; p.tile[b] = p.tile[p.nb]; }
    int best = 1, run = 1;  // most blocks per unit: a run of block starts inside one unit (+ the block that opened it)
    for (int b = 1; b < p.nb; ++b) {
        if (p.tile[b] > 0 && p.unit[b] == p.unit[b - 1]) ++run;
        else run = (p.tile[b] > 0) ? 2 : 1;
        if (run > best) best = run;
    }
    p.max_seg = best;
    // hardware block -> range: ranges sorted by the KV head of their first unit (stable), the sorted list dealt to the XCDs in contiguous eighths
    for (int b = 0; b < PLAN_MAX_BLOCKS; ++b) p.perm[b] = (uint16_t)b;
#ifndef KVZ_PLAN_PERM
#define KVZ_PLAN_PERM 1
#endif
    if (KVZ_PLAN_PERM && p.nb % 8 == 0 && Hkv > 1) {
        uint16_t sorted[PLAN_MAX_BLOCKS];
        int n = 0;
        for (int h = 0; h < Hkv; ++h)
            for (int b = 0; b < p.nb; ++b)
                if ((int)(p.unit[b] % (unsigned)Hkv) == h) sorted[n++] = (uint16_t)b;
        if (n == p.nb)
            for (int b = 0; b < p.nb; ++b) p.perm[b] = sorted[(b % 8) * (p.nb / 8) + b / 8];
    }
    return true;
}

template <typename T, int D, bool FAST>
__global__ __launch_bounds__(PA_WAVES * 64, PA_WAVES / 4) void score_rowstat2_kernel(ScoreArgs a, PaPlan plan) {
    constexpr int NWAVES = PA_WAVES;
    typedef ScoreCfg<D> C;
    typedef typename Mfma32<T>::v8 v8;
    constexpr int QG_BYTES = 32 * C::ROW_BYTES;  // one row group of one wave
    constexpr int RING = 3;  // key-tile buffers: tile p of the block's stream lives in buffer p % 3 (160 KiB of LDS at D = 128)
    __shared__ __attribute__((aligned(16))) char lds[RING * C::TILE_BYTES + NWAVES * PA_RG * QG_BYTES];
    constexpr int PIECES = C::TILE_BYTES / 1024 / NWAVES;  // LDS-DMA instructions per wave and tile
    constexpr float L2E = 1.44269504088896340736f;
    constexpr int NB = SC_TILE / 32;  // 32-key blocks per tile
    static_assert(NB == 4, "the block pipeline alternates two register sets over an even number of blocks per tile");

    const int R = a.G * a.q_len;
    const int KT = a.sink + a.m + a.q_len;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int diag0 = a.sink + a.m;  // first key that can be masked for some row
    const int off_ctx = a.start - a.sink;                       // virtual -> cache row, ctx segment
    const int off_rep = a.klen - a.q_len - a.sink - a.m;        // virtual -> cache row, repeat segment
    const uint32_t lane_off = stage_lane_offset<D, NWAVES>(wave, lane);

    struct Item { int k, h, rt, z, t_lo, t_hi; };  // unit, KV head, row tile, ordinal of the partial, key tiles [t_lo, t_hi)
    // exactly balanced static partition (PaPlan): this block's range of the tile sequence, walked as segments; Item.k = unit
    const int u_first = plan.unit[blockIdx.x], t_first = plan.tile[blockIdx.x];
    const int u_end = plan.unit[blockIdx.x + 1], t_end = plan.tile[blockIdx.x + 1];  // exclusive: (u_end, t_end)
    int ord0 = 0;  // ordinal of the first segment inside its unit = earlier blocks that also started inside it (+ the opener)
    if (t_first > 0) {
        ord0 = 1;
        for (int bb = (int)blockIdx.x - 1; bb > 0 && plan.unit[bb] == u_first && plan.tile[bb] > 0; --bb) ++ord0;
    }
    auto item_from = [&](int u) -> Item {
        Item it;
        it.k = u; it.h = it.rt = it.z = 0; it.t_lo = it.t_hi = 0;
        if (u > u_end || (u == u_end && t_end == 0)) return it;
        it.rt = a.dh.div(u);
        it.h = u - it.rt * a.n_kv_heads;
        const int r0 = it.rt * PA_ROWS, r1 = min(R - 1, r0 + PA_ROWS - 1);
        const int h0 = a.dq.div(r0), h1 = a.dq.div(r1);
        const int qmax = (h0 == h1) ? (r1 - h1 * a.q_len) : (a.q_len - 1);
        const int ntiles = (a.sink + a.m + qmax + 1 + SC_TILE - 1) / SC_TILE;
        it.t_lo = (u == u_first) ? t_first : 0;
        it.t_hi = (u == u_end) ? t_end : ntiles;
        it.z = (u == u_first) ? ord0 : 0;
        return it;
    };
    const int first_item = u_first;
    auto valid = [](const Item& it) { return it.t_lo < it.t_hi; };
    // tiles (128 consecutive virtual keys) that lie inside ONE segment are consecutive rows of the cache: [tc_lo, tc_hi) inside
    // the ctx chunk, [tr_lo, tr_hi) inside the repeat chunk, [0, ts_hi) inside the sink; every other tile straddles a boundary
    // (or the end) and takes the per-lane path
    const int ts_hi = a.sink / SC_TILE;
    const int tc_lo = (a.sink + SC_TILE - 1) / SC_TILE, tc_hi = (a.sink + a.m) / SC_TILE;
    const int tr_lo = (a.sink + a.m + SC_TILE - 1) / SC_TILE, tr_hi = KT / SC_TILE;
    const uint32_t lds0 = lds_addr(lds);
    const char* const kbase = reinterpret_cast<const char*>(a.k);
    const int64_t khs = a.k_head_stride * 2;
    auto stage = [&](int b, int h, int t) __attribute__((always_inline)) {
        const uint32_t dst = lds0 + (uint32_t)(b * C::TILE_BYTES);
        const char* kh = kbase + (int64_t)h * khs;
        const int kv0 = t * SC_TILE;
        int off = 0;
        bool linear = true;
        if (t >= tc_lo && t < tc_hi) off = off_ctx;
        else if (t >= tr_lo && t < tr_hi) off = off_rep;
        else if (t >= ts_hi) linear = false;
        if (linear) {
            stage_tile_linear_a<D, NWAVES>(dst, kh + (int64_t)(kv0 + off) * C::ROW_BYTES, lane_off, wave);
        } else {  // (inlined: a call would open with s_waitcnt vmcnt(0) and drain the tiles in flight)
            constexpr int ROWS_PER_INSTR = 1024 / C::ROW_BYTES;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int ci = i * NWAVES + wave;  // wave-uniform 1-KiB piece of the tile
                const int row = ci * ROWS_PER_INSTR + lane / C::CPR;
                const int pch = lane % C::CPR;
                const int chunk = (D == 128) ? (pch ^ (row & 15)) : (pch ^ ((row >> 1) & 7));
                const int kv = min(kv0 + row, KT - 1);
                const int crow = kv + (kv < a.sink ? 0 : (kv < a.sink + a.m ? off_ctx : off_rep));
                lds_dma16a(kh, (uint32_t)(crow * C::ROW_BYTES + chunk * 16), dst + (uint32_t)(ci * 1024));
            }
        }
    };
    auto stage_q = [&](const Item& it) __attribute__((always_inline)) {
        // 32 rows per group, same swizzle as a key tile; rows beyond R shadow row R-1.  The rows of a group lie in at most
        // two query heads of the KV head: one scalar division per group, none per lane.  (Inlined: a call drains the DMA queue.)
        const char* qh = reinterpret_cast<const char*>(a.q) + (int64_t)it.h * a.G * a.q_head_stride * 2;
        const int64_t hs = a.q_head_stride * 2;
        constexpr int ROWS_PER_INSTR = 1024 / C::ROW_BYTES;
#pragma unroll
        for (int g = 0; g < PA_RG; ++g) {
            const uint32_t buf = lds0 + (uint32_t)(RING * C::TILE_BYTES + (wave * PA_RG + g) * QG_BYTES);
            const int r0 = it.rt * PA_ROWS + (wave * PA_RG + g) * 32;
            const int rc0 = min(r0, R - 1);
            const int g0 = a.dq.div(rc0), qi0 = rc0 - g0 * a.q_len;  // wave-uniform
            if (a.q_len >= 32) {
                // at most two query heads per group: a wave-uniform base, a shift for the row and one select for the rows that
                // belong to the next head - no per-lane multiply or division (the generic form below costs ~38 instructions per
                // piece, 1 500 cycles per item switch in the in-kernel timeline)
                const uint32_t base = (uint32_t)(g0 * (int)hs + qi0 * C::ROW_BYTES);
                const uint32_t wrapd = (uint32_t)((int)hs - a.q_len * C::ROW_BYTES);
                const int nfirst = a.q_len - qi0;  // rows of the group that still lie in head g0
                const int tail = R - 1 - rc0;      // rows beyond the last one shadow it
                const int lrow = lane / C::CPR, pch = lane % C::CPR;
#pragma unroll
                for (int i = 0; i < 32 / ROWS_PER_INSTR; ++i) {
                    const int row = i * ROWS_PER_INSTR + lrow;
                    const int chunk = (D == 128) ? (pch ^ (row & 15)) : (pch ^ ((row >> 1) & 7));
                    const int rowc = min(row, tail);
                    const uint32_t voff = base + (uint32_t)(rowc * C::ROW_BYTES + chunk * 16) + (rowc >= nfirst ? wrapd : 0u);
                    lds_dma16a(qh, voff, buf + (uint32_t)(i * 1024));
                }
            } else {  // (tiny chunks only: a group of 32 rows spans several query heads)
#pragma unroll
                for (int i = 0; i < 32 / ROWS_PER_INSTR; ++i) {
                    const int row = i * ROWS_PER_INSTR + lane / C::CPR;
                    const int pch = lane % C::CPR;
                    const int chunk = (D == 128) ? (pch ^ (row & 15)) : (pch ^ ((row >> 1) & 7));
                    int gg = g0, qi = qi0 + min(row, R - 1 - rc0);  // clamp to the last row
                    const int dg = a.dq.div(qi);
                    gg += dg;
                    qi -= dg * a.q_len;
                    lds_dma16a(qh, (uint32_t)(gg * (int)hs + qi * C::ROW_BYTES + chunk * 16), buf + (uint32_t)(i * 1024));
                }
            }
        }
    };
    FragAddr<D> fa0;
    fa0.init(lds, l31, half);
    auto read_q = [&](v8 (&dst)[PA_RG][C::KK]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < PA_RG; ++g) {
            FragAddr<D> fq;
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) fq.a[kk] = fa0.a[kk] + (uint32_t)(RING * C::TILE_BYTES + (wave * PA_RG + g) * QG_BYTES);
            u32x4 tmp[C::KK];
            frag_load<D>(tmp, fq, 0);
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) dst[g][kk] = __builtin_bit_cast(v8, tmp[kk]);
        }
    };
    struct Rows { int r[PA_RG], limit[PA_RG]; };
    auto rows_of = [&](const Item& it) __attribute__((always_inline)) -> Rows {
        Rows w;
#pragma unroll
        for (int g = 0; g < PA_RG; ++g) {
            w.r[g] = it.rt * PA_ROWS + (wave * PA_RG + g) * 32 + l31;
            const int rc = min(w.r[g], R - 1);
            w.limit[g] = a.sink + a.m + a.dq.mod(rc);  // key j (virtual index) is visible to query i iff j <= sink + m + i  (score.py:67-85)
        }
        return w;
    };
    // fragments of block kb of LDS buffer b: both compile-time, so the buffer and block offsets fold into the ds_read immediates -
    // up to 64 KiB: the third buffer of the ring at D = 128 lies beyond the 16-bit offset field and is read through a second set of
    // bases (one address add per fragment register and read otherwise: 8 VALU instructions per 32-key block of that buffer)
    // (fp16 only: the bf16 rounding chain needs the 8 registers - with them the kernel sits at 256 VGPRs and spills)
    constexpr bool FAR_BASE = KVZ_PA_FAR_BASE && std::is_same<T, _Float16>::value && RING * C::TILE_BYTES > 65536;
    FragAddr<D> fa_far;
#pragma unroll
    for (int kk = 0; kk < C::KK; ++kk) {
        fa_far.a[kk] = fa0.a[kk] + 65536u;
        if (FAR_BASE) asm volatile("" : "+v"(fa_far.a[kk]));
    }
    auto load_frags = [&](u32x4 (&fr)[C::KK], auto b_tag, auto kb_tag) __attribute__((always_inline)) {
        constexpr int off = decltype(b_tag)::value * C::TILE_BYTES + decltype(kb_tag)::value * 32 * C::ROW_BYTES;
        if constexpr (FAR_BASE && off >= 65536) frag_load<D>(fr, fa_far, off - 65536);
        else frag_load<D>(fr, fa0, off);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;

    Item cur = item_from(first_item);
    if (!valid(cur)) return;
    Item nxt = item_from(cur.k + 1);
    bool sq_in_next = false, sq_done = false;
    int sq_t = cur.t_lo;
    auto sq_stage = [&](int b) __attribute__((always_inline)) {
        stage(b, sq_in_next ? nxt.h : cur.h, sq_t);
        ++sq_t;
        if (sq_t >= (sq_in_next ? nxt.t_hi : cur.t_hi)) {
            if (!sq_in_next && valid(nxt)) {
                sq_in_next = true;
                sq_t = nxt.t_lo;
            } else {
                sq_done = true;
            }
        }
    };
    stage_q(cur);
    sq_stage(0);
    int staged = 1;  // tiles of the block's stream staged so far (stream position p -> buffer p % 3)
    if (!sq_done) {
        sq_stage(1);
        staged = 2;
    }
    // the query rows and the first tile are needed now, the second tile at the first hand-over (which waits for it)
    if (staged == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else stage_wait();
    block_barrier();
    v8 bq[PA_RG][C::KK];
    read_q(bq);
    u32x4 fr[2][C::KK];
    load_frags(fr[0], I0{}, I0{});
    load_frags(fr[1], I0{}, I1{});
#pragma unroll
    for (int g = 0; g < PA_RG; ++g)
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
            asm volatile("" : "+v"(bq[g][kk]));  // the rows are in registers: the area is free
        }
    if (valid(nxt)) stage_q(nxt);
    Rows rows = rows_of(cur);

    int pbuf = 0;  // buffer of the tile being computed (= sp % 3)
    int sp = 0;    // its position in the block's stream
    int t = cur.t_lo;
    float m_ref[PA_RG], nml2_ref[PA_RG], l_run[PA_RG];  // reference (16-bit value), -fl(reference * log2e), sum of 2^(x*log2e + nml2)
    int wmin[PA_RG], t_hidden;  // t_hidden: first tile that no row of this wave sees
    auto start_item = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < PA_RG; ++g) {
            m_ref[g] = 0.f;  // reference 0 until a block says otherwise: its sum of exponentials leaves [2^-20, 2^16] (first block
            nml2_ref[g] = 0.f;  // of the item: both bounds, later blocks: the upper one) -> cold path, reference = block maximum
            l_run[g] = 0.f;
            // smallest / largest causal limit of the 32 rows of the group, in closed form (the rows are consecutive positions of
            // one query head, or run over into the next one): a shuffle reduction here costs 700-1 500 cycles of LDS latency per
            // item switch, paid at the first use of wmin (in-kernel timeline)
            const int r0 = cur.rt * PA_ROWS + (wave * PA_RG + g) * 32;
            const int rc0 = min(r0, R - 1), last = min(r0 + 31, R - 1);
            const int qi0 = a.dq.mod(rc0), n = last - rc0;
            const bool wraps = qi0 + n >= a.q_len;
            wmin[g] = a.sink + a.m + (wraps ? 0 : qi0);                       // keys <= wmin are visible to every row of the group,
            const int th = (a.sink + a.m + (wraps ? a.q_len - 1 : qi0 + n)) / SC_TILE + 1;  // keys beyond the largest limit to none
            t_hidden = (g == 0) ? th : max(t_hidden, th);
        }
        t_hidden = max(t_hidden, cur.t_lo + 1);
    };
    start_item();

    f16v acc[2][PA_RG];  // accumulators of block kb (index kb & 1)
    const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // One pipeline step: the matrix chains of the NEXT block (fragment set frn -> accn) are issued inside the epilogue of
    // the CURRENT block (accc, first key k0).  WITH_MFMA = false drains the pipeline (last block of an item).  `hook` runs
    // after the first quarter of the step (fragment prefetch / tile hand-over): by then the last MFMA of the previous step
    // has read the fragment registers that the prefetch overwrites.
    // (ps_low: a float, or std::false_type where the lower bound is known to be 0 - steps 1-3 of a tile: one compare and one scalar OR less)
    auto step = [&](f16v (&accn)[PA_RG], const f16v (&accc)[PA_RG], const u32x4 (&frn)[C::KK], int k0, auto ps_low, auto mask_tag,
                    auto mfma_tag, auto&& hook) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(mask_tag)::value;
        constexpr bool WITH_MFMA = decltype(mfma_tag)::value;
        uint32_t xp[PA_RG][8];
        float ps0[PA_RG], ps1[PA_RG];
        int rel[PA_RG];
#pragma unroll
        for (int g = 0; g < PA_RG; ++g) {
            rel[g] = rows.limit[g] - (k0 + 4 * half);  // key offset (i&3)+8*(i>>2) of accumulator i is visible iff <= rel
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            float arg[PA_RG][4];
            // -- first half: one MFMA per chain, conversions + rounding chain + exponent arguments of 4 logits per row group
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (WITH_MFMA) {
#pragma unroll
                for (int c = 0; c < MfmaSched<C::KK>::count(2 * qd); ++c)
#pragma unroll
                    for (int g = 0; g < PA_RG; ++g) {
                        const int kk = MfmaSched<C::KK>::first(2 * qd) + c;
                        accn[g] = Mfma32<T>::mfma(__builtin_bit_cast(v8, frn[kk]), bq[g][kk], kk == 0 ? zero16 : accn[g]);
                    }
            }
#pragma unroll
            for (int g = 0; g < PA_RG; ++g) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = qd * 4 + j;
                    v[j] = (!MASK || (i & 3) + 8 * (i >> 2) <= rel[g]) ? accc[g][i] : -INFINITY;  // -inf survives the chain
                }
                quad_args<T, FAST>(v[0], v[1], v[2], v[3], xp[g][2 * qd], xp[g][2 * qd + 1], arg[g], a.c, a.rcp, L2E, nml2_ref[g]);
            }
            // -- second half: (second MFMA,) the four exponentials and their sums
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (WITH_MFMA) {
#pragma unroll
                for (int c = 0; c < MfmaSched<C::KK>::count(2 * qd + 1); ++c)
#pragma unroll
                    for (int g = 0; g < PA_RG; ++g) {
                        const int kk = MfmaSched<C::KK>::first(2 * qd + 1) + c;
                        accn[g] = Mfma32<T>::mfma(__builtin_bit_cast(v8, frn[kk]), bq[g][kk], accn[g]);
                    }
            }
#pragma unroll
            for (int g = 0; g < PA_RG; ++g) {
                if (qd == 0) quad_sum_first(arg[g], ps0[g], ps1[g]);
                else quad_sum(arg[g], ps0[g], ps1[g]);
            }
            if (qd == 0) {
                __builtin_amdgcn_sched_barrier(0);
                hook();
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < PA_RG; ++g) {
            float ps = ps0[g] + ps1[g];
            bool off = !(ps <= PA2_SUM_LIMIT);
            if constexpr (std::is_same<decltype(ps_low), float>::value) off = off || ps < ps_low;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(off) != 0, 0)) {  // wave-uniform and rare: move the reference, redo
                asm volatile("" ::: "memory");                                                 // (keeps it a branch)
                const float tmax = max_packed16<T>(xp[g]);
                // up: a logit far above the reference; down (first block of an item only, nothing summed yet): all logits far below
                // it.  A lane whose 16 logits are all masked keeps its state (tmax = -inf).
                if (tmax > m_ref[g] || (l_run[g] == 0.f && tmax > -INFINITY)) {
                    const float nml2_new = -(tmax * L2E);
                    l_run[g] *= __builtin_amdgcn_exp2f(nml2_new - nml2_ref[g]);
                    m_ref[g] = tmax;
                    nml2_ref[g] = nml2_new;
                }
                ps = 0.f;
#pragma unroll
                for (int p = 0; p < 8; ++p)
                    ps += __builtin_amdgcn_exp2f(__builtin_fmaf(pair_lo<T>(xp[g][p]), L2E, nml2_ref[g])) +
                          __builtin_amdgcn_exp2f(__builtin_fmaf(pair_hi<T>(xp[g][p]), L2E, nml2_ref[g]));
            }
            l_run[g] += ps;
        }
    };
    // matrix chains of the first block of an item (nothing to overlap them with)
    auto chain0 = [&](f16v (&accn)[PA_RG], const u32x4 (&frn)[C::KK]) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk)
#pragma unroll
            for (int g = 0; g < PA_RG; ++g)
                accn[g] = Mfma32<T>::mfma(__builtin_bit_cast(v8, frn[kk]), bq[g][kk], kk == 0 ? zero16 : accn[g]);
        __builtin_amdgcn_sched_barrier(0);
    };

    bool next_ready = false;
    // Hand-over in the third step of a tile (B = its buffer).  Every fragment of the tile has been read by now (block 3 in the
    // second step).  With three buffers the tile two positions ahead goes into the buffer that the PREVIOUS hand-over freed,
    // so its DMA is issued BEFORE the barrier: a wave that arrives early issues while the others still compute, and after
    // the barrier only the fragment reads remain.  The counted wait leaves exactly the pieces issued here in flight.
    auto turnover = [&](auto b_tag) __attribute__((always_inline)) {
        constexpr int B = decltype(b_tag)::value;
        constexpr int B1 = (B + 1) % RING, B2 = (B + 2) % RING;
        int newer = 0;  // tiles staged here that come AFTER the next tile
        if (staged < sp + 2 && !sq_done) {  // the stream was starved (items of a single tile): the next tile itself is missing
            sq_stage(B1);
            ++staged;
        }
        if (staged < sp + 3 && staged >= sp + 2 && !sq_done) {
            sq_stage(B2);
            ++staged;
            newer = 1;
        }
        if (newer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");  // the next tile (and older pieces) landed
        else stage_wait();
        block_barrier();  // ... everybody's part has, and nobody reads tile B any more
        next_ready = staged >= sp + 2;
        if (next_ready) load_frags(fr[0], std::integral_constant<int, B1>{}, I0{});
    };
    auto tile_steps = [&](auto b_tag, auto mask_tag) __attribute__((always_inline)) {
        constexpr int B = decltype(b_tag)::value;
        constexpr int B1 = (B + 1) % RING;
        const int k0 = t * SC_TILE;
        // the second-dispatched half of the block loses VALU arbitration by age: it gets priority in steps 0 and 2 (the other
        // patterns - steps 0-2, always, steps 1 and 3 - measured within +-1 %, profiles/r3_passA_timeline.txt)
        const bool young = wave >= NWAVES / 2;
        if (young) __builtin_amdgcn_s_setprio(1);
        step(acc[1], acc[0], fr[1], k0, (t == cur.t_lo) ? PA2_SUM_LOW : 0.f, mask_tag, std::true_type{}, [&]() __attribute__((always_inline)) { load_frags(fr[0], b_tag, I2{}); });
        if (young) __builtin_amdgcn_s_setprio(0);
        step(acc[0], acc[1], fr[0], k0 + 32, std::false_type{}, mask_tag, std::true_type{}, [&]() __attribute__((always_inline)) { load_frags(fr[1], b_tag, I3{}); });
        if (young) __builtin_amdgcn_s_setprio(1);
        step(acc[1], acc[0], fr[1], k0 + 64, std::false_type{}, mask_tag, std::true_type{}, [&]() __attribute__((always_inline)) { turnover(b_tag); });
        if (young) __builtin_amdgcn_s_setprio(0);
        // the chain issued here belongs to block 0 of the next tile; after the last tile of an item it is simply not used
        // (one variant less of every tile body; the matrix pipe has the slack)
        step(acc[0], acc[1], fr[0], k0 + 96, std::false_type{}, mask_tag, std::true_type{}, [&]() __attribute__((always_inline)) {
            if (next_ready) load_frags(fr[1], std::integral_constant<int, B1>{}, I1{});
        });
    };
    auto tile_dispatch = [&](auto b_tag) __attribute__((always_inline)) {
        int wm = wmin[0];
#pragma unroll
        for (int g = 1; g < PA_RG; ++g) wm = min(wm, wmin[g]);
        const bool masked = t * SC_TILE + SC_TILE - 1 > wm;  // some key of the tile is hidden from some row of this wave
        if (masked) tile_steps(b_tag, std::true_type{});
        else tile_steps(b_tag, std::false_type{});
    };
    // A tile that lies entirely behind the causal limit of every row of this wave (the item's tile range is cut at the limit of
    // the LAST of its 256 rows; the waves holding the first rows see up to two such tiles per item, a whole key slice at worst):
    // nothing to compute - every term would be exp(-inf) = 0 - but the wave still stages its share of the tiles ahead and
    // takes part in the hand-over.  All later tiles of the item are hidden too, and the next item starts with its own chain0.
    // One copy of the code for all ring positions (fragment offsets at run time: 16 extra additions per skipped tile).
    auto tile_skip = [&]() __attribute__((always_inline)) {
        const int b1 = (pbuf == RING - 1) ? 0 : pbuf + 1, b2 = (b1 == RING - 1) ? 0 : b1 + 1;
        int newer = 0;
        if (staged < sp + 2 && !sq_done) {
            sq_stage(b1);
            ++staged;
        }
        if (staged < sp + 3 && staged >= sp + 2 && !sq_done) {
            sq_stage(b2);
            ++staged;
            newer = 1;
        }
        if (newer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else stage_wait();
        block_barrier();
        next_ready = staged >= sp + 2;
        if (next_ready) {
            frag_load<D>(fr[0], fa0, b1 * C::TILE_BYTES);
            frag_load<D>(fr[1], fa0, b1 * C::TILE_BYTES + 32 * C::ROW_BYTES);
        }
    };
    (void)diag0;

    chain0(acc[0], fr[0]);
    while (true) {
        if (t >= t_hidden) tile_skip();
        else if (pbuf == 0) tile_dispatch(I0{});
        else if (pbuf == 1) tile_dispatch(I1{});
        else tile_dispatch(I2{});
        pbuf = (pbuf == RING - 1) ? 0 : pbuf + 1;
        ++sp;
        ++t;
        if (t < cur.t_hi) continue;  // (acc[0] already holds block 0 of the next tile)

        // ---- item finished: partial statistics of this key slice (reference m_ref, sum relative to fl(m_ref*log2e)) ----
#pragma unroll
        for (int g = 0; g < PA_RG; ++g) {
            // merge the two half-waves (they saw disjoint keys of the same query row)
            const float m_o = __shfl_xor(m_ref[g], 32, 64);
            const float nml2_o = __shfl_xor(nml2_ref[g], 32, 64);
            const float l_o = __shfl_xor(l_run[g], 32, 64);
            const float M = fmaxf(m_ref[g], m_o);
            const float NML2 = (m_ref[g] >= m_o) ? nml2_ref[g] : nml2_o;
            const float Lp = l_run[g] * __builtin_amdgcn_exp2f(NML2 - nml2_ref[g]) + l_o * __builtin_amdgcn_exp2f(NML2 - nml2_o);
            if (half == 0 && rows.r[g] < R) {
                // (stored from assembly: a store the compiler knows about makes it wait on the counter that also holds the DMA)
                float2* dst = a.stats + ((int64_t)cur.z * a.n_kv_heads + cur.h) * a.stats_stride + rows.r[g];
                const float2 val = make_float2(M, Lp);
                asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(val) : "memory");
                if (cur.k != u_end) {
                    // this block walked the unit to its end: the slots beyond its own partial get the NEUTRAL statistic (-inf, 0), so
                    // that pass B can fetch a fixed number of slots per row without first reading how many the unit has (round 4: one
                    // dependent round trip less in the prologue of every pass-B block)
                    const float2 neutral = make_float2(-INFINITY, 0.f);
                    const int64_t slot_stride = (int64_t)a.n_kv_heads * a.stats_stride;   // (scalar: the next slot of the same row)
                    float2* dn = dst;
                    for (int sl = cur.z + 1; sl < a.max_seg; ++sl) {
                        dn += slot_stride;
                        asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dn), "v"(neutral) : "memory");
                    }
                }
            }
        }
        if (cur.k != u_end && threadIdx.x == 0) {
            // this block walked the unit to its end: the unit has cur.z + 1 partials (pass B merges exactly that many)
            int* dst = a.unit_nseg + cur.k;
            const int val = cur.z + 1;
            asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(val) : "memory");
        }
        if (!valid(nxt)) break;
        // ---- switch to the next item: fragments of its first two blocks are in registers, its query rows landed before the
        // last hand-over ----
        cur = nxt;
        read_q(bq);
#pragma unroll
        for (int g = 0; g < PA_RG; ++g)
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) asm volatile("" : "+v"(bq[g][kk]));
        chain0(acc[0], fr[0]);  // (the eight dependent MFMAs run under the index arithmetic below)
        nxt = item_from(cur.k + 1);
        t = cur.t_lo;
        rows = rows_of(cur);
        if (valid(nxt)) stage_q(nxt);
        if (sq_in_next) {  // the cursor was already inside the item that is now current
            sq_in_next = false;
            if (sq_done && valid(nxt)) {
                sq_done = false;
                sq_in_next = true;
                sq_t = nxt.t_lo;
            }
        }
        start_item();
    }
}

// ---- pass B: per-ctx-key maximum of the log-softmax over all query rows --------------------------------------------------
// ctx keys are the STATIONARY A operand (PB_RG groups of 32 per wave), query-row tiles stream through the three-buffer LDS ring
// and the query row is again the lane; the matrix chain of 32-row block b+1 rides inside the epilogue of block b (same in-wave
// pipeline as pass A).  Epilogue of four logits: one assembly block with the rounding chain and x - (m_r + log l_r) (see
// quad_args); blocks alternate between "hold" and "v_max3(best, hold, t)" so that two blocks share one maximum per key.
// Round 3: the partial statistics of pass A are merged in the PROLOGUE of this kernel (each block merges the segments of its own
// row slice into LDS: bit-identical to the former merge launch, which is gone together with the per-tile statistics DMA).
constexpr int PB_RG = 1;                         // groups of 32 stationary keys per wave
constexpr int PB_COLS = PB_WAVES * PB_RG * 32;   // stationary ctx keys per block
constexpr int PB_STAT_TILES = 56;                // row tiles per block whose merged statistics fit beside the ring (56 KiB)

// merged statistics (m_r, log l_r) of row r of KV head h from the partials of pass A; rows >= R: (+inf, 0) = never a maximum
__device__ static inline float2 merge_row_stats(const float2* __restrict__ stats, int64_t rows_total, int64_t i, int slices) {
    constexpr float L2E = 1.44269504088896340736f;
    float M = -INFINITY;
    for (int s = 0; s < slices; ++s) M = fmaxf(M, stats[s * rows_total + i].x);
    const float ML2 = M * L2E;
    float Lp = 0.f;
    for (int s = 0; s < slices; ++s) {
        const float2 ps = stats[s * rows_total + i];
        Lp += ps.y * __builtin_amdgcn_exp2f(ps.x * L2E - ML2);
    }
    const float delta = __builtin_fmaf(M, L2E, -ML2);
    return make_float2(M, logf(Lp) - delta * 0.69314718055994530942f);
}
template <typename T, int D, bool FAST>
__global__ __launch_bounds__(PB_WAVES * 64, PB_OCC) void score_colmax3_kernel(ScoreArgs a) {
    constexpr int NWAVES = PB_WAVES;
    typedef ScoreCfg<D> C;
    typedef typename Mfma32<T>::v8 v8;
    constexpr int RING = 3;  // query-row tile p of the block's slice lives in buffer p % 3 (see score_rowstat2_kernel)
    __shared__ __attribute__((aligned(16))) char lds[RING * C::TILE_BYTES + PB_STAT_TILES * SC_TILE * 8];
    __shared__ int nan_rows;  // some row of the slice has NaN statistics (inf / NaN in Q or K): the reference's softmax row is NaN and
                              // amax over the rows propagates it to EVERY key of the head (attention/score.py:59-63)
    float2* const lstat = reinterpret_cast<float2*>(lds + RING * C::TILE_BYTES);
    constexpr int PIECES = C::TILE_BYTES / 1024 / NWAVES;  // LDS-DMA instructions per wave and tile
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;

    const int SH = a.row_splits * a.n_kv_heads;
    const int bj = blockIdx.x % SH;            // (row slice, head)
    const int ctile = blockIdx.x / SH;         // ctx-key tile
    const int ysplit = bj % a.row_splits;
    const int h = bj / a.row_splits;
    const int R = a.G * a.q_len;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    // stationary operand: 32 ctx keys per group as the A operand (result row = key, 16 keys per lane)
    const int j0 = ctile * PB_COLS + wave * PB_RG * 32;
    v8 ak[PB_RG][C::KK];
#pragma unroll
    for (int g = 0; g < PB_RG; ++g) {
        const int j = min(j0 + g * 32 + l31, a.m - 1);
        const char* kp = reinterpret_cast<const char*>(a.k) + ((int64_t)h * a.k_head_stride + (int64_t)(a.start + j) * D) * 2 + half * 16;
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk)
            ak[g][kk] = __builtin_bit_cast(v8, *reinterpret_cast<const u32x4*>(kp + kk * 32));
    }  // (no wait here: the first two row tiles are staged while these loads are in flight)
    const int total_tiles = (R + SC_TILE - 1) / SC_TILE;
    const int per = (total_tiles + a.row_splits - 1) / a.row_splits;
    const int t_begin = ysplit * per;
    const int t_end = min(total_tiles, t_begin + per);

    const char* qbase = reinterpret_cast<const char*>(a.q) + (int64_t)h * a.G * a.q_head_stride * 2;
    auto rowptr = [&](int r) -> const char* {
        r = min(r, R - 1);
        const int g = r / a.q_len;
        const int qi = r - g * a.q_len;
        return qbase + ((int64_t)g * a.q_head_stride + (int64_t)qi * D) * 2;
    };
    const uint32_t lane_off = stage_lane_offset<D, NWAVES>(wave, lane);
    const uint32_t lds0 = lds_addr(lds);
    // staging cursor: tiles are staged in order, so the (query head, row in head) of a tile's first row is advanced
    // incrementally (one scalar division per block instead of one per tile)
    int sg_t = t_begin, sg_g = (t_begin * SC_TILE) / a.q_len, sg_qi = t_begin * SC_TILE - sg_g * a.q_len;
    auto stage = [&](int b) __attribute__((always_inline)) {  // stages tile sg_t into buffer b and advances the cursor
        const int t = sg_t;
        const uint32_t dst = lds0 + (uint32_t)(b * C::TILE_BYTES);
        const int r0 = t * SC_TILE;
        if (r0 + SC_TILE <= R && sg_qi + SC_TILE <= a.q_len) {
            stage_tile_linear_a<D, NWAVES>(dst, qbase + ((int64_t)sg_g * a.q_head_stride + (int64_t)sg_qi * D) * 2, lane_off, wave);
        } else {
            stage_tile<D, NWAVES>(lds + b * C::TILE_BYTES, r0, rowptr, wave, lane);
        }
        ++sg_t;
        sg_qi += SC_TILE;
        while (sg_qi >= a.q_len) { sg_qi -= a.q_len; ++sg_g; }
    };
    FragAddr<D> fa0;
    fa0.init(lds, l31, half);
    // the ds_read offset field holds 16 bits and the ring spans 96 KiB at D = 128: reads of the third buffer cost one address add per
    // fragment register (8 VALU instructions per 32-row block, found in the ISA: v_or_b32 in front of every ds_read_b128 of buffer 2).
    // This kernel has the registers for a second set of bases (191 of 256): buffers at or beyond 64 KiB are read through it.
    constexpr bool FAR_BASE = RING * C::TILE_BYTES > 65536;
    FragAddr<D> fa_far;
#pragma unroll
    for (int kk = 0; kk < C::KK; ++kk) {
        fa_far.a[kk] = fa0.a[kk] + 65536u;
        asm volatile("" : "+v"(fa_far.a[kk]));   // (opaque: the compiler would fold it back into fa0 + constant and re-add per read)
    }
    auto load_frags = [&](u32x4 (&fr)[C::KK], auto b_tag, auto kb_tag) __attribute__((always_inline)) {
        constexpr int off = decltype(b_tag)::value * C::TILE_BYTES + decltype(kb_tag)::value * 32 * C::ROW_BYTES;
        if constexpr (FAR_BASE && off >= 65536) frag_load<D>(fr, fa_far, off - 65536);
        else frag_load<D>(fr, fa0, off);
    };

    float best[PB_RG][16], hold[PB_RG][16];
#pragma unroll
    for (int g = 0; g < PB_RG; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) best[g][i] = hold[g][i] = -INFINITY;

    if (t_begin < t_end) {
        stage(0);
        // ---- merged statistics of the block's row slice -> LDS (the first tile is in flight meanwhile; the second one is issued
        // AFTER these loads: the prologue of all 256 blocks is one burst that fills a CU at ~11 B/clk, and the tile loop can start
        // as soon as the stationary keys, the first tile and the statistics are there - the second tile is awaited at the first
        // hand-over) ----
        {
            // two dependent round trips to memory per batch of rows: the units' partial counts, then all of their partials at once
            const int row_lo = t_begin * SC_TILE, nrows = (t_end - t_begin) * SC_TILE;
            if (threadIdx.x == 0) nan_rows = 0;
            __syncthreads();
            const int64_t rows_total = (int64_t)a.n_kv_heads * a.stats_stride;
            bool bad = false;
            constexpr int MAXS = 4;  // partials fetched in one go (more: the general loop below)
            if (a.max_seg <= MAXS) {
                // Round 4: the rows of a thread in batches of four, all sixteen partial loads of a batch issued before the first one
                // is used - ONE round trip (a row per loop iteration with its unit's partial count fetched first was two dependent
                // round trips per row, 3.5 rows per thread: the longest chain of the kernel's prologue)
                constexpr int RPT = 4;
                constexpr float L2E = 1.44269504088896340736f;
                for (int base = 0; base < nrows; base += NWAVES * 64 * RPT) {
                    int64_t ii[RPT];
#pragma unroll
                    for (int u = 0; u < RPT; ++u) {
                        const int idx = base + u * NWAVES * 64 + (int)threadIdx.x;
                        const int r = min(row_lo + idx, R - 1);               // (rows beyond R: loads of a valid row, result discarded)
                        ii[u] = (int64_t)h * a.stats_stride + r;
                    }
                    // (every row has a.max_seg slots: its unit's partials, then the neutral statistic written by the block that
                    // finished the unit - no per-unit count to fetch first)
                    const int nsl = a.max_seg;
                    float2 ps[RPT][MAXS];
#pragma unroll
                    for (int u = 0; u < RPT; ++u)
#pragma unroll
                        for (int sgm = 0; sgm < MAXS; ++sgm) ps[u][sgm] = a.stats[(int64_t)min(sgm, nsl - 1) * rows_total + ii[u]];
#pragma unroll
                    for (int u = 0; u < RPT; ++u) {
                        const int idx = base + u * NWAVES * 64 + (int)threadIdx.x;
                        if (idx >= nrows) continue;
                        float2 v = make_float2(INFINITY, 0.f);
                        if (row_lo + idx < R) {
                            float M = -INFINITY;
#pragma unroll
                            for (int sgm = 0; sgm < MAXS; ++sgm) if (sgm < nsl) M = fmaxf(M, ps[u][sgm].x);
                            const float ML2 = M * L2E;
                            float Lp = 0.f;
#pragma unroll
                            for (int sgm = 0; sgm < MAXS; ++sgm) if (sgm < nsl) Lp += ps[u][sgm].y * __builtin_amdgcn_exp2f(ps[u][sgm].x * L2E - ML2);
                            const float delta = __builtin_fmaf(M, L2E, -ML2);
                            v = make_float2(M, logf(Lp) - delta * 0.69314718055994530942f);
                            bad |= !(v.x == v.x) || !(v.y == v.y);
                        }
                        lstat[idx] = v;
                    }
                }
            } else {
                for (int idx = threadIdx.x; idx < nrows; idx += NWAVES * 64) {
                    const int r = row_lo + idx;
                    float2 v = make_float2(INFINITY, 0.f);
                    if (r < R) {
                        const int64_t i = (int64_t)h * a.stats_stride + r;
                        const int nseg = a.unit_nseg ? a.unit_nseg[(r / a.unit_rows) * a.n_kv_heads + h] : 1;
                        v = merge_row_stats(a.stats, rows_total, i, nseg);
                        bad |= !(v.x == v.x) || !(v.y == v.y);
                    }
                    lstat[idx] = v;
                }
            }
            if (bad) atomicOr(&nan_rows, 1);
        }

#pragma unroll
        for (int g = 0; g < PB_RG; ++g)
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) {
                asm volatile("" : "+v"(ak[g][kk]));  // the wait for the stationary keys belongs here
            }
        const bool two = t_begin + 1 < t_end;
        if (two) stage(1);  // (after every compiler-visible load has been waited for: its vmcnt(0) would wait for this tile as well)
        if (two) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");  // everything but the second tile
        else stage_wait();
        block_barrier();
        u32x4 fr[2][C::KK];
        load_frags(fr[0], I0{}, I0{});
        load_frags(fr[1], I0{}, I1{});
        f16v acc[2][PB_RG];
        const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int t = t_begin;
        auto mfma_step = [&](f16v (&accn)[PB_RG], const u32x4 (&frn)[C::KK], int kk) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < PB_RG; ++g) {
                accn[g] = Mfma32<T>::mfma(ak[g][kk], __builtin_bit_cast(v8, frn[kk]), kk == 0 ? zero16 : accn[g]);
            }
        };

        // one pipeline step: chain of the next block (frn -> accn) inside the epilogue of the current block (accc); ODD blocks
        // fold the held values of the previous block and their own into the running maxima
        auto step = [&](f16v (&accn)[PB_RG], const f16v (&accc)[PB_RG], const u32x4 (&frn)[C::KK], float2 st, auto odd_tag,
                        auto&& hook) __attribute__((always_inline)) {
            constexpr bool ODD = decltype(odd_tag)::value;
            // x - (m_r + log l_r): the two per-row statistics are added once per step and lane, the subtraction rides in the fma
            // of the rounding chain
            const float neg_mr = -(st.x + st.y);
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                float tv[PB_RG][4];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < MfmaSched<C::KK>::count(2 * qd); ++c) mfma_step(accn, frn, MfmaSched<C::KK>::first(2 * qd) + c);
#pragma unroll
                for (int g = 0; g < PB_RG; ++g) {
                    uint32_t xa, xb;
                    quad_args<T, FAST>(accc[g][4 * qd], accc[g][4 * qd + 1], accc[g][4 * qd + 2], accc[g][4 * qd + 3], xa, xb, tv[g], a.c, a.rcp,
                                       1.0f, neg_mr);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < MfmaSched<C::KK>::count(2 * qd + 1); ++c) mfma_step(accn, frn, MfmaSched<C::KK>::first(2 * qd + 1) + c);
#pragma unroll
                for (int g = 0; g < PB_RG; ++g) {
                    if constexpr (ODD) {
                        // (outputs NOT tied to the inputs: the allocator answered "+v" on the loop-carried maxima with a register copy
                        // per maximum and step; left to the compiler as plain fmaxf the maxima drift away from their block and spill)
                        // (one instruction per statement: an output without early-clobber may share a register with an input of
                        // its OWN instruction only)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float nb;
                            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(nb) : "v"(best[g][4 * qd + j]), "v"(hold[g][4 * qd + j]), "v"(tv[g][j]));
                            best[g][4 * qd + j] = nb;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) hold[g][4 * qd + j] = tv[g][j];  // (a renaming: the chain wrote the held values)
                    }
                }
                if (qd == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    hook();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        float2 st_nx[SC_TILE / 32];
        {
            const float2* const ls = lstat + l31;   // (first tile of the slice)
#pragma unroll
            for (int kb = 0; kb < SC_TILE / 32; ++kb) st_nx[kb] = ls[kb * 32];
        }
        auto tile_steps = [&](auto b_tag) __attribute__((always_inline)) {
            constexpr int B = decltype(b_tag)::value;
            constexpr int B1 = (B + 1) % RING, B2 = (B + 2) % RING;
            // (m_r, log l_r) of this lane's query row in each of the four 32-row blocks of the tile: read from LDS one tile AHEAD (last
            // step of the previous tile; round 4) - read at the top of the tile, the first rounding chain of step 0 waited for them
            float2 st[SC_TILE / 32];
#pragma unroll
            for (int kb = 0; kb < SC_TILE / 32; ++kb) st[kb] = st_nx[kb];
            // issue priority alternates between the two waves of a SIMD from step to step (the upper half of the block yields in
            // steps 0 and 2, leads in 1 and 3): -0.6 us per launch, measured same-box against no priorities and the opposite phase
            const bool young = wave >= NWAVES / 2;
            if (young) __builtin_amdgcn_s_setprio(0);
            step(acc[1], acc[0], fr[1], st[0], std::false_type{}, [&]() __attribute__((always_inline)) { load_frags(fr[0], b_tag, I2{}); });
            if (young) __builtin_amdgcn_s_setprio(1);
            step(acc[0], acc[1], fr[0], st[1], std::true_type{}, [&]() __attribute__((always_inline)) { load_frags(fr[1], b_tag, I3{}); });
            if (young) __builtin_amdgcn_s_setprio(0);
            step(acc[1], acc[0], fr[1], st[2], std::false_type{}, [&]() __attribute__((always_inline)) {
                // hand-over: the tile two positions ahead goes into the buffer the previous hand-over freed, its DMA is issued
                // BEFORE the barrier; the counted wait leaves exactly those pieces in flight
                if (t + 2 < t_end) {
                    stage(B2);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
                } else {
                    stage_wait();
                }
                block_barrier();  // the next tile has landed for everybody, every fragment of this tile has been read
                if (t + 1 < t_end) load_frags(fr[0], std::integral_constant<int, B1>{}, I0{});
            });
            // (after the last tile the chain issued here is not used)
            if (young) __builtin_amdgcn_s_setprio(1);
            step(acc[0], acc[1], fr[0], st[3], std::true_type{}, [&]() __attribute__((always_inline)) {
                if (t + 1 < t_end) {
                    load_frags(fr[1], std::integral_constant<int, B1>{}, I1{});
                    const float2* const ls = lstat + (t + 1 - t_begin) * SC_TILE + l31;
#pragma unroll
                    for (int kb = 0; kb < SC_TILE / 32; ++kb) st_nx[kb] = ls[kb * 32];
                }
            });
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) mfma_step(acc[0], fr[0], kk);
        __builtin_amdgcn_sched_barrier(0);
        // (the three ring positions follow each other in program order - one loop back-edge instead of three: a dispatch on the
        // ring position makes the allocator reconcile the registers of the 32-64 running maxima at the end of every variant)
        while (true) {
            tile_steps(I0{});
            if (++t >= t_end) break;
            tile_steps(I1{});
            if (++t >= t_end) break;
            tile_steps(I2{});
            if (++t >= t_end) break;
        }
    }
    // maximum over the 32 query-row lanes of each half-wave; lanes 16 / 48 then hold the 16 keys (i&3)+8*(i>>2)+4*half
    // (lane id recomputed: values kept alive across the loop for this epilogue would cost registers - one of them spilled)
    const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int l31_e = lane_e & 31, half_e = lane_e >> 5;
#pragma unroll
    for (int g = 0; g < PB_RG; ++g) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // all-reduce inside each row of 16 lanes with DPP (rotate by 8 and 4, then the two quad permutations: four VALU
            // instructions, no LDS), then one cross-row step
            float b = best[g][i];
            auto dpp = [](float v, auto ctrl_tag) __attribute__((always_inline)) -> float {
                return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl_tag)::value, 0xf, 0xf, false));
            };
            b = fmaxf(b, dpp(b, std::integral_constant<int, 0x128>{}));  // row_ror:8
            b = fmaxf(b, dpp(b, std::integral_constant<int, 0x124>{}));  // row_ror:4
            b = fmaxf(b, dpp(b, std::integral_constant<int, 0x4E>{}));   // quad_perm:[2,3,0,1]
            b = fmaxf(b, dpp(b, std::integral_constant<int, 0xB1>{}));   // quad_perm:[1,0,3,2]
            // rows 1 and 3 take in the maximum of the row before them (row_bcast:15, row mask 0b1010; rows 0 and 2 keep theirs): lanes
            // 16-31 / 48-63 now hold the maximum of their half-wave - no LDS round trip (ds_bpermute) per value
            b = fmaxf(b, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, b), __builtin_bit_cast(int, b), 0x142, 0xa, 0xf, false)));
            best[g][i] = b;
        }
        const bool poison = (t_begin < t_end) && nan_rows != 0;
        if (l31_e == 16) {
            if (a.log_out) {
                // log-softmax values are <= 0 (rounding may leave +1e-7: clamped, exp of either rounds to the same 16-bit 1.0), and
                // for non-positive floats "larger" is "smaller bit pattern": the maximum over the row slices is an unsigned minimum
                uint32_t* dst = a.log_out + (int64_t)h * a.log_head_stride;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int j = j0 + g * 32 + (i & 3) + 8 * (i >> 2) + 4 * half_e;
                    // (encoding: NaN -> 0 wins every minimum; a value >= 0 -> 1, the smallest positive pattern, exp of it is 1.0)
                    const uint32_t enc = poison ? 0u : (best[g][i] >= 0.f ? 1u : __builtin_bit_cast(uint32_t, best[g][i]));
                    if (j < a.m) atomicMin(dst + j, enc);
                }
            } else {
                float* dst = a.colpart + ((int64_t)ysplit * a.n_kv_heads + h) * a.m;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int j = j0 + g * 32 + (i & 3) + 8 * (i >> 2) + 4 * half_e;
                    if (j < a.m) dst[j] = poison ? __builtin_nanf("") : best[g][i];
                }
            }
        }
    }
}

// ---- pass A with one KEY per lane, exact pruning of pass B (round 5) ---------------------------------------------------------------------
// score_rowstatT2_kernel is score_rowstat2_kernel with the MFMA operands swapped: the 32 query rows of a wave are the A operand (registers),
// key tiles stream through the LDS ring as the B operand, so a lane holds ONE key and its 16 accumulators are 16 query rows
// (k -> row (k & 3) + 8 (k >> 2) + 4 half).  Row sums become 16 lane-partial accumulators, reduced across lanes once per item; what the layout
// buys is the per-key maximum over the 32 rows of the wave - lane-local over the 16 registers plus one half swap - written as u[g][j] for
// the ctx keys.  With n_r = -(m_r + log l_r):   max_{r in g} (x_rj + n_r) <= u_gj + max_g n   and   t_j >= max_g (u_gj + min_g n),
// so the column maximum t_j only has to be recomputed for the (32-row group, 32-key block) pairs that bound cannot rule out - exactly (the
// bounds use the fp32 addition the recomputation applies to a logit and a row's statistic, monotone in both arguments): 18 % of the pairs
// on the benchmark's Gaussian logits, 3-21 % on copy-like prompts (tools/prune_bound_sim.py).  score_merge_kernel merges the partial
// statistics into n_r and the group bounds, score_bounds2_kernel compacts the candidate pairs, score_colmax_sparse_kernel recomputes them.

// first half of the rounding chain for four logits: x = half(float(half(acc)) * rcp) (FAST) or the division; packed results only
// (round 6: v_fma_mixlo_f16 / v_fma_mixhi_f16 do the multiplication and the second rounding in one instruction per logit - six instead of
// eight per four logits, the same bits for all 65 536 inputs - and change nothing in the scoring loop: 836 k vs 840 k tokens/s,
// profiles/r6_mixlo_ab.txt.  The kernel pays for energy per logit, not for instruction slots.)
template <typename T, bool FAST>
__device__ static inline void quad_round(float a0, float a1, float a2, float a3, uint32_t& xa, uint32_t& xb, float c, float rcp) {
    if constexpr (std::is_same<T, _Float16>::value && FAST) {
        float g0, g1, g2, g3;
        asm("v_cvt_pk_f16_f32 %[xa], %[a0], %[a1]\n\t"
            "v_cvt_pk_f16_f32 %[xb], %[a2], %[a3]\n\t"
            "v_fma_mix_f32 %[g0], %[xa], %[r], 0 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g1], %[xa], %[r], 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g2], %[xb], %[r], 0 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g3], %[xb], %[r], 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_cvt_pk_f16_f32 %[xa], %[g0], %[g1]\n\t"
            "v_cvt_pk_f16_f32 %[xb], %[g2], %[g3]"
            : [xa] "=&v"(xa), [xb] "=&v"(xb), [g0] "=&v"(g0), [g1] "=&v"(g1), [g2] "=&v"(g2), [g3] "=&v"(g3)
            : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [r] "s"(rcp));
    } else if constexpr (std::is_same<T, __bf16>::value && FAST) {
        // (the first twelve instructions of quad_args' bf16 block: plain fp32 multiplications, see there)
        float g0, g1, g2, g3;
        asm("v_cvt_pk_bf16_f32 %[xa], %[a0], %[a1]\n\t"
            "v_cvt_pk_bf16_f32 %[xb], %[a2], %[a3]\n\t"
            "v_lshlrev_b32 %[g0], 16, %[xa]\n\t"
            "v_and_b32 %[g1], 0xffff0000, %[xa]\n\t"
            "v_lshlrev_b32 %[g2], 16, %[xb]\n\t"
            "v_and_b32 %[g3], 0xffff0000, %[xb]\n\t"
            "v_mul_f32 %[g0], %[r], %[g0]\n\t"
            "v_mul_f32 %[g1], %[r], %[g1]\n\t"
            "v_mul_f32 %[g2], %[r], %[g2]\n\t"
            "v_mul_f32 %[g3], %[r], %[g3]\n\t"
            "v_cvt_pk_bf16_f32 %[xa], %[g0], %[g1]\n\t"
            "v_cvt_pk_bf16_f32 %[xb], %[g2], %[g3]"
            : [xa] "=&v"(xa), [xb] "=&v"(xb), [g0] "=&v"(g0), [g1] "=&v"(g1), [g2] "=&v"(g2), [g3] "=&v"(g3)
            : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [r] "s"(rcp));
    } else {
        const T x0 = round_chain_h<T, FAST>(a0, c, rcp), x1 = round_chain_h<T, FAST>(a1, c, rcp);
        const T x2 = round_chain_h<T, FAST>(a2, c, rcp), x3 = round_chain_h<T, FAST>(a3, c, rcp);
        xa = bits16(x0) | (bits16(x1) << 16);
        xb = bits16(x2) | (bits16(x3) << 16);
    }
}
// second half: exponent arguments x * log2e + n_k with one addend per logit (four different rows), the four exponentials
template <typename T>
__device__ static inline void quad_exp4(uint32_t xa, uint32_t xb, float L2E, float n0, float n1, float n2, float n3, float (&e)[4]) {
    if constexpr (std::is_same<T, _Float16>::value) {
        float g0, g1, g2, g3;
        asm("v_fma_mix_f32 %[g0], %[xa], %[l2e], %[n0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g2], %[xb], %[l2e], %[n2] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g1], %[xa], %[l2e], %[n1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g3], %[xb], %[l2e], %[n3] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_exp_f32 %[e0], %[g0]\n\t"
            "v_exp_f32 %[e2], %[g2]\n\t"
            "v_exp_f32 %[e1], %[g1]\n\t"
            "v_exp_f32 %[e3], %[g3]"
            : [g0] "=&v"(g0), [g1] "=&v"(g1), [g2] "=&v"(g2), [g3] "=&v"(g3), [e0] "=&v"(e[0]), [e1] "=&v"(e[1]), [e2] "=&v"(e[2]), [e3] "=&v"(e[3])
            : [xa] "v"(xa), [xb] "v"(xb), [l2e] "v"(L2E), [n0] "v"(n0), [n1] "v"(n1), [n2] "v"(n2), [n3] "v"(n3));
    } else {
        e[0] = __builtin_amdgcn_exp2f(__builtin_fmaf(pair_lo<T>(xa), L2E, n0));
        e[1] = __builtin_amdgcn_exp2f(__builtin_fmaf(pair_hi<T>(xa), L2E, n1));
        e[2] = __builtin_amdgcn_exp2f(__builtin_fmaf(pair_lo<T>(xb), L2E, n2));
        e[3] = __builtin_amdgcn_exp2f(__builtin_fmaf(pair_hi<T>(xb), L2E, n3));
    }
}
// the same with ONE wave-uniform addend (scalar register) for all four logits (round 6: the key-per-lane pass keeps one reference per wave)
template <typename T>
__device__ static inline void quad_exp4s(uint32_t xa, uint32_t xb, float L2E, float n /* wave-uniform */, float (&e)[4]) {
    if constexpr (std::is_same<T, _Float16>::value) {
        float g0, g1, g2, g3;
        asm("v_fma_mix_f32 %[g0], %[xa], %[l2e], %[n] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g2], %[xb], %[l2e], %[n] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g1], %[xa], %[l2e], %[n] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[g3], %[xb], %[l2e], %[n] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_exp_f32 %[e0], %[g0]\n\t"
            "v_exp_f32 %[e2], %[g2]\n\t"
            "v_exp_f32 %[e1], %[g1]\n\t"
            "v_exp_f32 %[e3], %[g3]"
            : [g0] "=&v"(g0), [g1] "=&v"(g1), [g2] "=&v"(g2), [g3] "=&v"(g3), [e0] "=&v"(e[0]), [e1] "=&v"(e[1]), [e2] "=&v"(e[2]), [e3] "=&v"(e[3])
            : [xa] "v"(xa), [xb] "v"(xb), [l2e] "v"(L2E), [n] "s"(n));
    } else {
        e[0] = __builtin_amdgcn_exp2f(__builtin_fmaf(pair_lo<T>(xa), L2E, n));
        e[1] = __builtin_amdgcn_exp2f(__builtin_fmaf(pair_hi<T>(xa), L2E, n));
        e[2] = __builtin_amdgcn_exp2f(__builtin_fmaf(pair_lo<T>(xb), L2E, n));
        e[3] = __builtin_amdgcn_exp2f(__builtin_fmaf(pair_hi<T>(xb), L2E, n));
    }
}
// bf16 (round 6): the sixteen logits of a step are unpacked ONCE (shift / mask), the maximum for the bound u and the exponent block both
// read the fp32 values: four fma (plain fp32: packed fp32 arithmetic issues slower on this part, see quad_args) and four exponentials
__device__ static inline void quad_exp4f(float x0, float x1, float x2, float x3, float L2E, float n /* wave-uniform */, float (&e)[4]) {
    float g0, g1, g2, g3;
    asm("v_fma_f32 %[g0], %[x0], %[l2e], %[n]\n\t"
        "v_fma_f32 %[g2], %[x2], %[l2e], %[n]\n\t"
        "v_fma_f32 %[g1], %[x1], %[l2e], %[n]\n\t"
        "v_fma_f32 %[g3], %[x3], %[l2e], %[n]\n\t"
        "v_exp_f32 %[e0], %[g0]\n\t"
        "v_exp_f32 %[e2], %[g2]\n\t"
        "v_exp_f32 %[e1], %[g1]\n\t"
        "v_exp_f32 %[e3], %[g3]"
        : [g0] "=&v"(g0), [g1] "=&v"(g1), [g2] "=&v"(g2), [g3] "=&v"(g3), [e0] "=&v"(e[0]), [e1] "=&v"(e[1]), [e2] "=&v"(e[2]), [e3] "=&v"(e[3])
        : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [l2e] "v"(L2E), [n] "s"(n));
}
// NaN-propagating maximum of sixteen fp32 values
__device__ static inline float max16_f32(const float (&x)[16]) {
    float f;
    asm("v_maximum3_f32 %0, %1, %2, %3" : "=v"(f) : "v"(x[0]), "v"(x[1]), "v"(x[2]));
    asm("v_maximum3_f32 %0, %1, %2, %3" : "=v"(f) : "v"(f), "v"(x[3]), "v"(x[4]));
    asm("v_maximum3_f32 %0, %1, %2, %3" : "=v"(f) : "v"(f), "v"(x[5]), "v"(x[6]));
    asm("v_maximum3_f32 %0, %1, %2, %3" : "=v"(f) : "v"(f), "v"(x[7]), "v"(x[8]));
    asm("v_maximum3_f32 %0, %1, %2, %3" : "=v"(f) : "v"(f), "v"(x[9]), "v"(x[10]));
    asm("v_maximum3_f32 %0, %1, %2, %3" : "=v"(f) : "v"(f), "v"(x[11]), "v"(x[12]));
    asm("v_maximum3_f32 %0, %1, %2, %3" : "=v"(f) : "v"(f), "v"(x[13]), "v"(x[14]));
    asm("v_maximum3_f32 %0, %1, %2, %2" : "=v"(f) : "v"(f), "v"(x[15]));
    return f;
}
// maximum of the sixteen 16-bit values of eight packed registers: its 16-bit pattern in the LOW half of the result (the high half is
// unspecified).  NaN propagates through the three-operand maxima (v_pk_maximum3_f16 / v_maximum3_f32); the last fp16 step is an IEEE maxNum.
template <typename T> __device__ static inline uint32_t max16_bits(const uint32_t (&xp)[8]) {
    uint32_t m;
    if constexpr (std::is_same<T, _Float16>::value) {
        asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(m) : "v"(xp[0]), "v"(xp[1]), "v"(xp[2]));
        asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(xp[3]), "v"(xp[4]));
        asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(xp[5]), "v"(xp[6]));
        asm("v_pk_maximum3_f16 %0, %1, %2, %2" : "=v"(m) : "v"(m), "v"(xp[7]));
        uint32_t r;
        asm("v_max_f16_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(r) : "v"(m));
        return r;
    } else {
        // bf16 has no packed maximum: the sixteen values as fp32 (shift / mask - the exponent block unpacks them the same way), eight
        // three-operand maxima; the result is one of the inputs, so its low 16 bits are zero and the shift back is exact
        float xf[16];
#pragma unroll
        for (int p = 0; p < 8; ++p) { xf[2 * p] = pair_lo<T>(xp[p]); xf[2 * p + 1] = pair_hi<T>(xp[p]); }
        return __builtin_bit_cast(uint32_t, max16_f32(xf)) >> 16;
    }
}

template <typename T, int D, bool FAST>
__global__ __launch_bounds__(PA_WAVES * 64, PA_WAVES / 4) void score_rowstatT2_kernel(ScoreArgs a, PaPlan plan) {
    constexpr int NWAVES = PA_WAVES;
    typedef ScoreCfg<D> C;
    typedef typename Mfma32<T>::v8 v8;
    constexpr int QG_BYTES = 32 * C::ROW_BYTES;  // one row group of one wave
    constexpr int RING = 3;  // key-tile buffers: tile p of the block's stream lives in buffer p % 3 (160 KiB of LDS at D = 128)
    __shared__ __attribute__((aligned(16))) char lds[RING * C::TILE_BYTES + NWAVES * PA_RG * QG_BYTES];
    // (no LDS left for a flag word: the block's "redo" mask - items whose sums left the fp32 range, bit = ordinal of the item in the block -
    // lives in the workspace.  The block clears it here, long before its first item ends; the hand-over barrier of the prologue orders the
    // clear before every wave's flag atomics.)
    uint32_t* const redo_word = a.redo + blockIdx.x;
    if (threadIdx.x == 0) {
        atomicAnd(redo_word, 0u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    constexpr int PIECES = C::TILE_BYTES / 1024 / NWAVES;  // LDS-DMA instructions per wave and tile
    constexpr float L2E = 1.44269504088896340736f;
    constexpr int NB = SC_TILE / 32;  // 32-key blocks per tile
    static_assert(NB == 4, "the block pipeline alternates two register sets over an even number of blocks per tile");

    const int R = a.G * a.q_len;
    const int KT = a.sink + a.m + a.q_len;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int diag0 = a.sink + a.m;  // first key that can be masked for some row
    const int off_ctx = a.start - a.sink;                       // virtual -> cache row, ctx segment
    const int off_rep = a.klen - a.q_len - a.sink - a.m;        // virtual -> cache row, repeat segment
    const uint32_t lane_off = stage_lane_offset<D, NWAVES>(wave, lane);

    struct Item { int k, h, rt, z, t_lo, t_hi; };  // unit, KV head, row tile, ordinal of the partial, key tiles [t_lo, t_hi)
    // exactly balanced static partition (PaPlan): this block's range of the tile sequence, walked as segments; Item.k = unit
    const int pb = plan.perm[blockIdx.x];   // this block's range of the partition (dealt head by head over the XCDs: see PaPlan)
    const int u_first = plan.unit[pb], t_first = plan.tile[pb];
    const int u_end = plan.unit[pb + 1], t_end = plan.tile[pb + 1];  // exclusive: (u_end, t_end)
    int ord0 = 0;  // ordinal of the first segment inside its unit = earlier blocks that also started inside it (+ the opener)
    if (t_first > 0) {
        ord0 = 1;
        for (int bb = pb - 1; bb > 0 && plan.unit[bb] == u_first && plan.tile[bb] > 0; --bb) ++ord0;
    }
    auto item_from = [&](int u) -> Item {
        Item it;
        it.k = u; it.h = it.rt = it.z = 0; it.t_lo = it.t_hi = 0;
        if (u > u_end || (u == u_end && t_end == 0)) return it;
        it.rt = a.dh.div(u);
        it.h = u - it.rt * a.n_kv_heads;
        const int r0 = it.rt * PA_ROWS, r1 = min(R - 1, r0 + PA_ROWS - 1);
        const int h0 = a.dq.div(r0), h1 = a.dq.div(r1);
        const int qmax = (h0 == h1) ? (r1 - h1 * a.q_len) : (a.q_len - 1);
        const int ntiles = (a.sink + a.m + qmax + 1 + SC_TILE - 1) / SC_TILE;
        it.t_lo = (u == u_first) ? t_first : 0;
        it.t_hi = (u == u_end) ? t_end : ntiles;
        it.z = (u == u_first) ? ord0 : 0;
        return it;
    };
    const int first_item = u_first;
    auto valid = [](const Item& it) { return it.t_lo < it.t_hi; };
    // tiles (128 consecutive virtual keys) that lie inside ONE segment are consecutive rows of the cache: [tc_lo, tc_hi) inside
    // the ctx chunk, [tr_lo, tr_hi) inside the repeat chunk, [0, ts_hi) inside the sink; every other tile straddles a boundary
    // (or the end) and takes the per-lane path
    const int ts_hi = a.sink / SC_TILE;
    const int tc_lo = (a.sink + SC_TILE - 1) / SC_TILE, tc_hi = (a.sink + a.m) / SC_TILE;
    const int tr_lo = (a.sink + a.m + SC_TILE - 1) / SC_TILE, tr_hi = KT / SC_TILE;
    const uint32_t lds0 = lds_addr(lds);
    const char* const kbase = reinterpret_cast<const char*>(a.k);
    const int64_t khs = a.k_head_stride * 2;
    auto stage = [&](int b, int h, int t) __attribute__((always_inline)) {
        const uint32_t dst = lds0 + (uint32_t)(b * C::TILE_BYTES);
        const char* kh = kbase + (int64_t)h * khs;
        const int kv0 = t * SC_TILE;
        int off = 0;
        bool linear = true;
        if (t >= tc_lo && t < tc_hi) off = off_ctx;
        else if (t >= tr_lo && t < tr_hi) off = off_rep;
        else if (t >= ts_hi) linear = false;
        if (linear) {
            stage_tile_linear_a<D, NWAVES>(dst, kh + (int64_t)(kv0 + off) * C::ROW_BYTES, lane_off, wave);
        } else {  // (inlined: a call would open with s_waitcnt vmcnt(0) and drain the tiles in flight)
            constexpr int ROWS_PER_INSTR = 1024 / C::ROW_BYTES;
            // (this kernel has no register to spare for loop-invariant per-lane offsets of a path taken three times per item: the lane id
            // is made opaque here, so the offsets are computed where they are used instead of being hoisted - and spilled)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int ci = i * NWAVES + wave;  // wave-uniform 1-KiB piece of the tile
                const int row = ci * ROWS_PER_INSTR + lane_o / C::CPR;
                const int pch = lane_o % C::CPR;
                const int chunk = (D == 128) ? (pch ^ (row & 15)) : (pch ^ ((row >> 1) & 7));
                const int kv = min(kv0 + row, KT - 1);
                const int crow = kv + (kv < a.sink ? 0 : (kv < a.sink + a.m ? off_ctx : off_rep));
                lds_dma16a(kh, (uint32_t)(crow * C::ROW_BYTES + chunk * 16), dst + (uint32_t)(ci * 1024));
            }
        }
    };
    auto stage_q = [&](const Item& it) __attribute__((always_inline)) {
        // 32 rows per group, same swizzle as a key tile; rows beyond R shadow row R-1.  The rows of a group lie in at most
        // two query heads of the KV head: one scalar division per group, none per lane.  (Inlined: a call drains the DMA queue.)
        const char* qh = reinterpret_cast<const char*>(a.q) + (int64_t)it.h * a.G * a.q_head_stride * 2;
        const int64_t hs = a.q_head_stride * 2;
        constexpr int ROWS_PER_INSTR = 1024 / C::ROW_BYTES;
#pragma unroll
        for (int g = 0; g < PA_RG; ++g) {
            const uint32_t buf = lds0 + (uint32_t)(RING * C::TILE_BYTES + (wave * PA_RG + g) * QG_BYTES);
            const int r0 = it.rt * PA_ROWS + (wave * PA_RG + g) * 32;
            const int rc0 = min(r0, R - 1);
            const int g0 = a.dq.div(rc0), qi0 = rc0 - g0 * a.q_len;  // wave-uniform
            if (a.q_len >= 32) {
                // at most two query heads per group: a wave-uniform base, a shift for the row and one select for the rows that
                // belong to the next head - no per-lane multiply or division (the generic form below costs ~38 instructions per
                // piece, 1 500 cycles per item switch in the in-kernel timeline)
                const uint32_t base = (uint32_t)(g0 * (int)hs + qi0 * C::ROW_BYTES);
                const uint32_t wrapd = (uint32_t)((int)hs - a.q_len * C::ROW_BYTES);
                const int nfirst = a.q_len - qi0;  // rows of the group that still lie in head g0
                const int tail = R - 1 - rc0;      // rows beyond the last one shadow it
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));   // (not hoisted: see stage)
                const int lrow = lane_o / C::CPR, pch = lane_o % C::CPR;
#pragma unroll
                for (int i = 0; i < 32 / ROWS_PER_INSTR; ++i) {
                    const int row = i * ROWS_PER_INSTR + lrow;
                    const int chunk = (D == 128) ? (pch ^ (row & 15)) : (pch ^ ((row >> 1) & 7));
                    const int rowc = min(row, tail);
                    const uint32_t voff = base + (uint32_t)(rowc * C::ROW_BYTES + chunk * 16) + (rowc >= nfirst ? wrapd : 0u);
                    lds_dma16a(qh, voff, buf + (uint32_t)(i * 1024));
                }
            } else {  // (tiny chunks only: a group of 32 rows spans several query heads)
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
#pragma unroll
                for (int i = 0; i < 32 / ROWS_PER_INSTR; ++i) {
                    const int row = i * ROWS_PER_INSTR + lane_o / C::CPR;
                    const int pch = lane_o % C::CPR;
                    const int chunk = (D == 128) ? (pch ^ (row & 15)) : (pch ^ ((row >> 1) & 7));
                    int gg = g0, qi = qi0 + min(row, R - 1 - rc0);  // clamp to the last row
                    const int dg = a.dq.div(qi);
                    gg += dg;
                    qi -= dg * a.q_len;
                    lds_dma16a(qh, (uint32_t)(gg * (int)hs + qi * C::ROW_BYTES + chunk * 16), buf + (uint32_t)(i * 1024));
                }
            }
        }
    };
    FragAddr<D> fa0;
    fa0.init(lds, l31, half);
    auto read_q = [&](v8 (&dst)[PA_RG][C::KK]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < PA_RG; ++g) {
            FragAddr<D> fq;
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) fq.a[kk] = fa0.a[kk] + (uint32_t)(RING * C::TILE_BYTES + (wave * PA_RG + g) * QG_BYTES);
            u32x4 tmp[C::KK];
            frag_load<D>(tmp, fq, 0);
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) dst[g][kk] = __builtin_bit_cast(v8, tmp[kk]);
        }
    };
    // ONE set of fragment registers (the row-per-lane kernel has two): fragment kk of block n + 2 is read into the registers of fragment kk of
    // block n + 1 as soon as the MFMA that consumed them has been issued - the 32 registers are where the 16 per-row sums and addends of
    // this layout live
    typedef const __attribute__((address_space(3))) u32x4* lds_frag_t;
    // (the ds_read offset field holds 16 bits and the ring spans 96 KiB at D = 128: the third buffer is read through a second set of bases -
    // without it every fragment read of that buffer costs an address add)
    constexpr bool FAR_BASE = RING * C::TILE_BYTES > 65536;
    FragAddr<D> fa_far;
#pragma unroll
    for (int kk = 0; kk < C::KK; ++kk) {
        fa_far.a[kk] = fa0.a[kk] + 65536u;
        if (FAR_BASE) asm volatile("" : "+v"(fa_far.a[kk]));
    }
    auto load_frag1 = [&](u32x4& dst, auto b_tag, auto kb_tag, int kk) __attribute__((always_inline)) {
        constexpr int off = decltype(b_tag)::value * C::TILE_BYTES + decltype(kb_tag)::value * 32 * C::ROW_BYTES;
        if constexpr (FAR_BASE && off >= 65536) dst = *(lds_frag_t)(uintptr_t)(fa_far.a[kk] + (off - 65536));
        else dst = *(lds_frag_t)(uintptr_t)(fa0.a[kk] + off);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;

    Item cur = item_from(first_item);
    if (!valid(cur)) return;
    Item nxt = item_from(cur.k + 1);
    bool sq_in_next = false, sq_done = false;
    int sq_t = cur.t_lo;
    auto sq_stage = [&](int b) __attribute__((always_inline)) {
        stage(b, sq_in_next ? nxt.h : cur.h, sq_t);
        ++sq_t;
        if (sq_t >= (sq_in_next ? nxt.t_hi : cur.t_hi)) {
            if (!sq_in_next && valid(nxt)) {
                sq_in_next = true;
                sq_t = nxt.t_lo;
            } else {
                sq_done = true;
            }
        }
    };
    stage_q(cur);
    sq_stage(0);
    int staged = 1;  // tiles of the block's stream staged so far (stream position p -> buffer p % 3)
    if (!sq_done) {
        sq_stage(1);
        staged = 2;
    }
    // the query rows and the first tile are needed now, the second tile at the first hand-over (which waits for it)
    if (staged == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else stage_wait();
    block_barrier();
    v8 bq[PA_RG][C::KK];
    read_q(bq);
    u32x4 fr[C::KK];
#pragma unroll
    for (int kk = 0; kk < C::KK; ++kk) load_frag1(fr[kk], I0{}, I0{}, kk);
#pragma unroll
    for (int g = 0; g < PA_RG; ++g)
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
            asm volatile("" : "+v"(bq[g][kk]));  // the rows are in registers: the area is free
        }
    if (valid(nxt)) stage_q(nxt);

    int pbuf = 0;  // buffer of the tile being computed (= sp % 3)
    int sp = 0;    // its position in the block's stream
    int t = cur.t_lo;
    // per-row state of the lane's 16 rows (accumulator k = row (k & 3) + 8 (k >> 2) + 4 half of the wave's group): lane-partial sum of
    // 2^(x log2e + nm) and the addend nm = -fl(ref * log2e).  The references are LANE-PRIVATE: a lane sees one key per block, its 16 rows take
    // their logits of the item's first block as references (init_refs), so nothing crosses lanes inside the loop; the 32 lanes of a half are
    // merged like partial statistics when the item ends.  References never move inside the pipeline; an item whose sums leave the fp32 range
    // (a logit ~88 above its row's reference) is redone by the slow loop at the end of the kernel.
    constexpr float RL2E = 0.69314718055994530942f;
    float lsum[16];
    float refw = 0.f, nmw = 0.f;   // the wave's reference (a 16-bit value, or 0) and the addend -fl(refw * log2e): scalar registers
    int wmin, wmax, t_hidden;   // keys <= wmin are visible to every row of the wave, keys > wmax to none; t_hidden: first tile that no row of this wave sees
    int item_qi0, item_iw;      // position of the wave's first row in its query head; rows of the group before the wrap into the next head (64: none)
    int mask_a0, mask_w, mask_qiw;   // causal mask of the wave's group (see step)
    const uint16_t* urow;            // u of the wave's group: + (kb * n_groups) * 32 + lane
    uint32_t u_spare, u_lane;        // byte offsets from urow: the spare row behind the array; the lane's part of its key's offset
    int u_c;                         // (lane & 31) - sink: ctx index of the lane's key minus k0
    int tile_ctx = 1, tile_ctx_full = 0;   // the current tile holds ctx keys / nothing but ctx keys (scalar registers: a bool carried through
                                           // the loop comes back as a lane mask, and the branch on it costs two vector instructions per step)

    auto start_item = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 16; ++k) lsum[k] = 0.f;   // (nm, nm_max: init_refs, behind the item's first chain)
        const int r0 = cur.rt * PA_ROWS + wave * 32;
        const int rc0 = min(r0, R - 1), last = min(r0 + 31, R - 1);
        const int qi0 = a.dq.mod(rc0), n = last - rc0;
        const bool wraps = qi0 + n >= a.q_len;
        wmin = a.sink + a.m + (wraps ? 0 : qi0);
        wmax = a.sink + a.m + (wraps ? a.q_len - 1 : qi0 + n);
        t_hidden = max(wmax / SC_TILE + 1, cur.t_lo + 1);
        // with d = kv - sink - m the rows before the wrap into the next query head (i < iw) see key kv iff i >= d - qi0, the rows after it
        // iff i >= d + iw; rows beyond the last one (they shadow it) get the mask of the positions they would have - never stored
        const int iw = wraps ? a.q_len - qi0 : 64;
        item_qi0 = qi0;
        item_iw = iw;
        mask_a0 = -a.sink - a.m - qi0;   // + k0 + (lane & 31) - 4 half -> d - qi0 - 4 half
        mask_w = iw;                     // - 4 half
        mask_qiw = qi0 + iw;
        // u: [Hkv, nkb, n_groups, 2 halves, 32 keys] 16-bit; k0 is a multiple of 32, so with c = (lane & 31) - sink the ctx index k0 + c lies in
        // key block k0 / 32 + (c >> 5) at position c & 31
        urow = a.colu + ((int64_t)cur.h * a.nkb * a.n_groups + (cur.rt * PA_WAVES + wave)) * 64;
        u_spare = (uint32_t)((((int64_t)a.n_kv_heads - cur.h) * a.nkb * a.n_groups - (cur.rt * PA_WAVES + wave)) * 128) + (uint32_t)(lane * 2);
        u_c = l31 - a.sink;
        u_lane = (uint32_t)((u_c >> 5) * a.n_groups * 128 + half * 64 + (u_c & 31) * 2);
    };
    start_item();

    // visibility bits of this lane's key k0 + (lane & 31) for the 16 rows (bit c_k): the general form, all ones where nothing is hidden
    auto vis_bits = [&](int k0) __attribute__((always_inline)) -> uint32_t {
        auto ones_from = [](int x) __attribute__((always_inline)) -> uint32_t { return x >= 32 ? 0u : (0xFFFFFFFFu << max(x, 0)); };
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int h4 = (lane_o >> 5) * 4;
        const int mask_a = mask_a0 + k0 + (lane_o & 31) - h4;
        if (mask_w >= 64) return ones_from(mask_a);   // (wave-uniform: the group does not wrap into the next query head - rows >= d - qi0 see the key)
        const int mw = mask_w - h4, mask_b = max(mask_a + mask_qiw, mw);
        return (ones_from(mask_a) & ~ones_from(mw)) | ones_from(mask_b);
    };
    f16v acc[2];
    const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mfma_step = [&](f16v& accn, int kk) __attribute__((always_inline)) {
        accn = Mfma32<T>::mfma(bq[0][kk], __builtin_bit_cast(v8, fr[kk]), kk == 0 ? zero16 : accn);
    };

    // one pipeline step: chain of the next 32-key block (fr -> accn) inside the epilogue of the current one (accc, first key k0); the
    // fragments of the block after the next one (ring buffer LB, block LKB) follow each MFMA into its registers - unconditionally: where
    // that block does not exist (end of the stream) the reads return whatever the buffer holds and the chain on them is never used
    auto step = [&](f16v& accn, const f16v& accc, int k0, auto s_tag, auto mask_tag, auto lb_tag, auto lkb_tag, auto&& hook) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(mask_tag)::value;
        uint32_t xp[8];
        // (MASK) bit c of vism <=> the row with c_k = c sees this lane's key: [mask_a, mask_w) and [mask_b, 32)
        uint32_t vism = 0xFFFFFFFFu;
        if (MASK) {
            vism = vis_bits(k0);
            asm volatile("" : "+v"(vism));
        }
        // -- first half: rounding chain of the 16 logits (four per half-group), one MFMA per half-group
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < MfmaSched<C::KK>::count(qd); ++c) {
                const int kk = MfmaSched<C::KK>::first(qd) + c;
                mfma_step(accn, kk);
                if (qd != 0) load_frag1(fr[kk], lb_tag, lkb_tag, kk);   // (half-group 0: after the hook - the hand-over is there)
            }
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = qd * 4 + j;
                v[j] = accc[k];
                if (MASK) {
                    constexpr int ck = 0;   // (placeholder: the bit index is passed as an immediate below)
                    (void)ck;
                    int sel;
                    float vo;
                    const float ninf = -INFINITY;
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(vism), "n"((qd * 4 + j) % 4 + 8 * ((qd * 4 + j) / 4)));   // all ones: visible
                    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(vo) : "v"(sel), "v"(v[j]), "s"(ninf));                              // -inf survives the chain
                    v[j] = vo;
                }
            }
            quad_round<T, FAST>(v[0], v[1], v[2], v[3], xp[2 * qd], xp[2 * qd + 1], a.c, a.rcp);
            if (qd == 0) {
                __builtin_amdgcn_sched_barrier(0);
                hook();
#pragma unroll
                for (int c = 0; c < MfmaSched<C::KK>::count(0); ++c) load_frag1(fr[MfmaSched<C::KK>::first(0) + c], lb_tag, lkb_tag, MfmaSched<C::KK>::first(0) + c);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // -- the block's largest logit of this lane (16 rows of one key): reference check, and the bound u for ctx keys
        constexpr bool F16 = std::is_same<T, _Float16>::value;
        float xf[16];   // (bf16: the 16 logits as fp32, unpacked once for the maximum and the exponent blocks)
        if constexpr (!F16) {
#pragma unroll
            for (int p = 0; p < 8; ++p) { xf[2 * p] = pair_lo<T>(xp[p]); xf[2 * p + 1] = pair_hi<T>(xp[p]); }
        }
        if (tile_ctx) {   // (wave-uniform: the tile holds ctx keys; nothing but temporaries behind this branch)
            // (ctx keys are never masked.)  Each half stores the maximum of ITS 16 rows (the two are merged by the reader); lanes whose key is
            // not a ctx key store into the spare bytes behind the array
            uint32_t u16;
            if constexpr (F16) u16 = max16_bits<T>(xp);
            else u16 = __builtin_bit_cast(uint32_t, max16_f32(xf)) >> 16;
            // (round 6: the block term of the offset moved into the scalar base of the store - four vector instructions fewer per step - is
            // 1.7 % SLOWER in the scoring loop: 16 more scalar spills and a second branch inside the step; profiles/r6_passA_small_steps_ab.txt)
            uint32_t off = (uint32_t)((k0 >> 5) * a.n_groups * 128) + u_lane;
            if (!tile_ctx_full) off = ((uint32_t)(k0 + u_c) < (uint32_t)a.m) ? off : u_spare;   // (wave-uniform: only the tiles at the ends of the ctx range)
            asm volatile("global_store_short %0, %1, %2" ::"v"(off), "v"(u16), "s"(urow) : "memory");
        }
        // -- second half: exponentials against the row references, accumulated into the lane-partial row sums
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < MfmaSched<C::KK>::count(4 + qd); ++c) {
                const int kk = MfmaSched<C::KK>::first(4 + qd) + c;
                mfma_step(accn, kk);
                load_frag1(fr[kk], lb_tag, lkb_tag, kk);
            }
            float e[4];
            if constexpr (F16) quad_exp4s<T>(xp[2 * qd], xp[2 * qd + 1], L2E, nmw, e);
            else quad_exp4f(xf[4 * qd], xf[4 * qd + 1], xf[4 * qd + 2], xf[4 * qd + 3], L2E, nmw, e);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float nl;   // (one instruction per statement, outputs not tied: see pass B's running maxima)
                asm("v_add_f32 %0, %1, %2" : "=v"(nl) : "v"(lsum[4 * qd + j]), "v"(e[j]));
                lsum[4 * qd + j] = nl;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // first block of an item (ring buffer B of its first tile): its fragments, its chain (nothing to overlap it with), block 1 behind it
    auto chain0 = [&](auto b_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) load_frag1(fr[kk], b_tag, I0{}, kk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
            mfma_step(acc[0], kk);
            load_frag1(fr[kk], b_tag, I1{}, kk);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // rounding chain of the 16 logits in `av` (masked) -> packed pairs
    auto round16 = [&](const f16v av, uint32_t vism, uint32_t (&xq)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            float v[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int kx = qd * 4 + jj;
                const int ck = (kx & 3) + 8 * (kx >> 2);
                const int sel = __builtin_amdgcn_sbfe((int)vism, ck, 1);
                const float lg = av[kx];
                v[jj] = __builtin_bit_cast(float, (__builtin_bit_cast(int, lg) & sel) | (~sel & (int)0xFF800000));
            }
            quad_round<T, FAST>(v[0], v[1], v[2], v[3], xq[2 * qd], xq[2 * qd + 1], a.c, a.rcp);
        }
    };
    auto dpp = [](float v, auto ctrl_tag) __attribute__((always_inline)) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl_tag)::value, 0xf, 0xf, false));
    };
    typedef std::integral_constant<int, 0x128> ROR8;   // row_ror:8
    typedef std::integral_constant<int, 0x124> ROR4;   // row_ror:4
    typedef std::integral_constant<int, 0x4E> QP2;     // quad_perm:[2,3,0,1]
    typedef std::integral_constant<int, 0xB1> QP1;     // quad_perm:[1,0,3,2]
    // reference of a new item (round 6): ONE per wave - the largest logit of the item's first block (32 rows x 32 keys, acc[0], just computed
    // by chain0).  Softmax is shift invariant and the partial statistics carry their reference, so any value in reach of the fp32 range does;
    // a shared one makes the addend a scalar register (16 vector registers fewer) and the end of an item a plain sum over the 32 lanes of a
    // half (no common-reference search, no rescaling exponentials: ~700 -> ~200 VALU instructions per item and wave).  Nothing visible: 0;
    // a NaN logit: NaN (every sum becomes NaN; a NaN row poisons its whole KV head in the reference, attention/score.py:59-63).
    auto init_refs = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // (MFMA result -> assembly block: wait states the compiler does not count)
        uint32_t xq[8];
        round16(acc[0], vis_bits(t * SC_TILE), xq);
        const uint32_t mb = max16_bits<T>(xq) & 0xFFFFu;
        float mx = pair_lo<T>(mb);
        const bool isn = !(mx == mx);
        mx = fmaxf(mx, dpp(mx, ROR8{}));
        mx = fmaxf(mx, dpp(mx, ROR4{}));
        mx = fmaxf(mx, dpp(mx, QP2{}));
        mx = fmaxf(mx, dpp(mx, QP1{}));
        {
            const uint32_t b = __builtin_bit_cast(uint32_t, mx);
            const auto sw = __builtin_amdgcn_permlane16_swap(b, b, false, false);
            mx = fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
        }
        {
            const uint32_t b = __builtin_bit_cast(uint32_t, mx);
            const auto sw = __builtin_amdgcn_permlane32_swap(b, b, false, false);
            mx = fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
        }
        float r = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, mx)));
        if (__builtin_amdgcn_ballot_w64(isn) != 0) r = __builtin_nanf("");
        refw = (r == -INFINITY) ? 0.f : r;
        nmw = -(refw * L2E);
    };
    // Hand-over in the third step of a tile (B = its buffer).  Every fragment of the tile has been read by now (block 3 in the
    // second step).  With three buffers the tile two positions ahead goes into the buffer that the PREVIOUS hand-over freed,
    // so its DMA is issued BEFORE the barrier: a wave that arrives early issues while the others still compute, and after
    // the barrier only the fragment reads remain.  The counted wait leaves exactly the pieces issued here in flight.
    auto turnover = [&](auto b_tag) __attribute__((always_inline)) {
        constexpr int B = decltype(b_tag)::value;
        constexpr int B1 = (B + 1) % RING, B2 = (B + 2) % RING;
        int newer = 0;  // tiles staged here that come AFTER the next tile
        if (staged < sp + 2 && !sq_done) {  // the stream was starved (items of a single tile): the next tile itself is missing
            sq_stage(B1);
            ++staged;
        }
        if (staged < sp + 3 && staged >= sp + 2 && !sq_done) {
            sq_stage(B2);
            ++staged;
            newer = 1;
        }
        if (newer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");  // the next tile (and older pieces) landed
        else stage_wait();
        block_barrier();  // ... everybody's part has, and nobody reads tile B any more
    };
    auto tile_steps = [&](auto b_tag, auto mask_tag) __attribute__((always_inline)) {
        constexpr int B = decltype(b_tag)::value;
        constexpr int B1 = (B + 1) % RING;
        typedef std::integral_constant<int, B1> IB1;
        const int k0 = t * SC_TILE;
        tile_ctx = __builtin_amdgcn_readfirstlane((k0 < a.sink + a.m && k0 + SC_TILE > a.sink) ? 1 : 0);
        tile_ctx_full = __builtin_amdgcn_readfirstlane((k0 >= a.sink && k0 + SC_TILE <= a.sink + a.m) ? 1 : 0);
        const bool young = wave >= NWAVES / 2;
        auto nohook = [&]() __attribute__((always_inline)) {};
        // a tile that holds a causal limit of the wave's rows (round 6): only the 32-key blocks that CUT the rows pay for the masks (the
        // diagonal band of 32 rows touches one or two of the four); the blocks before it take the plain step, the blocks behind it are
        // hidden from every row of the wave - and so is everything after them in this item: nothing to compute, nothing to prefetch (the
        // next item opens with its own chain), only the hand-over in the third step
        auto one = [&](f16v& accn, const f16v& accc, int kb0, auto s_tag, auto lb_tag, auto lkb_tag, auto&& hook) __attribute__((always_inline)) {
            if constexpr (!decltype(mask_tag)::value) {
                step(accn, accc, kb0, s_tag, std::false_type{}, lb_tag, lkb_tag, hook);
            } else {
#if KVZ_T2_BLOCKMASK >= 2
                if (kb0 + 31 <= wmin) step(accn, accc, kb0, s_tag, std::false_type{}, lb_tag, lkb_tag, hook);
                else
#endif
#if KVZ_T2_BLOCKMASK >= 1
                if (kb0 <= wmax) {
                    step(accn, accc, kb0, s_tag, std::true_type{}, lb_tag, lkb_tag, hook);
                } else {
                    // (the next block's chain and the fragment reads behind it are issued all the same - nothing will use them, but both
                    // sides of the branch then define the same registers and the allocator has nothing to carry through the join)
#pragma unroll
                    for (int hg = 0; hg < 8; ++hg) {
#pragma unroll
                        for (int c = 0; c < MfmaSched<C::KK>::count(hg); ++c) {
                            const int kk = MfmaSched<C::KK>::first(hg) + c;
                            mfma_step(accn, kk);
                            if (hg != 0) load_frag1(fr[kk], lb_tag, lkb_tag, kk);
                        }
                        if (hg == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            hook();
#pragma unroll
                            for (int c = 0; c < MfmaSched<C::KK>::count(0); ++c) load_frag1(fr[MfmaSched<C::KK>::first(0) + c], lb_tag, lkb_tag, MfmaSched<C::KK>::first(0) + c);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#else
                step(accn, accc, kb0, s_tag, std::true_type{}, lb_tag, lkb_tag, hook);
#endif
            }
        };
        if (young) __builtin_amdgcn_s_setprio(1);
        one(acc[1], acc[0], k0, I0{}, b_tag, I2{}, nohook);
        if (young) __builtin_amdgcn_s_setprio(0);
        one(acc[0], acc[1], k0 + 32, I1{}, b_tag, I3{}, nohook);
        if (young) __builtin_amdgcn_s_setprio(1);
        one(acc[1], acc[0], k0 + 64, I2{}, IB1{}, I0{}, [&]() __attribute__((always_inline)) { turnover(b_tag); });
        if (young) __builtin_amdgcn_s_setprio(0);
        // the chain issued here belongs to block 0 of the next tile; after the last tile of an item it is simply not used
        one(acc[0], acc[1], k0 + 96, I3{}, IB1{}, I1{}, nohook);
    };
    auto tile_dispatch = [&](auto b_tag) __attribute__((always_inline)) {
        const bool masked = t * SC_TILE + SC_TILE - 1 > wmin;  // some key of the tile is hidden from some row of this wave
        if (masked) tile_steps(b_tag, std::true_type{});
        else tile_steps(b_tag, std::false_type{});
    };
    // a tile that lies entirely behind the causal limit of every row of this wave: nothing to compute, but the wave still stages its
    // share of the tiles ahead and takes part in the hand-over (see the row-per-lane kernel).  Every later tile of the item is hidden
    // too, and the next item starts with its own chain0: no fragments are read here.
    auto tile_skip = [&]() __attribute__((always_inline)) {
        const int b1 = (pbuf == RING - 1) ? 0 : pbuf + 1, b2 = (b1 == RING - 1) ? 0 : b1 + 1;
        int newer = 0;
        if (staged < sp + 2 && !sq_done) {
            sq_stage(b1);
            ++staged;
        }
        if (staged < sp + 3 && staged >= sp + 2 && !sq_done) {
            sq_stage(b2);
            ++staged;
            newer = 1;
        }
        if (newer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else stage_wait();
        block_barrier();
    };
    (void)diag0;

    // partial statistics of the item's key slice, stored as (reference, sum relative to fl(reference * log2e)) like the row-per-lane
    // kernel's: lane k < 16 of each half holds row (k & 3) + 8 (k >> 2) + 4 half of the wave's group
    auto store_stats = [&](float myM, float myL) __attribute__((always_inline)) {
        const int row = cur.rt * PA_ROWS + wave * 32 + (l31 & 3) + 8 * ((l31 >> 2) & 3) + 4 * half;
        if (l31 < 16 && row < R) {
            float2* dst = a.stats + ((int64_t)cur.z * a.n_kv_heads + cur.h) * a.stats_stride + row;
            const float2 val = make_float2(myM, myL);
            asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(val) : "memory");
            if (cur.k != u_end) {   // this block walked the unit to its end: neutral statistics in the slots beyond its own partial
                const float2 neutral = make_float2(-INFINITY, 0.f);
                const int64_t slot_stride = (int64_t)a.n_kv_heads * a.stats_stride;
                float2* dn = dst;
                for (int sl = cur.z + 1; sl < a.max_seg; ++sl) {
                    dn += slot_stride;
                    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dn), "v"(neutral) : "memory");
                }
            }
        }
        if (cur.k != u_end && threadIdx.x == 0) {
            int* dst = a.unit_nseg + cur.k;
            const int val = cur.z + 1;
            asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(val) : "memory");
        }
    };
    // end of an item of the pipeline: the 16 lane-partial sums of each half (all against the wave's reference) become 16 row sums - four
    // DPP steps inside the rows of 16 lanes, then the other row of the half (v_permlane16_swap).  The pipeline never moves the reference:
    // a row sum that left the safe part of the fp32 range (a logit ~88 above the reference: inf; every visible logit ~70 below it: the
    // terms lose their low bits; NaN inputs: NaN) flags the item, which is then redone at the end of the kernel by a slow loop that
    // follows the logits with lane-private references.
    auto finish_item = [&]() __attribute__((always_inline)) {
        float myL = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float sm = lsum[k];
            sm += dpp(sm, ROR8{});
            sm += dpp(sm, ROR4{});
            sm += dpp(sm, QP2{});
            sm += dpp(sm, QP1{});
            {
                const uint32_t b = __builtin_bit_cast(uint32_t, sm);
                const auto sw = __builtin_amdgcn_permlane16_swap(b, b, false, false);
                sm = __builtin_bit_cast(float, (uint32_t)sw[0]) + __builtin_bit_cast(float, (uint32_t)sw[1]);
            }
            if (l31 == k) myL = sm;
        }
        {
            const int c = (l31 & 3) + 8 * ((l31 >> 2) & 3) + 4 * half;   // row of the group
            const int qi = (c < item_iw) ? item_qi0 + c : c - item_iw;   // its position: keys <= sink + m + qi are visible
            const bool stored = l31 < 16 && cur.rt * PA_ROWS + wave * 32 + c < R;
            const bool sees = cur.t_lo * SC_TILE <= a.sink + a.m + qi;   // the first key of the item's slice
            const bool bad = stored && (!(myL < 1.2676506e30f) || (sees && !(myL > 7.8886091e-31f)));   // 2^100, 2^-100
            if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(redo_word, 1u << ((cur.k - u_first) & 31));
        }
        store_stats(refw, myL);
    };
    chain0(I0{});
    init_refs();
    while (true) {
        if (t >= t_hidden) tile_skip();
        else if (pbuf == 0) tile_dispatch(I0{});
        else if (pbuf == 1) tile_dispatch(I1{});
        else tile_dispatch(I2{});
        pbuf = (pbuf == RING - 1) ? 0 : pbuf + 1;
        ++sp;
        ++t;
        if (t < cur.t_hi) continue;

        // ---- item finished (finish_item flags it for the slow loop when a row sum left the safe range) ----
        finish_item();
        if (!valid(nxt)) break;
        // ---- switch to the next item: its query rows landed before the last hand-over; its first tile is in buffer pbuf ----
        cur = nxt;
        read_q(bq);
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) asm volatile("" : "+v"(bq[0][kk]));
        if (pbuf == 0) chain0(I0{});
        else if (pbuf == 1) chain0(I1{});
        else chain0(I2{});
        nxt = item_from(cur.k + 1);
        t = cur.t_lo;
        if (valid(nxt)) stage_q(nxt);
        if (sq_in_next) {  // the cursor was already inside the item that is now current
            sq_in_next = false;
            if (sq_done && valid(nxt)) {
                sq_done = false;
                sq_in_next = true;
                sq_t = nxt.t_lo;
            }
        }
        start_item();
        init_refs();
    }

    // ---- items whose sums left the fp32 range: once more, one tile at a time, no pipeline, the references updated in every block (every
    // exponential <= 1: nothing can overflow; NaN inputs end as NaN statistics, as in the reference).  The whole block takes part. ----
    stage_wait();      // (the flag atomics of this wave are done as well)
    block_barrier();
    const uint32_t redo = (uint32_t)__builtin_amdgcn_readfirstlane((int)atomicOr(redo_word, 0u));   // (read at the L2, where the atomics ran)
    if (__builtin_expect(redo != 0, 0)) {
        constexpr float NM_UNSET = 65504.f * L2E;   // reference -65504: "no logit seen yet"
        typedef const __attribute__((address_space(3))) u32x4* lp_t;
        for (int u = u_first;; ++u) {
            const Item it = item_from(u);
            if (!valid(it)) break;
            if (!((redo >> ((u - u_first) & 31)) & 1u)) continue;
            cur = it;
            stage_q(cur);
            stage_wait();
            block_barrier();
            read_q(bq);
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) asm volatile("" : "+v"(bq[0][kk]));
            start_item();
            // lane-private references (a lane sees one key per block): nm = -fl(ref * log2e), moved in every block
            float nm[16];
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) nm[k2] = NM_UNSET;
            for (int tt = cur.t_lo; tt < cur.t_hi; ++tt) {
                stage(0, cur.h, tt);
                stage_wait();
                block_barrier();
                if (tt < t_hidden) {
#pragma unroll 1
                    for (int sb = 0; sb < 4; ++sb) {
                        const uint32_t off = (uint32_t)(sb * 32 * C::ROW_BYTES);
#pragma unroll
                        for (int kk = 0; kk < C::KK; ++kk) fr[kk] = *(lp_t)(uintptr_t)(fa0.a[kk] + off);
#pragma unroll
                        for (int kk = 0; kk < C::KK; ++kk) mfma_step(acc[1], kk);
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
                        uint32_t xq[8];
                        round16(acc[1], vis_bits(tt * SC_TILE + sb * 32), xq);
#pragma unroll
                        for (int k2 = 0; k2 < 16; ++k2) {
                            const float xv = (k2 & 1) ? pair_hi<T>(xq[k2 >> 1]) : pair_lo<T>(xq[k2 >> 1]);
                            float nn = fminf(nm[k2], -(xv * L2E));   // (-inf / NaN logits leave the reference alone)
                            nn = (nn >= NM_UNSET) ? 0.f : nn;         // (nothing finite seen yet: reference 0)
                            lsum[k2] *= __builtin_amdgcn_exp2f(nn - nm[k2]);
                            nm[k2] = nn;
                        }
#pragma unroll
                        for (int qd = 0; qd < 4; ++qd) {
                            float e[4];
                            quad_exp4<T>(xq[2 * qd], xq[2 * qd + 1], L2E, nm[4 * qd], nm[4 * qd + 1], nm[4 * qd + 2], nm[4 * qd + 3], e);
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) lsum[4 * qd + jj] += e[jj];
                        }
                    }
                }
                block_barrier();
            }
            // the 32 lanes of a half hold (reference, sum) pairs of row k: common reference = the largest one (smallest addend), sums
            // rescaled to it and added (all-reduce over the half as in finish_item)
            float myM = 0.f, myL = 0.f;
#pragma unroll 1
            for (int k = 0; k < 16; ++k) {
                float nk = nm[0], lk = lsum[0];
#pragma unroll
                for (int k2 = 1; k2 < 16; ++k2) {   // (rolled loop: no dynamic register index)
                    nk = (k == k2) ? nm[k2] : nk;
                    lk = (k == k2) ? lsum[k2] : lk;
                }
                float nmin = nk;
                nmin = fminf(nmin, dpp(nmin, ROR8{}));
                nmin = fminf(nmin, dpp(nmin, ROR4{}));
                nmin = fminf(nmin, dpp(nmin, QP2{}));
                nmin = fminf(nmin, dpp(nmin, QP1{}));
                {
                    const uint32_t b = __builtin_bit_cast(uint32_t, nmin);
                    const auto sw = __builtin_amdgcn_permlane16_swap(b, b, false, false);
                    nmin = fminf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
                }
                float sm = lk * __builtin_amdgcn_exp2f(nmin - nk);
                sm += dpp(sm, ROR8{});
                sm += dpp(sm, ROR4{});
                sm += dpp(sm, QP2{});
                sm += dpp(sm, QP1{});
                {
                    const uint32_t b = __builtin_bit_cast(uint32_t, sm);
                    const auto sw = __builtin_amdgcn_permlane16_swap(b, b, false, false);
                    sm = __builtin_bit_cast(float, (uint32_t)sw[0]) + __builtin_bit_cast(float, (uint32_t)sw[1]);
                }
                if (l31 == k) { myM = nmin; myL = sm; }
            }
            {   // the reference itself: ref = 16-bit(-nm / log2e) exactly (fl(ref * log2e) / log2e is ref (1 +- 2^-23), the next 16-bit
                // value is at least 2^-11 away)
                const T rh = (T)(-myM * RL2E);
                myM = (float)rh;
            }
            store_stats(myM, myL);
        }
    }
}

// ---- merged statistics, group bounds (round 5) ----------------------------------------------------------------------------------------
// One block per (256-row tile, KV head), one thread per row: n_r = -(m_r + log l_r) from the unit's partial statistics (the merge of the
// column-maximum pass: same expression), -inf for the padding rows of the last tile; (max, min) of n over each 32-row group, (NaN, NaN)
// for a group with a NaN row.  Block 0 resets the candidate counter.
__device__ static inline void score_merge_body(const ScoreArgs& a, const int unit) {
    const int rt = a.dh.div(unit), h = unit - rt * a.n_kv_heads;
    const int R = a.G * a.q_len;
    const int r = rt * PA_ROWS + (int)threadIdx.x;
    float n = -INFINITY;
    bool bad = false;
    if (r < R) {
        const int nseg = a.unit_nseg[unit];
        const float2 v = merge_row_stats(a.stats, (int64_t)a.n_kv_heads * a.stats_stride, (int64_t)h * a.stats_stride + r, nseg);
        n = -(v.x + v.y);
        bad = !(n == n);
    }
    a.nrow[((int64_t)h * a.n_groups) * 32 + r] = n;
    float mx = (r < R) ? n : -INFINITY, mn = (r < R) ? n : INFINITY;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        mn = fminf(mn, __shfl_xor(mn, o, 64));
    }
    const uint64_t bal = __ballot(bad);
    const int l = threadIdx.x & 63;
    const bool gbad = ((l < 32) ? (uint32_t)bal : (uint32_t)(bal >> 32)) != 0;
    if ((l & 31) == 0) {
        const int grp = rt * (PA_ROWS / 32) + (int)(threadIdx.x >> 5);
        const float qnan = __builtin_nanf("");
        const bool any = rt * PA_ROWS + (int)(threadIdx.x & ~31u) < R;
        a.gbound[(int64_t)h * a.n_groups + grp] = gbad ? make_float2(qnan, qnan) : (any ? make_float2(mx, mn) : make_float2(-INFINITY, -INFINITY));
        a.gcount[(int64_t)h * a.n_groups + grp] = 0;
    }
    if (unit == 0 && (int)threadIdx.x < a.n_kv_heads) a.counter[threadIdx.x] = 0;
}
__global__ __launch_bounds__(PA_ROWS) void score_merge_kernel(ScoreArgs a) { score_merge_body(a, (int)blockIdx.x); }

// ---- candidate pairs of pass B, compacted (round 5) -----------------------------------------------------------------------------------
// One block per (KV head, 32-key block).  Its [n_groups][32] slab of u is contiguous (28 KiB at the headline shape): every thread reads 8
// keys of one group per pass, all passes in flight at once.  LB_j = max_g (u_gj + nmin_g); group g is a candidate of the block iff
// u_gj + nmax_g >= LB_j for some key j of the block (the fp32 addition pass B applies to a logit and its row's statistic: monotone in
// both arguments, so the test is exact; NaN bounds keep the pair).  The candidates go into ONE list (entries), those of a block contiguous.
// A head with a row of NaN statistics is poisoned as the reference's amax over rows poisons it: the block writes the NaN code itself.
constexpr int BD2_THREADS = 256;
constexpr int BD2_MAXPASS = 8;   // groups per block <= 64 * 8 (host checks)
template <typename T>
__global__ __launch_bounds__(BD2_THREADS) void score_bounds2_kernel(ScoreArgs a) {
    const int kb = blockIdx.x, h = blockIdx.y;
    const int tid = threadIdx.x;
    const int oct = tid & 3, grow = tid >> 2;   // 8 keys [8 oct, 8 oct + 8) of group (pass * 64 + grow)
    __shared__ float s_part[64][33];
    __shared__ float s_lb8[8][32];
    __shared__ float s_lb[32];
    __shared__ uint32_t s_list[64 * BD2_MAXPASS];
    __shared__ int s_n, s_base, s_poison;
    if (tid == 0) { s_n = 0; s_poison = 0; }
    __syncthreads();   // (the flags are initialised before any wave raises s_poison below)
    const int ng = a.n_groups;
    const int npass = (ng + 63) / 64;
    const uint4* ub = reinterpret_cast<const uint4*>(a.colu + ((int64_t)(h * a.nkb + kb) * ng) * 64);   // per group: 8 uint4 (two halves of 32 keys)
    const float2* gb = a.gbound + (int64_t)h * ng;
    uint4 uu[BD2_MAXPASS];
    float2 bb[BD2_MAXPASS];
#pragma unroll
    for (int p = 0; p < BD2_MAXPASS; ++p) {
        const int g = min(p * 64 + grow, ng - 1);
        if (p < npass) {
            const uint4 u0 = ub[g * 8 + oct], u1 = ub[g * 8 + 4 + oct];
            auto pkmax = [](uint32_t x, uint32_t y) __attribute__((always_inline)) -> uint32_t {
                uint32_t r;
                if constexpr (std::is_same<T, _Float16>::value) {
                    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
                } else {   // bf16: no packed maximum - both halves through fp32 (the result is one of the inputs: the way back is exact)
                    const float lo = fmaxf(pair_lo<T>(x), pair_lo<T>(y)), hi = fmaxf(pair_hi<T>(x), pair_hi<T>(y));
                    r = (__builtin_bit_cast(uint32_t, lo) >> 16) | (__builtin_bit_cast(uint32_t, hi) & 0xFFFF0000u);
                }
                return r;
            };
            uu[p] = make_uint4(pkmax(u0.x, u1.x), pkmax(u0.y, u1.y), pkmax(u0.z, u1.z), pkmax(u0.w, u1.w));
            bb[p] = gb[g];
        }
    }
    constexpr int DT = std::is_same<T, __bf16>::value ? KVZ_BF16 : KVZ_F16;
    auto key = [&](const uint4& v, int i) __attribute__((always_inline)) -> float {
        const uint32_t w = (i < 2) ? v.x : (i < 4) ? v.y : (i < 6) ? v.z : v.w;
        return half_bits_to_float((i & 1) ? (w >> 16) : (w & 0xFFFFu), DT);
    };
    float lb[8];
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) lb[i] = -INFINITY;
#pragma unroll
    for (int p = 0; p < BD2_MAXPASS; ++p) {
        if (p < npass && p * 64 + grow < ng) {
            bad |= !(bb[p].x == bb[p].x);
#pragma unroll
            for (int i = 0; i < 8; ++i) lb[i] = fmaxf(lb[i], key(uu[p], i) + bb[p].y);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s_part[grow][oct * 8 + i] = lb[i];
    if (bad) s_poison = 1;
    __syncthreads();
    {
        const int j = tid & 31, part = tid >> 5;
        float v = -INFINITY;
#pragma unroll
        for (int q = 0; q < 8; ++q) v = fmaxf(v, s_part[part * 8 + q][j]);
        s_lb8[part][j] = v;
    }
    __syncthreads();
    if (tid < 32) {
        float v = -INFINITY;
#pragma unroll
        for (int q = 0; q < 8; ++q) v = fmaxf(v, s_lb8[q][tid]);
        s_lb[tid] = v;
    }
    __syncthreads();
    if (s_poison) {
        if (tid < 32 && kb * 32 + tid < a.m) atomicMin(a.log_out + (int64_t)h * a.log_head_stride + kb * 32 + tid, 0u);
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) lb[i] = s_lb[oct * 8 + i];
#pragma unroll
    for (int p = 0; p < BD2_MAXPASS; ++p) {
        if (p < npass) {   // (uniform: every lane of a quad of threads takes part in the shuffles)
            const int g = p * 64 + grow;
            bool c = false;
            if (g < ng) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float ub1 = key(uu[p], i) + bb[p].x;
                    c |= (kb * 32 + oct * 8 + i < a.m) && (!(ub1 < lb[i]) || a.all_pairs);
                }
            }
            int ci = c ? 1 : 0;
            ci |= __shfl_xor(ci, 1, 64);
            ci |= __shfl_xor(ci, 2, 64);
            if (ci && oct == 0 && g < ng) s_list[atomicAdd(&s_n, 1)] = (uint32_t)g | ((uint32_t)kb << 11) | ((uint32_t)h << 25);
        }
    }
    __syncthreads();
    if (tid == 0) s_base = (int)atomicAdd(a.counter + h, (uint32_t)s_n);
    __syncthreads();
    uint32_t* const eh = a.entries + (int64_t)h * a.nkb * a.n_groups;
    for (int i = tid; i < s_n; i += BD2_THREADS) eh[s_base + i] = s_list[i];
}

// ---- pass B over the candidate pairs only (round 5) ---------------------------------------------------------------------------------
// No stationary tile, no LDS, no barrier: the list of candidate (32-row group, 32-key block) pairs is cut into equal shares, one per
// wave; a wave keeps the 32 keys of the current key block in registers (B operand: one KEY per lane, as in the key-per-lane pass A),
// reads the 32 query rows of a pair straight from memory as the A operand (the rows of the next pair are in flight while this one is
// computed), and holds ONE running maximum per lane; when the key block changes (or the share ends) the two halves of the wave are
// merged and the 32 maxima go out with the same atomic minimum on the log-score patterns that merges the row slices of the dense pass.
constexpr int SB_WAVES = 4;
#ifndef KVZ_SB_NBUF
#define KVZ_SB_NBUF 2   // staging buffers per wave of the pair-level sparse pass (knob score_prune = 5)
#endif
#ifndef KVZ_SB_OCC1
#define KVZ_SB_OCC1 3   // blocks per CU of the one-buffer sparse pass (4: 128 registers per lane - two of them spill)
#endif
// log-score of four logits: t_k = float(x_k) + n_k (fp16: the mixed-precision fma reads the 16-bit half and adds in one instruction -
// x * 1.0 + n is the same single rounding as the conversion followed by the addition)
template <typename T>
__device__ static inline void quad_addn(uint32_t xa, uint32_t xb, float n0, float n1, float n2, float n3, float (&t)[4]) {
    if constexpr (std::is_same<T, _Float16>::value) {
        asm("v_fma_mix_f32 %[t0], %[xa], 1.0, %[n0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[t2], %[xb], 1.0, %[n2] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[t1], %[xa], 1.0, %[n1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[t3], %[xb], 1.0, %[n3] op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3])
            : [xa] "v"(xa), [xb] "v"(xb), [n0] "v"(n0), [n1] "v"(n1), [n2] "v"(n2), [n3] "v"(n3));
    } else {
        t[0] = pair_lo<T>(xa) + n0; t[1] = pair_hi<T>(xa) + n1; t[2] = pair_lo<T>(xb) + n2; t[3] = pair_hi<T>(xb) + n3;
    }
}
// NBUF = 2 (round 5): a wave stages the rows of the next pair into its second buffer while it computes the current one; 67 KiB of LDS per
// block: two blocks = eight waves per CU.  NBUF = 1 (round 6): ONE buffer per wave - the rows of the next pair are staged into it as soon
// as the fragments of the current pair sit in registers (before its MFMAs: the same distance ahead), half the LDS, SIXTEEN waves per CU.
template <typename T, int D, bool FAST, int NBUF>
__global__ __launch_bounds__(SB_WAVES * 64, NBUF == 1 ? KVZ_SB_OCC1 : 2) void score_colmax_sparse_kernel(ScoreArgs a) {
    typedef ScoreCfg<D> C;
    typedef typename Mfma32<T>::v8 v8;
    constexpr int QG_BYTES = 32 * C::ROW_BYTES;      // the 32 query rows of a pair, swizzled like a key tile
    constexpr int PB_BYTES = QG_BYTES + 256;         // + their 32 statistics n_r (written twice: one 64-lane dword DMA)
    __shared__ __attribute__((aligned(16))) char lds[SB_WAVES * NBUF * PB_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int R = a.G * a.q_len;
    // the heads' lists, one after the other, cut into equal shares; blocks are dispatched round-robin over the 8 XCDs: the blocks of one XCD
    // take one contiguous eighth of the pairs (a head's query rows and keys stay in that XCD's L2)
    const uint32_t cnt_l = (lane < a.n_kv_heads) ? a.counter[lane] : 0u;   // (Hkv < 128: lanes 0..63 hold heads 0..63, the second word below)
    const uint32_t cnt_h = (lane + 64 < a.n_kv_heads) ? a.counter[lane + 64] : 0u;
    auto head_count = [&](int hh) __attribute__((always_inline)) -> int {
        return hh < 64 ? __builtin_amdgcn_readlane((int)cnt_l, hh) : __builtin_amdgcn_readlane((int)cnt_h, hh - 64);
    };
    int total = 0;
    for (int hh = 0; hh < a.n_kv_heads; ++hh) total += head_count(hh);
    const int nb8 = (int)gridDim.x >> 3;
    const int lb = (nb8 > 0 && (gridDim.x & 7u) == 0) ? (int)(blockIdx.x & 7u) * nb8 + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int W = gridDim.x * SB_WAVES, w = lb * SB_WAVES + wave;
    const int per = (total + W - 1) / W;
    const int lo = w * per, hi = min(total, lo + per);
    if (lo >= hi) return;
    const int64_t head_cap = (int64_t)a.nkb * a.n_groups;
    const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    char* const wl = lds + wave * NBUF * PB_BYTES;
    const uint32_t wl0 = lds_addr(wl);
    FragAddr<D> fa0;
    fa0.init(wl, l31, half);
    const uint32_t nb0 = wl0 + (uint32_t)QG_BYTES + (uint32_t)(4 * half) * 4u;   // + 32 qd: the statistics of accumulators 4 qd .. 4 qd + 3

    // rows and statistics of pair e -> LDS buffer b of this wave (9 DMA instructions, nothing through registers)
    constexpr int ROWS_PER_INSTR = 1024 / C::ROW_BYTES;
    const int lrow = lane / C::CPR, pch = lane % C::CPR;
    const int64_t hs = a.q_head_stride * 2;
    auto stage = [&](int b, uint32_t e) __attribute__((always_inline)) {
        const int g = (int)(e & 2047u), h = (int)(e >> 25);
        const char* qh = reinterpret_cast<const char*>(a.q) + (int64_t)h * a.G * hs;
        const uint32_t buf = wl0 + (uint32_t)(b * PB_BYTES);
        const int rc0 = min(g * 32, R - 1);
        const int g0 = a.dq.div(rc0), qi0 = rc0 - g0 * a.q_len;   // (wave-uniform; q_len >= 32: at most two query heads per group)
        const uint32_t base = (uint32_t)(g0 * (int)hs + qi0 * C::ROW_BYTES);
        const uint32_t wrapd = (uint32_t)((int)hs - a.q_len * C::ROW_BYTES);
        const int nfirst = a.q_len - qi0, tail = R - 1 - rc0;
#pragma unroll
        for (int i = 0; i < 32 / ROWS_PER_INSTR; ++i) {
            const int row = i * ROWS_PER_INSTR + lrow;
            const int chunk = (D == 128) ? (pch ^ (row & 15)) : (pch ^ ((row >> 1) & 7));
            const int rowc = min(row, tail);
            const uint32_t voff = base + (uint32_t)(rowc * C::ROW_BYTES + chunk * 16) + (rowc >= nfirst ? wrapd : 0u);
            lds_dma16a(qh, voff, buf + (uint32_t)(i * 1024));
        }
        {
            const char* np = reinterpret_cast<const char*>(a.nrow + ((int64_t)h * a.n_groups + g) * 32);
            const uint32_t la = __builtin_amdgcn_readfirstlane(buf + (uint32_t)QG_BYTES);
            const uint64_t bs = (uint64_t)(uintptr_t)np;
            const uint64_t bss = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(bs >> 32)) << 32) |
                                 (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)bs);
            const uint32_t voff = (uint32_t)(l31 * 4);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(bss), "s"(la) : "memory");
        }
    };
    constexpr int STAGE_OPS = 32 / ROWS_PER_INSTR + 1;

    u32x4 bk[C::KK];
    uint32_t cur = 0xFFFFFFFFu;   // (h, kb) of the keys in bk
    float best = -INFINITY;
    auto flush = [&]() __attribute__((always_inline)) {
        const int kb = (int)(cur & 0x3FFFu), h = (int)(cur >> 14);
        const uint32_t bb = __builtin_bit_cast(uint32_t, best);
        const auto sw = __builtin_amdgcn_permlane32_swap(bb, bb, false, false);
        const float b = fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
        const int j = kb * 32 + l31;
        if (half == 0 && j < a.m) {
            const uint32_t enc = (b >= 0.f) ? 1u : __builtin_bit_cast(uint32_t, b);   // (the encoding of the dense pass; NaN cannot come here)
            atomicMin(a.log_out + (int64_t)h * a.log_head_stride + j, enc);
        }
    };
    auto keys_for = [&](uint32_t e) __attribute__((always_inline)) {
        const uint32_t hk = e >> 11;
        if (hk != cur) {
            if (cur != 0xFFFFFFFFu) flush();
            cur = hk;
            best = -INFINITY;
            const int kb = (int)(hk & 0x3FFFu), h = (int)(hk >> 14);
            const int j = min(kb * 32 + l31, a.m - 1);
            const char* kp = reinterpret_cast<const char*>(a.k) + ((int64_t)h * a.k_head_stride + (int64_t)(a.start + j) * D) * 2 + half * 16;
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) bk[kk] = *reinterpret_cast<const u32x4*>(kp + kk * 32);
        }
    };
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) f4v* lds_f4_t;
    auto compute = [&](auto b_tag, auto&& restage) __attribute__((always_inline)) {
        constexpr int B = decltype(b_tag)::value;
        u32x4 fq[C::KK];
        frag_load<D>(fq, fa0, B * PB_BYTES);
        f4v nn[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) nn[qd] = *(lds_f4_t)(uintptr_t)(nb0 + (uint32_t)(B * PB_BYTES + 32 * qd));
        if constexpr (NBUF == 1) {   // the buffer is free once the reads have returned: the rows of the next pair go into it now
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            restage();
        }
        f16v acc = zero16;
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) acc = Mfma32<T>::mfma(__builtin_bit_cast(v8, fq[kk]), __builtin_bit_cast(v8, bk[kk]), acc);
        // the rounding chain is an assembly block: the compiler does not count the wait states between the last MFMA of the chain and the
        // first read of its result (18 for a 16-pass MFMA) - the first quad read stale accumulators without them
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            uint32_t xa, xb;
            quad_round<T, FAST>(acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3], xa, xb, a.c, a.rcp);
            // accumulators 4 qd .. 4 qd + 3 are rows 8 qd + 4 half + (0..3)
            float t[4];
            quad_addn<T>(xa, xb, nn[qd].x, nn[qd].y, nn[qd].z, nn[qd].w, t);
            best = fmaxf(fmaxf(best, t[0]), t[1]);   // (two v_max3_f32)
            best = fmaxf(fmaxf(best, t[2]), t[3]);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    // the share in batches of 64 pairs: one entry per lane, read with v_readlane (a load the compiler sees inside the loop would make it
    // wait for everything in flight, the staged rows of the next pair included)
    // (a batch lies inside one head's list)
    int hcur = 0, hbase = 0;   // head of position b0, global position of its first pair
    for (int b0 = lo; b0 < hi;) {
        while (b0 >= hbase + head_count(hcur)) { hbase += head_count(hcur); ++hcur; }
        const int nb = min(min(64, hi - b0), hbase + head_count(hcur) - b0);
        const uint32_t ev = a.entries[hcur * head_cap + (b0 - hbase) + min(lane, nb - 1)];
        auto entry = [&](int i) __attribute__((always_inline)) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)ev, i); };
        stage(0, entry(0));
        if constexpr (NBUF == 1) {
            for (int i = 0; i < nb; ++i) {
                keys_for(entry(i));   // (a key block's loads are issued before the wait for the staged rows: one round trip, not two)
                stage_wait();
                compute(I0{}, [&]() __attribute__((always_inline)) { if (i + 1 < nb) stage(0, entry(i + 1)); });
            }
        } else {
            auto none = [&]() __attribute__((always_inline)) {};
            for (int i = 0; i < nb; i += 2) {
                const bool has1 = i + 1 < nb;
                if (has1) stage(1, entry(i + 1));
                keys_for(entry(i));
                if (has1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STAGE_OPS) : "memory");
                else stage_wait();
                compute(I0{}, none);
                if (!has1) break;
                const bool has2 = i + 2 < nb;
                if (has2) stage(0, entry(i + 2));
                keys_for(entry(i + 1));
                if (has2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STAGE_OPS) : "memory");
                else stage_wait();
                compute(I1{}, none);
            }
        }
        b0 += nb;
    }
    flush();
}

// ---- candidates at KEY granularity (round 6) ------------------------------------------------------------------------------------------------
// A (32-row group, 32-key block) pair is recomputed in full - 1024 logits - although all it is needed for is, typically, ONE key: every key
// has its argmax group (and the one or two groups whose bound cannot rule them out), so a key block of 32 keys names ~55 of the 448 groups
// (18 % of the pairs on the benchmark's inputs).  Turned around, a row group is named by ~31 KEYS of the whole chunk: score_bounds3_kernel
// applies the same exact test per (group, key) - u_gj + nmax_g >= LB_j - and appends key j to the list of group g; score_colmax_keys_kernel
// takes a group's 32 query rows as the A operand and GATHERS 32 of its candidate keys as the B operand (one key per lane), so one 32x32
// block of logits serves 32 candidates: ~1.6 blocks per group instead of ~11 pairs, a seventh of the matrix and rounding-chain work.  The
// logit of a (row, key) pair does not depend on where it sits in an MFMA tile, so the column maxima are the bits of the full pass B.
struct Bounds3Smem {
    float part[64][33];
    float lb8[8][32];
    float lb[32];
    uint32_t list[64 * BD2_MAXPASS * 2];   // work items opened by this block: at most two per group (32 keys)
    int n, base, poison;
};
template <typename T>
__device__ static inline void score_bounds3_body(const ScoreArgs& a, const int kb, const int h, Bounds3Smem& sm) {
    const int tid = threadIdx.x;
    const int oct = tid & 3, grow = tid >> 2;   // 8 keys [8 oct, 8 oct + 8) of group (pass * 64 + grow)
    auto& s_part = sm.part;
    auto& s_lb8 = sm.lb8;
    auto& s_lb = sm.lb;
    auto& s_list = sm.list;
    int& s_n = sm.n;
    int& s_base = sm.base;
    int& s_poison = sm.poison;
    if (tid == 0) { s_n = 0; s_poison = 0; }
    __syncthreads();   // (the flags are initialised before any wave touches them below)
    const int ng = a.n_groups;
    const int npass = (ng + 63) / 64;
    const uint4* ub = reinterpret_cast<const uint4*>(a.colu + ((int64_t)(h * a.nkb + kb) * ng) * 64);   // per group: 8 uint4 (two halves of 32 keys)
    const float2* gb = a.gbound + (int64_t)h * ng;
    float kf[BD2_MAXPASS][8];   // the thread's 8 bounds u of every pass as fp32: converted once, read by both phases below
    float2 bb[BD2_MAXPASS];
#pragma unroll
    for (int p = 0; p < BD2_MAXPASS; ++p) {
        const int g = min(p * 64 + grow, ng - 1);
        if (p < npass) {
            const uint4 u0 = ub[g * 8 + oct], u1 = ub[g * 8 + 4 + oct];
            auto pkmax = [](uint32_t x, uint32_t y) __attribute__((always_inline)) -> uint32_t {
                uint32_t r;
                if constexpr (std::is_same<T, _Float16>::value) {
                    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
                } else {   // bf16: no packed maximum - both halves through fp32 (the result is one of the inputs: the way back is exact)
                    const float lo = fmaxf(pair_lo<T>(x), pair_lo<T>(y)), hi = fmaxf(pair_hi<T>(x), pair_hi<T>(y));
                    r = (__builtin_bit_cast(uint32_t, lo) >> 16) | (__builtin_bit_cast(uint32_t, hi) & 0xFFFF0000u);
                }
                return r;
            };
            const uint32_t w0 = pkmax(u0.x, u1.x), w1 = pkmax(u0.y, u1.y), w2 = pkmax(u0.z, u1.z), w3 = pkmax(u0.w, u1.w);
            kf[p][0] = pair_lo<T>(w0); kf[p][1] = pair_hi<T>(w0); kf[p][2] = pair_lo<T>(w1); kf[p][3] = pair_hi<T>(w1);
            kf[p][4] = pair_lo<T>(w2); kf[p][5] = pair_hi<T>(w2); kf[p][6] = pair_lo<T>(w3); kf[p][7] = pair_hi<T>(w3);
            bb[p] = gb[g];
        }
    }
    float lb[8];
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) lb[i] = -INFINITY;
#pragma unroll
    for (int p = 0; p < BD2_MAXPASS; ++p) {
        if (p < npass && p * 64 + grow < ng) {
            bad |= !(bb[p].x == bb[p].x);
#pragma unroll
            for (int i = 0; i < 8; ++i) lb[i] = fmaxf(lb[i], kf[p][i] + bb[p].y);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s_part[grow][oct * 8 + i] = lb[i];
    if (bad) s_poison = 1;
    __syncthreads();
    {
        const int j = tid & 31, part = tid >> 5;
        float v = -INFINITY;
#pragma unroll
        for (int q = 0; q < 8; ++q) v = fmaxf(v, s_part[part * 8 + q][j]);
        s_lb8[part][j] = v;
    }
    __syncthreads();
    if (tid < 32) {
        float v = -INFINITY;
#pragma unroll
        for (int q = 0; q < 8; ++q) v = fmaxf(v, s_lb8[q][tid]);
        s_lb[tid] = v;
    }
    __syncthreads();
    if (s_poison) {   // a row with NaN statistics poisons its KV head as the reference's amax over rows does: the NaN code, by this block
        if (tid < 32 && kb * 32 + tid < a.m) atomicMin(a.log_out + (int64_t)h * a.log_head_stride + kb * 32 + tid, 0u);
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) lb[i] = s_lb[oct * 8 + i];
    // two phases, so that the atomics of a thread's passes are all in flight together (one after the other, each waiting for its position,
    // they were the longest thing in the kernel): first every pass's candidates and ONE atomic per quad of threads that has any ...
    uint32_t hits_[BD2_MAXPASS], base_[BD2_MAXPASS];
    auto qb = [](int v, auto sel) __attribute__((always_inline)) -> int { return __builtin_amdgcn_update_dpp(0, v, decltype(sel)::value, 0xf, 0xf, false); };
#pragma unroll
    for (int p = 0; p < BD2_MAXPASS; ++p) {
        hits_[p] = 0; base_[p] = 0;
        if (p < npass) {   // (uniform: the four threads of a group take part in the quad broadcasts)
            const int g = p * 64 + grow;
            uint32_t hits = 0;
            if (g < ng) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float ub1 = kf[p][i] + bb[p].x;
                    if ((kb * 32 + oct * 8 + i < a.m) && (!(ub1 < lb[i]) || a.all_pairs)) hits |= 1u << i;
                }
            }
            hits_[p] = hits;
            // (most quads have no candidate at all - a key names two or three of the 448 groups)
            if (__builtin_amdgcn_ballot_w64(hits != 0) == 0) continue;   // (wave-uniform)
            const int c = __builtin_popcount(hits);
            const int tot = qb(c, std::integral_constant<int, 0x00>{}) + qb(c, std::integral_constant<int, 0x55>{}) +
                            qb(c, std::integral_constant<int, 0xAA>{}) + qb(c, std::integral_constant<int, 0xFF>{});
            if (oct == 0 && tot) base_[p] = atomicAdd(a.gcount + (int64_t)h * ng + min(g, ng - 1), (uint32_t)tot);
        }
    }
    // ... then the group's (up to 32) candidate keys of this block go to its list behind the quad's position (prefix of the four counts)
#pragma unroll
    for (int p = 0; p < BD2_MAXPASS; ++p) {
        if (p < npass) {
            const uint32_t hits = hits_[p];
            if (__builtin_amdgcn_ballot_w64(hits != 0) == 0) continue;   // (wave-uniform)
            const int g = p * 64 + grow;
            const int c = __builtin_popcount(hits);
            const int c0 = qb(c, std::integral_constant<int, 0x00>{}), c1 = qb(c, std::integral_constant<int, 0x55>{});
            const int c2 = qb(c, std::integral_constant<int, 0xAA>{}), c3 = qb(c, std::integral_constant<int, 0xFF>{});
            const int tot = c0 + c1 + c2 + c3;
            const int before = (oct > 0 ? c0 : 0) + (oct > 1 ? c1 : 0) + (oct > 2 ? c2 : 0);
            const uint32_t base = (uint32_t)qb((int)base_[p], std::integral_constant<int, 0x00>{});
            if (tot) {
                uint32_t pos = base + (uint32_t)before;
                uint32_t* const dst = a.klist + ((int64_t)h * ng + min(g, ng - 1)) * a.kcap;
                for (uint32_t hb = hits; hb; hb &= hb - 1) dst[pos++] = (uint32_t)(kb * 32 + oct * 8 + __builtin_ctz(hb));
                // a list is consumed in chunks of 32 keys: whoever got the position that OPENS a chunk announces it as a work item
                if (oct == 0) {
                    for (uint32_t k = (base + 31u) >> 5; (k << 5) < base + (uint32_t)tot; ++k)
                        s_list[atomicAdd(&s_n, 1)] = (uint32_t)g | (k << 11) | ((uint32_t)h << 25);
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) s_base = (int)atomicAdd(a.counter + h, (uint32_t)s_n);
    __syncthreads();
    uint32_t* const eh = a.entries + (int64_t)h * a.nkb * a.n_groups;
    for (int i = tid; i < s_n; i += BD2_THREADS) eh[s_base + i] = s_list[i];
}
template <typename T>
__global__ __launch_bounds__(BD2_THREADS) void score_bounds3_kernel(ScoreArgs a) {
    __shared__ Bounds3Smem sm;
    score_bounds3_body<T>(a, (int)blockIdx.x, (int)blockIdx.y, sm);
}

// The work items - (KV head, row group, chunk of 32 candidate keys), announced by score_bounds3_kernel in per-head queues - are cut into equal
// shares, one per wave (the blocks of one XCD take one contiguous eighth, as in the pair-level kernel).  Per item: the group's 32 query rows
// are staged into the wave's LDS buffer (LDS-DMA, swizzled like a key tile) and read into registers as the A operand; the chunk's keys are
// gathered as the B operand - lane (j, half) loads the eight 16-byte chunks of ITS key row straight from the cache - and one chain of MFMAs
// gives every lane the logits of its key against 16 of the rows; rounding chain, x + n_r, maximum over the lane's rows, the two halves
// merged, one atomic minimum per key on the log-score patterns (the merge of the dense pass).  The next item's candidate indices are
// requested while the current one is computed.
constexpr int SK_WAVES = 4;
#ifndef KVZ_SK_BLOCKS_PER_CU
#define KVZ_SK_BLOCKS_PER_CU 4
#endif
template <int D> struct SkSmem {
    static constexpr int QG_BYTES = 32 * ScoreCfg<D>::ROW_BYTES;      // the 32 query rows of a group, swizzled like a key tile
    static constexpr int PB_BYTES = QG_BYTES + 256;                   // + their 32 statistics n_r (written twice: one 64-lane dword DMA)
    static constexpr int BYTES = SK_WAVES * PB_BYTES;
};
// (block bid of nblk: the stand-alone launch passes its own index and grid, the fused tail launch the index inside its candidate-key part)
template <typename T, int D, bool FAST>
__device__ static inline void score_colmax_keys_body(const ScoreArgs& a, const uint32_t bid, const uint32_t nblk, char* const lds) {
    typedef ScoreCfg<D> C;
    typedef typename Mfma32<T>::v8 v8;
    constexpr int QG_BYTES = SkSmem<D>::QG_BYTES;
    constexpr int PB_BYTES = SkSmem<D>::PB_BYTES;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int R = a.G * a.q_len;
    const int ng = a.n_groups;
    const uint32_t cnt_l = (lane < a.n_kv_heads) ? a.counter[lane] : 0u;   // (Hkv < 128: lanes 0..63 hold heads 0..63, the second word below)
    const uint32_t cnt_h = (lane + 64 < a.n_kv_heads) ? a.counter[lane + 64] : 0u;
    auto head_count = [&](int hh) __attribute__((always_inline)) -> int {
        return hh < 64 ? __builtin_amdgcn_readlane((int)cnt_l, hh) : __builtin_amdgcn_readlane((int)cnt_h, hh - 64);
    };
    int total = 0;
    for (int hh = 0; hh < a.n_kv_heads; ++hh) total += head_count(hh);
    const int nb8 = (int)nblk >> 3;
    const int lb = (nb8 > 0 && (nblk & 7u) == 0) ? (int)(bid & 7u) * nb8 + (int)(bid >> 3) : (int)bid;
    const int W = nblk * SK_WAVES, w = lb * SK_WAVES + wave;
    const int per = (total + W - 1) / W;
    const int lo = w * per, hi = min(total, lo + per);
    if (lo >= hi) return;
    const int64_t head_cap = (int64_t)a.nkb * ng;
    const f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    char* const wl = lds + wave * PB_BYTES;
    const uint32_t wl0 = lds_addr(wl);
    FragAddr<D> fa0;
    fa0.init(wl, l31, half);
    const uint32_t nb0 = wl0 + (uint32_t)QG_BYTES + (uint32_t)(4 * half) * 4u;   // + 32 qd: the statistics of accumulators 4 qd .. 4 qd + 3
    constexpr int ROWS_PER_INSTR = 1024 / C::ROW_BYTES;
    const int lrow = lane / C::CPR, pch = lane % C::CPR;
    const int64_t hs = a.q_head_stride * 2;
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) f4v* lds_f4_t;
    // candidate index of this lane in item e and the number of candidates the chunk holds: the list slot is read WITHOUT knowing the list's
    // length (the two loads travel together: one round trip, not two); the lanes behind the end of the list take the chunk's first
    // candidate (computing a pair twice changes nothing) and do not store
    auto cand = [&](uint32_t e, int& nvalid) __attribute__((always_inline)) -> uint32_t {
        const int g = (int)(e & 2047u), k = (int)((e >> 11) & 0x3FFFu), h = (int)(e >> 25);
        const int64_t gi = (int64_t)h * ng + g;
        const uint32_t jraw = a.klist[gi * a.kcap + k * 32 + l31];
        const int n = min((int)__builtin_amdgcn_readfirstlane((int)a.gcount[gi]), a.kcap);
        nvalid = min(32, n - k * 32);
        const uint32_t j0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)jraw);   // (slot k * 32: written by whoever opened the chunk)
        return min(l31 < nvalid ? jraw : j0, (uint32_t)(a.m - 1));
    };
    int hcur = 0, hbase = 0;   // head of position b0, global position of its first item
    for (int b0 = lo; b0 < hi;) {
        while (b0 >= hbase + head_count(hcur)) { hbase += head_count(hcur); ++hcur; }
        const int nb = min(min(64, hi - b0), hbase + head_count(hcur) - b0);   // (a batch lies inside one head's queue)
        const uint32_t ev = a.entries[hcur * head_cap + (b0 - hbase) + min(lane, nb - 1)];
        auto entry = [&](int i) __attribute__((always_inline)) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)ev, i); };
        int nv_next = 0;
        uint32_t j_next = cand(entry(0), nv_next);
        for (int i = 0; i < nb; ++i) {
            const uint32_t e = entry(i);
            const int g = (int)(e & 2047u), h = (int)(e >> 25);
            const uint32_t j = j_next;
            const int nv = nv_next;
            {   // rows and statistics of group g -> the wave's LDS buffer (nothing through registers)
                const char* qh = reinterpret_cast<const char*>(a.q) + (int64_t)h * a.G * hs;
                const int rc0 = min(g * 32, R - 1);
                const int g0 = a.dq.div(rc0), qi0 = rc0 - g0 * a.q_len;   // (wave-uniform; q_len >= 32: at most two query heads per group)
                const uint32_t base = (uint32_t)(g0 * (int)hs + qi0 * C::ROW_BYTES);
                const uint32_t wrapd = (uint32_t)((int)hs - a.q_len * C::ROW_BYTES);
                const int nfirst = a.q_len - qi0, tail = R - 1 - rc0;
#pragma unroll
                for (int r = 0; r < 32 / ROWS_PER_INSTR; ++r) {
                    const int row = r * ROWS_PER_INSTR + lrow;
                    const int chunk = (D == 128) ? (pch ^ (row & 15)) : (pch ^ ((row >> 1) & 7));
                    const int rowc = min(row, tail);
                    const uint32_t voff = base + (uint32_t)(rowc * C::ROW_BYTES + chunk * 16) + (rowc >= nfirst ? wrapd : 0u);
                    lds_dma16a(qh, voff, wl0 + (uint32_t)(r * 1024));
                }
                const char* np = reinterpret_cast<const char*>(a.nrow + ((int64_t)h * ng + g) * 32);
                const uint32_t la = __builtin_amdgcn_readfirstlane(wl0 + (uint32_t)QG_BYTES);
                const uint64_t bs = (uint64_t)(uintptr_t)np;
                const uint64_t bss = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(bs >> 32)) << 32) |
                                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)bs);
                const uint32_t voff = (uint32_t)(l31 * 4);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(bss), "s"(la) : "memory");
            }
            u32x4 bk[C::KK];
            {
                const char* kp = reinterpret_cast<const char*>(a.k) + ((int64_t)h * a.k_head_stride + (int64_t)(a.start + (int64_t)j) * D) * 2 + half * 16;
#pragma unroll
                for (int kk = 0; kk < C::KK; ++kk) bk[kk] = *reinterpret_cast<const u32x4*>(kp + kk * 32);
            }
            if (i + 1 < nb) j_next = cand(entry(i + 1), nv_next);   // (in flight behind this item's loads)
            stage_wait();
            u32x4 fq[C::KK];
            frag_load<D>(fq, fa0, 0);
            f4v nn[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) nn[qd] = *(lds_f4_t)(uintptr_t)(nb0 + (uint32_t)(32 * qd));
            f16v acc = zero16;
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) acc = Mfma32<T>::mfma(__builtin_bit_cast(v8, fq[kk]), __builtin_bit_cast(v8, bk[kk]), acc);
            // (wait states between the last MFMA of the chain and the assembly block that reads its result: the compiler does not count them)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            float best = -INFINITY;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                uint32_t xa, xb;
                quad_round<T, FAST>(acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3], xa, xb, a.c, a.rcp);
                float t[4];   // accumulators 4 qd .. 4 qd + 3 are rows 8 qd + 4 half + (0..3)
                quad_addn<T>(xa, xb, nn[qd].x, nn[qd].y, nn[qd].z, nn[qd].w, t);
                best = fmaxf(fmaxf(best, t[0]), t[1]);
                best = fmaxf(fmaxf(best, t[2]), t[3]);
            }
            const uint32_t bb = __builtin_bit_cast(uint32_t, best);
            const auto sw = __builtin_amdgcn_permlane32_swap(bb, bb, false, false);
            const float b = fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
            if (half == 0 && l31 < nv) {
                const uint32_t enc = (b >= 0.f) ? 1u : __builtin_bit_cast(uint32_t, b);   // (the encoding of the dense pass; NaN cannot come here)
                atomicMin(a.log_out + (int64_t)h * a.log_head_stride + j, enc);
            }
        }
        b0 += nb;
    }
}
template <typename T, int D, bool FAST>
__global__ __launch_bounds__(SK_WAVES * 64, 4) void score_colmax_keys_kernel(ScoreArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[SkSmem<D>::BYTES];
    score_colmax_keys_body<T, D, FAST>(a, blockIdx.x, gridDim.x, lds);
}

// ---- the tail of the pruned call as ONE launch, pipelined over the calls of a stream (round 6) ------------------------------------------------
// merge -> bounds -> candidate keys are three launches that each need the one before it; in the scoring loop a row-statistics block takes
// a whole CU, so every one of them waits for the moment some other stream's row-statistics kernel retires blocks, and the CUs they occupy
// meanwhile start their next row-statistics block late (8-10 us of an 84-us call).  The phases of DIFFERENT calls do not depend on each
// other: the launch behind the row-statistics kernel of call i of a stream runs the merge of call i, the bounds of call i-1 and the
// candidate-key pass of call i-2 of that stream side by side (each call on its own third of the workspace, order by the stream alone, no
// synchronisation inside the launch) - one burst of short blocks per call instead of three.  The host keeps the arguments of the last two
// calls per workspace (TailState) and flushes them when the scores are read (kvz_score_tail_flush).
template <typename T, int D, bool FAST>
__global__ __launch_bounds__(256, 4) void score_tail_kernel(ScoreArgs am, ScoreArgs ab, ScoreArgs ak, int n_keys, int n_bounds) {
    static_assert(SK_WAVES * 64 == 256 && BD2_THREADS == 256 && PA_ROWS == 256, "the three phases share one block size");
    constexpr int BYTES = SkSmem<D>::BYTES > (int)sizeof(Bounds3Smem) ? SkSmem<D>::BYTES : (int)sizeof(Bounds3Smem);
    __shared__ __attribute__((aligned(16))) char lds[BYTES];
    const int b = blockIdx.x;   // (wave-uniform branches: a block belongs to one phase; bounds first or merge first: no difference, profiles/r6_tail_blocks_ab.txt)
    if (b < n_keys) {
        score_colmax_keys_body<T, D, FAST>(ak, (uint32_t)b, (uint32_t)n_keys, lds);
    } else if (b < n_keys + n_bounds) {
        const int v = b - n_keys;
        const int h = v / ab.nkb;
        score_bounds3_body<T>(ab, v - h * ab.nkb, h, *reinterpret_cast<Bounds3Smem*>(lds));
    } else {
        score_merge_body(am, b - n_keys - n_bounds);
    }
}

template <typename T>
__global__ void score_finalize_kernel(const float* __restrict__ colpart, int splits, int Hkv, int m, T* __restrict__ out,
                                      int64_t out_head_stride) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (j >= m) return;
    float t = -INFINITY;
    bool nan = false;
    for (int s = 0; s < splits; ++s) {
        const float v = colpart[((int64_t)s * Hkv + h) * m + j];
        nan |= !(v == v);  // (fmaxf drops a NaN operand; the reference's amax propagates it)
        t = fmaxf(t, v);
    }
    out[(int64_t)h * out_head_stride + j] = (T)(nan ? __builtin_nanf("") : expf(t));
}

// log buffer -> scores: entries still holding the fill pattern (-inf: never scored) leave `out` untouched
constexpr uint32_t SC_LOG_EMPTY = 0xFF800000u;
template <typename T>
__global__ void score_finalize_log_kernel(const uint32_t* __restrict__ log, int64_t n, T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = log[i];
    if (b != SC_LOG_EMPTY) out[i] = (T)(b == 0u ? __builtin_nanf("") : expf(__builtin_bit_cast(float, b)));  // 0 = NaN (see pass B)
}

// the same launch, and the histogram of the top 11 bits of the order key of EVERY entry of `out` as it stands afterwards (entries
// never scored contribute the value `out` already holds): the first pass of the global-threshold selection (kvz_select.hip) rides
// on the launch that streams all scores anyway.  LDS-privatised histogram, one global atomic per non-empty bin and block.
template <typename T>
__global__ __launch_bounds__(1024) void score_finalize_log_hist_kernel(const uint32_t* __restrict__ log, int64_t n, T* __restrict__ out,
                                                                     uint32_t* __restrict__ hist_hi) {
    __shared__ uint32_t lh[SEL_HI_BINS];
    for (int i = threadIdx.x; i < SEL_HI_BINS; i += 1024) lh[i] = 0;
    __syncthreads();
    auto conv = [](uint32_t b) -> uint32_t { return bits16<T>((T)(b == 0u ? __builtin_nanf("") : expf(__builtin_bit_cast(float, b)))); };
    const int64_t nvec = n >> 3;
    const int64_t stride = (int64_t)gridDim.x * 1024;
    constexpr int U = 2;   // 8-element groups in flight per thread (4 x 16-byte loads: one dependent load per iteration runs at 1.4 TB/s)
    auto one = [&](int64_t i, const u32x4& l0, const u32x4& l1) {
        const uint32_t lw[8] = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        bool any_empty = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) any_empty |= lw[j] == SC_LOG_EMPTY;
        u32x4 o = {0, 0, 0, 0};
        if (any_empty) o = reinterpret_cast<const u32x4*>(out)[i];  // (rare: a buffer that was not scored to the end)
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t old = (o[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
            v[j] = lw[j] == SC_LOG_EMPTY ? old : conv(lw[j]);
            atomicAdd(&lh[order_key16(v[j]) >> 5], 1u);
        }
        const u32x4 w = {v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
        reinterpret_cast<u32x4*>(out)[i] = w;
    };
    const u32x4* lv = reinterpret_cast<const u32x4*>(log);
    int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    for (; i + (U - 1) * stride < nvec; i += U * stride) {
        u32x4 a0[U], a1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { a0[u] = lv[2 * (i + u * stride)]; a1[u] = lv[2 * (i + u * stride) + 1]; }
#pragma unroll
        for (int u = 0; u < U; ++u) one(i + u * stride, a0[u], a1[u]);
    }
    for (; i < nvec; i += stride) one(i, lv[2 * i], lv[2 * i + 1]);
    if (blockIdx.x == 0) {  // tail (< 8 elements)
        for (int64_t i = (nvec << 3) + threadIdx.x; i < n; i += 1024) {
            const uint32_t b = log[i];
            uint32_t v = bits16<T>(out[i]);
            if (b != SC_LOG_EMPTY) {
                v = conv(b);
                reinterpret_cast<uint16_t*>(out)[i] = (uint16_t)v;
            }
            atomicAdd(&lh[order_key16(v) >> 5], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL_HI_BINS; i += 1024) {
        const uint32_t c = lh[i];
        if (c) atomicAdd(&hist_hi[i], c);
    }
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// number of row slices of pass B: enough blocks to fill 256 CUs once, no empty slice, and a slice's merged statistics fit LDS
static inline int score_row_splits(int Hkv, int G, int q_len, int m) {
    const int ctiles = (m + PB_COLS - 1) / PB_COLS;
    const int rtiles = (G * q_len + SC_TILE - 1) / SC_TILE;
    int splits = 256 / (ctiles * Hkv);  // one round of resident blocks (256 CUs) when the shape allows
    const int need = (rtiles + PB_STAT_TILES - 1) / PB_STAT_TILES;
    if (splits < need) splits = need;
    if (splits > rtiles) splits = rtiles;
    if (splits < 1) splits = 1;
    const int per = (rtiles + splits - 1) / splits;
    return (rtiles + per - 1) / per;
}

// the partition of pass A is a pure function of the shape: memoised (a scoring pass repeats two or three shapes thousands of
// times, and the workspace check of every call needs max_seg as well)
struct PlanKey { int sink, m, q_len, G, Hkv; };
static bool get_plan(PaPlan& out, int sink, int m, int q_len, int G, int Hkv) {
    static std::mutex mu;
    static std::vector<std::pair<PlanKey, PaPlan>> cache;
    static std::vector<PlanKey> bad;
    std::lock_guard<std::mutex> lk(mu);
    for (auto& e : cache)
        if (e.first.sink == sink && e.first.m == m && e.first.q_len == q_len && e.first.G == G && e.first.Hkv == Hkv) {
            out = e.second;
            return true;
        }
    if (!make_plan(out, PA_ROWS, sink, m, q_len, G, Hkv)) return false;
    if (cache.size() >= 64) cache.erase(cache.begin());
    cache.push_back({PlanKey{sink, m, q_len, G, Hkv}, out});
    return true;
}

// ---- host: exhaustive search for an exact reciprocal constant ----------------------------------------------
static inline uint16_t f32_to_f16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t ex = (u >> 23) & 0xFFu;
    uint32_t man = u & 0x7FFFFFu;
    if (ex == 0xFF) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u : 0));
    const int e = (int)ex - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - e;  // 14..24
        uint32_t r = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (r & 1))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)e << 10) | (man >> 13);
    const uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;  // may carry into the exponent (correct)
    return (uint16_t)(sign | r);
}
static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)((u >> 16) | ((u & 0xFFFFu) ? 0x40u : 0));
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static inline float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, ex = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu, u;
    if (ex == 0) {
        if (!man) u = sign;
        else {
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            u = sign | ((uint32_t)(112 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (ex == 31) u = sign | 0x7F800000u | (man << 13);
    else u = sign | ((ex + 112u) << 23) | (man << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// returns r with  half(x * r) == half(x / c)  for every finite 16-bit x (zero sign included), or 0 if none of the
// neighbours of 1/c qualifies.
static float find_exact_reciprocal_search(float c, int dtype);
// memoised per (dtype, D): four fixed slots, each initialised exactly once (callers may come from several host threads:
// ctypes releases the GIL) - the 5 x 65536-case search runs at most four times per process, never on a later launch path
float score_exact_reciprocal(int D, int dtype);
static float find_exact_reciprocal(int D, int dtype) {
    static std::once_flag once[2][2];
    static float value[2][2];
    const int di = (D == 128) ? 1 : 0, ti = (dtype == KVZ_BF16) ? 1 : 0;
    std::call_once(once[ti][di], [&] { value[ti][di] = find_exact_reciprocal_search(sqrtf((float)D), dtype); });
    return value[ti][di];
}
float score_exact_reciprocal(int D, int dtype) { return (D == 64 || D == 128) ? find_exact_reciprocal(D, dtype) : 0.f; }
static float find_exact_reciprocal_search(float c, int dtype) {
    const float base = 1.0f / c;
    float cand[5] = {base, nextafterf(base, 1.f), nextafterf(base, 0.f), 0.f, 0.f};
    cand[3] = nextafterf(cand[1], 1.f);
    cand[4] = nextafterf(cand[2], 0.f);
    float found = 0.f;
    for (int ci = 0; ci < 5 && found == 0.f; ++ci) {
        bool ok = true;
        for (uint32_t b = 0; b < 65536 && ok; ++b) {
            float x;
            if (dtype == KVZ_BF16) {
                const uint32_t u = b << 16;
                memcpy(&x, &u, 4);
            } else x = f16_bits_to_f32((uint16_t)b);
            if (!(x - x == 0.f)) continue;  // inf / nan
            volatile float dq = x / c, mq = x * cand[ci];
            const uint16_t a16 = dtype == KVZ_BF16 ? f32_to_bf16_rne(dq) : f32_to_f16_rne(dq);
            const uint16_t b16 = dtype == KVZ_BF16 ? f32_to_bf16_rne(mq) : f32_to_f16_rne(mq);
            ok = (a16 == b16);
        }
        if (ok) found = cand[ci];
    }
    return found;
}

// ---- host state of the pipelined tail (score_tail_kernel; knob score_prune = 6) -------------------------------------------------------------
// Per workspace (= per side stream of a cache object): the arguments of the call whose bounds phase and of the call whose candidate-key
// phase are still to run, and which third of the workspace the next call takes.  Everything is ordered by the ONE stream the calls of a
// workspace are launched on.
struct TailState {
    ScoreArgs bounds_a, keys_a;
    bool need_bounds = false, need_keys = false;
    int code = -1;          // (dtype, head dim, chain variant) of the pending calls: the fused launch is one template instance
    int next_set = 0;
    hipStream_t stream = nullptr;
};
static std::mutex g_tail_mu;
static std::map<const void*, TailState> g_tail;
static inline int tail_code(int dtype, int D, bool fast) { return (dtype == KVZ_BF16 ? 4 : 0) + (D == 128 ? 2 : 0) + (fast ? 1 : 0); }

// one tail launch on st.stream: the merge of `am` (null: none), the bounds of st.bounds_a, the candidate keys of st.keys_a; the state moves on
// candidate-key blocks of the tail launch: one per CU.  In the burst of short blocks that a tail launch is, 1 024 blocks (what the stand-alone launch
// uses) cost 3 % of the scoring loop against 256; 128 / 64 / 32 gain another 0.5 % on Gaussian inputs and lose 5 % on repeat-prompt-like ones (more
// candidates per group): profiles/r6_tail_blocks_ab.txt
#ifndef KVZ_SK_TAIL_BLOCKS
#define KVZ_SK_TAIL_BLOCKS device_cus()
#endif
template <typename T, int D, bool FAST>
static int launch_tail(const ScoreArgs* am, TailState& st) {
    const ScoreArgs zero{};
    const int n_keys = st.need_keys ? KVZ_SK_TAIL_BLOCKS : 0;
    const int n_bounds = st.need_bounds ? st.bounds_a.nkb * st.bounds_a.n_kv_heads : 0;   // (two / four key blocks per block: -6 % / -14 % in the loop, profiles/r6_tail_blocks_ab.txt)
    const int n_merge = am ? (am->G * am->q_len + PA_ROWS - 1) / PA_ROWS * am->n_kv_heads : 0;   // (as its own launch - its blocks fit beside row-statistics blocks - +0.3 %, +4 us of host time: not worth the launch)
    if (n_keys + n_bounds + n_merge > 0) {
        ProfScope ps("score_tail", st.stream);
        hipLaunchKernelGGL((score_tail_kernel<T, D, FAST>), dim3(n_keys + n_bounds + n_merge), dim3(256), 0, st.stream, am ? *am : zero,
                           st.need_bounds ? st.bounds_a : zero, st.need_keys ? st.keys_a : zero, n_keys, n_bounds);
        KVZ_CHECK_LAUNCH("score_tail_kernel");
    }
    st.keys_a = st.bounds_a;
    st.need_keys = st.need_bounds;
    st.need_bounds = am != nullptr;
    if (am) st.bounds_a = *am;
    return KVZ_OK;
}
// the pending phases of a workspace, to the end (two launches at most)
static int tail_flush_locked(TailState& st) {
    while (st.need_bounds || st.need_keys) {
        int rc;
        switch (st.code) {
            case 0: rc = launch_tail<_Float16, 64, false>(nullptr, st); break;
            case 1: rc = launch_tail<_Float16, 64, true>(nullptr, st); break;
            case 2: rc = launch_tail<_Float16, 128, false>(nullptr, st); break;
            case 3: rc = launch_tail<_Float16, 128, true>(nullptr, st); break;
            case 4: rc = launch_tail<__bf16, 64, false>(nullptr, st); break;
            case 5: rc = launch_tail<__bf16, 64, true>(nullptr, st); break;
            case 6: rc = launch_tail<__bf16, 128, false>(nullptr, st); break;
            case 7: rc = launch_tail<__bf16, 128, true>(nullptr, st); break;
            default: set_error("kvz_score_tail_flush: corrupt state"); return KVZ_EINVAL;
        }
        if (rc != KVZ_OK) return rc;
    }
    return KVZ_OK;
}
// can this call shape take the pruned call at all?  (launch_score_impl applies the same test)
static inline bool prune_shape_ok(int n_groups, int nkb, int Hkv, int q_len) {
    return n_groups <= 64 * BD2_MAXPASS && nkb < 16384 && Hkv < 128 && q_len >= 32;
}

template <typename T, int D, bool FAST>
static int launch_score_impl(ScoreArgs a, int Hkv, hipStream_t stream, TailState* tail) {
    a.n_kv_heads = Hkv;
    PaPlan plan;
    if (!get_plan(plan, a.sink, a.m, a.q_len, a.G, Hkv)) {
        set_error("kvz_score_chunk: more than 65535 (head, row tile) units");
        return KVZ_EUNSUPPORTED;
    }
    a.unit_rows = PA_ROWS;
    a.max_seg = plan.max_seg;
    a.row_splits = score_row_splits(Hkv, a.G, a.q_len, a.m);
    const int ctiles = (a.m + PB_COLS - 1) / PB_COLS;
    // exact pruning of pass B (round 5; knob score_prune: 0 = two full passes; 1 = key-per-lane pass A, every block of pass B (checks);
    // 3 = candidate pairs only; 4 = every pair through the sparse kernel (checks)).  Both dtypes (round 6); deferred-log path.
    int prune = 0;
    if (a.unit_nseg && prune_shape_ok(a.n_groups, a.nkb, Hkv, a.q_len)) prune = tunable(TUNE_SCORE_PRUNE);
    if (prune >= 3 && !a.log_out) prune = 1;   // (the sparse pass merges through the log buffer)
    if (tail && prune != 6) {
        set_error("kvz_score_chunk: internal - a pipelined tail without the knob");
        return KVZ_EINVAL;
    }
    if (prune == 6 && !tail) prune = 3;        // (6 = 3 with the tail pipelined over the calls of a stream: asynchronous entry points only)
    if (prune) {
        {
            ProfScope ps("score_rowstat", stream);
            hipLaunchKernelGGL((score_rowstatT2_kernel<T, D, FAST>), dim3(plan.nb), dim3(PA_WAVES * 64), 0, stream, a, plan);
            KVZ_CHECK_LAUNCH("score_rowstatT2_kernel");
        }
        if (prune == 6) {
            a.all_pairs = 0;
            return launch_tail<T, D, FAST>(&a, *tail);
        }
        if (prune >= 3) {
            a.all_pairs = (prune == 4);
            if (prune != 5) {   // candidates at key granularity (round 6)
                // (measurement only, results unusable: score_prune = 16 + mask leaves launches out - 1: the merge, 2: the bounds, 4: the candidate-key pass)
                const int skip = prune >= 16 ? (prune & 7) : 0;
                {
                    ProfScope ps("score_bounds", stream);
                    if (!(skip & 1)) hipLaunchKernelGGL(score_merge_kernel, dim3((a.G * a.q_len + PA_ROWS - 1) / PA_ROWS * Hkv), dim3(PA_ROWS), 0, stream, a);
                    KVZ_CHECK_LAUNCH("score_merge_kernel");
                    if (!(skip & 2)) hipLaunchKernelGGL((score_bounds3_kernel<T>), dim3(a.nkb, Hkv), dim3(BD2_THREADS), 0, stream, a);
                    KVZ_CHECK_LAUNCH("score_bounds3_kernel");
                }
                if (!(skip & 4)) {
                    ProfScope ps("score_colmax", stream);
                    // (four blocks = sixteen waves per CU: an item is mostly latency - queue entry, candidate index, key row - and there are ~1.6 per (head, group))
                    hipLaunchKernelGGL((score_colmax_keys_kernel<T, D, FAST>), dim3(KVZ_SK_BLOCKS_PER_CU * device_cus()), dim3(SK_WAVES * 64), 0, stream, a);
                    KVZ_CHECK_LAUNCH("score_colmax_keys_kernel");
                }
                return KVZ_OK;
            }
            {
                ProfScope ps("score_bounds", stream);
                hipLaunchKernelGGL(score_merge_kernel, dim3((a.G * a.q_len + PA_ROWS - 1) / PA_ROWS * Hkv), dim3(PA_ROWS), 0, stream, a);
                KVZ_CHECK_LAUNCH("score_merge_kernel");
                hipLaunchKernelGGL((score_bounds2_kernel<T>), dim3(a.nkb, Hkv), dim3(BD2_THREADS), 0, stream, a);
                KVZ_CHECK_LAUNCH("score_bounds2_kernel");
            }
            {
                ProfScope ps("score_colmax", stream);
                hipLaunchKernelGGL((score_colmax_sparse_kernel<T, D, FAST, KVZ_SB_NBUF>), dim3((KVZ_SB_NBUF == 1 ? KVZ_SB_OCC1 : 2) * device_cus()), dim3(SB_WAVES * 64), 0, stream, a);
                KVZ_CHECK_LAUNCH("score_colmax_sparse_kernel");
            }
            return KVZ_OK;
        }
    } else if (a.unit_nseg) {  // (null: the row statistics came out of the scoring forward's attention kernel - kvz_flash_fwd_window)
        ProfScope ps("score_rowstat", stream);
        hipLaunchKernelGGL((score_rowstat2_kernel<T, D, FAST>), dim3(plan.nb), dim3(PA_WAVES * 64), 0, stream, a, plan);
        KVZ_CHECK_LAUNCH("score_rowstat2_kernel");
    } else {
        a.max_seg = 1;
    }
    {
        ProfScope ps("score_colmax", stream);
        hipLaunchKernelGGL((score_colmax3_kernel<T, D, FAST>), dim3(ctiles * a.row_splits * Hkv), dim3(PB_WAVES * 64), 0, stream, a);
        KVZ_CHECK_LAUNCH("score_colmax3_kernel");
    }
    if (a.log_out) return KVZ_OK;  // (the row slices were merged by the atomics; kvz_score_finalize_log turns the buffer into scores)
    hipLaunchKernelGGL((score_finalize_kernel<T>), dim3((a.m + 255) / 256, Hkv), dim3(256), 0, stream, a.colpart,
                       a.row_splits, Hkv, a.m, reinterpret_cast<T*>(a.out), a.out_head_stride);
    KVZ_CHECK_LAUNCH("score_finalize_kernel");
    return KVZ_OK;
}
template <typename T, int D>
static int launch_score(ScoreArgs a, int Hkv, hipStream_t stream, TailState* tail) {
    return a.rcp != 0.f ? launch_score_impl<T, D, true>(a, Hkv, stream, tail) : launch_score_impl<T, D, false>(a, Hkv, stream, tail);
}

}  // namespace kvz

using namespace kvz;


static inline int score_stats_stride(int G, int q_len) { return (G * q_len + SC_TILE - 1) / SC_TILE * SC_TILE; }
static inline size_t score_stats_bytes(int Hkv, int G, int q_len, int m, int sink) {
    int slices = 1;
    PaPlan plan;  // one partial per block that touches a row tile
    if (get_plan(plan, sink, m, q_len, G, Hkv)) slices = plan.max_seg;
    return align256((size_t)slices * Hkv * score_stats_stride(G, q_len) * sizeof(float2));
}

static inline size_t score_colpart_bytes(int Hkv, int G, int q_len, int m) {
    return align256((size_t)score_row_splits(Hkv, G, q_len, m) * Hkv * m * sizeof(float));
}
static inline size_t score_nseg_bytes(int Hkv, int G, int q_len) {
    return align256((size_t)((G * q_len + PA_ROWS - 1) / PA_ROWS) * Hkv * sizeof(int));
}

// pruning of pass B: 32-row groups (whole row tiles of the key-per-lane pass) and the buffers of that path
static inline int score_n_groups(int G, int q_len) { return (G * q_len + PA_ROWS - 1) / PA_ROWS * PA_WAVES; }
static inline size_t score_colu_bytes(int Hkv, int G, int q_len, int m) { return align256((size_t)Hkv * score_n_groups(G, q_len) * ((m + 31) / 32) * 64 * sizeof(uint16_t) + 128); }
static inline size_t score_gbound_bytes(int Hkv, int G, int q_len) { return align256((size_t)Hkv * score_n_groups(G, q_len) * sizeof(float2)); }
static inline size_t score_nrow_bytes(int Hkv, int G, int q_len) { return align256((size_t)Hkv * score_n_groups(G, q_len) * 32 * sizeof(float)); }
static inline size_t score_entries_bytes(int Hkv, int G, int q_len, int m) {
    return align256(((size_t)Hkv * ((m + 31) / 32) * score_n_groups(G, q_len) + 128 + PLAN_MAX_BLOCKS) * sizeof(uint32_t));   // (per-head counters, redo words, entries)
}
// candidates at key granularity: one counter and one list of up to nkb * 32 ctx indices per (KV head, row group)
static inline size_t score_klist_bytes(int Hkv, int G, int q_len, int m) {
    return align256((size_t)Hkv * score_n_groups(G, q_len) * (1 + (size_t)((m + 31) / 32) * 32) * sizeof(uint32_t));
}

extern "C" size_t kvz_score_workspace_bytes(int Hkv, int G, int q_len, int m, int sink) {
    if (Hkv <= 0 || G <= 0 || q_len <= 0 || m <= 0 || sink < 0) return 0;
    return score_stats_bytes(Hkv, G, q_len, m, sink) + score_colpart_bytes(Hkv, G, q_len, m) + score_nseg_bytes(Hkv, G, q_len) +
           score_colu_bytes(Hkv, G, q_len, m) + score_gbound_bytes(Hkv, G, q_len) + score_nrow_bytes(Hkv, G, q_len) +
           score_entries_bytes(Hkv, G, q_len, m) + score_klist_bytes(Hkv, G, q_len, m);
}

static int score_chunk_impl(const void* q, int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen,
                            int sink, int start, int end, int q_len, int Hkv, int G, int D, int dtype, void* out,
                            int64_t out_head_stride, uint32_t* log_out, int64_t log_head_stride, void* ws, size_t ws_bytes,
                            kvz_stream_t stream_, const float* merged_stats = nullptr, int64_t merged_stride = 0, bool defer = false);

extern "C" int kvz_score_chunk(const void* q, int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen,
                               int sink, int start, int end, int q_len, int Hkv, int G, int D, int dtype, void* out,
                               int64_t out_head_stride, void* ws, size_t ws_bytes, kvz_stream_t stream_) {
    KVZ_REQUIRE(out, KVZ_EINVAL, "kvz_score_chunk: null pointer");
    return score_chunk_impl(q, q_head_stride, k, k_head_stride, klen, sink, start, end, q_len, Hkv, G, D, dtype, out, out_head_stride,
                            nullptr, 0, ws, ws_bytes, stream_);
}

extern "C" int kvz_score_chunk_log(const void* q, int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen,
                                   int sink, int start, int end, int q_len, int Hkv, int G, int D, int dtype, uint32_t* log_out,
                                   int64_t log_head_stride, void* ws, size_t ws_bytes, kvz_stream_t stream_) {
    KVZ_REQUIRE(log_out && (reinterpret_cast<uintptr_t>(log_out) & 3u) == 0, KVZ_EINVAL, "kvz_score_chunk_log: bad log buffer");
    return score_chunk_impl(q, q_head_stride, k, k_head_stride, klen, sink, start, end, q_len, Hkv, G, D, dtype, nullptr, 0, log_out,
                            log_head_stride, ws, ws_bytes, stream_);
}

extern "C" int kvz_score_from_stats_log(const void* q, int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen,
                                       int sink, int start, int end, int q_len, int Hkv, int G, int D, int dtype, const float* stats,
                                       int64_t stats_head_stride, uint32_t* log_out, int64_t log_head_stride, kvz_stream_t stream_) {
    KVZ_REQUIRE(stats && log_out && (reinterpret_cast<uintptr_t>(log_out) & 3u) == 0, KVZ_EINVAL, "kvz_score_from_stats_log: bad buffers");
    KVZ_REQUIRE(stats_head_stride < (1ll << 31), KVZ_EUNSUPPORTED, "kvz_score_from_stats_log: statistics stride too large");
    return score_chunk_impl(q, q_head_stride, k, k_head_stride, klen, sink, start, end, q_len, Hkv, G, D, dtype, nullptr, 0, log_out,
                            log_head_stride, nullptr, 0, stream_, stats, stats_head_stride);
}

extern "C" int kvz_score_log_fill(uint32_t* log, int64_t n, kvz_stream_t stream_) {
    KVZ_REQUIRE(log && n >= 0, KVZ_EINVAL, "kvz_score_log_fill: bad arguments");
    if (n == 0) return KVZ_OK;
    KVZ_REQUIRE(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(log), (int)SC_LOG_EMPTY, (size_t)n, (hipStream_t)stream_) == hipSuccess,
                KVZ_ELAUNCH, "kvz_score_log_fill: hipMemsetD32Async failed");
    return KVZ_OK;
}

extern "C" int kvz_score_finalize_log(const uint32_t* log, int64_t n, void* out, int dtype, kvz_stream_t stream_) {
    KVZ_REQUIRE(log && out && n >= 0, KVZ_EINVAL, "kvz_score_finalize_log: bad arguments");
    KVZ_REQUIRE(dtype == KVZ_F16 || dtype == KVZ_BF16, KVZ_EINVAL, "kvz_score_finalize_log: bad dtype %d", dtype);
    if (n == 0) return KVZ_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    ProfScope ps("score_finalize_log", stream);
    if (dtype == KVZ_F16) hipLaunchKernelGGL((score_finalize_log_kernel<_Float16>), grid, block, 0, stream, log, n, reinterpret_cast<_Float16*>(out));
    else hipLaunchKernelGGL((score_finalize_log_kernel<__bf16>), grid, block, 0, stream, log, n, reinterpret_cast<__bf16*>(out));
    KVZ_CHECK_LAUNCH("score_finalize_log_kernel");
    return KVZ_OK;
}

extern "C" int kvz_score_finalize_log_hist(const uint32_t* log, int64_t n, void* out, int dtype, void* select_ws, size_t select_ws_bytes,
                                           kvz_stream_t stream_) {
    KVZ_REQUIRE(log && out && select_ws && n > 0, KVZ_EINVAL, "kvz_score_finalize_log_hist: bad arguments");
    KVZ_REQUIRE(dtype == KVZ_F16 || dtype == KVZ_BF16, KVZ_EINVAL, "kvz_score_finalize_log_hist: bad dtype %d", dtype);
    KVZ_REQUIRE(select_ws_bytes >= SEL_WS_WORDS * sizeof(uint32_t) && (reinterpret_cast<uintptr_t>(select_ws) & 3u) == 0, KVZ_EWORKSPACE,
                "kvz_score_finalize_log_hist: selection workspace too small (kvz_select_workspace_bytes)");
    KVZ_REQUIRE(aligned16(log) && aligned16(out), KVZ_EINVAL, "kvz_score_finalize_log_hist: log / out must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    KVZ_REQUIRE(hipMemsetAsync(select_ws, 0, SEL_WS_WORDS * sizeof(uint32_t), stream) == hipSuccess, KVZ_ELAUNCH,
                "kvz_score_finalize_log_hist: hipMemsetAsync failed");
    int blocks = (int)(((n + 7) / 8 + 1023) / 1024);
    if (blocks > device_cus()) blocks = device_cus();  // one 1024-thread block per CU: every block flushes its non-empty bins with global atomics (kvz_select.hip)
    if (blocks < 1) blocks = 1;
    uint32_t* hist = reinterpret_cast<uint32_t*>(select_ws);
    ProfScope ps("score_finalize_log", stream);
    if (dtype == KVZ_F16) hipLaunchKernelGGL((score_finalize_log_hist_kernel<_Float16>), dim3(blocks), dim3(1024), 0, stream, log, n, reinterpret_cast<_Float16*>(out), hist);
    else hipLaunchKernelGGL((score_finalize_log_hist_kernel<__bf16>), dim3(blocks), dim3(1024), 0, stream, log, n, reinterpret_cast<__bf16*>(out), hist);
    KVZ_CHECK_LAUNCH("score_finalize_log_hist_kernel");
    return KVZ_OK;
}

static int score_chunk_impl(const void* q, int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen,
                            int sink, int start, int end, int q_len, int Hkv, int G, int D, int dtype, void* out,
                            int64_t out_head_stride, uint32_t* log_out, int64_t log_head_stride, void* ws, size_t ws_bytes,
                            kvz_stream_t stream_, const float* merged_stats, int64_t merged_stride, bool defer) {
    hipStream_t stream = (hipStream_t)stream_;
    KVZ_REQUIRE(q && k && (out || log_out) && (ws || merged_stats), KVZ_EINVAL, "kvz_score_chunk: null pointer");
    KVZ_REQUIRE(Hkv > 0 && Hkv <= 65535 && G > 0 && q_len > 0, KVZ_EINVAL, "kvz_score_chunk: bad shape");
    KVZ_REQUIRE(D == 64 || D == 128, KVZ_EUNSUPPORTED, "kvz_score_chunk: head_dim %d unsupported (64 or 128)", D);
    KVZ_REQUIRE(dtype == KVZ_F16 || dtype == KVZ_BF16, KVZ_EINVAL, "kvz_score_chunk: bad dtype %d", dtype);
    const int m = end - start;
    KVZ_REQUIRE(sink >= 0 && start >= sink && m > 0 && end <= klen - q_len, KVZ_EINVAL,
                "kvz_score_chunk: bad window sink=%d start=%d end=%d klen=%d q_len=%d", sink, start, end, klen, q_len);
    KVZ_REQUIRE(aligned16(q) && aligned16(k), KVZ_EINVAL, "kvz_score_chunk: q/k must be 16-byte aligned");
    KVZ_REQUIRE((q_head_stride * 2) % 16 == 0 && (k_head_stride * 2) % 16 == 0, KVZ_EINVAL,
                "kvz_score_chunk: head strides must be multiples of 8 elements");
    KVZ_REQUIRE((int64_t)G * q_head_stride * 2 < (1ll << 31) && (int64_t)klen * D * 2 < (1ll << 31), KVZ_EUNSUPPORTED,
                "kvz_score_chunk: a KV head (and the query heads of its group) must span less than 2 GiB");
    KVZ_REQUIRE(merged_stats || ws_bytes >= kvz_score_workspace_bytes(Hkv, G, q_len, m, sink), KVZ_EWORKSPACE,
                "kvz_score_chunk: workspace too small");
    // pipelined tail (score_prune = 6, asynchronous log entry points, a workspace of three sets): this call takes the next third of the
    // workspace; phases another call left pending on this workspace run first when this call cannot continue their pipeline
    TailState* tail = nullptr;
    std::unique_lock<std::mutex> tail_lock(g_tail_mu, std::defer_lock);
    if (ws && !merged_stats) {
        tail_lock.lock();
        const size_t third = (ws_bytes / 3) & ~(size_t)255;
        const int code = tail_code(dtype, D, find_exact_reciprocal(D, dtype) != 0.f);
        const bool want = defer && log_out && tunable(TUNE_SCORE_PRUNE) == 6 && third >= kvz_score_workspace_bytes(Hkv, G, q_len, m, sink) &&
                          prune_shape_ok(score_n_groups(G, q_len), (m + 31) / 32, Hkv, q_len);
        auto it = g_tail.find(ws);
        if (it != g_tail.end() && (it->second.need_bounds || it->second.need_keys) &&
            (!want || it->second.code != code || it->second.stream != stream)) {
            TailState& st = it->second;
            const int rc = tail_flush_locked(st);
            if (rc != KVZ_OK) return rc;
            // (another stream takes the workspace over: rare - a rebuilt stream pool - and not worth an event)
            if (st.stream != stream) KVZ_REQUIRE(hipStreamSynchronize(st.stream) == hipSuccess, KVZ_ELAUNCH, "kvz_score_chunk: hipStreamSynchronize failed");
        }
        if (want) {
            TailState& st = (it != g_tail.end()) ? it->second : g_tail[ws];
            st.code = code;
            st.stream = stream;
            ws = reinterpret_cast<char*>(ws) + (size_t)st.next_set * third;
            st.next_set = (st.next_set + 1) % 3;
            tail = &st;
        } else {
            tail_lock.unlock();
        }
    }
    ScoreArgs a{};
    a.q = q; a.k = k; a.q_head_stride = q_head_stride; a.k_head_stride = k_head_stride;
    a.klen = klen; a.sink = sink; a.start = start; a.m = m; a.q_len = q_len; a.G = G;
    a.stats = reinterpret_cast<float2*>(ws);
    a.stats_stride = score_stats_stride(G, q_len);
    a.colpart = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + score_stats_bytes(Hkv, G, q_len, m, sink));
    a.unit_nseg = reinterpret_cast<int*>(reinterpret_cast<char*>(a.colpart) + score_colpart_bytes(Hkv, G, q_len, m));
    a.colu = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(a.unit_nseg) + score_nseg_bytes(Hkv, G, q_len));
    a.gbound = reinterpret_cast<float2*>(reinterpret_cast<char*>(a.colu) + score_colu_bytes(Hkv, G, q_len, m));
    a.nrow = reinterpret_cast<float*>(reinterpret_cast<char*>(a.gbound) + score_gbound_bytes(Hkv, G, q_len));
    a.counter = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(a.nrow) + score_nrow_bytes(Hkv, G, q_len));
    a.redo = a.counter + 128;   // (one counter per KV head: Hkv < 128)
    a.entries = a.redo + PLAN_MAX_BLOCKS;
    a.gcount = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(a.counter) + score_entries_bytes(Hkv, G, q_len, m));
    a.klist = a.gcount + (size_t)Hkv * score_n_groups(G, q_len);
    a.kcap = ((m + 31) / 32) * 32;
    a.n_groups = score_n_groups(G, q_len);
    a.all_pairs = 0;
    a.nkb = (m + 31) / 32;
    if (merged_stats) {  // pass B only: merged (m_r, l'_r) per row [Hkv, merged_stride], no partials, no workspace
        KVZ_REQUIRE(log_out, KVZ_EINVAL, "kvz_score_from_stats: the log buffer is the only output of this path");
        KVZ_REQUIRE(merged_stride >= (int64_t)G * q_len && (reinterpret_cast<uintptr_t>(merged_stats) & 7u) == 0, KVZ_EINVAL,
                    "kvz_score_from_stats: bad statistics buffer");
        a.stats = reinterpret_cast<float2*>(const_cast<float*>(merged_stats));
        a.stats_stride = (int)merged_stride;
        a.colpart = nullptr;
        a.unit_nseg = nullptr;
    }
    a.out = out; a.out_head_stride = out_head_stride;
    a.log_out = log_out; a.log_head_stride = log_head_stride;
    a.dq = make_fastdiv(q_len);
    a.dh = make_fastdiv(Hkv);
    a.c = sqrtf((float)D);  // == float32(math.sqrt(D)) for D in {64, 128}
    a.rcp = find_exact_reciprocal(D, dtype);
    if (dtype == KVZ_F16) {
        if (D == 128) return launch_score<_Float16, 128>(a, Hkv, stream, tail);
        return launch_score<_Float16, 64>(a, Hkv, stream, tail);
    }
    if (D == 128) return launch_score<__bf16, 128>(a, Hkv, stream, tail);
    return launch_score<__bf16, 64>(a, Hkv, stream, tail);
}

// the log entry point of the asynchronous calls (kvz_api.hip): the same call, allowed to leave phases of its tail to the next calls on
// this workspace and stream (score_prune = 6)
namespace kvz {
int score_chunk_log_deferred(const void* q, int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen, int sink, int start,
                             int end, int q_len, int Hkv, int G, int D, int dtype, uint32_t* log_out, int64_t log_head_stride, void* ws,
                             size_t ws_bytes, kvz_stream_t stream_) {
    KVZ_REQUIRE(log_out && (reinterpret_cast<uintptr_t>(log_out) & 3u) == 0, KVZ_EINVAL, "kvz_score_chunk_log: bad log buffer");
    return score_chunk_impl(q, q_head_stride, k, k_head_stride, klen, sink, start, end, q_len, Hkv, G, D, dtype, nullptr, 0, log_out,
                            log_head_stride, ws, ws_bytes, stream_, nullptr, 0, true);
}
}  // namespace kvz

// Runs what earlier calls on this workspace left pending (on the stream they were launched on).  Returns 1 when something was launched,
// 0 when nothing was pending.  The log scores of a workspace's calls are complete behind this on its stream.
extern "C" int kvz_score_tail_flush(const void* ws) {
    std::lock_guard<std::mutex> lk(g_tail_mu);
    auto it = g_tail.find(ws);
    if (it == g_tail.end()) return 0;
    const bool pending = it->second.need_bounds || it->second.need_keys;
    const int rc = pending ? tail_flush_locked(it->second) : KVZ_OK;
    // a drained workspace is forgotten: nothing of it is referenced any more, and an address that comes back (a freed workspace, another
    // cache object) starts from a clean state instead of inheriting this one's stream and set counter
    if (rc == KVZ_OK) g_tail.erase(it);
    return rc != KVZ_OK ? rc : (pending ? 1 : 0);
}

// test hook: the rounding chain on raw 16-bit patterns (exhaustive-check of the exact-reciprocal path on the device)
namespace kvz {
template <typename T, bool FAST>
__global__ void chain_probe_kernel(const uint16_t* in, int n, float c, float rcp, uint16_t* out, int path) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T x;
    uint16_t b = in[i];
    __builtin_memcpy(&x, &b, 2);
    // acc = float(x) is exactly representable, so half(acc) == x: the probe isolates the division step
    if (path == 1) {   // the kernels' own chain of four logits (fp16: the assembly block of quad_args); lane 0 of the quad is the probe
        uint32_t xa, xb;
        float arg[4];
        quad_args<T, FAST>((float)x, 1.f, 2.f, 3.f, xa, xb, arg, c, rcp, 1.44269504088896340736f, 0.f);
        out[i] = (uint16_t)(xa & 0xffffu);
        return;
    }
    if (path == 2 || path == 3) {   // the pruned call's chain (quad_round), low / high half of a pair
        uint32_t xa, xb;
        if (path == 2) quad_round<T, FAST>((float)x, 1.f, 2.f, 3.f, xa, xb, c, rcp);
        else quad_round<T, FAST>(1.f, (float)x, 2.f, 3.f, xa, xb, c, rcp);
        out[i] = (uint16_t)(path == 2 ? (xa & 0xffffu) : (xa >> 16));
        return;
    }
    const float r = round_chain<T, FAST>((float)x, c, rcp);
    const T h = (T)r;
    __builtin_memcpy(&b, &h, 2);
    out[i] = b;
}
}  // namespace kvz
extern "C" int kvz_debug_round_chain(const void* in_bits, int n, int D, int dtype, int force_division, void* out_bits,
                                     float* rcp_used, kvz_stream_t stream_) {
    // force_division: 0 = exact-reciprocal multiply (round_chain), 1 = IEEE division, 2 = the reciprocal through the kernels' own
    // four-logit chain (quad_args: for fp16 the assembly block)
    // 3 / 4: the same through the pruned call's chain (quad_round), low / high half of a pair
    const int path = force_division >= 2 ? force_division - 1 : 0;
    if (force_division >= 2) force_division = 0;
    hipStream_t stream = (hipStream_t)stream_;
    KVZ_REQUIRE(in_bits && out_bits && n > 0, KVZ_EINVAL, "kvz_debug_round_chain: bad arguments");
    const float c = sqrtf((float)D);
    KVZ_REQUIRE(D == 64 || D == 128, KVZ_EUNSUPPORTED, "kvz_debug_round_chain: head_dim %d unsupported", D);
    const float rcp = force_division ? 0.f : find_exact_reciprocal(D, dtype);
    if (rcp_used) *rcp_used = rcp;
    dim3 grid((n + 255) / 256), block(256);
    const uint16_t* in = reinterpret_cast<const uint16_t*>(in_bits);
    uint16_t* out = reinterpret_cast<uint16_t*>(out_bits);
    if (dtype == KVZ_F16) {
        if (rcp != 0.f) hipLaunchKernelGGL((chain_probe_kernel<_Float16, true>), grid, block, 0, stream, in, n, c, rcp, out, path);
        else hipLaunchKernelGGL((chain_probe_kernel<_Float16, false>), grid, block, 0, stream, in, n, c, rcp, out, path);
    } else {
        if (rcp != 0.f) hipLaunchKernelGGL((chain_probe_kernel<__bf16, true>), grid, block, 0, stream, in, n, c, rcp, out, path);
        else hipLaunchKernelGGL((chain_probe_kernel<__bf16, false>), grid, block, 0, stream, in, n, c, rcp, out, path);
    }
    KVZ_CHECK_LAUNCH("chain_probe_kernel");
    return KVZ_OK;
}


// test hook (host only, no GPU needed): the static partition of pass A for a geometry.  unit / tile: 257 entries each.
extern "C" int kvz_debug_score_plan(int sink, int m, int q_len, int G, int Hkv, uint16_t* unit, uint16_t* tile, int* n_blocks,
                                    int* max_seg, int* rows_per_unit) {
    KVZ_REQUIRE(unit && tile && n_blocks && max_seg && rows_per_unit, KVZ_EINVAL, "kvz_debug_score_plan: null pointer");
    KVZ_REQUIRE(sink >= 0 && m > 0 && q_len > 0 && G > 0 && Hkv > 0, KVZ_EINVAL, "kvz_debug_score_plan: bad shape");
    PaPlan p;
    const int rows = PA_ROWS;
    KVZ_REQUIRE(make_plan(p, rows, sink, m, q_len, G, Hkv), KVZ_EUNSUPPORTED, "kvz_debug_score_plan: too many units");
    for (int b = 0; b <= PLAN_MAX_BLOCKS; ++b) { unit[b] = p.unit[b]; tile[b] = p.tile[b]; }
    *n_blocks = p.nb;
    *max_seg = p.max_seg;
    *rows_per_unit = rows;
    return KVZ_OK;
}

// test hook (host only): quotient of the invariant-divisor arithmetic used inside the kernels (FastDiv), n in [0, 2^31)
extern "C" int kvz_debug_fastdiv(int d, int n, int* quotient, int* remainder) {
    KVZ_REQUIRE(d > 0 && n >= 0 && quotient && remainder, KVZ_EINVAL, "kvz_debug_fastdiv: bad arguments");
    const FastDiv f = make_fastdiv(d);
    *quotient = f.div(n);
    *remainder = f.mod(n);
    return KVZ_OK;
}

