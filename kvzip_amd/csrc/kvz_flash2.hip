// kvz_flash2.hip — the dense causal GQA forward of prefill and of the scoring pass on the 32x32x16 matrix cores (round 3).
//
// Same contract as kvz_flash.hip (reference attention/attn.py:75-89, flash-attn's dense kernel; the first generation step of
// attention/attn.py:56-73 when it has enough rows): for every KV head h, R = q_len*G query rows against the head's key segment,
// causal mask aligned bottom-right.  What changes is the shape of the work, chosen for the scoring forward (2 026 x 7 rows
// against 133 k keys per KV head, 98 % of the attention time of a scoring pass - profiles/r2_flash_probe.txt):
//   * a block owns 256 query rows of a head (8 waves x 32 rows: twice the reuse of every K / V tile of the 16-row kernel) and
//     walks the keys once in 64-key tiles;
//   * S^T = K.Q^T and O^T += V^T.P^T on v_mfma_f32_32x32x16: the query row is the COLUMN of both results, i.e. one row per lane,
//     softmax state is a per-lane scalar and P never leaves the registers: the 16 accumulator registers of a lane are keys
//     {8*(i/4) + 4*half + i%4}, and the transposed V reads (ds_read_b64_tr_b16) are ADDRESSED in exactly that key order, so the
//     packed P registers are the B operand as they are (no cross-lane exchange);
//   * K and V tiles come in by LDS-DMA (global_load_lds_dwordx4 from assembly) into a ring of three 32-KiB slots, two tiles
//     ahead, one fence-free barrier per tile with a counted s_waitcnt; K is XOR-swizzled on 16-byte chunks for conflict-free
//     ds_read_b128 fragments, V on 64-byte blocks for conflict-free transposed reads (both applied on the SOURCE address, the
//     DMA writes LDS linearly);
//   * online softmax in the exp2 domain with the scale folded into one fma per logit and a deferred rescale: the running
//     maximum only moves (and O is only rescaled) when a tile exceeds it by more than 2^F2_DEFER.
//   * round 4: the LAST ROUND of blocks is split along the keys.  One block per (head, 256-row tile) unit leaves CUs idle whenever the
//     units are not a multiple of the CUs: the scoring forward of the headline geometry has 56 row tiles x 4 heads = 224 units for
//     256 CUs, and G = 5, Hkv = 8 has 320: two rounds for 1.25 rounds of work.  With a workspace for partials, the units of an XCD
//     group beyond its last full round of CUs/8 blocks are cut into key ranges so that all blocks of that round get the same
//     number of key tiles: few helpers (28 units, 4 free CUs) -> every unit gives its last 1/8 of the keys to a tail block, the tail
//     blocks queue behind the first parts and each free CU works through seven of them; many helpers (8 units, 24 free CUs) ->
//     every unit is cut into four equal parts (up to 32: a call with 16 units fills the chip that way, which is why such calls
//     come here instead of going to the 16-row kernel).  The blocks that own the FIRST part of their units still walk the same
//     keys at the same time, so the L2 of an XCD keeps serving a key tile to all of them (a partition into arbitrary equal ranges
//     was measured first: every block then streams its own keys from HBM and the kernel is 8-39 % slower than without any split).
//     A part stores an unnormalised partial (m, l, O^T in fp32) and a second launch merges the parts of a unit.
// The 16-row kernel stays for what this one does not take: head dim 64, fewer than 16 units, calls without a workspace (kvz_flash.hip).
#include "kvz_mfma_lds.h"

#include <math.h>

#include <type_traits>

namespace kvz {

typedef short f2s4 __attribute__((ext_vector_type(4)));
typedef short f2s8 __attribute__((ext_vector_type(8)));

constexpr int F2_WAVES = 8, F2_THREADS = F2_WAVES * 64;
constexpr int F2_ROWS = F2_WAVES * 32;   // query rows per block
constexpr int F2_KT = 64;                // keys per tile
constexpr int F2_RING = 3;
constexpr int F2_MAXH = 64;
constexpr float F2_DEFER = 6.f;          // exp2-domain slack before the running maximum moves (P <= 2^6: exact in fp16 / bf16 ranges)

struct Flash2Args {
    const void* q;
    const void* k;
    const void* v;
    void* out;
    float* lse;                       // optional [Hkv, R] natural-log LSE of the scaled logits
    const int32_t* k_start;           // device arrays (used when n_meta == 0)
    const int32_t* k_len;
    int32_t m_start[F2_MAXH], m_len[F2_MAXH];  // host copies (n_meta = Hkv)
    int n_meta;
    int k_len_offset;
    int64_t q_sh, q_sg, q_si;         // element strides of (head, group member, position) in q
    int64_t o_sh, o_sg, o_si;         // ... and in out
    int Hkv, G, q_len, causal;
    float scale;
    int n_rt;                         // row tiles per head
    int xcd_mode, xcd_par;            // block -> (head, row tile) mapping, see the kernel
    // scoring window (f2: the row statistics of KVScore._get_score come out of the forward's own QK^T tiles): keys [0, win_sink) ++
    // [win_start, win_end) ++ the last q_len keys of the segment; win_stats [Hkv, win_stats_stride] float2 (m_r, l'_r), row index
    // g*q_len + i (the layout the column-maximum pass reads)
    int win_sink, win_start, win_end;
    float2* win_stats;
    int64_t win_stats_stride;
    float win_c, win_rcp;             // sqrt(D) and its exact reciprocal (0: divide)
    // split last round: split_blocks = blocks of a round over all XCD groups (0: one block per unit); partial slot of part p of the
    // i-th unit of group c's last round: (c * 2 * blocks per group) + i * parts + p
    FastDiv dG;                       // G
    int split_blocks;
    float2* part_ml;                  // [2*split_blocks][F2_ROWS] (m in the exp2 domain, l)
    float4* part_o;                   // [2*split_blocks][D/4][F2_ROWS]: O^T, four consecutive d per element, rows contiguous
};
constexpr int F2_MAXPARTS = 32;

// ---- the split last round of an XCD group: nu units, nb blocks per round, rem = nu % nb units left for the last round ----
// many helpers (nb - rem >= rem): each unit is cut into parts = 1 + helpers per unit equal parts (at most F2_MAXPARTS);
// few helpers: two parts, the tail is the 1 / (1 + tpu) end of the keys; a free CU takes tpu tails one after the other (the tail
// blocks are dispatched after the first parts).
struct F2Round {
    int full, rem, parts, hpu, tpu;   // hpu > 0: many helpers; tpu > 0: few helpers; rem == 0: nothing to split
};
__device__ static inline F2Round f2_round(int nu, int nb) {
    F2Round r;
    r.rem = nb > 0 ? nu % nb : 0;
    r.full = nu - r.rem;
    r.parts = 1; r.hpu = 0; r.tpu = 0;
    const int helpers = nb - r.rem;
    if (r.rem == 0) return r;
    if (helpers >= r.rem) {
        r.hpu = min(helpers / r.rem, F2_MAXPARTS - 1);
        r.parts = 1 + r.hpu;
    } else {
        r.tpu = (r.rem + helpers - 1) / helpers;
        r.parts = 2;
    }
    return r;
}
// key tiles [t0, t1) of part p of a unit of n key tiles
__device__ static inline void f2_part(const F2Round& r, int n, int p, int& t0, int& t1) {
    if (r.hpu) {
        t0 = (int)((int64_t)n * p / r.parts);
        t1 = (int)((int64_t)n * (p + 1) / r.parts);
    } else {
        const int cut = (int)((int64_t)n * r.tpu / (r.tpu + 1));
        t0 = p ? cut : 0;
        t1 = p ? n : cut;
    }
}

// ---- the (head, row tile) units of an XCD group and their key tiles (shared by the forward and the merge kernel) ----
// group c = blockIdx % 8 with the XCD-aware orders (workgroup b runs on XCD b % 8), one group otherwise
__device__ static inline int f2_len(const Flash2Args& a, int h) { return (a.n_meta ? a.m_len[h] : a.k_len[h]) + a.k_len_offset; }
__device__ static inline int f2_units(const Flash2Args& a, int c) {
    if (a.xcd_mode == 1) {            // xcd_par = XCDs per head: the row tiles of head c / par with xt = c % par (mod par)
        const int cc = c % a.xcd_par;
        return a.n_rt > cc ? (a.n_rt - cc + a.xcd_par - 1) / a.xcd_par : 0;
    }
    if (a.xcd_mode == 2) return a.xcd_par * a.n_rt;  // xcd_par = heads per XCD
    return a.Hkv * a.n_rt;                            // head-major (what the 2-D grid of round 3 did)
}
__device__ static inline void f2_unit(const Flash2Args& a, int c, int ui, int& h, int& xt) {
    if (a.xcd_mode == 1) {
        h = c / a.xcd_par;
        xt = ui * a.xcd_par + (c - h * a.xcd_par);
    } else if (a.xcd_mode == 2) {
        const int hs = ui / a.n_rt;
        h = c * a.xcd_par + hs;
        xt = ui - hs * a.n_rt;
    } else {
        h = ui / a.n_rt;
        xt = ui - h * a.n_rt;
    }
}
// key tiles row tile xt has to walk: up to the limit of its last row (xt counts from the LAST row tile: the heaviest causal
// prefixes come first in dispatch order)
__device__ static inline int f2_tiles(const Flash2Args& a, int xt, int len) {
    const int rt = a.n_rt - 1 - xt, R = a.q_len * a.G;
    const int rl = min(R - 1, rt * F2_ROWS + F2_ROWS - 1);
    int blk_limit = a.causal ? a.dG.div(rl) + len - a.q_len : len - 1;
    blk_limit = min(blk_limit, len - 1);
    return blk_limit >= 0 ? blk_limit / F2_KT + 1 : 0;
}

// One part of one unit: key tiles [t_begin, te_u) of row tile xt (counted from the last) of head h, whose keys end after n_real tiles
// (te_u can exceed it only for the one virtual tile of a unit that sees no key).  `whole`: the part is the unit - normalise and store;
// otherwise (SPLIT kernels only) the unnormalised partial goes to `slot`.
template <typename T, bool WIN, bool FAST, bool SPLIT>
__device__ __attribute__((always_inline)) static void f2_walk(const Flash2Args& a, char* lds, int h, int xt, int len, int n_real, int t_begin,
                                                               int te_u, bool whole, int slot) {
    typedef typename Mfma32<T>::v8 v8;
    constexpr int D = 128, ROW_BYTES = D * 2, KK = D / 16, DB = D / 32;
    constexpr int TILE_BYTES = F2_KT * ROW_BYTES;                 // 16 KiB (K or V)
    constexpr int V_BASE = F2_RING * TILE_BYTES;                  // the three K tiles first (ds_read immediates stay below 64 KiB)
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    const int R = a.q_len * a.G;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    // ---- staging: tile t -> ring slot s; 16 + 16 one-KiB pieces, two of each per wave ----
    const uint32_t lds0 = lds_addr(lds);
    const int srow = wave * 4 + (lane >> 4);                       // row of the piece this lane writes (first piece; second: +32)
    const int sp = lane & 15;                                      // 16-byte position inside the row
    const uint32_t k_lane_off = (uint32_t)(srow * ROW_BYTES + ((sp ^ (srow & 15)) << 4));
    const uint32_t v_lane_off = (uint32_t)(srow * ROW_BYTES + ((sp ^ ((srow & 3) << 2)) << 4));
    // ---- fragment addresses ----
    // K (A operand of S^T = K.Q^T): row l31 of a 32-key sub-block, 16-byte chunk kk*2 + half, XOR-swizzled with (row & 15)
    uint32_t kaddr[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) kaddr[kk] = lds0 + (uint32_t)(l31 * ROW_BYTES + (((kk * 2 + half) ^ (l31 & 15)) << 4));
    // V^T (A operand of O^T += V^T.P^T) by transposed reads: the 16 lanes of a group read a [4 keys][16 d] block, lane j the 8 bytes
    // d = 4*(j&3).. of key row j>>2, and receive d = j of the four keys.  Groups: d half (lane>>4)&1, key half = `half`.
    const int j16 = lane & 15, rho = j16 >> 2;
    const uint32_t vaddr0 = lds0 + (uint32_t)(V_BASE + (4 * half + rho) * ROW_BYTES + ((((lane >> 4) & 1) * 2 + ((j16 & 3) >> 1)) << 4) + (j16 & 1) * 8);
    uint32_t vx[DB];  // + swizzled 64-byte block of d-block db
#pragma unroll
    for (int db = 0; db < DB; ++db) vx[db] = vaddr0 + (uint32_t)((db ^ rho) << 6);

    const float sl2 = a.scale * 1.44269504088896340736f;

    const int t_end = min(te_u, n_real);
    const int rt = a.n_rt - 1 - xt;
    const int64_t seg = a.n_meta ? a.m_start[h] : a.k_start[h];
    const char* kbase = reinterpret_cast<const char*>(a.k) + seg * ROW_BYTES;
    const char* vbase = reinterpret_cast<const char*>(a.v) + seg * ROW_BYTES;
    const int rep0 = len - a.q_len;   // first key of the repeat chunk

    // ---- this lane's query row (column of S^T and O^T): r = i*G + g inside head h ----
    const int r_wave = rt * F2_ROWS + wave * 32;
    const int r = r_wave + l31;
    const bool rvalid = r < R;
    const int rc = rvalid ? r : R - 1;
    const int qi = a.dG.div(rc), qg = rc - qi * a.G;
    v8 qf[KK];
    {
        const T* qp = reinterpret_cast<const T*>(a.q) + h * a.q_sh + qg * a.q_sg + qi * a.q_si + half * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) qf[kk] = __builtin_bit_cast(v8, *reinterpret_cast<const u32x4*>(qp + kk * 16));
    }
    int limit = a.causal ? qi + len - a.q_len : len - 1;  // last visible key (bottom-right aligned mask)
    if (limit > len - 1) limit = len - 1;
    if (!rvalid) limit = -1;
    // wave-uniform bounds of the limits (rows of a wave are consecutive, the limit is monotone in the row)
    int wmin = -1, wmax = -1;
    if (r_wave < R) {
        const int q_first = a.dG.div(r_wave), q_last = a.dG.div(min(r_wave + 31, R - 1));
        wmin = min(len - 1, a.causal ? q_first + len - a.q_len : len - 1);
        wmax = min(len - 1, a.causal ? q_last + len - a.q_len : len - 1);
    }
    // window statistics of this lane's half of the row (attention/score.py:57-61 on the forward's accumulators): reference value
    // wm (a 16-bit logit), wl = sum of 2^(x*log2e - fl(wm*log2e)) over the window keys seen so far
    float wm = -INFINITY, wml2 = 0.f, wl = 0.f;
    float m2 = -INFINITY;   // running maximum in the exp2 domain (scaled logits * log2e)
    float l_run = 0.f;
    f16v o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;

    auto stage = [&](int t, int s) __attribute__((always_inline)) {
        const uint32_t kd = lds0 + (uint32_t)(s * TILE_BYTES + wave * 1024), vd = kd + (uint32_t)V_BASE;
        const int t0 = t * F2_KT;
        if (t0 + F2_KT <= len) {
            const char* kb = kbase + (int64_t)t0 * ROW_BYTES;
            const char* vb = vbase + (int64_t)t0 * ROW_BYTES;
            lds_dma16a(kb, k_lane_off, kd);
            lds_dma16a(kb + 32 * ROW_BYTES, k_lane_off, kd + 8 * 1024);
            lds_dma16a(vb, v_lane_off, vd);
            lds_dma16a(vb + 32 * ROW_BYTES, v_lane_off, vd + 8 * 1024);
        } else {  // last tile of the segment: rows beyond it shadow the last row (masked by the limits)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int key = min(t0 + 32 * i + srow, len - 1);
                lds_dma16a(kbase, (uint32_t)(key * ROW_BYTES + ((sp ^ (srow & 15)) << 4)), kd + (uint32_t)(i * 8 * 1024));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int key = min(t0 + 32 * i + srow, len - 1);
                lds_dma16a(vbase, (uint32_t)(key * ROW_BYTES + ((sp ^ ((srow & 3) << 2)) << 4)), vd + (uint32_t)(i * 8 * 1024));
            }
        }
    };

    // one tile: slot B of the ring.  MASK: some key of the tile is hidden from some row of this wave
    auto tile = [&](int t, auto b_tag, auto mask_tag) __attribute__((always_inline)) {
        constexpr int B = decltype(b_tag)::value;
        constexpr bool MASK = decltype(mask_tag)::value;
        typedef const __attribute__((address_space(3))) u32x4* lp16;
        typedef __attribute__((address_space(3))) f2s4* lp8;
        // ---- S^T: two sub-blocks of 32 keys ----
        f16v s[2];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const u32x4 kf = *(lp16)(uintptr_t)(kaddr[kk] + (uint32_t)(B * TILE_BYTES + sb * 32 * ROW_BYTES));
                if (kk == 0) {
                    const f16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    s[sb] = Mfma32<T>::mfma(__builtin_bit_cast(v8, kf), qf[kk], z);
                } else {
                    s[sb] = Mfma32<T>::mfma(__builtin_bit_cast(v8, kf), qf[kk], s[sb]);
                }
            }
        }
        // ---- online softmax, one query row per lane (its other 32 keys of the tile live in lane ^ 32) ----
        if constexpr (MASK) {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int rel = limit - (t * F2_KT + sb * 32 + 4 * half);  // key offset (i&3)+8*(i>>2) is visible iff <= rel
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((i & 3) + 8 * (i >> 2) > rel) s[sb][i] = -INFINITY;
            }
        }
        if constexpr (WIN) {
            const int t0w = t * F2_KT;
            const bool hit = t0w < a.win_sink || (t0w < a.win_end && t0w + F2_KT > a.win_start) || t0w + F2_KT > rep0;
            if (hit) {  // (wave-uniform; 3 % of the tiles of a scoring forward)
                constexpr float L2E = 1.44269504088896340736f;
                float x[2][16];
                float xmax = -INFINITY;
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = t0w + sb * 32 + 8 * (i >> 2) + 4 * half + (i & 3);
                        const bool in = key < a.win_sink || (key >= a.win_start && key < a.win_end) || key >= rep0;
                        // (hidden keys of the repeat chunk are already -inf when the tile took the masked path; a window tile that
                        // reaches into the causal zone always does)
                        const float xv = round_chain<T, FAST>(s[sb][i], a.win_c, a.win_rcp);
                        x[sb][i] = (in && key <= limit) ? xv : -INFINITY;
                        xmax = fmaxf(xmax, x[sb][i]);
                    }
                if (xmax > wm) {  // the reference only moves up (per lane: the two halves of a row are merged at the end)
                    const float nml2 = xmax * L2E;
                    wl *= (wm == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(wml2 - nml2);
                    wm = xmax;
                    wml2 = nml2;
                }
                const float nw = (wm == -INFINITY) ? 0.f : -wml2;
                float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {
                        acc0 += __builtin_amdgcn_exp2f(__builtin_fmaf(x[sb][i], L2E, nw));
                        acc1 += __builtin_amdgcn_exp2f(__builtin_fmaf(x[sb][i + 1], L2E, nw));
                    }
                wl += acc0 + acc1;
            }
        }
        float tmax = s[0][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, s[0][i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, s[1][i]);
        {   // the row's other half-wave: v_permlane32_swap exchanges lanes 32-63 of one operand with lanes 0-31 of the other
            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, tmax), __builtin_bit_cast(unsigned, tmax), false, false);
            tmax = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
        }
        const float t2 = tmax * sl2;
        // deferred rescale: move the maximum only when some row of the wave outgrew it by more than 2^F2_DEFER
        if (__builtin_amdgcn_ballot_w64(t2 > m2 + F2_DEFER) != 0) {
            const float m_new = fmaxf(m2, t2);
            const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m2 - m_new);  // (m2 = -inf: alpha = 0, O and l are 0)
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
            m2 = m_new;
        }
        const float nm = (m2 == -INFINITY) ? 0.f : -m2;
        float ps0 = 0.f, ps1 = 0.f;
        v8 pb[2][2];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sb][i], sl2, nm));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sb][i + 1], sl2, nm));
                ps0 += p0;
                ps1 += p1;
                pb[sb][i >> 3][i & 7] = (T)p0;
                pb[sb][i >> 3][(i & 7) + 1] = (T)p1;
            }
        l_run += ps0 + ps1;
        // ---- O^T += V^T.P^T: four steps of 16 keys; key order of a step = the accumulator order of P ----
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t koff = (uint32_t)(B * TILE_BYTES + (sb * 32 + 16 * e) * ROW_BYTES);
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const f2s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp8)(uintptr_t)(vx[db] + koff));
                    const f2s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp8)(uintptr_t)(vx[db] + koff + 8 * ROW_BYTES));
                    const f2s8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[db] = Mfma32<T>::mfma(__builtin_bit_cast(v8, both), pb[sb][e], o[db]);
                }
            }
    };
    // hand-over at the end of tile t of a segment [t_begin, t_end): tile t+1 has landed for everybody, nobody reads the slot of tile t
    // any more -> tile t+3 goes there
    auto turnover = [&](int t, int slot) __attribute__((always_inline)) {
        if (t + 2 < t_end) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // (tile t+2 may still be in flight)
        else stage_wait();
        block_barrier();
        if (t + 3 < t_end) stage(t + 3, slot);
    };
    auto step = [&](int t, auto b_tag) __attribute__((always_inline)) {
        const int t0 = t * F2_KT;
        if (t0 <= wmax) {  // (a tile behind the limit of every row of this wave: nothing to add, but the wave still stages and syncs)
            if (t0 + F2_KT - 1 > wmin) tile(t, b_tag, std::true_type{});
            else tile(t, b_tag, std::false_type{});
        }
        turnover(t, decltype(b_tag)::value);
    };

    const int ns = t_end - t_begin;
    if (ns > 0) {
        // (every wave has passed the last barrier of the previous segment: the ring is free)
        stage(t_begin, 0);
        if (ns > 1) stage(t_begin + 1, 1);
        if (ns > 2) stage(t_begin + 2, 2);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) asm volatile("" : "+v"(qf[kk]));  // (the wait for the query rows belongs here)
        if (ns > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ns > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else stage_wait();
        block_barrier();
        int t = t_begin;
        while (true) {
            step(t, I0{});
            if (++t >= t_end) break;
            step(t, I1{});
            if (++t >= t_end) break;
            step(t, I2{});
            if (++t >= t_end) break;
        }
    }

    if constexpr (WIN) {
        // merge the two half-waves of a row (disjoint keys), as the row-statistics pass does: (m, l' relative to fl(m*log2e))
        const float m_o = __shfl_xor(wm, 32, 64), ml2_o = __shfl_xor(wml2, 32, 64), l_o = __shfl_xor(wl, 32, 64);
        const float M = fmaxf(wm, m_o);
        const float ML2 = (wm >= m_o) ? wml2 : ml2_o;
        const float la = (wm == -INFINITY) ? 0.f : wl * __builtin_amdgcn_exp2f(wml2 - ML2);
        const float lb = (m_o == -INFINITY) ? 0.f : l_o * __builtin_amdgcn_exp2f(ml2_o - ML2);
        if (rvalid && half == 0) a.win_stats[(int64_t)h * a.win_stats_stride + (int64_t)qg * a.q_len + qi] = make_float2(M, la + lb);
    }
    l_run += __shfl_xor(l_run, 32, 64);
    if (SPLIT && !whole) {
        // ---- partial of a split unit: unnormalised, lane holds O^T[d = db*32 + 8*(i>>2) + 4*half + (i&3)][row] -> float4 of four
        // consecutive d, rows contiguous (coalesced 512-byte runs per half-wave); every row of the tile is written
        const int row = wave * 32 + l31;
        float4* po = a.part_o + (int64_t)slot * (D / 4) * F2_ROWS + row;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                po[(db * 8 + 2 * g4 + half) * F2_ROWS] = make_float4(o[db][4 * g4], o[db][4 * g4 + 1], o[db][4 * g4 + 2], o[db][4 * g4 + 3]);
        if (half == 0) a.part_ml[(int64_t)slot * F2_ROWS + row] = make_float2(m2, l_run);
    } else if (rvalid) {
        // ---- normalise and store: lane holds O^T[d = db*32 + 8*(i>>2) + 4*half + (i&3)][row l31] ----
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        T* op = reinterpret_cast<T*>(a.out) + h * a.o_sh + qg * a.o_sg + qi * a.o_si + 4 * half;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                T w[4] = {(T)(o[db][4 * g4] * inv), (T)(o[db][4 * g4 + 1] * inv), (T)(o[db][4 * g4 + 2] * inv), (T)(o[db][4 * g4 + 3] * inv)};
                *reinterpret_cast<uint2*>(op + db * 32 + 8 * g4) = *reinterpret_cast<const uint2*>(w);
            }
        if (a.lse && half == 0)
            a.lse[(int64_t)h * R + r] = (l_run > 0.f) ? (m2 + log2f(l_run)) * 0.69314718055994530942f : -INFINITY;
    }
}

template <typename T, bool WIN, bool FAST, bool SPLIT>
__global__ __launch_bounds__(F2_THREADS, 1) void flash2_fwd_kernel(Flash2Args a) {
    constexpr int TILE_BYTES = F2_KT * 128 * 2;
    __shared__ __attribute__((aligned(16))) char lds[2 * F2_RING * TILE_BYTES];

    // ---- block -> units, XCD-aware (round 4).  Workgroup b runs on XCD b % 8 and every XCD has its own 4-MiB L2: with
    // the row tiles of a head spread over all eight XCDs (the plain 2-D grid of round 3) every XCD pulled every head's K and V
    // through the fabric - 2.46 GB of fabric reads for 0.30 GB of inputs at the scoring forward's shape
    // (profiles/r3_pmc_traffic.json).  Now a head's row tiles run on 8 / Hkv XCDs (Hkv <= 8) or an XCD owns Hkv / 8 whole heads;
    // inside a head's XCD group the row tiles are dealt round-robin, so the causal prefixes stay balanced over the XCDs, and
    // in dispatch order the heaviest row tiles (longest causal prefixes) still come first.
    // Block (c, sl) owns unit sl of group c - except, in the SPLIT kernel, in the group's last round (see the head of the file): there
    // it owns one part of one unit.
    const int b = (int)blockIdx.x;
    const int c = a.xcd_mode ? (b & 7) : 0, sl = a.xcd_mode ? (b >> 3) : b;
    const int nu = f2_units(a, c);
    if constexpr (!SPLIT) {
        if (sl >= nu) return;  // (padding blocks when 8 / Hkv does not divide the row tiles)
        int h, xt;
        f2_unit(a, c, sl, h, xt);
        const int len = f2_len(a, h), n = f2_tiles(a, xt, len);
        f2_walk<T, WIN, FAST, false>(a, lds, h, xt, len, n, 0, n, true, 0);
    } else {
        const int nb_g = a.split_blocks / (a.xcd_mode ? 8 : 1);
        const F2Round rd = f2_round(nu, nb_g);
        int ui = sl, part = 0;                       // part `part` of unit ui of the group
        bool split_unit = false;
        if (sl >= rd.full) {
            const int j = sl - rd.full;              // block of the last round
            if (rd.rem == 0) return;                 // (padding blocks: the grid is sized for the largest group)
            split_unit = true;
            if (j < rd.rem) {
                ui = rd.full + j; part = 0;
            } else if (rd.hpu) {
                const int k = j - rd.rem, i = k / rd.hpu;
                if (i >= rd.rem) return;             // (helpers that do not divide evenly stay idle)
                ui = rd.full + i; part = 1 + (k - i * rd.hpu);
            } else {                                 // one block per tail, queued behind the first parts: the CUs that get no first
                if (j >= 2 * rd.rem) return;         // part take tpu of them each, one after the other
                ui = rd.full + (j - rd.rem); part = 1;
            }
        }
        int h, xt;
        f2_unit(a, c, ui, h, xt);
        const int len = f2_len(a, h);
        const int n_real = f2_tiles(a, xt, len), n_u = max(1, n_real);  // (a unit whose rows see no key still has to write its zeros)
        int t_begin = 0, te_u = n_u;
        if (split_unit) f2_part(rd, n_u, part, t_begin, te_u);
        if (t_begin >= te_u) return;                 // (an empty part of a short unit)
        f2_walk<T, WIN, FAST, true>(a, lds, h, xt, len, n_real, t_begin, te_u, t_begin == 0 && te_u == n_u,
                                    c * 2 * nb_g + (ui - rd.full) * rd.parts + part);
    }
}

// Merge of the parts of the units of a split last round: one block per such unit (block (c, i) = i-th unit of group c's last round);
// a unit with a single non-empty part has its result already.  blockIdx.y = one of the eight 32-row groups of the unit (a unit cut into
// 32 parts is 4 MiB of partials: one block per unit left the merge of a few units slower than their forward); thread = (row, 8 of d).
template <typename T>
__global__ __launch_bounds__(F2_THREADS) void flash2_merge_kernel(Flash2Args a) {
    constexpr int D = 128, MAXSEG = F2_MAXPARTS;
    const int b = (int)blockIdx.x;
    const int n_groups = a.xcd_mode ? 8 : 1;
    const int c = a.xcd_mode ? (b & 7) : 0, i = a.xcd_mode ? (b >> 3) : b;
    const int nu = f2_units(a, c);
    const int nb_g = a.split_blocks / n_groups;
    const F2Round rd = f2_round(nu, nb_g);
    if (i >= rd.rem) return;
    int h, xt;
    f2_unit(a, c, rd.full + i, h, xt);
    const int n_u = max(1, f2_tiles(a, xt, f2_len(a, h)));
    int slots[MAXSEG], nseg = 0;
    for (int p = 0; p < rd.parts; ++p) {
        int t0, t1;
        f2_part(rd, n_u, p, t0, t1);
        if (t0 < t1) slots[nseg++] = c * 2 * nb_g + i * rd.parts + p;
    }
    if (nseg <= 1) return;  // one part walked the whole unit: stored by the forward
    const int R = a.q_len * a.G;
    const int rt = a.n_rt - 1 - xt;
    const int row = (int)blockIdx.y * 32 + (threadIdx.x & 31), dh = threadIdx.x >> 5;  // 512 threads: 32 rows x 16 pairs of d-quads
    const int r = rt * F2_ROWS + row;
    if (r >= R) return;
    float M = -INFINITY;
    for (int sgm = 0; sgm < nseg; ++sgm) M = fmaxf(M, a.part_ml[(int64_t)slots[sgm] * F2_ROWS + row].x);
    float w[MAXSEG], L = 0.f;
    for (int sgm = 0; sgm < nseg; ++sgm) {
        const float2 ml = a.part_ml[(int64_t)slots[sgm] * F2_ROWS + row];
        w[sgm] = (ml.x == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(ml.x - M);
        L += w[sgm] * ml.y;
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
    const int qi = a.dG.div(r), qg = r - qi * a.G;
    T* op = reinterpret_cast<T*>(a.out) + h * a.o_sh + qg * a.o_sg + qi * a.o_si;
    for (int dq = dh; dq < D / 4; dq += 16) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sgm = 0; sgm < nseg; ++sgm) {
            const float4 p = a.part_o[((int64_t)slots[sgm] * (D / 4) + dq) * F2_ROWS + row];
            acc.x += w[sgm] * p.x; acc.y += w[sgm] * p.y; acc.z += w[sgm] * p.z; acc.w += w[sgm] * p.w;
        }
        T o4[4] = {(T)(acc.x * inv), (T)(acc.y * inv), (T)(acc.z * inv), (T)(acc.w * inv)};
        *reinterpret_cast<uint2*>(op + dq * 4) = *reinterpret_cast<const uint2*>(o4);
    }
    if (a.lse && dh == 0) a.lse[(int64_t)h * R + r] = (L > 0.f) ? (M + log2f(L)) * 0.69314718055994530942f : -INFINITY;
}

// Does the 32-row kernel take this call?  Head dim 128, and enough (head, 256-row tile) units: without a workspace every unit is one
// block that walks all of its head's keys, so the units have to fill the chip by themselves (below that the 16-row kernel with its
// key splits does better); with the workspace of the split last round two units per XCD are enough - their keys are cut into up
// to F2_MAXPARTS parts.  Measured (profiles/r4_flash_small_shapes.txt, fp16, 28 / 4 heads): 16 units against 8 k keys 65 us, the
// 16-row kernel 63; against 133 k keys 317 and 558; 56 units 1 033 and 2 035.  (8 units: 89 and 54 at 8 k keys, 216 and 287 at 133 k.)
constexpr int F2_MIN_UNITS_SPLIT = 16;
bool flash2_takes(int Hkv, int G, int q_len, int D, bool with_ws) {
    if (D != 128 || Hkv <= 0 || G <= 0 || q_len <= 0) return false;
    const int64_t blocks = (int64_t)((q_len * (int64_t)G + F2_ROWS - 1) / F2_ROWS) * Hkv;
    int need = tunable(TUNE_FLASH2_MIN_BLOCKS);
    // (a knob raised above its default keeps the 32-row kernel out altogether: that is how the tests force the 16-row kernel)
    if (with_ws && tunable(TUNE_FLASH2_SPLIT) != 0 && need > F2_MIN_UNITS_SPLIT && need <= 128) need = F2_MIN_UNITS_SPLIT;
    return blocks >= need;
}

// ---- balanced partition: grid and workspace ----
// One block per CU (96 KiB of LDS), a multiple of 8 so that every XCD group has the same number of blocks.
static int f2_cus() { const int cus = device_cus(); return cus >= 8 ? cus / 8 * 8 : 256; }
constexpr size_t F2_SLOT_BYTES = (size_t)F2_ROWS * sizeof(float2) + (size_t)F2_ROWS * 128 * sizeof(float);
static inline size_t f2_align256(size_t x) { return (x + 255) & ~(size_t)255; }
// XCD-aware order of the units (see the kernel): mode, parameter, number of groups
static void f2_order(int Hkv, int& mode, int& par, int& groups) {
    mode = 0; par = 1; groups = 1;
    if (tunable(TUNE_FLASH2_XCD) != 0) {
        if (Hkv <= 8 && 8 % Hkv == 0) { mode = 1; par = 8 / Hkv; groups = 8; }
        else if (Hkv % 8 == 0) { mode = 2; par = Hkv / 8; groups = 8; }
    }
}
// Blocks of a round: one per CU.
static int f2_split_blocks() { return f2_cus(); }
// Does this call split its last round of blocks along the keys (and therefore touch the workspace)?  One predicate for the
// workspace size and the launch (round 5, ADVICE: the size used to be 2 x CUs x 133 KiB = 68 MB for EVERY head-dim-128 call, which
// ops.flash_fwd then allocated per layer although full-chip prefill shapes never split).
static bool f2_will_split(int Hkv, int n_rt, int* groups_out, int* nu_max_out, int* nb_g_out) {
    int xcd_mode, xcd_par, groups;
    f2_order(Hkv, xcd_mode, xcd_par, groups);
    const int nb_g = f2_split_blocks() / groups;
    const int nu_max = xcd_mode == 1 ? (n_rt + xcd_par - 1) / xcd_par : (xcd_mode == 2 ? xcd_par * n_rt : n_rt * Hkv);
    const int rounds = (nu_max + nb_g - 1) / nb_g;
    if (groups_out) *groups_out = groups;
    if (nu_max_out) *nu_max_out = nu_max;
    if (nb_g_out) *nb_g_out = nb_g;
    if (tunable(TUNE_FLASH2_SPLIT) == 0) return false;
    return nu_max % nb_g != 0 && (tunable(TUNE_FLASH2_SPLIT) >= 2 || 10 * rounds * nb_g >= 11 * nu_max);
}
size_t flash2_workspace_bytes(int Hkv, int G, int q_len, int D) {
    if (D != 128 || Hkv <= 0 || G <= 0 || q_len <= 0) return 0;
    const int n_rt = (int)(((int64_t)q_len * G + F2_ROWS - 1) / F2_ROWS);
    if (!f2_will_split(Hkv, n_rt, nullptr, nullptr, nullptr)) return 0;
    return f2_align256((size_t)2 * f2_split_blocks() * F2_SLOT_BYTES);
}

// (same arguments as kvz_flash_fwd, which dispatches here; ws: NULL or flash2_workspace_bytes() bytes, no initialisation needed -
// without it every (head, row tile) is one block)
int flash2_fwd(const void* q, int64_t q_stride_head, int64_t q_stride_group, int64_t q_stride_pos, const void* k,
               const void* v, const int32_t* k_start, const int32_t* k_len, int k_len_offset, const int32_t* k_meta_host,
               int Hkv, int G, int q_len, float scale, int causal, int dtype, void* out, int64_t o_stride_head,
               int64_t o_stride_group, int64_t o_stride_pos, float* lse_out, void* ws, size_t ws_bytes, hipStream_t stream, int win_sink,
               int win_start, int win_end, float* win_stats, int64_t win_stats_head_stride) {
    Flash2Args a{};
    a.q = q; a.k = k; a.v = v; a.out = out; a.lse = lse_out;
    a.k_start = k_start; a.k_len = k_len; a.k_len_offset = k_len_offset;
    a.n_meta = 0;
    if (k_meta_host && Hkv <= F2_MAXH) {
        for (int h = 0; h < Hkv; ++h) { a.m_start[h] = k_meta_host[h]; a.m_len[h] = k_meta_host[Hkv + h]; }
        a.n_meta = Hkv;
    }
    a.q_sh = q_stride_head; a.q_sg = q_stride_group; a.q_si = q_stride_pos;
    a.o_sh = o_stride_head; a.o_sg = o_stride_group; a.o_si = o_stride_pos;
    a.Hkv = Hkv; a.G = G; a.q_len = q_len; a.causal = causal; a.scale = scale;
    a.dG = make_fastdiv(G);
    const int R = q_len * G;
    a.n_rt = (R + F2_ROWS - 1) / F2_ROWS;
    a.win_sink = win_sink; a.win_start = win_start; a.win_end = win_end;
    a.win_stats = reinterpret_cast<float2*>(win_stats); a.win_stats_stride = win_stats_head_stride;
    a.win_c = sqrtf(128.f);
    a.win_rcp = win_stats ? score_exact_reciprocal(128, dtype) : 0.f;
    int groups;
    f2_order(Hkv, a.xcd_mode, a.xcd_par, groups);
    // one block per unit: n_rt * Hkv blocks (with the per-head XCD groups: 8 x the units of the largest group, padding blocks exit)
    const int unit_blocks = a.xcd_mode == 1 ? 8 * ((a.n_rt + a.xcd_par - 1) / a.xcd_par) : a.n_rt * Hkv;
    int n_blocks = unit_blocks;
    // the split last round needs its workspace, does not carry the scoring window's statistics, and must pay for itself: the chip runs
    // at its power budget, so CUs that one block per unit leaves idle give most of their share back as clock (224 units on 256 CUs:
    // +2.4 % from the split where the CU count says +14 %), and the parts cost a merge launch.  Rounds of blocks needed without / with
    // the split: ceil(x) against x = units / blocks per round - split from a ratio of 1.1 (320 units: 2 against 1.25, +51 %).
    a.split_blocks = 0;
    const size_t need = win_stats ? 0 : flash2_workspace_bytes(Hkv, G, q_len, 128);
    if (ws && need && ws_bytes >= need) {
        KVZ_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15u) == 0, KVZ_EINVAL, "kvz_flash_fwd: workspace must be 16-byte aligned");
        int nb_g = 0, nu_max = 0;
        if (f2_will_split(Hkv, a.n_rt, nullptr, &nu_max, &nb_g)) {   // (need != 0 says so already; the numbers are needed below)
            a.split_blocks = f2_split_blocks();
            a.part_ml = reinterpret_cast<float2*>(ws);
            a.part_o = reinterpret_cast<float4*>(reinterpret_cast<char*>(ws) + (size_t)2 * a.split_blocks * F2_ROWS * sizeof(float2));
            // per group: the whole units of the full rounds, then one round of blocks for what is left (sized for the largest group)
            n_blocks = groups * (nu_max + nb_g);
        }
    }
    const dim3 grid(n_blocks), block(F2_THREADS);
    ProfScope ps("flash_fwd", stream);
#define KVZ_F2_LAUNCH(T, WIN, FAST, SPLIT) hipLaunchKernelGGL((flash2_fwd_kernel<T, WIN, FAST, SPLIT>), grid, block, 0, stream, a)
    if (a.split_blocks) {
        if (dtype == KVZ_F16) KVZ_F2_LAUNCH(_Float16, false, true, true); else KVZ_F2_LAUNCH(__bf16, false, true, true);
    } else if (!win_stats) {
        if (dtype == KVZ_F16) KVZ_F2_LAUNCH(_Float16, false, true, false); else KVZ_F2_LAUNCH(__bf16, false, true, false);
    } else if (a.win_rcp != 0.f) {
        if (dtype == KVZ_F16) KVZ_F2_LAUNCH(_Float16, true, true, false); else KVZ_F2_LAUNCH(__bf16, true, true, false);
    } else {
        if (dtype == KVZ_F16) KVZ_F2_LAUNCH(_Float16, true, false, false); else KVZ_F2_LAUNCH(__bf16, true, false, false);
    }
#undef KVZ_F2_LAUNCH
    KVZ_CHECK_LAUNCH("flash2_fwd_kernel");
    if (a.split_blocks) {
        const dim3 mgrid(a.split_blocks, F2_ROWS / 32);  // (at most blocks-per-round - 1 units per group are split)
        if (dtype == KVZ_F16) hipLaunchKernelGGL((flash2_merge_kernel<_Float16>), mgrid, block, 0, stream, a);
        else hipLaunchKernelGGL((flash2_merge_kernel<__bf16>), mgrid, block, 0, stream, a);
        KVZ_CHECK_LAUNCH("flash2_merge_kernel");
    }
    return KVZ_OK;
}

}  // namespace kvz
