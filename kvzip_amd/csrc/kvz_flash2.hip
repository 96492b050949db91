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
// The 16-row kernel stays for what this one does not take: head dim 64, key splits for short row counts (kvz_flash.hip).
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
};

template <typename T, bool WIN, bool FAST>
__global__ __launch_bounds__(F2_THREADS, 1) void flash2_fwd_kernel(Flash2Args a) {
    typedef typename Mfma32<T>::v8 v8;
    constexpr int D = 128, ROW_BYTES = D * 2, KK = D / 16, DB = D / 32;
    constexpr int TILE_BYTES = F2_KT * ROW_BYTES;                 // 16 KiB (K or V)
    constexpr int V_BASE = F2_RING * TILE_BYTES;                  // the three K tiles first (ds_read immediates stay below 64 KiB)
    __shared__ __attribute__((aligned(16))) char lds[2 * F2_RING * TILE_BYTES];
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;

    // ---- block -> (head, row tile), XCD-aware (round 4).  Workgroup b runs on XCD b % 8 and every XCD has its own 4-MiB L2: with
    // the row tiles of a head spread over all eight XCDs (the plain 2-D grid of round 3) every XCD pulled every head's K and V
    // through the fabric - 2.46 GB of fabric reads for 0.30 GB of inputs at the scoring forward's shape
    // (profiles/r3_pmc_traffic.json).  Now a head's row tiles run on 8 / Hkv XCDs (Hkv <= 8) or an XCD owns Hkv / 8 whole heads;
    // inside a head's XCD group the row tiles are dealt round-robin, so the causal prefixes stay balanced over the XCDs, and
    // in dispatch order the heaviest row tiles (longest causal prefixes) still come first.
    int h, xt;
    {
        const int b = (int)blockIdx.x, c = b & 7, sl = b >> 3;
        if (a.xcd_mode == 1) {        // xcd_par = XCDs per head
            h = c / a.xcd_par;
            xt = sl * a.xcd_par + (c - h * a.xcd_par);
        } else if (a.xcd_mode == 2) { // xcd_par = heads per XCD
            const int hs = sl / a.n_rt;
            h = c * a.xcd_par + hs;
            xt = sl - hs * a.n_rt;
        } else {                      // head-major (what the 2-D grid did)
            h = b / a.n_rt;
            xt = b - h * a.n_rt;
        }
        if (xt >= a.n_rt || h >= a.Hkv) return;  // (padding blocks of mode 1 when 8 / Hkv does not divide the row tiles)
    }
    const int rt = a.n_rt - 1 - xt;
    const int R = a.q_len * a.G;
    const int len = (a.n_meta ? a.m_len[h] : a.k_len[h]) + a.k_len_offset;
    const int64_t seg = a.n_meta ? a.m_start[h] : a.k_start[h];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    // ---- this lane's query row (column of S^T and O^T): r = i*G + g inside head h ----
    const int r_wave = rt * F2_ROWS + wave * 32;
    const int r = r_wave + l31;
    const bool rvalid = r < R;
    const int rc = rvalid ? r : R - 1;
    const int qi = rc / a.G, qg = rc - qi * a.G;
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
        const int q_first = r_wave / a.G, q_last = min(r_wave + 31, R - 1) / a.G;
        wmin = min(len - 1, a.causal ? q_first + len - a.q_len : len - 1);
        wmax = min(len - 1, a.causal ? q_last + len - a.q_len : len - 1);
    }
    // keys this block has to walk: up to the limit of its last row
    int n_tiles;
    {
        const int rl = min(R - 1, rt * F2_ROWS + F2_ROWS - 1);
        int blk_limit = a.causal ? rl / a.G + len - a.q_len : len - 1;
        blk_limit = min(blk_limit, len - 1);
        n_tiles = blk_limit >= 0 ? blk_limit / F2_KT + 1 : 0;
    }

    // ---- staging: tile t -> ring slot s; 16 + 16 one-KiB pieces, two of each per wave ----
    const char* kbase = reinterpret_cast<const char*>(a.k) + seg * ROW_BYTES;
    const char* vbase = reinterpret_cast<const char*>(a.v) + seg * ROW_BYTES;
    const uint32_t lds0 = lds_addr(lds);
    const int srow = wave * 4 + (lane >> 4);                       // row of the piece this lane writes (first piece; second: +32)
    const int sp = lane & 15;                                      // 16-byte position inside the row
    const uint32_t k_lane_off = (uint32_t)(srow * ROW_BYTES + ((sp ^ (srow & 15)) << 4));
    const uint32_t v_lane_off = (uint32_t)(srow * ROW_BYTES + ((sp ^ ((srow & 3) << 2)) << 4));
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

    // window statistics of this lane's half of the row (attention/score.py:57-61 on the forward's accumulators): reference value
    // wm (a 16-bit logit), wl = sum of 2^(x*log2e - fl(wm*log2e)) over the window keys seen so far
    float wm = -INFINITY, wml2 = 0.f, wl = 0.f;
    const int rep0 = len - a.q_len;   // first key of the repeat chunk
    float m2 = -INFINITY;   // running maximum in the exp2 domain (scaled logits * log2e)
    float l_run = 0.f;
    f16v o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    const float sl2 = a.scale * 1.44269504088896340736f;

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
    // hand-over at the end of tile t: tile t+1 has landed for everybody, nobody reads slot t % 3 any more -> tile t+3 goes there
    auto turnover = [&](int t, int slot) __attribute__((always_inline)) {
        if (t + 2 < n_tiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // (tile t+2 may still be in flight)
        else stage_wait();
        block_barrier();
        if (t + 3 < n_tiles) stage(t + 3, slot);
    };
    auto step = [&](int t, auto b_tag) __attribute__((always_inline)) {
        const int t0 = t * F2_KT;
        if (t0 <= wmax) {  // (a tile behind the limit of every row of this wave: nothing to add, but the wave still stages and syncs)
            if (t0 + F2_KT - 1 > wmin) tile(t, b_tag, std::true_type{});
            else tile(t, b_tag, std::false_type{});
        }
        turnover(t, decltype(b_tag)::value);
    };

    if (n_tiles > 0) {
        stage(0, 0);
        if (n_tiles > 1) stage(1, 1);
        if (n_tiles > 2) stage(2, 2);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) asm volatile("" : "+v"(qf[kk]));  // (the wait for the query rows belongs here)
        if (n_tiles > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n_tiles > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else stage_wait();
        block_barrier();
        int t = 0;
        while (true) {
            step(t, I0{});
            if (++t >= n_tiles) break;
            step(t, I1{});
            if (++t >= n_tiles) break;
            step(t, I2{});
            if (++t >= n_tiles) break;
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
    // ---- normalise and store: lane holds O^T[d = db*32 + 8*(i>>2) + 4*half + (i&3)][row l31] ----
    l_run += __shfl_xor(l_run, 32, 64);
    if (rvalid) {
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

// Does the 32-row kernel take this call?  Head dim 128, and enough (head, 256-row tile) blocks to fill the chip without key splits.
bool flash2_takes(int Hkv, int G, int q_len, int D) {
    if (D != 128 || Hkv <= 0 || G <= 0 || q_len <= 0) return false;
    const int64_t blocks = (int64_t)((q_len * (int64_t)G + F2_ROWS - 1) / F2_ROWS) * Hkv;
    return blocks >= tunable(TUNE_FLASH2_MIN_BLOCKS);  // fewer blocks: the 16-row kernel with key splits fills the chip better
}

// (same arguments as kvz_flash_fwd, which dispatches here; no workspace)
int flash2_fwd(const void* q, int64_t q_stride_head, int64_t q_stride_group, int64_t q_stride_pos, const void* k,
               const void* v, const int32_t* k_start, const int32_t* k_len, int k_len_offset, const int32_t* k_meta_host,
               int Hkv, int G, int q_len, float scale, int causal, int dtype, void* out, int64_t o_stride_head,
               int64_t o_stride_group, int64_t o_stride_pos, float* lse_out, hipStream_t stream, int win_sink, int win_start,
               int win_end, float* win_stats, int64_t win_stats_head_stride) {
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
    const int R = q_len * G;
    a.n_rt = (R + F2_ROWS - 1) / F2_ROWS;
    a.win_sink = win_sink; a.win_start = win_start; a.win_end = win_end;
    a.win_stats = reinterpret_cast<float2*>(win_stats); a.win_stats_stride = win_stats_head_stride;
    a.win_c = sqrtf(128.f);
    a.win_rcp = win_stats ? score_exact_reciprocal(128, dtype) : 0.f;
    int n_blocks = a.n_rt * Hkv;
    a.xcd_mode = 0; a.xcd_par = 1;
    if (tunable(TUNE_FLASH2_XCD) != 0) {
        if (Hkv <= 8 && 8 % Hkv == 0) {
            a.xcd_mode = 1; a.xcd_par = 8 / Hkv;
            n_blocks = 8 * ((a.n_rt + a.xcd_par - 1) / a.xcd_par);
        } else if (Hkv % 8 == 0) {
            a.xcd_mode = 2; a.xcd_par = Hkv / 8;
        }
    }
    const dim3 grid(n_blocks), block(F2_THREADS);
    ProfScope ps("flash_fwd", stream);
#define KVZ_F2_LAUNCH(T, WIN, FAST) hipLaunchKernelGGL((flash2_fwd_kernel<T, WIN, FAST>), grid, block, 0, stream, a)
    if (!win_stats) {
        if (dtype == KVZ_F16) KVZ_F2_LAUNCH(_Float16, false, true); else KVZ_F2_LAUNCH(__bf16, false, true);
    } else if (a.win_rcp != 0.f) {
        if (dtype == KVZ_F16) KVZ_F2_LAUNCH(_Float16, true, true); else KVZ_F2_LAUNCH(__bf16, true, true);
    } else {
        if (dtype == KVZ_F16) KVZ_F2_LAUNCH(_Float16, true, false); else KVZ_F2_LAUNCH(__bf16, true, false);
    }
#undef KVZ_F2_LAUNCH
    KVZ_CHECK_LAUNCH("flash2_fwd_kernel");
    return KVZ_OK;
}

}  // namespace kvz
