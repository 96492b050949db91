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
// Tiling (both passes): the operand whose index is reduced over is STREAMED through LDS in 128-row tiles
// and used as the MFMA A operand (rows of the 32x32 result live in registers, so the reduction is
// lane-local); the other operand is STATIONARY in registers as the B operand (32 columns per wave,
// one column per lane).  LDS tiles are XOR-swizzled on 16-byte chunks so that ds_read_b128 fragment
// reads are bank-conflict free.  Bound: MFMA co-limited by the VALU rounding chain.
#include "kvz_common.h"

#include <math.h>
#include <string.h>

#include <type_traits>

namespace kvz {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma32;
template <> struct Mfma32<_Float16> {
    typedef h8 v8;
    __device__ static inline f16v mfma(v8 a, v8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma32<__bf16> {
    typedef b8 v8;
    __device__ static inline f16v mfma(v8 a, v8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

constexpr int SC_TILE = 128;     // streamed rows per LDS tile (2 x 32 KiB LDS buffers per block -> 2 blocks per CU)
// pass A: 4 waves / block, 2 waves per SIMD, up to 256 VGPRs: deep register prefetch of the A fragments
// pass B: 8 waves / block, 4 waves per SIMD, <= 128 VGPRs
constexpr int PA_WAVES = 4, PB_WAVES = 8;
#ifndef KVZ_KSPLIT_TILES
#define KVZ_KSPLIT_TILES 8
#endif
constexpr int SC_KSPLIT_TILES = KVZ_KSPLIT_TILES;  // pass A: key tiles per block (load balance under the causal mask)

struct ScoreArgs {
    const void* q;       // [Hkv*G, q_len, D]
    const void* k;       // [Hkv, klen, D]
    int64_t q_head_stride, k_head_stride;  // elements
    int klen, sink, start, m, q_len, G;
    float2* stats;       // [key_splits, Hkv, G*q_len]  partial (m_r, l'_r) of each key slice (l' relative to fl(m*log2e))
    int key_splits;      // pass A: slices of SC_KSPLIT_TILES key tiles
    float* colpart;      // [row_splits, Hkv, m]  per-slice column maxima of the log-softmax
    void* out;           // [Hkv, m] half
    int64_t out_head_stride;
    int row_splits;      // pass B
    int n_kv_heads;
    float c;             // float32(sqrt(D))
    float rcp;           // reciprocal constant r such that half(x*r) == half(x/c) for EVERY 16-bit x (0 = none found)
};

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

// the same chain, result kept as the 16-bit value (maxima are taken on 16-bit values, the exp2 / subtraction
// arguments read it through the mixed-precision fma: no separate conversion back to fp32)
template <typename T, bool FAST>
__device__ static inline T round_chain_h(float acc, float c, float rcp) {
    const T h1 = (T)acc;
    const float d = FAST ? (float)h1 * rcp : (float)h1 / c;
    return (T)d;
}
// maximum of 16 values of T as fp32 (fp16: packed v_pk_max_f16 on pairs)
template <typename T>
__device__ static inline float max16(const T (&hx)[16]) {
    if constexpr (std::is_same<T, _Float16>::value) {
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        h2v m = {hx[0], hx[1]};
#pragma unroll
        for (int i = 2; i < 16; i += 2) m = __builtin_elementwise_max(m, h2v{hx[i], hx[i + 1]});
        return fmaxf((float)m[0], (float)m[1]);
    } else {
        float m = (float)hx[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmaxf(m, (float)hx[i]);
        return m;
    }
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

// ---- staging: one 128-row tile, HBM/L2 -> LDS directly (global_load_lds_dwordx4, no VGPR round trip) ---------
// One wave-instruction writes 1 KiB = 64 lanes x 16 B LINEARLY (wave-uniform base + lane*16).  The XOR swizzle of
// the tile is therefore applied on the SOURCE side: LDS position p of a row receives global chunk p ^ f(row), and
// fragment reads use lds_off(row, chunk) = position chunk ^ f(row) (same involution on both sides).
// rowptr clamps out-of-range rows to a valid row: loads are UNCONDITIONAL; out-of-range rows are neutralised
// downstream (causal limit in pass A, m = +inf statistics in pass B).
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
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
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(buf + ci * 1024), 16, 0, 0);
    }
}

// ---- pass A: per-query-row softmax statistics ------------------------------------------------------
// Online softmax in the exp2 domain: the running pseudo-maximum ml2 = fl(m * log2e) is an fp32 number, every term
// is exp2(fma(x, log2e, -ml2)) (one rounding), and the common factor 2^(m*log2e - ml2) that this introduces
// into l is removed exactly at the end (delta = fma(m, log2e, -ml2)).
template <typename T, int D, bool FAST>
__global__ __launch_bounds__(PA_WAVES * 64, 2) void score_rowstat_kernel(ScoreArgs a) {
    constexpr int NWAVES = PA_WAVES;
    constexpr int SC_COLS = NWAVES * 32;  // stationary query rows per block (32 per wave)
    typedef ScoreCfg<D> C;
    typedef typename Mfma32<T>::v8 v8;
    __shared__ __attribute__((aligned(16))) char lds[2 * C::TILE_BYTES];
    constexpr float L2E = 1.44269504088896340736f;

    const int h = blockIdx.y;
    const int R = a.G * a.q_len;
    const int KT = a.sink + a.m + a.q_len;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    // stationary operand: 32 query rows per wave, one per lane (B operand: col = row index)
    const int r = blockIdx.x * SC_COLS + wave * 32 + l31;
    const bool rvalid = r < R;
    const int rc = min(r, R - 1);  // out-of-range lanes shadow the last row; their result is never stored
    const int g = rc / a.q_len;
    const int qi = rc - g * a.q_len;
    v8 bq[C::KK];
    {
        const char* qp = reinterpret_cast<const char*>(a.q) +
                         (((int64_t)h * a.G + g) * a.q_head_stride + (int64_t)qi * D) * 2 + half * 16;
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk)
            bq[kk] = __builtin_bit_cast(v8, *reinterpret_cast<const u32x4*>(qp + kk * 32));
    }
    // key j (virtual index) is visible to query i iff j <= sink + m + i  (reference score.py:67-85)
    const int limit = a.sink + a.m + qi;

    // block-uniform loop bound: the largest limit of any row in the block; this block owns key tiles [t_lo, t_hi)
    int kend;
    {
        const int r0 = blockIdx.x * SC_COLS;
        const int r1 = min(R - 1, r0 + SC_COLS - 1);
        const int qmax = (r0 / a.q_len == r1 / a.q_len) ? (r1 % a.q_len) : (a.q_len - 1);
        kend = min(KT, a.sink + a.m + qmax + 1);
    }
    const int ntiles = (kend + SC_TILE - 1) / SC_TILE;
    // dispatch order follows blockIdx.z: run the LAST key slices (masked, diagonal tiles) first so that the tail of
    // the launch consists of the cheapest blocks
    const int zslice = (int)gridDim.z - 1 - (int)blockIdx.z;
    const int t_lo = zslice * SC_KSPLIT_TILES;
    const int t_hi = min(ntiles, t_lo + SC_KSPLIT_TILES);

    const char* kh = reinterpret_cast<const char*>(a.k) + (int64_t)h * a.k_head_stride * 2;
    const int off_ctx = a.start - a.sink;                       // virtual -> cache row, ctx segment
    const int off_rep = a.klen - a.q_len - a.sink - a.m;        // virtual -> cache row, repeat segment
    auto keyptr = [&](int kv) -> const char* {
        kv = min(kv, KT - 1);
        const int row = kv + (kv < a.sink ? 0 : (kv < a.sink + a.m ? off_ctx : off_rep));
        return kh + (int64_t)row * C::ROW_BYTES;
    };

    float m_run = -INFINITY, ml2_run = 0.f, l_run = 0.f;
    const int diag0 = a.sink + a.m;  // first key that can be masked for some row

    // A fragments (key rows) of one 32-key block: all LDS reads are issued together, one block AHEAD of their use, so
    // that the matrix chain never waits for LDS latency (sched_barrier pins the order: the compiler would otherwise
    // sink every ds_read next to its MFMA and serialise read latency + MFMA eight times per block)
    auto load_frags = [&](u32x4 (&fr)[C::KK], const char* buf, int kb) {
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk)
            fr[kk] = *reinterpret_cast<const u32x4*>(buf + C::lds_off(kb * 32 + l31, kk * 2 + half));
    };
    // rounding chain + online softmax of one 32-key block whose first key is k0; MASK = per-logit causal test
    auto epilogue = [&](const f16v& acc, int k0, auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        T x[16];
        const int rel = limit - (k0 + 4 * half);  // key offset (i&3)+8*(i>>2) visible iff <= rel
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            T v = round_chain_h<T, FAST>(acc[i], a.c, a.rcp);
            if (MASK) v = ((i & 3) + 8 * (i >> 2) <= rel) ? v : (T)(-INFINITY);
            x[i] = v;
        }
        const float tmax = max16<T>(x);
        if (tmax > m_run) {  // new running maximum: rescale the partial sum
            const float ml2_new = tmax * L2E;
            l_run *= __builtin_amdgcn_exp2f(ml2_run - ml2_new);  // m_run = -inf: l_run is 0 and ml2_run finite
            m_run = tmax;
            ml2_run = ml2_new;
        }
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            ps0 += __builtin_amdgcn_exp2f(__builtin_fmaf((float)x[i], L2E, -ml2_run));
            ps1 += __builtin_amdgcn_exp2f(__builtin_fmaf((float)x[i + 1], L2E, -ml2_run));
        }
        l_run += ps0 + ps1;  // (masked keys: x = -inf -> exp2(-inf) = 0)
    };
    // causal limits of this wave's 32 rows (wave-uniform): a 32-key block is fully visible to the wave if its last key
    // <= wmin, fully masked if its first key > wmax; only the blocks in between need the per-logit test
    int wmin = limit, wmax = limit;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        wmin = min(wmin, __shfl_xor(wmin, o, 64));
        wmax = max(wmax, __shfl_xor(wmax, o, 64));
    }
    wmin = __builtin_amdgcn_readfirstlane(wmin);
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    // one 128-key tile = 4 blocks of 32 keys; DIAG = tile straddles / lies beyond the causal diagonal or the key range
    auto tile_body = [&](const char* buf, int t, auto diag_tag) {
        constexpr bool DIAG = decltype(diag_tag)::value;
        u32x4 fr[C::KK];
        load_frags(fr, buf, 0);
#pragma unroll
        for (int kb = 0; kb < SC_TILE / 32; ++kb) {
            const int k0 = t * SC_TILE + kb * 32;
            if (DIAG && k0 > wmax) {  // nothing of this block is visible to any row of the wave
                if (kb + 1 < SC_TILE / 32) load_frags(fr, buf, kb + 1);
                continue;
            }
            f16v acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) acc = Mfma32<T>::mfma(__builtin_bit_cast(v8, fr[kk]), bq[kk], acc);
            if (kb + 1 < SC_TILE / 32) load_frags(fr, buf, kb + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (DIAG && k0 + 31 > wmin) epilogue(acc, k0, std::true_type{});
            else epilogue(acc, k0, std::false_type{});
        }
    };

    if (t_lo < t_hi) {
        stage_tile<D, NWAVES>(lds, t_lo * SC_TILE, keyptr, wave, lane);
        __syncthreads();
        for (int t = t_lo; t < t_hi; ++t) {
            const int cur = (t - t_lo) & 1;
            const char* buf = lds + cur * C::TILE_BYTES;
            if (t + 1 < t_hi) stage_tile<D, NWAVES>(lds + (cur ^ 1) * C::TILE_BYTES, (t + 1) * SC_TILE, keyptr, wave, lane);
            if ((t * SC_TILE + SC_TILE - 1) > diag0) tile_body(buf, t, std::true_type{});   // also covers kv >= KT
            else tile_body(buf, t, std::false_type{});
            __syncthreads();  // next tile landed (vmcnt drained) and everybody is done reading this one
        }
    }
    // merge the two half-waves (they saw disjoint keys of the same query row)
    const float m_o = __shfl_xor(m_run, 32, 64);
    const float ml2_o = __shfl_xor(ml2_run, 32, 64);
    const float l_o = __shfl_xor(l_run, 32, 64);
    const float M = fmaxf(m_run, m_o);
    const float ML2 = (m_run >= m_o) ? ml2_run : ml2_o;
    const float Lp = l_run * __builtin_amdgcn_exp2f(ml2_run - ML2) + l_o * __builtin_amdgcn_exp2f(ml2_o - ML2);
    // partial statistics of this key slice (empty slice: m = -inf, l' = 0)
    if (half == 0 && rvalid) a.stats[((int64_t)zslice * gridDim.y + h) * R + r] = make_float2(M, Lp);
}

// merge the key slices of pass A:  stats[0] <- (m_r, log l_r).  l'_s is relative to fl(m_s*log2e); the common factor
// 2^(M*log2e - fl(M*log2e)) is removed exactly (delta).
__global__ void score_merge_stats_kernel(float2* __restrict__ stats, int key_splits, int64_t rows_total) {
    constexpr float L2E = 1.44269504088896340736f;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_total) return;
    float M = -INFINITY;
    for (int s = 0; s < key_splits; ++s) M = fmaxf(M, stats[s * rows_total + i].x);
    const float ML2 = M * L2E;
    float Lp = 0.f;
    for (int s = 0; s < key_splits; ++s) {
        const float2 ps = stats[s * rows_total + i];
        Lp += ps.y * __builtin_amdgcn_exp2f(ps.x * L2E - ML2);
    }
    const float delta = __builtin_fmaf(M, L2E, -ML2);
    stats[i] = make_float2(M, logf(Lp) - delta * 0.69314718055994530942f);
}

// ---- pass B: per-ctx-key maximum of the log-softmax over all query rows --------------------------------
template <typename T, int D, bool FAST>
__global__ __launch_bounds__(PB_WAVES * 64, 4) void score_colmax_kernel(ScoreArgs a) {
    constexpr int NWAVES = PB_WAVES;
    constexpr int SC_COLS = NWAVES * 32;  // stationary ctx keys per block (32 per wave)
    typedef ScoreCfg<D> C;
    typedef typename Mfma32<T>::v8 v8;
    __shared__ __attribute__((aligned(16))) char lds[2 * C::TILE_BYTES + 2 * SC_TILE * 8];
    float2* lstat = reinterpret_cast<float2*>(lds + 2 * C::TILE_BYTES);  // [2][128]

    // XCD-aware block order: workgroup b runs on XCD b % 8 and every XCD has its own L2.  The blocks that stream the SAME
    // query-row tiles (same row slice and head, different ctx-key tile) get ids that are congruent mod 8 whenever
    // row_splits*Hkv is a multiple of 8, so each query tile is fetched into ONE L2 instead of eight.
    const int SH = a.row_splits * a.n_kv_heads;
    const int bj = blockIdx.x % SH;            // (row slice, head)
    const int ctile = blockIdx.x / SH;         // ctx-key tile
    const int ysplit = bj % a.row_splits;
    const int h = bj / a.row_splits;
    const int R = a.G * a.q_len;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    // stationary operand: 32 ctx keys per wave (B operand)
    const int j = ctile * SC_COLS + wave * 32 + l31;
    const bool jvalid = j < a.m;
    v8 bk[C::KK];
    {
        const char* kp = reinterpret_cast<const char*>(a.k) +
                         ((int64_t)h * a.k_head_stride + (int64_t)(a.start + (jvalid ? j : 0)) * D) * 2 + half * 16;
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk)
            bk[kk] = __builtin_bit_cast(v8, *reinterpret_cast<const u32x4*>(kp + kk * 32));
    }
    // this block's slice of the query rows (tiles of 128); the host picks row_splits so that no slice is empty
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
    const float2* stats_h = a.stats + (int64_t)h * R;  // merged (m_r, log l_r)
    auto load_stat = [&](int t) -> float2 {
        const int r = t * SC_TILE + (int)(threadIdx.x & (SC_TILE - 1));
        const float2 v = stats_h[min(r, R - 1)];
        // rows beyond R get (m = +inf): x - inf = -inf never wins the max
        return (r < R) ? v : make_float2(INFINITY, 0.f);
    };

    float best = -INFINITY;
    if (t_begin < t_end) {
        stage_tile<D, NWAVES>(lds, t_begin * SC_TILE, rowptr, wave, lane);
        float2 sst = load_stat(t_begin);
        if (threadIdx.x < SC_TILE) lstat[threadIdx.x] = sst;
        __syncthreads();

        for (int t = t_begin; t < t_end; ++t) {
            const int cur = (t - t_begin) & 1;
            const char* buf = lds + cur * C::TILE_BYTES;
            const float2* ls = lstat + cur * SC_TILE;
            if (t + 1 < t_end) {
                stage_tile<D, NWAVES>(lds + (cur ^ 1) * C::TILE_BYTES, (t + 1) * SC_TILE, rowptr, wave, lane);
                sst = load_stat(t + 1);
            }
#pragma unroll 1
            for (int kb = 0; kb < SC_TILE / 32; ++kb) {
                f16v acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int kk = 0; kk < C::KK; ++kk) {
                    const u32x4 raw = *reinterpret_cast<const u32x4*>(buf + C::lds_off(kb * 32 + l31, kk * 2 + half));
                    acc = Mfma32<T>::mfma(__builtin_bit_cast(v8, raw), bk[kk], acc);
                }
                float b0 = -INFINITY, b1 = -INFINITY;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const int rr = kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                    const float2 s0 = ls[rr], s1 = ls[rr + 1];
                    b0 = fmaxf(b0, ((float)round_chain_h<T, FAST>(acc[i], a.c, a.rcp) - s0.x) - s0.y);
                    b1 = fmaxf(b1, ((float)round_chain_h<T, FAST>(acc[i + 1], a.c, a.rcp) - s1.x) - s1.y);
                }
                best = fmaxf(best, fmaxf(b0, b1));
            }
            if (t + 1 < t_end && threadIdx.x < SC_TILE) lstat[(cur ^ 1) * SC_TILE + threadIdx.x] = sst;
            __syncthreads();
        }
    }
    best = fmaxf(best, __shfl_xor(best, 32, 64));
    if (half == 0 && jvalid) a.colpart[((int64_t)ysplit * a.n_kv_heads + h) * a.m + j] = best;
}

template <typename T>
__global__ void score_finalize_kernel(const float* __restrict__ colpart, int splits, int Hkv, int m, T* __restrict__ out,
                                      int64_t out_head_stride) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (j >= m) return;
    float t = -INFINITY;
    for (int s = 0; s < splits; ++s) t = fmaxf(t, colpart[((int64_t)s * Hkv + h) * m + j]);
    out[(int64_t)h * out_head_stride + j] = (T)expf(t);
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// key slices of pass A
static inline int score_key_splits(int sink, int m, int q_len) {
    const int ntiles = (sink + m + q_len + SC_TILE - 1) / SC_TILE;
    return (ntiles + SC_KSPLIT_TILES - 1) / SC_KSPLIT_TILES;
}
// number of row slices of pass B: enough blocks to fill 256 CUs about twice, no empty slice
static inline int score_row_splits(int Hkv, int G, int q_len, int m) {
    const int ctiles = (m + PB_WAVES * 32 - 1) / (PB_WAVES * 32);
    const int rtiles = (G * q_len + SC_TILE - 1) / SC_TILE;
    int splits = 512 / (ctiles * Hkv);  // 256 CUs x 2 resident blocks: exactly one round when the shape allows
    if (splits > rtiles) splits = rtiles;
    if (splits < 1) splits = 1;
    const int per = (rtiles + splits - 1) / splits;
    return (rtiles + per - 1) / per;
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
static float find_exact_reciprocal(float c, int dtype) {
    static float cache[2][3] = {{0, 0, 0}, {0, 0, 0}};  // [dtype][{c, r, valid}]
    if (cache[dtype][2] != 0.f && cache[dtype][0] == c) return cache[dtype][1];
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
    cache[dtype][0] = c; cache[dtype][1] = found; cache[dtype][2] = 1.f;
    return found;
}

template <typename T, int D, bool FAST>
static int launch_score_impl(ScoreArgs a, int Hkv, hipStream_t stream) {
    const int R = a.G * a.q_len;
    {
        ProfScope ps("score_rowstat", stream);
        hipLaunchKernelGGL((score_rowstat_kernel<T, D, FAST>), dim3((R + PA_WAVES * 32 - 1) / (PA_WAVES * 32), Hkv, a.key_splits),
                           dim3(PA_WAVES * 64), 0, stream, a);
    }
    KVZ_CHECK_LAUNCH("score_rowstat_kernel");
    {
        const int64_t rows_total = (int64_t)Hkv * R;
        hipLaunchKernelGGL(score_merge_stats_kernel, dim3((unsigned)((rows_total + 255) / 256)), dim3(256), 0, stream, a.stats,
                           a.key_splits, rows_total);
    }
    KVZ_CHECK_LAUNCH("score_merge_stats_kernel");
    const int ctiles = (a.m + PB_WAVES * 32 - 1) / (PB_WAVES * 32);
    a.row_splits = score_row_splits(Hkv, a.G, a.q_len, a.m);
    a.n_kv_heads = Hkv;
    {
        ProfScope ps("score_colmax", stream);
        hipLaunchKernelGGL((score_colmax_kernel<T, D, FAST>), dim3(ctiles * a.row_splits * Hkv), dim3(PB_WAVES * 64), 0, stream, a);
    }
    KVZ_CHECK_LAUNCH("score_colmax_kernel");
    hipLaunchKernelGGL((score_finalize_kernel<T>), dim3((a.m + 255) / 256, Hkv), dim3(256), 0, stream, a.colpart,
                       a.row_splits, Hkv, a.m, reinterpret_cast<T*>(a.out), a.out_head_stride);
    KVZ_CHECK_LAUNCH("score_finalize_kernel");
    return KVZ_OK;
}
template <typename T, int D>
static int launch_score(ScoreArgs a, int Hkv, hipStream_t stream) {
    return a.rcp != 0.f ? launch_score_impl<T, D, true>(a, Hkv, stream) : launch_score_impl<T, D, false>(a, Hkv, stream);
}

}  // namespace kvz

using namespace kvz;


static inline size_t score_stats_bytes(int Hkv, int G, int q_len, int m, int sink) {
    return align256((size_t)score_key_splits(sink, m, q_len) * Hkv * G * q_len * sizeof(float2));
}

extern "C" size_t kvz_score_workspace_bytes(int Hkv, int G, int q_len, int m, int sink) {
    if (Hkv <= 0 || G <= 0 || q_len <= 0 || m <= 0 || sink < 0) return 0;
    return score_stats_bytes(Hkv, G, q_len, m, sink) +
           align256((size_t)score_row_splits(Hkv, G, q_len, m) * Hkv * m * sizeof(float));
}

extern "C" int kvz_score_chunk(const void* q, int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen,
                               int sink, int start, int end, int q_len, int Hkv, int G, int D, int dtype, void* out,
                               int64_t out_head_stride, void* ws, size_t ws_bytes, kvz_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    KVZ_REQUIRE(q && k && out && ws, KVZ_EINVAL, "kvz_score_chunk: null pointer");
    KVZ_REQUIRE(Hkv > 0 && Hkv <= 65535 && G > 0 && q_len > 0, KVZ_EINVAL, "kvz_score_chunk: bad shape");
    KVZ_REQUIRE(D == 64 || D == 128, KVZ_EUNSUPPORTED, "kvz_score_chunk: head_dim %d unsupported (64 or 128)", D);
    KVZ_REQUIRE(dtype == KVZ_F16 || dtype == KVZ_BF16, KVZ_EINVAL, "kvz_score_chunk: bad dtype %d", dtype);
    const int m = end - start;
    KVZ_REQUIRE(sink >= 0 && start >= sink && m > 0 && end <= klen - q_len, KVZ_EINVAL,
                "kvz_score_chunk: bad window sink=%d start=%d end=%d klen=%d q_len=%d", sink, start, end, klen, q_len);
    KVZ_REQUIRE(aligned16(q) && aligned16(k), KVZ_EINVAL, "kvz_score_chunk: q/k must be 16-byte aligned");
    KVZ_REQUIRE((q_head_stride * 2) % 16 == 0 && (k_head_stride * 2) % 16 == 0, KVZ_EINVAL,
                "kvz_score_chunk: head strides must be multiples of 8 elements");
    KVZ_REQUIRE(ws_bytes >= kvz_score_workspace_bytes(Hkv, G, q_len, m, sink), KVZ_EWORKSPACE,
                "kvz_score_chunk: workspace too small");
    ScoreArgs a{};
    a.q = q; a.k = k; a.q_head_stride = q_head_stride; a.k_head_stride = k_head_stride;
    a.klen = klen; a.sink = sink; a.start = start; a.m = m; a.q_len = q_len; a.G = G;
    a.stats = reinterpret_cast<float2*>(ws);
    a.key_splits = score_key_splits(sink, m, q_len);
    a.colpart = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + score_stats_bytes(Hkv, G, q_len, m, sink));
    a.out = out; a.out_head_stride = out_head_stride;
    a.c = sqrtf((float)D);  // == float32(math.sqrt(D)) for D in {64, 128}
    a.rcp = find_exact_reciprocal(a.c, dtype);
    if (dtype == KVZ_F16) {
        if (D == 128) return launch_score<_Float16, 128>(a, Hkv, stream);
        return launch_score<_Float16, 64>(a, Hkv, stream);
    }
    if (D == 128) return launch_score<__bf16, 128>(a, Hkv, stream);
    return launch_score<__bf16, 64>(a, Hkv, stream);
}

// test hook: the rounding chain on raw 16-bit patterns (exhaustive-check of the exact-reciprocal path on the device)
namespace kvz {
template <typename T, bool FAST>
__global__ void chain_probe_kernel(const uint16_t* in, int n, float c, float rcp, uint16_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T x;
    uint16_t b = in[i];
    __builtin_memcpy(&x, &b, 2);
    // acc = float(x) is exactly representable, so half(acc) == x: the probe isolates the division step
    const float r = round_chain<T, FAST>((float)x, c, rcp);
    const T h = (T)r;
    __builtin_memcpy(&b, &h, 2);
    out[i] = b;
}
}  // namespace kvz
extern "C" int kvz_debug_round_chain(const void* in_bits, int n, int D, int dtype, int force_division, void* out_bits,
                                     float* rcp_used, kvz_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    KVZ_REQUIRE(in_bits && out_bits && n > 0, KVZ_EINVAL, "kvz_debug_round_chain: bad arguments");
    const float c = sqrtf((float)D);
    const float rcp = force_division ? 0.f : find_exact_reciprocal(c, dtype);
    if (rcp_used) *rcp_used = rcp;
    dim3 grid((n + 255) / 256), block(256);
    const uint16_t* in = reinterpret_cast<const uint16_t*>(in_bits);
    uint16_t* out = reinterpret_cast<uint16_t*>(out_bits);
    if (dtype == KVZ_F16) {
        if (rcp != 0.f) hipLaunchKernelGGL((chain_probe_kernel<_Float16, true>), grid, block, 0, stream, in, n, c, rcp, out);
        else hipLaunchKernelGGL((chain_probe_kernel<_Float16, false>), grid, block, 0, stream, in, n, c, rcp, out);
    } else {
        if (rcp != 0.f) hipLaunchKernelGGL((chain_probe_kernel<__bf16, true>), grid, block, 0, stream, in, n, c, rcp, out);
        else hipLaunchKernelGGL((chain_probe_kernel<__bf16, false>), grid, block, 0, stream, in, n, c, rcp, out);
    }
    KVZ_CHECK_LAUNCH("chain_probe_kernel");
    return KVZ_OK;
}
