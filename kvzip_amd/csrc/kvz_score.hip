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

constexpr int SC_THREADS = 256;
constexpr int SC_TILE = 128;   // streamed rows per LDS tile
constexpr int SC_COLS = 128;   // stationary columns per block (32 per wave)

struct ScoreArgs {
    const void* q;       // [Hkv*G, q_len, D]
    const void* k;       // [Hkv, klen, D]
    int64_t q_head_stride, k_head_stride;  // elements
    int klen, sink, start, m, q_len, G;
    float2* stats;       // [Hkv, G*q_len]  (m_r, log l_r)
    int32_t* colmax;     // [Hkv, m]  order-encoded float
    void* out;           // [Hkv, m] half
    int64_t out_head_stride;
    int row_splits;      // pass B
    float inv_c, c;      // float(sqrt(D)) and its reciprocal
};

// reference rounding chain (attention/score.py:57): half(matmul) / sqrt(D) -> half
template <typename T>
__device__ static inline float round_chain(float acc, float c) {
    const T h1 = (T)acc;
    const float d = (float)h1 / c;  // IEEE fp32 division (no fast-math)
    const T h2 = (T)d;
    return (float)h2;
}

// order-preserving float <-> int encoding for atomicMax
__device__ static inline int32_t f2ord(float f) {
    int32_t i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ static inline float ord2f(int32_t i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

template <int D> struct ScoreCfg {
    static constexpr int ROW_BYTES = D * 2;
    static constexpr int CPR = ROW_BYTES / 16;                       // 16-byte chunks per row
    static constexpr int KK = D / 16;                                // MFMA k-steps
    static constexpr int TILE_BYTES = SC_TILE * ROW_BYTES;
    static constexpr int LOADS = SC_TILE * CPR / SC_THREADS;         // 16-byte loads per thread per tile
    // swizzled byte offset of 16-byte chunk `chunk` of tile row `row`
    __device__ static inline int lds_off(int row, int chunk) {
        if (D == 128) return row * ROW_BYTES + ((chunk ^ (row & 15)) << 4);
        return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4);
    }
};

// ---- staging: one 128-row tile, global -> registers -> swizzled LDS --------------------------------
template <int D, typename RowPtr>
__device__ static inline void stage_load(u32x4 (&regs)[ScoreCfg<D>::LOADS], int row0, RowPtr rowptr) {
    typedef ScoreCfg<D> C;
#pragma unroll
    for (int it = 0; it < C::LOADS; ++it) {
        const int c = it * SC_THREADS + threadIdx.x;
        const int row = c / C::CPR, chunk = c % C::CPR;
        const char* p = rowptr(row0 + row);
        regs[it] = p ? *reinterpret_cast<const u32x4*>(p + chunk * 16) : u32x4{0, 0, 0, 0};
    }
}
template <int D>
__device__ static inline void stage_store(const u32x4 (&regs)[ScoreCfg<D>::LOADS], char* buf) {
    typedef ScoreCfg<D> C;
#pragma unroll
    for (int it = 0; it < C::LOADS; ++it) {
        const int c = it * SC_THREADS + threadIdx.x;
        const int row = c / C::CPR, chunk = c % C::CPR;
        *reinterpret_cast<u32x4*>(buf + C::lds_off(row, chunk)) = regs[it];
    }
}

// ---- pass A: per-query-row softmax statistics ------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(SC_THREADS, 2) void score_rowstat_kernel(ScoreArgs a) {
    typedef ScoreCfg<D> C;
    typedef typename Mfma32<T>::v8 v8;
    __shared__ __attribute__((aligned(16))) char lds[2 * C::TILE_BYTES];

    const int h = blockIdx.y;
    const int R = a.G * a.q_len;
    const int KT = a.sink + a.m + a.q_len;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    // stationary operand: 32 query rows per wave, one per lane (B operand: col = row index)
    const int r = blockIdx.x * SC_COLS + wave * 32 + l31;
    const bool rvalid = r < R;
    const int g = rvalid ? r / a.q_len : 0;
    const int qi = rvalid ? r - g * a.q_len : 0;
    v8 bq[C::KK];
    {
        const char* qp = reinterpret_cast<const char*>(a.q) +
                         (((int64_t)h * a.G + g) * a.q_head_stride + (int64_t)qi * D) * 2 + half * 16;
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
            u32x4 raw = rvalid ? *reinterpret_cast<const u32x4*>(qp + kk * 32) : u32x4{0, 0, 0, 0};
            bq[kk] = __builtin_bit_cast(v8, raw);
        }
    }
    // key j (virtual index) is visible to query i iff j <= sink + m + i  (reference score.py:67-85)
    const int limit = rvalid ? a.sink + a.m + qi : -1;

    // block-uniform loop bound: the largest limit of any row in the block
    int kend;
    {
        const int r0 = blockIdx.x * SC_COLS;
        const int r1 = min(R - 1, r0 + SC_COLS - 1);
        const int qmax = (r0 / a.q_len == r1 / a.q_len) ? (r1 % a.q_len) : (a.q_len - 1);
        kend = min(KT, a.sink + a.m + qmax + 1);
    }
    const int ntiles = (kend + SC_TILE - 1) / SC_TILE;

    const char* kh = reinterpret_cast<const char*>(a.k) + (int64_t)h * a.k_head_stride * 2;
    auto keyptr = [&](int kv) -> const char* {
        if (kv >= KT) return nullptr;
        int row;
        if (kv < a.sink) row = kv;
        else if (kv < a.sink + a.m) row = a.start + (kv - a.sink);
        else row = a.klen - a.q_len + (kv - a.sink - a.m);
        return kh + (int64_t)row * C::ROW_BYTES;
    };

    u32x4 st[C::LOADS];
    stage_load<D>(st, 0, keyptr);
    stage_store<D>(st, lds);
    __syncthreads();

    float m_run = -INFINITY, l_run = 0.f;
    const int diag0 = a.sink + a.m;  // first key that can be masked for some row

    for (int t = 0; t < ntiles; ++t) {
        const char* buf = lds + (t & 1) * C::TILE_BYTES;
        if (t + 1 < ntiles) stage_load<D>(st, (t + 1) * SC_TILE, keyptr);
        const bool need_mask = (t * SC_TILE + SC_TILE - 1) > diag0;  // also covers kv >= KT
#pragma unroll
        for (int kb = 0; kb < SC_TILE / 32; ++kb) {
            f16v acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(buf + C::lds_off(kb * 32 + l31, kk * 2 + half));
                acc = Mfma32<T>::mfma(__builtin_bit_cast(v8, raw), bq[kk], acc);
            }
            float x[16];
            float tmax = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = round_chain<T>(acc[i], a.c);
                if (need_mask) {
                    const int kv = t * SC_TILE + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                    v = (kv <= limit) ? v : -INFINITY;
                }
                x[i] = v;
                tmax = fmaxf(tmax, v);
            }
            const float m_new = fmaxf(m_run, tmax);
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) psum += __expf(x[i] - m_safe);
            l_run = l_run * __expf(m_run - m_safe) + psum;
            m_run = m_new;
        }
        if (t + 1 < ntiles) stage_store<D>(st, lds + ((t + 1) & 1) * C::TILE_BYTES);
        __syncthreads();
    }
    // merge the two half-waves (they saw disjoint keys of the same query row)
    const float m_o = __shfl_xor(m_run, 32, 64);
    const float l_o = __shfl_xor(l_run, 32, 64);
    const float M = fmaxf(m_run, m_o);
    const float Ms = (M == -INFINITY) ? 0.f : M;
    const float L = l_run * __expf(m_run - Ms) + l_o * __expf(m_o - Ms);
    if (half == 0 && rvalid) a.stats[(int64_t)h * R + r] = make_float2(M, logf(L));
}

// ---- pass B: per-ctx-key maximum of the log-softmax over all query rows --------------------------------
template <typename T, int D>
__global__ __launch_bounds__(SC_THREADS, 2) void score_colmax_kernel(ScoreArgs a) {
    typedef ScoreCfg<D> C;
    typedef typename Mfma32<T>::v8 v8;
    __shared__ __attribute__((aligned(16))) char lds[2 * C::TILE_BYTES + 2 * SC_TILE * 8];
    float2* lstat = reinterpret_cast<float2*>(lds + 2 * C::TILE_BYTES);  // [2][128]

    const int h = blockIdx.z;
    const int R = a.G * a.q_len;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    // stationary operand: 32 ctx keys per wave (B operand)
    const int j = blockIdx.x * SC_COLS + wave * 32 + l31;
    const bool jvalid = j < a.m;
    v8 bk[C::KK];
    {
        const char* kp = reinterpret_cast<const char*>(a.k) +
                         ((int64_t)h * a.k_head_stride + (int64_t)(a.start + (jvalid ? j : 0)) * D) * 2 + half * 16;
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
            u32x4 raw = jvalid ? *reinterpret_cast<const u32x4*>(kp + kk * 32) : u32x4{0, 0, 0, 0};
            bk[kk] = __builtin_bit_cast(v8, raw);
        }
    }
    // this block's slice of the query rows (tiles of 128)
    const int total_tiles = (R + SC_TILE - 1) / SC_TILE;
    const int per = (total_tiles + a.row_splits - 1) / a.row_splits;
    const int t_begin = blockIdx.y * per;
    const int t_end = min(total_tiles, t_begin + per);
    if (t_begin >= t_end) return;

    const char* qbase = reinterpret_cast<const char*>(a.q) + (int64_t)h * a.G * a.q_head_stride * 2;
    auto rowptr = [&](int r) -> const char* {
        if (r >= R) return nullptr;
        const int g = r / a.q_len;
        const int qi = r - g * a.q_len;
        return qbase + ((int64_t)g * a.q_head_stride + (int64_t)qi * D) * 2;
    };
    const float2* stats_h = a.stats + (int64_t)h * R;
    auto load_stat = [&](int t) -> float2 {
        const int r = t * SC_TILE + (int)threadIdx.x;
        // rows beyond R get (m = +inf): x - inf = -inf never wins the max
        return (threadIdx.x < SC_TILE) ? ((r < R) ? stats_h[r] : make_float2(INFINITY, 0.f)) : make_float2(0.f, 0.f);
    };

    u32x4 st[C::LOADS];
    stage_load<D>(st, t_begin * SC_TILE, rowptr);
    float2 sst = load_stat(t_begin);
    stage_store<D>(st, lds);
    if (threadIdx.x < SC_TILE) lstat[threadIdx.x] = sst;
    __syncthreads();

    float best = -INFINITY;
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        const char* buf = lds + cur * C::TILE_BYTES;
        const float2* ls = lstat + cur * SC_TILE;
        if (t + 1 < t_end) {
            stage_load<D>(st, (t + 1) * SC_TILE, rowptr);
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
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float2 s = ls[kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * half];
                const float v = round_chain<T>(acc[i], a.c);
                best = fmaxf(best, (v - s.x) - s.y);
            }
        }
        if (t + 1 < t_end) {
            stage_store<D>(st, lds + (cur ^ 1) * C::TILE_BYTES);
            if (threadIdx.x < SC_TILE) lstat[(cur ^ 1) * SC_TILE + threadIdx.x] = sst;
        }
        __syncthreads();
    }
    best = fmaxf(best, __shfl_xor(best, 32, 64));
    if (half == 0 && jvalid) atomicMax(&a.colmax[(int64_t)h * a.m + j], f2ord(best));
}

__global__ void score_init_kernel(int32_t* colmax, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) colmax[i] = f2ord(-INFINITY);
}

template <typename T>
__global__ void score_finalize_kernel(const int32_t* __restrict__ colmax, int m, T* __restrict__ out,
                                      int64_t out_head_stride) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (j >= m) return;
    const float t = ord2f(colmax[(int64_t)h * m + j]);
    out[(int64_t)h * out_head_stride + j] = (T)expf(t);
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename T, int D>
static int launch_score(ScoreArgs a, int Hkv, hipStream_t stream) {
    const int R = a.G * a.q_len;
    const int64_t ncol = (int64_t)Hkv * a.m;
    hipLaunchKernelGGL(score_init_kernel, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, stream, a.colmax, ncol);
    KVZ_CHECK_LAUNCH("score_init_kernel");
    hipLaunchKernelGGL((score_rowstat_kernel<T, D>), dim3((R + SC_COLS - 1) / SC_COLS, Hkv), dim3(SC_THREADS), 0, stream, a);
    KVZ_CHECK_LAUNCH("score_rowstat_kernel");
    const int ctiles = (a.m + SC_COLS - 1) / SC_COLS;
    const int rtiles = (R + SC_TILE - 1) / SC_TILE;
    // enough blocks to fill 256 CUs twice over
    int splits = (1024 + ctiles * Hkv - 1) / (ctiles * Hkv);
    if (splits > rtiles) splits = rtiles;
    if (splits < 1) splits = 1;
    a.row_splits = splits;
    hipLaunchKernelGGL((score_colmax_kernel<T, D>), dim3(ctiles, splits, Hkv), dim3(SC_THREADS), 0, stream, a);
    KVZ_CHECK_LAUNCH("score_colmax_kernel");
    hipLaunchKernelGGL((score_finalize_kernel<T>), dim3((a.m + 255) / 256, Hkv), dim3(256), 0, stream, a.colmax, a.m,
                       reinterpret_cast<T*>(a.out), a.out_head_stride);
    KVZ_CHECK_LAUNCH("score_finalize_kernel");
    return KVZ_OK;
}

}  // namespace kvz

using namespace kvz;

extern "C" size_t kvz_score_workspace_bytes(int Hkv, int G, int q_len, int m) {
    if (Hkv <= 0 || G <= 0 || q_len <= 0 || m <= 0) return 0;
    return align256((size_t)Hkv * G * q_len * sizeof(float2)) + align256((size_t)Hkv * m * sizeof(int32_t));
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
    KVZ_REQUIRE(ws_bytes >= kvz_score_workspace_bytes(Hkv, G, q_len, m), KVZ_EWORKSPACE,
                "kvz_score_chunk: workspace too small");
    ScoreArgs a{};
    a.q = q; a.k = k; a.q_head_stride = q_head_stride; a.k_head_stride = k_head_stride;
    a.klen = klen; a.sink = sink; a.start = start; a.m = m; a.q_len = q_len; a.G = G;
    a.stats = reinterpret_cast<float2*>(ws);
    a.colmax = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + align256((size_t)Hkv * G * q_len * sizeof(float2)));
    a.out = out; a.out_head_stride = out_head_stride;
    a.c = sqrtf((float)D);  // float32(math.sqrt(D)): sqrt of 64/128 rounds identically in float and double->float
    a.inv_c = 1.0f / a.c;
    if (dtype == KVZ_F16) {
        if (D == 128) return launch_score<_Float16, 128>(a, Hkv, stream);
        return launch_score<_Float16, 64>(a, Hkv, stream);
    }
    if (D == 128) return launch_score<__bf16, 128>(a, Hkv, stream);
    return launch_score<__bf16, 64>(a, Hkv, stream);
}
