// kvz_flash.hip — multi-row (q_len > 1) causal GQA attention for gfx950: one kernel for
//   * the FIRST generation step on a pruned cache (q_len = len(query), reference model/wrapper.py:271-274 ->
//     attention/attn.py:56-73, flash_attn_varlen_func with max_seqlen_q > 1): ragged per-head key segments, and
//   * the dense pre-prune forward of prefill and of the scoring pass (reference attention/attn.py:75-89, flash-attn's dense
//     kernel): every head owns `klen` rows of the dense cache.
// Both are "for every KV head h: R = q_len*G query rows against len_h keys, causal mask aligned to the bottom-right corner".
// Unlike the decode kernel (kvz_attn.hip: 16 rows per block, keys split over blocks) a block here owns 128 query rows of one
// head (8 waves x 16 rows) and walks the head's keys ONCE in 64-key tiles that all waves share through LDS: K/V traffic is
// len/128 per row instead of len/16.  S^T = K.Q^T and O^T += V^T.P^T on v_mfma_f32_16x16x32 (K tile XOR-swizzled for
// conflict-free ds_read_b128 A fragments, V tile padded for ds_read_b64_tr_b16), online softmax with one query row per lane
// column, next tile prefetched into registers while the current one is computed.  Optionally emits the row log-sum-exp.
// Query / output addressing is by strides (head, group member, position), so the [Hkv*q_len, G, D] layout of the varlen call
// and the [H, q_len, D] layout of the dense forward need no re-layout copy.
#include "kvz_common.h"

namespace kvz {

typedef _Float16 fh8 __attribute__((ext_vector_type(8)));
typedef __bf16 fb8 __attribute__((ext_vector_type(8)));
typedef short fs4 __attribute__((ext_vector_type(4)));
typedef short fs8 __attribute__((ext_vector_type(8)));
typedef float ff4 __attribute__((ext_vector_type(4)));
typedef uint32_t fu4 __attribute__((ext_vector_type(4)));

template <typename T> struct FlashTraits;
template <> struct FlashTraits<_Float16> {
    typedef fh8 v8;
    __device__ static inline ff4 mfma(v8 a, v8 b, ff4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct FlashTraits<__bf16> {
    typedef fb8 v8;
    __device__ static inline ff4 mfma(v8 a, v8 b, ff4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

constexpr int FL_THREADS = 512, FL_WAVES = 8;
constexpr int FL_ROWS = FL_WAVES * 16;  // query rows per block
constexpr int FL_KT = 64;               // keys per tile
constexpr int FL_MAXH = 64;

struct FlashArgs {
    const void* q;
    const void* k;
    const void* v;
    void* out;
    float* lse;                       // optional [Hkv, R] natural-log LSE of the scaled logits
    const int32_t* k_start;           // device arrays (used when n_meta == 0)
    const int32_t* k_len;
    int32_t m_start[FL_MAXH], m_len[FL_MAXH];  // host copies (n_meta = Hkv)
    int n_meta;
    int k_len_offset;
    int64_t q_sh, q_sg, q_si;         // element strides of (head, group member, position) in q
    int64_t o_sh, o_sg, o_si;         // ... and in out
    int Hkv, G, q_len, causal;
    float scale;
    int ns;                           // key splits (blockIdx.z); > 1: unnormalised partials go to part_o / part_ml
    float* part_o;                    // [Hkv*R, ns, D]
    float* part_ml;                   // [Hkv*R, ns, 2]  (running max in the exp2 domain, sum)
};

template <typename T, int D>
__global__ __launch_bounds__(FL_THREADS) void flash_fwd_kernel(FlashArgs a) {
    typedef typename FlashTraits<T>::v8 v8;
    constexpr int ROW_BYTES = D * 2, CPR = ROW_BYTES / 16, KK = D / 32, DB = D / 16;
    constexpr int VSTRIDE = ROW_BYTES + 32;                       // padded V rows: conflict-free transpose reads
    constexpr int K_BYTES = FL_KT * ROW_BYTES, V_BYTES = FL_KT * VSTRIDE;
    constexpr int LOADS = FL_KT * CPR / FL_THREADS;               // 16-byte loads per thread and tile (K and V each)
    static_assert(LOADS >= 1 && (FL_KT * CPR) % FL_THREADS == 0, "tile / thread layout");
    __shared__ __attribute__((aligned(16))) char lds[2 * (K_BYTES + V_BYTES)];

    const int h = blockIdx.y, rt = blockIdx.x;
    const int R = a.q_len * a.G;
    const int len = (a.n_meta ? a.m_len[h] : a.k_len[h]) + a.k_len_offset;
    const int64_t seg = a.n_meta ? a.m_start[h] : a.k_start[h];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, quad = lane >> 4;

    // this lane's query row (column of S^T): r = i*G + g inside head h
    const int r = rt * FL_ROWS + wave * 16 + l15;
    const bool rvalid = r < R;
    const int rc = rvalid ? r : R - 1;
    const int qi = rc / a.G, qg = rc - qi * a.G;
    v8 qf[KK];
    {
        const T* qp = reinterpret_cast<const T*>(a.q) + h * a.q_sh + qg * a.q_sg + qi * a.q_si + quad * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) qf[kk] = __builtin_bit_cast(v8, *reinterpret_cast<const fu4*>(qp + kk * 32));
    }
    int limit = a.causal ? qi + len - a.q_len : len - 1;  // last visible key (bottom-right aligned mask)
    if (!rvalid) limit = -1;
    // keys this block has to walk: up to the limit of its last row
    int blk_limit;
    {
        const int rl = min(R - 1, rt * FL_ROWS + FL_ROWS - 1);
        blk_limit = a.causal ? rl / a.G + len - a.q_len : len - 1;
        blk_limit = min(blk_limit, len - 1);
    }
    int n_tiles = blk_limit >= 0 ? blk_limit / FL_KT + 1 : 0;
    int t_first = 0;
    if (a.ns > 1) {  // this block's share of the key tiles
        const int per = (n_tiles + a.ns - 1) / a.ns;
        t_first = min(n_tiles, (int)blockIdx.z * per);
        n_tiles = min(n_tiles, t_first + per);
    }

    const char* kbase = reinterpret_cast<const char*>(a.k) + seg * ROW_BYTES;
    const char* vbase = reinterpret_cast<const char*>(a.v) + seg * ROW_BYTES;
    fu4 kreg[LOADS], vreg[LOADS];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int c = i * FL_THREADS + threadIdx.x;
            int key = t * FL_KT + c / CPR;
            key = key < len ? key : len - 1;  // clamp (masked below)
            kreg[i] = *reinterpret_cast<const fu4*>(kbase + (int64_t)key * ROW_BYTES + (c % CPR) * 16);
            vreg[i] = *reinterpret_cast<const fu4*>(vbase + (int64_t)key * ROW_BYTES + (c % CPR) * 16);
        }
    };
    auto store_tile = [&](int b) {
        char* kl = lds + b * (K_BYTES + V_BYTES);
        char* vl = kl + K_BYTES;
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int c = i * FL_THREADS + threadIdx.x;
            const int row = c / CPR, ch = c % CPR;
            *reinterpret_cast<fu4*>(kl + row * ROW_BYTES + ((ch ^ (row & (CPR - 1) & 15)) << 4)) = kreg[i];
            *reinterpret_cast<fu4*>(vl + row * VSTRIDE + ch * 16) = vreg[i];
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    ff4 o[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) o[i] = ff4{0.f, 0.f, 0.f, 0.f};
    const float sl2 = a.scale * 1.44269504088896340736f;  // exp2 domain

    if (t_first < n_tiles) {
        load_tile(t_first);
        store_tile(t_first & 1);
    }
    __syncthreads();
    for (int t = t_first; t < n_tiles; ++t) {
        const int b = t & 1;
        if (t + 1 < n_tiles) load_tile(t + 1);  // in flight while this tile is computed
        const char* kl = lds + b * (K_BYTES + V_BYTES);
        const char* vl = kl + K_BYTES;
        const int t0 = t * FL_KT;
        // ---- S^T = K.Q^T: 4 sub-tiles of 16 keys; rows = keys (quad*4 + reg), col = query row ----
        ff4 s[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            s[sub] = ff4{0.f, 0.f, 0.f, 0.f};
            const int row = sub * 16 + l15;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const fu4 kf = *reinterpret_cast<const fu4*>(kl + row * ROW_BYTES + (((kk * 4 + quad) ^ (row & (CPR - 1) & 15)) << 4));
                s[sub] = FlashTraits<T>::mfma(__builtin_bit_cast(v8, kf), qf[kk], s[sub]);
            }
        }
        // ---- online softmax (exp2 domain), one query row per lane column ----
        float sv[16];
        float tmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = t0 + (j >> 2) * 16 + quad * 4 + (j & 3);
            float x = s[j >> 2][j & 3] * sl2;
            x = (key <= limit) ? x : -INFINITY;
            sv[j] = x;
            tmax = fmaxf(tmax, x);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = exp2f(m_run - m_safe);
        float psum = 0.f;
        v8 pb[2];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float p = exp2f(sv[j] - m_safe);
            psum += p;
            pb[j >> 3][j & 7] = (T)p;
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < DB; ++i) {
            o[i][0] *= alpha; o[i][1] *= alpha; o[i][2] *= alpha; o[i][3] *= alpha;
        }
        // ---- O^T += V^T.P^T: two groups of 32 keys; A = V^T by transposed LDS reads, B = P^T in registers ----
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const char* trp = vl + (g2 * 32 + quad * 4 + (l15 >> 2)) * VSTRIDE + (l15 & 3) * 8;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                fs4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fs4*)(trp + db * 32));
                fs4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fs4*)(trp + 16 * VSTRIDE + db * 32));
                fs8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o[db] = FlashTraits<T>::mfma(__builtin_bit_cast(v8, both), pb[g2], o[db]);
            }
        }
        if (t + 1 < n_tiles) store_tile(b ^ 1);  // (nobody reads that buffer: the barrier of the previous trip is behind us)
        __syncthreads();
    }
    // ---- normalise and store: lane holds O^T[d = db*16 + quad*4 + reg][row l15] ----
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (a.ns > 1) {
        if (rvalid) {
            const int64_t slot = ((int64_t)h * R + r) * a.ns + blockIdx.z;
            float* po = a.part_o + slot * D + quad * 4;
#pragma unroll
            for (int db = 0; db < DB; ++db) *reinterpret_cast<ff4*>(po + db * 16) = o[db];
            if (quad == 0) *reinterpret_cast<float2*>(a.part_ml + slot * 2) = make_float2(m_run, l_run);
        }
        return;
    }
    if (rvalid) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        T* op = reinterpret_cast<T*>(a.out) + h * a.o_sh + qg * a.o_sg + qi * a.o_si + quad * 4;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            T w[4] = {(T)(o[db][0] * inv), (T)(o[db][1] * inv), (T)(o[db][2] * inv), (T)(o[db][3] * inv)};
            *reinterpret_cast<uint2*>(op + db * 16) = *reinterpret_cast<const uint2*>(w);
        }
        if (a.lse && quad == 0)
            a.lse[(int64_t)h * R + r] = (l_run > 0.f) ? (m_run + log2f(l_run)) * 0.69314718055994530942f : -INFINITY;
    }
}

// merge of the key splits: one thread per (row, 4 output dims)
template <typename T, int D>
__global__ __launch_bounds__(256) void flash_combine_kernel(FlashArgs a) {
    constexpr int TPR = D / 4;  // threads per row
    const int R = a.q_len * a.G;
    const int64_t row = (int64_t)blockIdx.x * (256 / TPR) + threadIdx.x / TPR;
    if (row >= (int64_t)a.Hkv * R) return;
    const int c = threadIdx.x % TPR;
    const float2* ml = reinterpret_cast<const float2*>(a.part_ml) + row * a.ns;
    float M = -INFINITY;
    for (int s = 0; s < a.ns; ++s) M = fmaxf(M, ml[s].x);
    const float Ms = (M == -INFINITY) ? 0.f : M;
    float L = 0.f;
    ff4 acc = ff4{0.f, 0.f, 0.f, 0.f};
    const float* po = a.part_o + row * a.ns * D + c * 4;
    for (int s = 0; s < a.ns; ++s) {
        const float2 x = ml[s];
        if (x.y <= 0.f) continue;  // (a split that saw no visible key)
        const float w = exp2f(x.x - Ms);
        L += x.y * w;
        const ff4 ov = *reinterpret_cast<const ff4*>(po + (int64_t)s * D);
        acc[0] += ov[0] * w; acc[1] += ov[1] * w; acc[2] += ov[2] * w; acc[3] += ov[3] * w;
    }
    const int h = (int)(row / R), r = (int)(row - (int64_t)h * R);
    const int qi = r / a.G, qg = r - qi * a.G;
    const float inv = L > 0.f ? 1.f / L : 0.f;
    T w4[4] = {(T)(acc[0] * inv), (T)(acc[1] * inv), (T)(acc[2] * inv), (T)(acc[3] * inv)};
    *reinterpret_cast<uint2*>(reinterpret_cast<T*>(a.out) + h * a.o_sh + qg * a.o_sg + qi * a.o_si + c * 4) =
        *reinterpret_cast<const uint2*>(w4);
    if (a.lse && c == 0) a.lse[row] = (L > 0.f) ? (Ms + log2f(L)) * 0.69314718055994530942f : -INFINITY;
}

template <typename T, int D>
static int launch_flash(const FlashArgs& a, hipStream_t stream) {
    const int R = a.q_len * a.G;
    const dim3 grid((R + FL_ROWS - 1) / FL_ROWS, a.Hkv, a.ns), block(FL_THREADS);
    ProfScope ps("flash_fwd", stream);
    hipLaunchKernelGGL((flash_fwd_kernel<T, D>), grid, block, 0, stream, a);
    KVZ_CHECK_LAUNCH("flash_fwd_kernel");
    if (a.ns > 1) {
        const int64_t rows = (int64_t)a.Hkv * R;
        const int rpb = 256 / (D / 4);
        hipLaunchKernelGGL((flash_combine_kernel<T, D>), dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, stream, a);
        KVZ_CHECK_LAUNCH("flash_combine_kernel");
    }
    return KVZ_OK;
}

// key splits for (Hkv, R): enough blocks for two per CU, at most 64 splits; 1 when the row tiles alone fill the chip
static int flash_splits(int Hkv, int R) {
    const int64_t blocks = (int64_t)((R + FL_ROWS - 1) / FL_ROWS) * Hkv;
    if (blocks >= 256) return 1;
    const int ns = (int)((512 + blocks - 1) / blocks);
    return ns > 64 ? 64 : ns;
}

}  // namespace kvz

using namespace kvz;

static inline size_t fl_align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t flash16_workspace_bytes(int Hkv, int G, int q_len, int D) {
    const int R = q_len * G, ns = flash_splits(Hkv, R);
    if (ns <= 1) return 0;
    const size_t slots = (size_t)Hkv * R * ns;
    return fl_align256(slots * 2 * sizeof(float)) + fl_align256(slots * D * sizeof(float));
}

extern "C" size_t kvz_flash_workspace_bytes(int Hkv, int G, int q_len, int D) {
    if (Hkv <= 0 || G <= 0 || q_len <= 0 || D <= 0) return 0;
    const size_t w16 = flash16_workspace_bytes(Hkv, G, q_len, D);
    if (!flash2_takes(Hkv, G, q_len, D, true)) return w16;
    const size_t w32 = flash2_workspace_bytes(Hkv, G, q_len, D);  // (partials of its balanced partition)
    return w32 > w16 ? w32 : w16;
}

extern "C" int kvz_flash_fwd(const void* q, int64_t q_stride_head, int64_t q_stride_group, int64_t q_stride_pos, const void* k,
                             const void* v, const int32_t* k_start, const int32_t* k_len, int k_len_offset,
                             const int32_t* k_meta_host, int Hkv, int G, int q_len, int D, float scale, int causal, int dtype,
                             void* out, int64_t o_stride_head, int64_t o_stride_group, int64_t o_stride_pos, float* lse_out,
                             void* ws, size_t ws_bytes, kvz_stream_t stream_) {
    KVZ_REQUIRE(q && k && v && out, KVZ_EINVAL, "kvz_flash_fwd: null pointer");
    KVZ_REQUIRE((k_start && k_len) || k_meta_host, KVZ_EINVAL, "kvz_flash_fwd: head segments neither on the device nor on the host");
    KVZ_REQUIRE(Hkv > 0 && Hkv <= 65535 && G > 0 && q_len > 0, KVZ_EINVAL, "kvz_flash_fwd: bad shape");
    KVZ_REQUIRE(k_meta_host == nullptr || Hkv <= FL_MAXH || (k_start && k_len), KVZ_EINVAL,
                "kvz_flash_fwd: more than %d heads need the device arrays", FL_MAXH);
    KVZ_REQUIRE(D == 64 || D == 128, KVZ_EUNSUPPORTED, "kvz_flash_fwd: head_dim %d unsupported (64 or 128)", D);
    KVZ_REQUIRE(dtype == KVZ_F16 || dtype == KVZ_BF16, KVZ_EINVAL, "kvz_flash_fwd: bad dtype %d", dtype);
    KVZ_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0, KVZ_EINVAL,
                "kvz_flash_fwd: q/k/v must be 16-byte aligned, out 8-byte aligned");
    KVZ_REQUIRE(q_stride_head % 8 == 0 && q_stride_group % 8 == 0 && q_stride_pos % 8 == 0, KVZ_EINVAL,
                "kvz_flash_fwd: query strides must be multiples of 8 elements");
    KVZ_REQUIRE(o_stride_head % 4 == 0 && o_stride_group % 4 == 0 && o_stride_pos % 4 == 0, KVZ_EINVAL,
                "kvz_flash_fwd: output strides must be multiples of 4 elements");
    const size_t w32 = flash2_workspace_bytes(Hkv, G, q_len, D);
    if (flash2_takes(Hkv, G, q_len, D, ws != nullptr && w32 != 0 && ws_bytes >= w32) && (k_meta_host == nullptr || Hkv <= FL_MAXH))
        return flash2_fwd(q, q_stride_head, q_stride_group, q_stride_pos, k, v, k_start, k_len, k_len_offset, k_meta_host, Hkv, G, q_len,
                          scale, causal, dtype, out, o_stride_head, o_stride_group, o_stride_pos, lse_out, ws, ws_bytes, (hipStream_t)stream_);
    FlashArgs a{};
    a.q = q; a.k = k; a.v = v; a.out = out; a.lse = lse_out;
    a.k_start = k_start; a.k_len = k_len; a.k_len_offset = k_len_offset;
    a.n_meta = 0;
    if (k_meta_host && Hkv <= FL_MAXH) {
        for (int h = 0; h < Hkv; ++h) { a.m_start[h] = k_meta_host[h]; a.m_len[h] = k_meta_host[Hkv + h]; }
        a.n_meta = Hkv;
    }
    a.q_sh = q_stride_head; a.q_sg = q_stride_group; a.q_si = q_stride_pos;
    a.o_sh = o_stride_head; a.o_sg = o_stride_group; a.o_si = o_stride_pos;
    a.Hkv = Hkv; a.G = G; a.q_len = q_len; a.causal = causal; a.scale = scale;
    // key splits only with a workspace that holds their partials (ws == NULL: every block walks all of its head's keys)
    a.ns = 1;
    const size_t need = flash16_workspace_bytes(Hkv, G, q_len, D);
    if (ws && need) {
        KVZ_REQUIRE(aligned16(ws), KVZ_EINVAL, "kvz_flash_fwd: workspace must be 16-byte aligned");
        KVZ_REQUIRE(ws_bytes >= need, KVZ_EWORKSPACE, "kvz_flash_fwd: workspace too small (%zu < %zu)", ws_bytes, need);
        a.ns = flash_splits(Hkv, q_len * G);
        a.part_ml = reinterpret_cast<float*>(ws);
        a.part_o = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + fl_align256((size_t)Hkv * q_len * G * a.ns * 2 * sizeof(float)));
    }
    hipStream_t stream = (hipStream_t)stream_;
    if (dtype == KVZ_F16) return D == 128 ? launch_flash<_Float16, 128>(a, stream) : launch_flash<_Float16, 64>(a, stream);
    return D == 128 ? launch_flash<__bf16, 128>(a, stream) : launch_flash<__bf16, 64>(a, stream);
}


// f2: the dense forward of a scoring pass that also emits the row statistics of KVScore._get_score from its own QK^T tiles
extern "C" int kvz_flash_fwd_window(const void* q, int64_t q_stride_head, int64_t q_stride_group, int64_t q_stride_pos, const void* k,
                                    const void* v, const int32_t* k_meta_host, int Hkv, int G, int q_len, int D, float scale, int dtype,
                                    void* out, int64_t o_stride_head, int64_t o_stride_group, int64_t o_stride_pos, int win_sink,
                                    int win_start, int win_end, float* win_stats, int64_t win_stats_head_stride, kvz_stream_t stream_) {
    KVZ_REQUIRE(q && k && v && out && k_meta_host && win_stats, KVZ_EINVAL, "kvz_flash_fwd_window: null pointer");
    KVZ_REQUIRE(Hkv > 0 && Hkv <= FL_MAXH && G > 0 && q_len > 0, KVZ_EINVAL, "kvz_flash_fwd_window: bad shape");
    KVZ_REQUIRE(dtype == KVZ_F16 || dtype == KVZ_BF16, KVZ_EINVAL, "kvz_flash_fwd_window: bad dtype %d", dtype);
    KVZ_REQUIRE(flash2_takes(Hkv, G, q_len, D, false), KVZ_EUNSUPPORTED,
                "kvz_flash_fwd_window: only the 32-row kernel emits the statistics (head_dim 128, >= %d row blocks)", tunable(TUNE_FLASH2_MIN_BLOCKS));
    KVZ_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0 &&
                    (reinterpret_cast<uintptr_t>(win_stats) & 7u) == 0, KVZ_EINVAL, "kvz_flash_fwd_window: alignment");
    KVZ_REQUIRE(q_stride_head % 8 == 0 && q_stride_group % 8 == 0 && q_stride_pos % 8 == 0 && o_stride_head % 4 == 0 &&
                    o_stride_group % 4 == 0 && o_stride_pos % 4 == 0, KVZ_EINVAL, "kvz_flash_fwd_window: strides");
    for (int h = 0; h < Hkv; ++h)
        KVZ_REQUIRE(win_sink >= 0 && win_start >= win_sink && win_end >= win_start && win_end <= k_meta_host[Hkv + h] - q_len, KVZ_EINVAL,
                    "kvz_flash_fwd_window: bad window sink=%d start=%d end=%d len=%d q_len=%d", win_sink, win_start, win_end,
                    k_meta_host[Hkv + h], q_len);
    KVZ_REQUIRE(win_stats_head_stride >= (int64_t)G * q_len, KVZ_EINVAL, "kvz_flash_fwd_window: statistics stride too small");
    return flash2_fwd(q, q_stride_head, q_stride_group, q_stride_pos, k, v, nullptr, nullptr, 0, k_meta_host, Hkv, G, q_len, scale, 1,
                      dtype, out, o_stride_head, o_stride_group, o_stride_pos, nullptr, nullptr, 0, (hipStream_t)stream_, win_sink, win_start, win_end,
                      win_stats, win_stats_head_stride);
}
