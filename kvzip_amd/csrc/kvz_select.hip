// kvz_select.hip — exact order-statistic selection on 16-bit scores (gfx950).
//
// Replaces KVScore._threshold (reference attention/score.py:88-102), which sorts all
// L*Hkv*N scores to read ONE order statistic, and KVScore._threshold_uniform
// (attention/score.py:104-120).  Scores are fp16/bf16, i.e. 16-bit patterns, so the k-th largest
// value is found exactly with a two-level radix histogram (11 + 5 bits) in two streaming reads,
// followed by one streaming read that emits the boolean mask.  HBM-bound: 5 B per score.
// Round 4: THREE launches (histogram of the top bits; histogram of the low bits; mask) - every block of the second and third
// launch locates the bin itself from the global histogram of the launch before (2048 + 32 counters: a few hundred cycles)
// instead of waiting for a one-block "pick" launch, and the second launch also clears the counters the third one adds to.
// When the scores come out of the deferred scoring path the first histogram rides in the launch that turns the log buffer into
// 16-bit scores (kvz_score_finalize_log_hist, kvz_score.hip), which streams every score anyway: two launches here.
//
// Workspace layout (uint32 words):  [0,2048) hist_hi   [2048,2080) hist_lo   [2080,2084) unused
#include "kvz_common.h"

namespace kvz {

constexpr int HI_BINS = SEL_HI_BINS;  // top 11 bits of the order key
constexpr int LO_BINS = SEL_LO_BINS;  // low 5 bits
constexpr int SEL_THREADS = 256;        // row maximum
constexpr int HIST_THREADS = 1024;      // histogram and mask passes: one big block per CU - every block ends with one global atomic per non-empty
                                        // bin, and those serialise on a few hundred addresses: the three passes took 51 / 58 / 78 /
                                        // 131 / 224 us with 256 / 512 / 1024 / 2048 / 4096 blocks of 256 threads (profiles/r4_select.txt)
constexpr int SEL_UNROLL = 8;         // 16-byte loads a thread keeps in flight in the streaming passes
constexpr size_t SELECT_WS_WORDS = SEL_WS_WORDS;
constexpr uint32_t SEL_BIN_NONE = 0xFFFFFFFFu;   // find_bin_wave: the rank lies beyond the histogram's total

typedef kvz_u32x4 u32x4;

__device__ static inline void unpack8(const u32x4& v, uint32_t (&bits)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bits[2 * i] = v[i] & 0xFFFFu;
        bits[2 * i + 1] = v[i] >> 16;
    }
}

// ---- pass 1: histogram of the top 11 key bits ---------------------------------------------
__global__ __launch_bounds__(HIST_THREADS) void select_hist_hi_kernel(const uint16_t* __restrict__ scores,
                                                                    int64_t n, uint32_t* __restrict__ hist_hi) {
    __shared__ uint32_t lh[HI_BINS];
    for (int i = threadIdx.x; i < HI_BINS; i += HIST_THREADS) lh[i] = 0;
    __syncthreads();

    const int64_t nvec = n >> 3;
    stream_vec16<SEL_UNROLL>(reinterpret_cast<const u32x4*>(scores), nvec, [&](int64_t, const u32x4& v) {
        uint32_t b[8];
        unpack8(v, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&lh[order_key16(b[j]) >> 5], 1u);
    });
    // tail (< 8 elements) handled by block 0
    if (blockIdx.x == 0) {
        for (int64_t i = (nvec << 3) + threadIdx.x; i < n; i += HIST_THREADS)
            atomicAdd(&lh[order_key16(scores[i]) >> 5], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HI_BINS; i += HIST_THREADS) {
        uint32_t c = lh[i];
        if (c) atomicAdd(&hist_hi[i], c);
    }
}

// Search of a descending cumulative histogram by ONE wave: the bin `b` and the residual rank `r` such that (number of elements in
// bins > b) <= idx < that + hist[b].  It is the prologue of the streaming launches (round 4: every block locates the bin itself; a
// block-wide form with ~10 barriers cost a few microseconds in front of a 10-us streaming pass).  Lane l owns the BINS / 64
// consecutive bins of descending position l*PER .. l*PER+PER-1; 64-bit counts (n may exceed 2^32).  One barrier to publish the result.
template <int BINS>
__device__ static inline void find_bin_wave(const uint32_t* __restrict__ hist, uint64_t idx, uint32_t* out_bin, uint64_t* out_rank) {
    constexpr int PER = (BINS + 63) / 64;
    __shared__ uint32_t s_bin;
    __shared__ uint64_t s_rank;
    if (threadIdx.x < 64) {
        const int t = threadIdx.x;
        // idx >= the histogram's total (a workspace that does not belong to these scores, kvz_select_threshold_prehist): no lane
        // finds the rank - the sentinel makes the threshold NaN (nothing kept, the host raises) instead of whatever LDS held
        if (t == 0) { s_bin = SEL_BIN_NONE; s_rank = 0; }
        uint32_t loc[PER];
        uint64_t sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int jj = t * PER + j;  // position in DEscending order
            loc[j] = (jj < BINS) ? hist[BINS - 1 - jj] : 0u;
            sum += loc[j];
        }
        uint64_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t nb = __shfl_up(inc, o, 64);
            if (t >= o) inc += nb;
        }
        uint64_t above = inc - sum;
        if (idx >= above && idx < above + sum) {   // exactly one lane (idx < total)
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (idx >= above && idx < above + loc[j]) {
                    s_bin = (uint32_t)(BINS - 1 - (t * PER + j));
                    s_rank = idx - above;
                }
                above += loc[j];
            }
        }
    }
    __syncthreads();
    *out_bin = s_bin;
    *out_rank = s_rank;
}

// ---- pass 2: histogram of the low 5 bits inside the top bin that holds rank idx (every block finds that bin itself) -----------
__global__ __launch_bounds__(HIST_THREADS) void select_hist_lo_kernel(const uint16_t* __restrict__ scores, int64_t n, uint64_t idx,
                                                                    const uint32_t* __restrict__ hist_hi,
                                                                    uint32_t* __restrict__ hist_lo, int32_t* __restrict__ row_counts,
                                                                    int64_t rows, unsigned long long* __restrict__ kept_dev) {
    __shared__ uint32_t ll[LO_BINS];
    uint32_t bin;
    uint64_t rank;
    find_bin_wave<HI_BINS>(hist_hi, idx, &bin, &rank);
    if (threadIdx.x < LO_BINS) ll[threadIdx.x] = 0;
    // the counters the mask launch adds to (nothing else touches them before it)
    for (int64_t i = (int64_t)blockIdx.x * HIST_THREADS + threadIdx.x; row_counts && i < rows; i += (int64_t)gridDim.x * HIST_THREADS)
        row_counts[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *kept_dev = 0ull;
    __syncthreads();

    const int64_t nvec = n >> 3;
    stream_vec16<SEL_UNROLL>(reinterpret_cast<const u32x4*>(scores), nvec, [&](int64_t, const u32x4& v) {
        uint32_t b[8];
        unpack8(v, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t key = order_key16(b[j]);
            if ((key >> 5) == bin) atomicAdd(&ll[key & 31u], 1u);
        }
    });
    if (blockIdx.x == 0) {
        for (int64_t i = (nvec << 3) + threadIdx.x; i < n; i += HIST_THREADS) {
            uint32_t key = order_key16(scores[i]);
            if ((key >> 5) == bin) atomicAdd(&ll[key & 31u], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < LO_BINS) {
        uint32_t c = ll[threadIdx.x];
        if (c) atomicAdd(&hist_lo[threadIdx.x], c);
    }
}

// ---- pass 3: emit valid = score > thres, per-row counts, total kept --------------------------
// grid = (blocks_per_row, rows) when row_counts != nullptr, each block covering a slice of ONE row;
// otherwise rows == 1 and row_len == n.
__global__ __launch_bounds__(HIST_THREADS) void select_emit_kernel(
    const uint16_t* __restrict__ scores, int64_t row_len, int dtype, uint64_t idx, const uint32_t* __restrict__ hist_hi,
    const uint32_t* __restrict__ hist_lo, uint8_t* __restrict__ valid_out, int32_t* __restrict__ row_counts,
    float* __restrict__ thres_dev, unsigned long long* __restrict__ kept_dev) {
    // the 16-bit key of the threshold: top bin from the first histogram, low bits from the second (every block, redundantly)
    uint32_t bin, lo;
    uint64_t rank, rank2;
    find_bin_wave<HI_BINS>(hist_hi, idx, &bin, &rank);
    find_bin_wave<LO_BINS>(hist_lo, rank, &lo, &rank2);
    const uint32_t tkey = (bin << 5) | lo;
    const uint32_t tbits = order_key16_inv(tkey);
    const bool lost = bin == SEL_BIN_NONE || lo == SEL_BIN_NONE;   // the histograms do not hold rank idx: they are not of these scores
    const float thres = lost ? __builtin_nanf("") : half_bits_to_float(tbits, dtype);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *thres_dev = thres;

    const int64_t row = blockIdx.y;
    const uint16_t* srow = scores + row * row_len;
    uint8_t* vrow = valid_out + row * row_len;
    int cnt = 0;
    const bool vec_ok = ((row_len & 7) == 0);
    if (vec_ok) {
        const int64_t nvec = row_len >> 3;
        uint2* vv = reinterpret_cast<uint2*>(vrow);
        stream_vec16<SEL_UNROLL>(reinterpret_cast<const u32x4*>(srow), nvec, [&](int64_t i, const u32x4& v) {
            uint32_t b[8];
            unpack8(v, b);
            uint32_t m[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                m[j] = (half_bits_to_float(b[j], dtype) > thres) ? 1u : 0u;
                cnt += (int)m[j];
            }
            uint2 o;
            o.x = m[0] | (m[1] << 8) | (m[2] << 16) | (m[3] << 24);
            o.y = m[4] | (m[5] << 8) | (m[6] << 16) | (m[7] << 24);
            vv[i] = o;
        });
    } else {
        const int64_t stride = (int64_t)gridDim.x * HIST_THREADS;
        for (int64_t i = (int64_t)blockIdx.x * HIST_THREADS + threadIdx.x; i < row_len; i += stride) {
            uint32_t mm = (half_bits_to_float(srow[i], dtype) > thres) ? 1u : 0u;
            cnt += (int)mm;
            vrow[i] = (uint8_t)mm;
        }
    }
    // block reduction of cnt -> one atomic per block
    __shared__ int wsum[HIST_THREADS / WAVE];
    int w = wave_reduce_sum(cnt);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int i = 0; i < HIST_THREADS / WAVE; ++i) tot += wsum[i];
        if (tot) {
            if (row_counts) atomicAdd(&row_counts[row], tot);
            atomicAdd(kept_dev, (unsigned long long)tot);
        }
    }
}

// ratio >= 1: everything kept
__global__ void select_all_kernel(int64_t rows, int64_t row_len, int32_t* row_counts, float* thres_dev,
                                  unsigned long long* kept_dev) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row_counts && i < rows) row_counts[i] = (int32_t)row_len;
    if (i == 0) {
        if (thres_dev) *thres_dev = 0.f;
        if (kept_dev) *kept_dev = (unsigned long long)(rows * row_len);
    }
}

// ---- per-row exact top-k (uniform head budgets) --------------------------------------------------
constexpr int TOPK_THREADS = 1024;

template <int BINS>
__device__ static inline void find_bin_desc_1024(const uint32_t* hist, uint32_t idx, uint32_t* out_bin,
                                                 uint32_t* out_rank, uint32_t* out_above) {
    // BINS <= 2048; executed by the first 64 lanes, everybody syncs
    __shared__ uint32_t s_bin, s_rank, s_above;
    constexpr int PER = (BINS + 63) / 64;
    if (threadIdx.x < 64) {
        const int t = threadIdx.x;
        uint32_t loc[PER];
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            int jj = t * PER + j;
            loc[j] = (jj < BINS) ? hist[BINS - 1 - jj] : 0u;
            sum += loc[j];
        }
        uint32_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t nb = __shfl_up(inc, o, 64);
            if (t >= o) inc += nb;
        }
        uint32_t above = inc - sum;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            int jj = t * PER + j;
            if (jj < BINS && idx >= above && idx < above + loc[j]) {
                s_bin = (uint32_t)(BINS - 1 - jj);
                s_rank = idx - above;
                s_above = above;
            }
            above += loc[j];
        }
    }
    __syncthreads();
    *out_bin = s_bin;
    *out_rank = s_rank;
    *out_above = s_above;
    __syncthreads();
}

__global__ __launch_bounds__(TOPK_THREADS) void select_topk_rows_kernel(const uint16_t* __restrict__ scores,
                                                                       int64_t row_len, uint32_t k,
                                                                       uint8_t* __restrict__ valid_out,
                                                                       int32_t* __restrict__ row_counts) {
    __shared__ uint32_t hh[HI_BINS];
    __shared__ uint32_t hl[LO_BINS];
    __shared__ uint32_t wtot[TOPK_THREADS / WAVE];
    __shared__ uint32_t s_running;
    const int64_t row = blockIdx.x;
    const uint16_t* srow = scores + row * row_len;
    uint8_t* vrow = valid_out + row * row_len;
    const int t = threadIdx.x;

    for (int i = t; i < HI_BINS; i += TOPK_THREADS) hh[i] = 0;
    if (t < LO_BINS) hl[t] = 0;
    if (t == 0) s_running = 0;
    __syncthreads();
    for (int64_t i = t; i < row_len; i += TOPK_THREADS) atomicAdd(&hh[order_key16(srow[i]) >> 5], 1u);
    __syncthreads();
    uint32_t bin, rank, above;
    find_bin_desc_1024<HI_BINS>(hh, k - 1, &bin, &rank, &above);
    for (int64_t i = t; i < row_len; i += TOPK_THREADS) {
        uint32_t key = order_key16(srow[i]);
        if ((key >> 5) == bin) atomicAdd(&hl[key & 31u], 1u);
    }
    __syncthreads();
    uint32_t lo, rank2, above2;
    find_bin_desc_1024<LO_BINS>(hl, rank, &lo, &rank2, &above2);
    const uint32_t tkey = (bin << 5) | lo;
    // elements with key > tkey: above + above2;  ties to keep (lowest index first):
    const uint32_t need = k - (above + above2);

    // ordered emit: chunks of TOPK_THREADS elements, running count of ties seen so far
    for (int64_t base = 0; base < row_len; base += TOPK_THREADS) {
        const int64_t i = base + t;
        uint32_t key = (i < row_len) ? order_key16(srow[i]) : 0u;
        const bool in = (i < row_len);
        const uint32_t is_tie = (in && key == tkey) ? 1u : 0u;
        // block exclusive scan of is_tie
        uint64_t bal = __ballot(is_tie);
        uint32_t lane = (uint32_t)lane_id();
        uint32_t wpre = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[t >> 6] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = s_running;
        for (int w = 0; w < (t >> 6); ++w) before += wtot[w];
        if (in) {
            uint8_t keep = (key > tkey) ? 1 : ((is_tie && (before + wpre) < need) ? 1 : 0);
            vrow[i] = keep;
        }
        __syncthreads();
        if (t == 0) {
            uint32_t tot = 0;
            for (int w = 0; w < TOPK_THREADS / WAVE; ++w) tot += wtot[w];
            s_running += tot;
        }
        __syncthreads();
    }
    if (t == 0 && row_counts) row_counts[row] = (int32_t)k;
}

// ---- head-level selection: one score per (layer, KV head), expanded over N context tokens by the reference ---------
// (model/wrapper.py:40-58 expands [L,Hkv] to [L,1,Hkv,N] and sorts L*Hkv*N values; every head value appears N times,
// so the order statistic of rank idx in the expanded tensor is the head value of rank idx / N.)  One block does the whole
// selection for up to 65536 values: 11+5-bit LDS histograms, pick, emit.
__global__ __launch_bounds__(TOPK_THREADS) void select_small_kernel(const uint16_t* __restrict__ scores, int n, uint32_t rank,
                                                                   int dtype, int64_t weight, uint8_t* __restrict__ valid_out,
                                                                   int32_t* __restrict__ row_counts, float* __restrict__ thres_dev,
                                                                   unsigned long long* __restrict__ kept_dev) {
    __shared__ uint32_t hh[HI_BINS];
    __shared__ uint32_t hl[LO_BINS];
    __shared__ uint32_t s_kept;
    const int t = threadIdx.x;
    for (int i = t; i < HI_BINS; i += TOPK_THREADS) hh[i] = 0;
    if (t < LO_BINS) hl[t] = 0;
    if (t == 0) s_kept = 0;
    __syncthreads();
    for (int i = t; i < n; i += TOPK_THREADS) atomicAdd(&hh[order_key16(scores[i]) >> 5], 1u);
    __syncthreads();
    uint32_t bin, r1, above;
    find_bin_desc_1024<HI_BINS>(hh, rank, &bin, &r1, &above);
    for (int i = t; i < n; i += TOPK_THREADS) {
        const uint32_t key = order_key16(scores[i]);
        if ((key >> 5) == bin) atomicAdd(&hl[key & 31u], 1u);
    }
    __syncthreads();
    uint32_t lo, r2, above2;
    find_bin_desc_1024<LO_BINS>(hl, r1, &lo, &r2, &above2);
    const float thres = half_bits_to_float(order_key16_inv((bin << 5) | lo), dtype);
    uint32_t cnt = 0;
    for (int i = t; i < n; i += TOPK_THREADS) {
        const uint32_t keep = half_bits_to_float(scores[i], dtype) > thres ? 1u : 0u;  // float compare: strict >, -0 == +0
        valid_out[i] = (uint8_t)keep;
        if (row_counts) row_counts[i] = keep ? (int32_t)weight : 0;
        cnt += keep;
    }
    cnt = (uint32_t)wave_reduce_sum((int)cnt);
    if ((t & 63) == 0 && cnt) atomicAdd(&s_kept, cnt);
    __syncthreads();
    if (t == 0) {
        *thres_dev = thres;
        *kept_dev = (unsigned long long)s_kept * (unsigned long long)weight;
    }
}

// ---- per-row maximum of 16-bit scores (head-score production, reference test.py:22-25) ----------------------------
__global__ __launch_bounds__(SEL_THREADS) void rowmax16_kernel(const uint16_t* __restrict__ scores, int64_t row_len, int dtype,
                                                              uint16_t* __restrict__ out) {
    const uint16_t* srow = scores + (int64_t)blockIdx.x * row_len;
    float best = -INFINITY;
    uint32_t bits = 0xFC00u;  // -inf (fp16); rewritten below for bf16
    if (dtype == KVZ_BF16) bits = 0xFF80u;
    for (int64_t i = threadIdx.x; i < row_len; i += SEL_THREADS) {
        const uint32_t b = srow[i];
        const float f = half_bits_to_float(b, dtype);
        if (f > best || (f != f)) { best = f; bits = b; }  // a NaN wins, as in torch.amax
    }
    __shared__ float sb[SEL_THREADS];
    __shared__ uint32_t sbits[SEL_THREADS];
    sb[threadIdx.x] = best;
    sbits[threadIdx.x] = bits;
    __syncthreads();
    for (int o = SEL_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float f = sb[threadIdx.x + o];
            if (f > sb[threadIdx.x] || (f != f)) { sb[threadIdx.x] = f; sbits[threadIdx.x] = sbits[threadIdx.x + o]; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (uint16_t)sbits[0];
}

__global__ void fill_rows_kernel(uint8_t* valid, int64_t total, uint8_t value, int32_t* row_counts, int64_t rows,
                                 int32_t count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = i; j < total; j += stride) valid[j] = value;
    if (row_counts && i < rows) row_counts[i] = count;
}

}  // namespace kvz

using namespace kvz;

extern "C" size_t kvz_select_workspace_bytes(void) { return SELECT_WS_WORDS * sizeof(uint32_t); }

static int select_threshold_impl(const void* scores, int64_t n, double ratio, int dtype, uint8_t* valid_out,
                                 int64_t row_len, int32_t* row_counts, float* thres_dev, int64_t* kept_dev,
                                 void* ws, size_t ws_bytes, kvz_stream_t stream_, bool prehist);
extern "C" int kvz_select_threshold(const void* scores, int64_t n, double ratio, int dtype, uint8_t* valid_out,
                                    int64_t row_len, int32_t* row_counts, float* thres_dev, int64_t* kept_dev,
                                    void* ws, size_t ws_bytes, kvz_stream_t stream_) {
    return select_threshold_impl(scores, n, ratio, dtype, valid_out, row_len, row_counts, thres_dev, kept_dev, ws, ws_bytes, stream_, false);
}
extern "C" int kvz_select_threshold_prehist(const void* scores, int64_t n, double ratio, int dtype, uint8_t* valid_out,
                                            int64_t row_len, int32_t* row_counts, float* thres_dev, int64_t* kept_dev,
                                            void* ws, size_t ws_bytes, kvz_stream_t stream_) {
    return select_threshold_impl(scores, n, ratio, dtype, valid_out, row_len, row_counts, thres_dev, kept_dev, ws, ws_bytes, stream_, true);
}
static int select_threshold_impl(const void* scores, int64_t n, double ratio, int dtype, uint8_t* valid_out,
                                 int64_t row_len, int32_t* row_counts, float* thres_dev, int64_t* kept_dev,
                                 void* ws, size_t ws_bytes, kvz_stream_t stream_, bool prehist) {
    hipStream_t stream = (hipStream_t)stream_;
    KVZ_REQUIRE(scores && valid_out && thres_dev && kept_dev && ws, KVZ_EINVAL, "kvz_select_threshold: null pointer");
    KVZ_REQUIRE(n > 0, KVZ_EINVAL, "kvz_select_threshold: n must be > 0 (got %lld)", (long long)n);
    KVZ_REQUIRE(dtype == KVZ_F16 || dtype == KVZ_BF16, KVZ_EINVAL, "kvz_select_threshold: bad dtype %d", dtype);
    KVZ_REQUIRE(ws_bytes >= kvz_select_workspace_bytes(), KVZ_EWORKSPACE, "kvz_select_threshold: workspace too small");
    KVZ_REQUIRE(aligned16(scores), KVZ_EINVAL, "kvz_select_threshold: scores must be 16-byte aligned");
    KVZ_REQUIRE((reinterpret_cast<uintptr_t>(valid_out) & 7u) == 0, KVZ_EINVAL,
                "kvz_select_threshold: valid_out must be 8-byte aligned");
    if (row_counts) {
        KVZ_REQUIRE(row_len > 0 && n % row_len == 0, KVZ_EINVAL, "kvz_select_threshold: n %% row_len != 0");
    } else {
        row_len = n;
    }
    const int64_t rows = n / row_len;
    KVZ_REQUIRE(rows <= 65535, KVZ_EINVAL, "kvz_select_threshold: too many rows (%lld)", (long long)rows);
    if (!(ratio < 1.0)) {  // reference: `if ratio < 1: ... else: all ones, thres = 0.`
        (void)hipMemsetAsync(valid_out, 1, (size_t)n, stream);
        int blocks = (int)((rows + 255) / 256);
        hipLaunchKernelGGL(select_all_kernel, dim3(blocks), dim3(256), 0, stream, rows, row_len, row_counts,
                           thres_dev, reinterpret_cast<unsigned long long*>(kept_dev));
        KVZ_CHECK_LAUNCH("select_all_kernel");
        return KVZ_OK;
    }
    // idx = max(int(n * ratio) - 1, 0)   (Python: int*float -> double product, truncation)
    double prod = (double)n * ratio;
    int64_t idx = (int64_t)prod - 1;
    if (idx < 0) idx = 0;

    uint32_t* hist_hi = reinterpret_cast<uint32_t*>(ws);
    uint32_t* hist_lo = hist_hi + HI_BINS;
    // prehist: hist_hi already holds the histogram of exactly these scores and hist_lo is zero (kvz_score_finalize_log_hist)
    if (!prehist) (void)hipMemsetAsync(ws, 0, SELECT_WS_WORDS * sizeof(uint32_t), stream);

    const int64_t nvec = (n + 7) >> 3;
    int blocks = (int)((nvec + HIST_THREADS - 1) / HIST_THREADS);
    const int cap = tunable(TUNE_SEL_BLOCKS) > 0 ? tunable(TUNE_SEL_BLOCKS) : device_cus();
    if (blocks > cap) blocks = cap;  // (one 1024-thread block per CU: every block flushes its non-empty bins with global atomics)
    if (blocks < 1) blocks = 1;
    const uint16_t* s16 = reinterpret_cast<const uint16_t*>(scores);
    ProfScope ps("select", stream);  // the streaming passes
    if (!prehist) {
        hipLaunchKernelGGL(select_hist_hi_kernel, dim3(blocks), dim3(HIST_THREADS), 0, stream, s16, n, hist_hi);
        KVZ_CHECK_LAUNCH("select_hist_hi_kernel");
    }
    hipLaunchKernelGGL(select_hist_lo_kernel, dim3(blocks), dim3(HIST_THREADS), 0, stream, s16, n, (uint64_t)idx, hist_hi, hist_lo,
                       row_counts, rows, reinterpret_cast<unsigned long long*>(kept_dev));
    KVZ_CHECK_LAUNCH("select_hist_lo_kernel");

    // rows whose start is not 16-byte aligned take the scalar path inside the kernel (row_len % 8 != 0)
    const int64_t per_row_vec = ((row_len & 7) == 0) ? (row_len >> 3) : row_len;
    int bx = (int)((per_row_vec + HIST_THREADS - 1) / HIST_THREADS);
    int max_bx = (int)((tunable(TUNE_EMIT_BLOCKS) > 0 ? tunable(TUNE_EMIT_BLOCKS) : device_cus()) / rows);
    if (max_bx < 1) max_bx = 1;
    if (bx > max_bx) bx = max_bx;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(select_emit_kernel, dim3(bx, (unsigned)rows), dim3(HIST_THREADS), 0, stream, s16, row_len, dtype,
                       (uint64_t)idx, hist_hi, hist_lo, valid_out, row_counts, thres_dev,
                       reinterpret_cast<unsigned long long*>(kept_dev));
    KVZ_CHECK_LAUNCH("select_emit_kernel");
    return KVZ_OK;
}

extern "C" int kvz_select_topk_rows(const void* scores, int64_t rows, int64_t row_len, int64_t k, int dtype,
                                    uint8_t* valid_out, int32_t* row_counts, kvz_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    KVZ_REQUIRE(scores && valid_out, KVZ_EINVAL, "kvz_select_topk_rows: null pointer");
    KVZ_REQUIRE(rows > 0 && row_len > 0, KVZ_EINVAL, "kvz_select_topk_rows: empty input");
    KVZ_REQUIRE(dtype == KVZ_F16 || dtype == KVZ_BF16, KVZ_EINVAL, "kvz_select_topk_rows: bad dtype %d", dtype);
    KVZ_REQUIRE(k >= 0, KVZ_EINVAL, "kvz_select_topk_rows: negative k");
    KVZ_REQUIRE(row_len < (1ll << 31), KVZ_EINVAL, "kvz_select_topk_rows: row too long");
    if (k == 0 || k >= row_len) {
        const int64_t total = rows * row_len;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        int need = (int)((rows + 255) / 256);
        if (blocks < need) blocks = need;
        hipLaunchKernelGGL(fill_rows_kernel, dim3(blocks), dim3(256), 0, stream, valid_out, total,
                           (uint8_t)(k == 0 ? 0 : 1), row_counts, rows, (int32_t)(k == 0 ? 0 : row_len));
        KVZ_CHECK_LAUNCH("fill_rows_kernel");
        return KVZ_OK;
    }
    hipLaunchKernelGGL(select_topk_rows_kernel, dim3((unsigned)rows), dim3(TOPK_THREADS), 0, stream,
                       reinterpret_cast<const uint16_t*>(scores), row_len, (uint32_t)k, valid_out, row_counts);
    KVZ_CHECK_LAUNCH("select_topk_rows_kernel");
    return KVZ_OK;
}

extern "C" int kvz_select_heads(const void* head_scores, int rows, int64_t N, double ratio, int dtype, uint8_t* valid_heads,
                                int32_t* row_counts, float* thres_dev, int64_t* kept_dev, kvz_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    KVZ_REQUIRE(head_scores && valid_heads && thres_dev && kept_dev, KVZ_EINVAL, "kvz_select_heads: null pointer");
    KVZ_REQUIRE(rows > 0 && rows <= 65536 && N > 0, KVZ_EINVAL, "kvz_select_heads: bad shape rows=%d N=%lld", rows, (long long)N);
    KVZ_REQUIRE(dtype == KVZ_F16 || dtype == KVZ_BF16, KVZ_EINVAL, "kvz_select_heads: bad dtype %d", dtype);
    if (!(ratio < 1.0)) {  // reference: all ones, thres = 0.
        (void)hipMemsetAsync(valid_heads, 1, (size_t)rows, stream);
        hipLaunchKernelGGL(select_all_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, (int64_t)rows, N, row_counts,
                           thres_dev, reinterpret_cast<unsigned long long*>(kept_dev));
        KVZ_CHECK_LAUNCH("select_all_kernel");
        return KVZ_OK;
    }
    // idx = max(int(rows*N*ratio) - 1, 0) in the EXPANDED tensor (score.py:92-94); its head rank is idx / N
    const double prod = (double)((int64_t)rows * N) * ratio;
    int64_t idx = (int64_t)prod - 1;
    if (idx < 0) idx = 0;
    const uint32_t rank = (uint32_t)(idx / N);
    ProfScope ps("select_heads", stream);
    hipLaunchKernelGGL(select_small_kernel, dim3(1), dim3(TOPK_THREADS), 0, stream, reinterpret_cast<const uint16_t*>(head_scores),
                       rows, rank, dtype, N, valid_heads, row_counts, thres_dev, reinterpret_cast<unsigned long long*>(kept_dev));
    KVZ_CHECK_LAUNCH("select_small_kernel");
    return KVZ_OK;
}

extern "C" int kvz_rowmax16(const void* scores, int64_t rows, int64_t row_len, int dtype, void* out, kvz_stream_t stream_) {
    KVZ_REQUIRE(scores && out, KVZ_EINVAL, "kvz_rowmax16: null pointer");
    KVZ_REQUIRE(rows > 0 && rows <= 0x7fffffffll && row_len > 0, KVZ_EINVAL, "kvz_rowmax16: bad shape");
    KVZ_REQUIRE(dtype == KVZ_F16 || dtype == KVZ_BF16, KVZ_EINVAL, "kvz_rowmax16: bad dtype %d", dtype);
    hipLaunchKernelGGL(rowmax16_kernel, dim3((unsigned)rows), dim3(SEL_THREADS), 0, (hipStream_t)stream_,
                       reinterpret_cast<const uint16_t*>(scores), row_len, dtype, reinterpret_cast<uint16_t*>(out));
    KVZ_CHECK_LAUNCH("rowmax16_kernel");
    return KVZ_OK;
}
