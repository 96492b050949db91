// kvz_common.h — shared device/host helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/kvzip_hip.h"
#include "../../include/kvzip_hip_debug.h"

namespace kvz {

constexpr int WAVE = 64;

// ---- error plumbing ---------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define KVZ_REQUIRE(cond, code, ...)          \
    do {                                      \
        if (!(cond)) {                        \
            kvz::set_error(__VA_ARGS__);      \
            return (code);                    \
        }                                     \
    } while (0)

#define KVZ_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            kvz::set_error("%s: launch failed: %s", (name), hipGetErrorString(e_));   \
            return KVZ_ELAUNCH;                                                       \
        }                                                                             \
    } while (0)

// ---- optional per-kernel profiler (kvz_prof_enable): hipEvents on the launch stream ------------------
bool prof_enabled();
void prof_begin(const char* name, hipStream_t stream, int* slot, size_t* idx);
void prof_end(hipStream_t stream, int slot, size_t idx);
struct ProfScope {
    hipStream_t stream;
    int slot = -1;
    size_t idx = 0;
    ProfScope(const char* name, hipStream_t s) : stream(s) {
        if (prof_enabled()) prof_begin(name, s, &slot, &idx);
    }
    ~ProfScope() {
        if (slot >= 0) prof_end(stream, slot, idx);
    }
};

// division of a non-negative int (< 2^31) by a launch-invariant divisor: q = mulhi(n, m) >> sh with m = ceil(2^(31+l) / d),
// l = ceil(log2 d) (Granlund-Montgomery round-up method, exact for 31-bit numerators).  A 32-bit division costs ~20 instructions;
// the item switch of pass A had nine of them, executed by all eight waves.
struct FastDiv {
    uint32_t d, m, sh;  // sh = l - 1; d == 1 is the identity (m = 0)
    __host__ __device__ inline int div(int n) const { return m ? (int)(__umulhi_((uint32_t)n, m) >> sh) : n; }
    __host__ __device__ inline int mod(int n) const { return n - div(n) * (int)d; }
    __host__ __device__ static inline uint32_t __umulhi_(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
};
static inline FastDiv make_fastdiv(int d) {
    FastDiv f;
    f.d = (uint32_t)d;
    if (d <= 1) { f.m = 0; f.sh = 0; return f; }
    int l = 0;
    while ((1ll << l) < d) ++l;
    f.m = (uint32_t)((((uint64_t)1 << (31 + l)) + (uint64_t)d - 1) / (uint64_t)d);
    f.sh = (uint32_t)(l - 1);
    return f;
}

// ---- tuning knobs with a test hook (kvz_debug_set_tunable; ONE environment variable, read when the library is loaded: KVZIP_SCORE_PRUNE presets score_prune) ----
enum Tunable { TUNE_ATTN_ITEMS = 0, TUNE_FLASH_MIN_ROWS = 1, TUNE_FLASH2_MIN_BLOCKS = 2, TUNE_FLASH2_XCD = 3, TUNE_SEL_BLOCKS = 4, TUNE_EMIT_BLOCKS = 5, TUNE_FLASH2_SPLIT = 6, TUNE_SCORE_PRUNE = 7, TUNE_COUNT = 8 };
int tunable(Tunable t);
// compute units of the current device (cached hipDeviceAttributeMultiprocessorCount; 256 on MI355X): "one block per CU" launch sizes
// - decode work items, selection passes, the finalize + histogram launch, the dense forward's rounds - all derive from this one number
int device_cus();

// kvz_score_chunk_log for the asynchronous entry points (kvz_api.hip): may leave phases of the call's tail to the next calls on the same
// workspace and stream (score_prune = 6, kvz_score.hip: score_tail_kernel)
int score_chunk_log_deferred(const void* q, int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen, int sink, int start,
                             int end, int q_len, int Hkv, int G, int D, int dtype, uint32_t* log_out, int64_t log_head_stride, void* ws,
                             size_t ws_bytes, kvz_stream_t stream);

// exact-reciprocal constant of the scoring rounding chain (kvz_score.hip): half(x * r) == half(x / sqrt(D)) for every 16-bit x, or 0
float score_exact_reciprocal(int D, int dtype);

// the 32-row dense forward (kvz_flash2.hip), reached through kvz_flash_fwd
bool flash2_takes(int Hkv, int G, int q_len, int D, bool with_ws);
size_t flash2_workspace_bytes(int Hkv, int G, int q_len, int D);
int flash2_fwd(const void* q, int64_t q_stride_head, int64_t q_stride_group, int64_t q_stride_pos, const void* k, const void* v,
               const int32_t* k_start, const int32_t* k_len, int k_len_offset, const int32_t* k_meta_host, int Hkv, int G, int q_len,
               float scale, int causal, int dtype, void* out, int64_t o_stride_head, int64_t o_stride_group, int64_t o_stride_pos,
               float* lse_out, void* ws, size_t ws_bytes, hipStream_t stream, int win_sink = 0, int win_start = 0, int win_end = 0,
               float* win_stats = nullptr, int64_t win_stats_head_stride = 0);

// selection workspace shared by kvz_select.hip and the fused finalize + histogram launch of kvz_score.hip (uint32 words):
// [0, 2048) histogram of the top 11 bits of the order key, [2048, 2080) histogram of the low 5 bits inside the picked bin, 4 spare
constexpr int SEL_HI_BINS = 2048, SEL_LO_BINS = 32;
constexpr size_t SEL_WS_WORDS = SEL_HI_BINS + SEL_LO_BINS + 4;

// Streaming read of `nvec` 16-byte vectors by a whole grid (x dimension), U loads in flight per thread before the first one is
// consumed: a grid-stride loop with ONE dependent 16-byte load per iteration keeps ~8 KB per CU in flight and runs at 1.4 TB/s on
// an L2 / MALL-resident input (profiles/r4_select.txt); f(i, v) is called once per vector.
typedef uint32_t kvz_u32x4 __attribute__((ext_vector_type(4)));
template <int U, typename F>
__device__ static inline void stream_vec16(const kvz_u32x4* __restrict__ src, int64_t nvec, F&& f) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += U * stride) {
        kvz_u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {  // (the last batch re-reads its first vector where it runs past the end: the loads stay unconditional)
            const int64_t j = i + u * stride;
            v[u] = src[j < nvec ? j : i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * stride < nvec) f(i + u * stride, v[u]);
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- half / bf16 bit helpers ------------------------------------------------------------
// Monotone map from the 16-bit pattern of an fp16 / bf16 value to an unsigned key such that
// a > b (as floats)  <=>  key(a) > key(b).  (-0 maps just below +0; both compare equal to 0 as floats,
// see select kernels for why that is harmless.)
__host__ __device__ static inline uint32_t order_key16(uint32_t bits) {
    return (bits & 0x8000u) ? (~bits & 0xFFFFu) : (bits | 0x8000u);
}
__host__ __device__ static inline uint32_t order_key16_inv(uint32_t key) {
    return (key & 0x8000u) ? (key & 0x7FFFu) : (~key & 0xFFFFu);
}

__device__ static inline float half_bits_to_float(uint32_t bits, int dtype) {
    if (dtype == KVZ_BF16) return __uint_as_float(bits << 16);
    _Float16 h;
    unsigned short s = (unsigned short)bits;
    __builtin_memcpy(&h, &s, 2);
    return (float)h;
}

// ---- wave primitives ----------------------------------------------------------------------
__device__ static inline int lane_id() { return threadIdx.x & 63; }

__device__ static inline int wave_reduce_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ static inline float wave_reduce_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// inclusive prefix sum across the 64 lanes of a wave
__device__ static inline int wave_inclusive_scan(int v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int n = __shfl_up(v, o, 64);
        if (lane_id() >= o) v += n;
    }
    return v;
}

}  // namespace kvz
