// kvz_compact.hip — eviction = order-preserving, head-major compaction of K and V (gfx950).
//
// Replaces EvictCache._get_valid + prepare_init (reference attention/kvcache.py:140-185), which runs
// 2*L boolean-mask gathers (nonzero + index_select each), and the append kernel
// update_flatten_view (reference csrc/csrc/cuda_api.cu:15-111).
//
// Pure data movement, HBM-bound.  A K/V row is D*2 bytes (256 B for D = 128): one row is moved by
// D*2/16 lanes with 16-byte loads/stores, so a wave moves 4 rows per instruction; source rows are whole
// 128-byte lines and destination rows are consecutive, i.e. every store is fully coalesced and evicted
// rows are never touched.  Algorithmic bytes: 2 * kept_rows * row_bytes * 2 (K and V) + mask bytes.
#include "kvz_common.h"

namespace kvz {

constexpr int CT = KVZ_COMPACT_TILE;  // tokens per tile
constexpr int CP_THREADS = 256;
constexpr int CP_PER_THREAD = CT / CP_THREADS;  // 4 mask positions per thread
static_assert(CP_PER_THREAD == 4, "tile/threads layout");

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// full-mask value of position p in [0, klen): ones(sink) ++ valid ++ ones(rest)
// tok1 = 1: one mask byte per context token; tok1 = 0: ONE byte per (layer, head) row (head-level eviction,
// reference model/wrapper.py:40-58: the whole context of a head is kept or dropped)
__device__ static inline uint32_t full_mask4(const uint8_t* __restrict__ vrow, int p0, int sink, int N, int klen, int tok1) {
    // returns 4 mask bits (bit j = position p0+j), positions >= klen are 0
    uint32_t bits = 0;
    const int c0 = p0 - sink;
    if (tok1 && p0 >= sink && p0 + 3 < sink + N && ((reinterpret_cast<uintptr_t>(vrow + c0) & 3u) == 0)) {
        uint32_t w = *reinterpret_cast<const uint32_t*>(vrow + c0);
        bits = ((w & 0xFFu) ? 1u : 0u) | ((w & 0xFF00u) ? 2u : 0u) | ((w & 0xFF0000u) ? 4u : 0u) |
               ((w & 0xFF000000u) ? 8u : 0u);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = p0 + j;
            uint32_t b;
            if (p >= klen) b = 0;
            else if (p < sink || p >= sink + N) b = 1;
            else b = vrow[tok1 ? p - sink : 0] ? 1u : 0u;
            bits |= b << j;
        }
    }
    return bits;
}

// ---- plan, step 1: kept tokens per (row, tile) ------------------------------------------------
__global__ __launch_bounds__(CP_THREADS) void compact_tile_count_kernel(const uint8_t* __restrict__ valid, int N,
                                                                       int sink, int klen, int ntiles, int tok1,
                                                                       int32_t* __restrict__ tile_cnt) {
    const int tile = blockIdx.x;
    const int row = blockIdx.y;
    const uint8_t* vrow = valid + (int64_t)row * (tok1 ? N : 1);
    const int p0 = tile * CT + threadIdx.x * CP_PER_THREAD;
    int c = __popc(full_mask4(vrow, p0, sink, N, klen, tok1));
    __shared__ int ws[CP_THREADS / WAVE];
    int w = wave_reduce_sum(c);
    if (lane_id() == 0) ws[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[(int64_t)row * ntiles + tile] = ws[0] + ws[1] + ws[2] + ws[3];
}

// ---- plan, step 2: per layer, scan tiles of every head + head metadata ---------------------------
// one block (256 threads) per layer.  In-place: tile_base[row][t] becomes the exclusive prefix.
__global__ __launch_bounds__(CP_THREADS) void compact_plan_scan_kernel(int Hkv, int ntiles, int slack,
                                                                      int32_t* __restrict__ tile_base,
                                                                      int32_t* __restrict__ len_k,
                                                                      int32_t* __restrict__ cu_len_k,
                                                                      int32_t* __restrict__ seg_start,
                                                                      int32_t* __restrict__ max_len_k) {
    const int layer = blockIdx.x;
    __shared__ int wtot[CP_THREADS / WAVE];
    __shared__ int s_carry;
    __shared__ int s_len[1024];  // Hkv <= 1024
    for (int h = 0; h < Hkv; ++h) {
        int32_t* tb = tile_base + ((int64_t)layer * Hkv + h) * ntiles;
        if (threadIdx.x == 0) s_carry = 0;
        __syncthreads();
        for (int base = 0; base < ntiles; base += CP_THREADS) {
            const int i = base + threadIdx.x;
            int v = (i < ntiles) ? tb[i] : 0;
            int inc = wave_inclusive_scan(v);
            if (lane_id() == 63) wtot[threadIdx.x >> 6] = inc;
            __syncthreads();
            int pre = s_carry;
            for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) pre += wtot[w];
            if (i < ntiles) tb[i] = pre + inc - v;
            __syncthreads();
            if (threadIdx.x == 0) s_carry += wtot[0] + wtot[1] + wtot[2] + wtot[3];
            __syncthreads();
        }
        if (threadIdx.x == 0) s_len[h] = s_carry;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int cu = 0, seg = 0, mx = 0;
        cu_len_k[(int64_t)layer * (Hkv + 1)] = 0;
        for (int h = 0; h < Hkv; ++h) {
            const int l = s_len[h];
            len_k[(int64_t)layer * Hkv + h] = l;
            seg_start[(int64_t)layer * Hkv + h] = seg;
            cu += l;
            seg += l + slack;
            cu_len_k[(int64_t)layer * (Hkv + 1) + h + 1] = cu;
            mx = l > mx ? l : mx;
        }
        max_len_k[layer] = mx;
    }
}

// ---- gather -------------------------------------------------------------------------------------
struct CompactArgs {
    const void* const* k_ptrs;   // device pointer tables (batched) or nullptr
    const void* const* v_ptrs;
    void* const* k_out_ptrs;
    void* const* v_out_ptrs;
    const void* k_single;        // used when tables are null
    const void* v_single;
    void* k_out_single;
    void* v_out_single;
    const uint8_t* valid;        // [layers*Hkv, N]  (tok1 = 0: [layers*Hkv], one byte per head)
    int tok1;
    const int32_t* tile_base;    // [layers*Hkv, ntiles]
    const int32_t* seg_start;    // [layers*Hkv]
    int64_t in_head_stride_bytes;
    int Hkv, N, sink, klen, ntiles;
    int row_bytes;               // D * elem_bytes, multiple of 16
};

// LPR = lanes per row = row_bytes / 16 (power of two, 4..64)
template <int LPR>
__global__ __launch_bounds__(CP_THREADS) void compact_gather_kernel(CompactArgs a) {
    const int tile = blockIdx.x;
    const int h = blockIdx.y;
    const int layer = blockIdx.z;
    const int row = layer * a.Hkv + h;

    __shared__ uint16_t list[CT];
    __shared__ int wtot[CP_THREADS / WAVE];

    const uint8_t* vrow = a.valid + (int64_t)row * (a.tok1 ? a.N : 1);
    const int p0 = tile * CT + threadIdx.x * CP_PER_THREAD;
    const uint32_t bits = full_mask4(vrow, p0, a.sink, a.N, a.klen, a.tok1);
    const int c = __popc(bits);
    const int inc = wave_inclusive_scan(c);
    if (lane_id() == 63) wtot[threadIdx.x >> 6] = inc;
    __syncthreads();
    int pre = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) pre += wtot[w];
    const int total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    int o = pre + inc - c;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (bits & (1u << j)) list[o++] = (uint16_t)(threadIdx.x * CP_PER_THREAD + j);
    __syncthreads();
    if (total == 0) return;

    const char* ksrc = reinterpret_cast<const char*>(a.k_ptrs ? a.k_ptrs[layer] : a.k_single) +
                       (int64_t)h * a.in_head_stride_bytes + (int64_t)tile * CT * a.row_bytes;
    const char* vsrc = reinterpret_cast<const char*>(a.v_ptrs ? a.v_ptrs[layer] : a.v_single) +
                       (int64_t)h * a.in_head_stride_bytes + (int64_t)tile * CT * a.row_bytes;
    const int64_t dst_row0 = (int64_t)a.seg_start[row] + a.tile_base[(int64_t)row * a.ntiles + tile];
    char* kdst = reinterpret_cast<char*>(a.k_out_ptrs ? a.k_out_ptrs[layer] : a.k_out_single) + dst_row0 * a.row_bytes;
    char* vdst = reinterpret_cast<char*>(a.v_out_ptrs ? a.v_out_ptrs[layer] : a.v_out_single) + dst_row0 * a.row_bytes;

    constexpr int RPI = CP_THREADS / LPR;  // rows per block-iteration
    const int rsub = threadIdx.x / LPR;
    const int coff = (threadIdx.x % LPR) * 16;
    constexpr int UNROLL = 8;
    for (int j0 = 0; j0 < total; j0 += RPI * UNROLL) {
        u32x4 kv[UNROLL], vv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int j = j0 + u * RPI + rsub;
            if (j < total) {
                const int64_t so = (int64_t)list[j] * a.row_bytes + coff;
                // streamed exactly once: non-temporal, do not displace anything useful from L2 / MALL
                kv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ksrc + so));
                vv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vsrc + so));
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int j = j0 + u * RPI + rsub;
            if (j < total) {
                const int64_t dof = (int64_t)j * a.row_bytes + coff;
                __builtin_nontemporal_store(kv[u], reinterpret_cast<u32x4*>(kdst + dof));
                __builtin_nontemporal_store(vv[u], reinterpret_cast<u32x4*>(vdst + dof));
            }
        }
    }
}

// ---- reference-exact update_flatten_view (out-of-place rebuild) ----------------------------------
__global__ __launch_bounds__(256) void update_flatten_view_kernel(const char* __restrict__ cache,
                                                                 const char* __restrict__ state,
                                                                 const int32_t* __restrict__ headlens,
                                                                 const int32_t* __restrict__ cu_headlens, int t,
                                                                 int row_bytes, char* __restrict__ out) {
    const int h = blockIdx.y;
    const int64_t hl = headlens[h];
    const int64_t src_row0 = cu_headlens[h];
    const int64_t dst_row0 = src_row0 + (int64_t)h * t;          // rows inserted for earlier heads
    const int64_t ins_row0 = (int64_t)cu_headlens[h + 1] + (int64_t)h * t;  // insertion point (reference :34)
    const int64_t old_chunks = hl * row_bytes / 16;
    const int64_t new_chunks = (int64_t)t * row_bytes / 16;
    const u32x4* src = reinterpret_cast<const u32x4*>(cache + src_row0 * row_bytes);
    u32x4* dst = reinterpret_cast<u32x4*>(out + dst_row0 * row_bytes);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < old_chunks; i += stride) dst[i] = src[i];
    const u32x4* ssrc = reinterpret_cast<const u32x4*>(state + (int64_t)h * t * row_bytes);
    u32x4* idst = reinterpret_cast<u32x4*>(out + ins_row0 * row_bytes);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < new_chunks; i += stride) idst[i] = ssrc[i];
}

// ---- O(t) in-place append into per-head slack -----------------------------------------------------
// state rows may be strided (e.g. V straight out of the projection: [b, t, Hkv, D] viewed as [b, Hkv, t, D])
__global__ __launch_bounds__(256) void append_inplace_kernel(char* __restrict__ kc, char* __restrict__ vc,
                                                            const char* __restrict__ ks, const char* __restrict__ vs,
                                                            int64_t k_head_stride_bytes, int64_t k_row_stride_bytes,
                                                            int64_t v_head_stride_bytes, int64_t v_row_stride_bytes,
                                                            const int32_t* __restrict__ seg_start,
                                                            const int32_t* __restrict__ base_len, int len_offset,
                                                            int t, int row_bytes) {
    const int h = blockIdx.y;
    const int64_t dst_row0 = (int64_t)seg_start[h] + base_len[h] + len_offset;
    const int cpr = row_bytes / 16;
    const int64_t chunks = (int64_t)t * cpr;
    const char* k_src = ks + (int64_t)h * k_head_stride_bytes;
    const char* v_src = vs + (int64_t)h * v_head_stride_bytes;
    char* k_dst = kc + dst_row0 * row_bytes;
    char* v_dst = vc + dst_row0 * row_bytes;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += stride) {
        const int64_t row = i / cpr;
        const int c = (int)(i % cpr) * 16;
        *reinterpret_cast<u32x4*>(k_dst + row * row_bytes + c) = *reinterpret_cast<const u32x4*>(k_src + row * k_row_stride_bytes + c);
        *reinterpret_cast<u32x4*>(v_dst + row * row_bytes + c) = *reinterpret_cast<const u32x4*>(v_src + row * v_row_stride_bytes + c);
    }
}

// ---- append into the DENSE (pre-prune) cache: head h owns rows h*cache_head_rows .., the new rows go to row `fill` of every head.
// No device-resident metadata (the slack-layout append reads seg_start / base_len from the device): one launch per layer and
// chunk in the scoring loop, so the host side is a plain call with scalars.
__global__ __launch_bounds__(256) void dense_append_kernel(char* __restrict__ kc, char* __restrict__ vc,
                                                          const char* __restrict__ ks, const char* __restrict__ vs,
                                                          int64_t cache_head_stride_bytes, int64_t k_head_stride_bytes,
                                                          int64_t k_row_stride_bytes, int64_t v_head_stride_bytes,
                                                          int64_t v_row_stride_bytes, int fill, int t, int row_bytes) {
    const int h = blockIdx.y;
    const int cpr = row_bytes / 16;
    const int64_t chunks = (int64_t)t * cpr;
    const char* k_src = ks + (int64_t)h * k_head_stride_bytes;
    const char* v_src = vs + (int64_t)h * v_head_stride_bytes;
    char* k_dst = kc + (int64_t)h * cache_head_stride_bytes + (int64_t)fill * row_bytes;
    char* v_dst = vc + (int64_t)h * cache_head_stride_bytes + (int64_t)fill * row_bytes;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += stride) {
        const int64_t row = i / cpr;
        const int c = (int)(i % cpr) * 16;
        *reinterpret_cast<u32x4*>(k_dst + row * row_bytes + c) = *reinterpret_cast<const u32x4*>(k_src + row * k_row_stride_bytes + c);
        *reinterpret_cast<u32x4*>(v_dst + row * row_bytes + c) = *reinterpret_cast<const u32x4*>(v_src + row * v_row_stride_bytes + c);
    }
}

static int launch_gather(const CompactArgs& a, int layers, hipStream_t stream) {
    dim3 grid((unsigned)a.ntiles, (unsigned)a.Hkv, (unsigned)layers);
    dim3 block(CP_THREADS);
    ProfScope ps("compact_gather", stream);
    switch (a.row_bytes / 16) {
        case 4: hipLaunchKernelGGL(compact_gather_kernel<4>, grid, block, 0, stream, a); break;
        case 8: hipLaunchKernelGGL(compact_gather_kernel<8>, grid, block, 0, stream, a); break;
        case 16: hipLaunchKernelGGL(compact_gather_kernel<16>, grid, block, 0, stream, a); break;
        case 32: hipLaunchKernelGGL(compact_gather_kernel<32>, grid, block, 0, stream, a); break;
        case 64: hipLaunchKernelGGL(compact_gather_kernel<64>, grid, block, 0, stream, a); break;
        default:
            set_error("kvz_compact: row_bytes %d unsupported (D*elem_bytes must be 64..1024, power of two)", a.row_bytes);
            return KVZ_EUNSUPPORTED;
    }
    KVZ_CHECK_LAUNCH("compact_gather_kernel");
    return KVZ_OK;
}

}  // namespace kvz

using namespace kvz;

static inline int ntiles_of(int klen) { return (klen + CT - 1) / CT; }

extern "C" size_t kvz_compact_plan_bytes(int layers, int Hkv, int klen) {
    return (size_t)layers * Hkv * ntiles_of(klen) * sizeof(int32_t);
}

static int compact_plan_impl(const uint8_t* valid, int tok1, int layers, int Hkv, int N, int sink, int klen, int slack,
                             int32_t* len_k, int32_t* cu_len_k, int32_t* seg_start, int32_t* max_len_k,
                             int32_t* tile_base, kvz_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    KVZ_REQUIRE(valid && len_k && cu_len_k && seg_start && max_len_k && tile_base, KVZ_EINVAL,
                "kvz_compact_plan: null pointer");
    KVZ_REQUIRE(layers > 0 && Hkv > 0 && Hkv <= 1024 && N >= 0 && sink >= 0 && slack >= 0, KVZ_EINVAL,
                "kvz_compact_plan: bad shape");
    KVZ_REQUIRE(klen >= sink + N && klen > 0, KVZ_EINVAL, "kvz_compact_plan: klen %d < sink %d + N %d", klen, sink, N);
    KVZ_REQUIRE((int64_t)layers * Hkv <= 65535, KVZ_EINVAL, "kvz_compact_plan: too many rows");
    const int nt = ntiles_of(klen);
    hipLaunchKernelGGL(compact_tile_count_kernel, dim3(nt, layers * Hkv), dim3(CP_THREADS), 0, stream, valid, N, sink,
                       klen, nt, tok1, tile_base);
    KVZ_CHECK_LAUNCH("compact_tile_count_kernel");
    hipLaunchKernelGGL(compact_plan_scan_kernel, dim3(layers), dim3(CP_THREADS), 0, stream, Hkv, nt, slack, tile_base,
                       len_k, cu_len_k, seg_start, max_len_k);
    KVZ_CHECK_LAUNCH("compact_plan_scan_kernel");
    return KVZ_OK;
}

extern "C" int kvz_compact_plan(const uint8_t* valid, int layers, int Hkv, int N, int sink, int klen, int slack,
                                int32_t* len_k, int32_t* cu_len_k, int32_t* seg_start, int32_t* max_len_k,
                                int32_t* tile_base, kvz_stream_t stream) {
    return compact_plan_impl(valid, 1, layers, Hkv, N, sink, klen, slack, len_k, cu_len_k, seg_start, max_len_k, tile_base, stream);
}
extern "C" int kvz_compact_plan_heads(const uint8_t* valid_heads, int layers, int Hkv, int N, int sink, int klen, int slack,
                                      int32_t* len_k, int32_t* cu_len_k, int32_t* seg_start, int32_t* max_len_k,
                                      int32_t* tile_base, kvz_stream_t stream) {
    return compact_plan_impl(valid_heads, 0, layers, Hkv, N, sink, klen, slack, len_k, cu_len_k, seg_start, max_len_k, tile_base, stream);
}

static int check_rows(const char* who, int D, int elem_bytes, int64_t in_head_stride) {
    KVZ_REQUIRE(elem_bytes == 1 || elem_bytes == 2 || elem_bytes == 4, KVZ_EINVAL, "%s: bad elem_bytes %d", who, elem_bytes);
    const int rb = D * elem_bytes;
    KVZ_REQUIRE(rb >= 64 && rb <= 1024 && (rb & (rb - 1)) == 0, KVZ_EUNSUPPORTED,
                "%s: D*elem_bytes = %d must be a power of two in [64,1024]", who, rb);
    KVZ_REQUIRE((in_head_stride * elem_bytes) % 16 == 0, KVZ_EINVAL, "%s: head stride not 16-byte aligned", who);
    return KVZ_OK;
}

extern "C" int kvz_compact_layer(const void* k, const void* v, int64_t in_head_stride, const uint8_t* valid,
                                 const int32_t* tile_base, const int32_t* seg_start, int Hkv, int N, int sink,
                                 int klen, int D, int elem_bytes, void* k_out, void* v_out, kvz_stream_t stream_) {
    KVZ_REQUIRE(k && v && valid && tile_base && seg_start && k_out && v_out, KVZ_EINVAL, "kvz_compact_layer: null pointer");
    KVZ_REQUIRE(aligned16(k) && aligned16(v) && aligned16(k_out) && aligned16(v_out), KVZ_EINVAL,
                "kvz_compact_layer: K/V pointers must be 16-byte aligned");
    KVZ_REQUIRE(klen >= sink + N && Hkv > 0, KVZ_EINVAL, "kvz_compact_layer: bad shape");
    int rc = check_rows("kvz_compact_layer", D, elem_bytes, in_head_stride);
    if (rc) return rc;
    CompactArgs a{};
    a.k_single = k; a.v_single = v; a.k_out_single = k_out; a.v_out_single = v_out;
    a.valid = valid; a.tok1 = 1; a.tile_base = tile_base; a.seg_start = seg_start;
    a.in_head_stride_bytes = in_head_stride * elem_bytes;
    a.Hkv = Hkv; a.N = N; a.sink = sink; a.klen = klen; a.ntiles = ntiles_of(klen);
    a.row_bytes = D * elem_bytes;
    return launch_gather(a, 1, (hipStream_t)stream_);
}

static int compact_layers_impl(const void* const* k_ptrs, const void* const* v_ptrs, int64_t in_head_stride,
                               const uint8_t* valid, int tok1, const int32_t* tile_base, const int32_t* seg_start, int layers,
                               int Hkv, int N, int sink, int klen, int D, int elem_bytes, void* const* k_out_ptrs,
                               void* const* v_out_ptrs, kvz_stream_t stream_) {
    KVZ_REQUIRE(k_ptrs && v_ptrs && valid && tile_base && seg_start && k_out_ptrs && v_out_ptrs, KVZ_EINVAL,
                "kvz_compact_layers: null pointer");
    KVZ_REQUIRE(klen >= sink + N && Hkv > 0 && layers > 0 && layers <= 65535, KVZ_EINVAL, "kvz_compact_layers: bad shape");
    int rc = check_rows("kvz_compact_layers", D, elem_bytes, in_head_stride);
    if (rc) return rc;
    CompactArgs a{};
    a.k_ptrs = k_ptrs; a.v_ptrs = v_ptrs; a.k_out_ptrs = k_out_ptrs; a.v_out_ptrs = v_out_ptrs;
    a.valid = valid; a.tok1 = tok1; a.tile_base = tile_base; a.seg_start = seg_start;
    a.in_head_stride_bytes = in_head_stride * elem_bytes;
    a.Hkv = Hkv; a.N = N; a.sink = sink; a.klen = klen; a.ntiles = ntiles_of(klen);
    a.row_bytes = D * elem_bytes;
    return launch_gather(a, layers, (hipStream_t)stream_);
}

extern "C" int kvz_compact_layers(const void* const* k_ptrs, const void* const* v_ptrs, int64_t in_head_stride,
                                  const uint8_t* valid, const int32_t* tile_base, const int32_t* seg_start, int layers,
                                  int Hkv, int N, int sink, int klen, int D, int elem_bytes, void* const* k_out_ptrs,
                                  void* const* v_out_ptrs, kvz_stream_t stream) {
    return compact_layers_impl(k_ptrs, v_ptrs, in_head_stride, valid, 1, tile_base, seg_start, layers, Hkv, N, sink, klen, D,
                               elem_bytes, k_out_ptrs, v_out_ptrs, stream);
}
// head-level eviction: `valid_heads` holds ONE byte per (layer, head); a kept head moves all of its rows, a dropped head
// only its sink rows (and whatever follows the context) - whole-segment copies, no per-token mask is ever materialised
extern "C" int kvz_compact_layers_heads(const void* const* k_ptrs, const void* const* v_ptrs, int64_t in_head_stride,
                                        const uint8_t* valid_heads, const int32_t* tile_base, const int32_t* seg_start,
                                        int layers, int Hkv, int N, int sink, int klen, int D, int elem_bytes,
                                        void* const* k_out_ptrs, void* const* v_out_ptrs, kvz_stream_t stream) {
    return compact_layers_impl(k_ptrs, v_ptrs, in_head_stride, valid_heads, 0, tile_base, seg_start, layers, Hkv, N, sink, klen,
                               D, elem_bytes, k_out_ptrs, v_out_ptrs, stream);
}

extern "C" int kvz_update_flatten_view(const void* cache, const void* state, const int32_t* headlens,
                                       const int32_t* cu_headlens, int Hkv, int t, int D, int elem_bytes, void* out,
                                       kvz_stream_t stream_) {
    KVZ_REQUIRE(cache && state && headlens && cu_headlens && out, KVZ_EINVAL, "kvz_update_flatten_view: null pointer");
    KVZ_REQUIRE(Hkv > 0 && Hkv <= 65535 && t >= 0, KVZ_EINVAL, "kvz_update_flatten_view: bad shape");
    KVZ_REQUIRE(aligned16(cache) && aligned16(state) && aligned16(out), KVZ_EINVAL,
                "kvz_update_flatten_view: pointers must be 16-byte aligned");
    const int rb = D * elem_bytes;
    KVZ_REQUIRE(rb > 0 && rb % 16 == 0, KVZ_EUNSUPPORTED, "kvz_update_flatten_view: row bytes %d not a multiple of 16", rb);
    hipLaunchKernelGGL(update_flatten_view_kernel, dim3(256, Hkv), dim3(256), 0, (hipStream_t)stream_,
                       reinterpret_cast<const char*>(cache), reinterpret_cast<const char*>(state), headlens,
                       cu_headlens, t, rb, reinterpret_cast<char*>(out));
    KVZ_CHECK_LAUNCH("update_flatten_view_kernel");
    return KVZ_OK;
}

extern "C" int kvz_append_inplace(void* k_cache, void* v_cache, const void* k_state, const void* v_state,
                                  int64_t k_head_stride, int64_t k_row_stride, int64_t v_head_stride,
                                  int64_t v_row_stride, const int32_t* seg_start, const int32_t* base_len,
                                  int len_offset, int Hkv, int t, int D, int elem_bytes, kvz_stream_t stream_) {
    KVZ_REQUIRE(k_cache && v_cache && k_state && v_state && seg_start && base_len, KVZ_EINVAL,
                "kvz_append_inplace: null pointer");
    KVZ_REQUIRE(Hkv > 0 && Hkv <= 65535 && t > 0, KVZ_EINVAL, "kvz_append_inplace: bad shape");
    const int rb = D * elem_bytes;
    KVZ_REQUIRE(rb > 0 && rb % 16 == 0, KVZ_EUNSUPPORTED, "kvz_append_inplace: row bytes %d not a multiple of 16", rb);
    KVZ_REQUIRE(aligned16(k_cache) && aligned16(v_cache) && aligned16(k_state) && aligned16(v_state), KVZ_EINVAL,
                "kvz_append_inplace: pointers must be 16-byte aligned");
    KVZ_REQUIRE((k_head_stride * elem_bytes) % 16 == 0 && (k_row_stride * elem_bytes) % 16 == 0 &&
                    (v_head_stride * elem_bytes) % 16 == 0 && (v_row_stride * elem_bytes) % 16 == 0,
                KVZ_EINVAL, "kvz_append_inplace: state strides must be multiples of 16 bytes");
    int64_t chunks = (int64_t)t * rb / 16;
    int bx = (int)((chunks + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(append_inplace_kernel, dim3(bx, Hkv), dim3(256), 0, (hipStream_t)stream_,
                       reinterpret_cast<char*>(k_cache), reinterpret_cast<char*>(v_cache),
                       reinterpret_cast<const char*>(k_state), reinterpret_cast<const char*>(v_state),
                       k_head_stride * elem_bytes, k_row_stride * elem_bytes, v_head_stride * elem_bytes,
                       v_row_stride * elem_bytes, seg_start, base_len, len_offset, t, rb);
    KVZ_CHECK_LAUNCH("append_inplace_kernel");
    return KVZ_OK;
}

extern "C" int kvz_dense_append(void* k_cache, void* v_cache, int64_t cache_head_stride, int fill, const void* k_state,
                                const void* v_state, int64_t k_head_stride, int64_t k_row_stride, int64_t v_head_stride,
                                int64_t v_row_stride, int Hkv, int t, int D, int elem_bytes, kvz_stream_t stream_) {
    KVZ_REQUIRE(k_cache && v_cache && k_state && v_state, KVZ_EINVAL, "kvz_dense_append: null pointer");
    KVZ_REQUIRE(Hkv > 0 && Hkv <= 65535 && t > 0 && fill >= 0, KVZ_EINVAL, "kvz_dense_append: bad shape");
    const int rb = D * elem_bytes;
    KVZ_REQUIRE(rb > 0 && rb % 16 == 0, KVZ_EUNSUPPORTED, "kvz_dense_append: row bytes %d not a multiple of 16", rb);
    KVZ_REQUIRE((int64_t)(fill + t) * D <= cache_head_stride, KVZ_EINVAL, "kvz_dense_append: rows %d..%d exceed the head capacity",
                fill, fill + t);
    KVZ_REQUIRE(aligned16(k_cache) && aligned16(v_cache) && aligned16(k_state) && aligned16(v_state), KVZ_EINVAL,
                "kvz_dense_append: pointers must be 16-byte aligned");
    KVZ_REQUIRE((cache_head_stride * elem_bytes) % 16 == 0 && (k_head_stride * elem_bytes) % 16 == 0 &&
                    (k_row_stride * elem_bytes) % 16 == 0 && (v_head_stride * elem_bytes) % 16 == 0 &&
                    (v_row_stride * elem_bytes) % 16 == 0,
                KVZ_EINVAL, "kvz_dense_append: strides must be multiples of 16 bytes");
    int64_t chunks = (int64_t)t * rb / 16;
    int bx = (int)((chunks + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(dense_append_kernel, dim3(bx, Hkv), dim3(256), 0, (hipStream_t)stream_, reinterpret_cast<char*>(k_cache),
                       reinterpret_cast<char*>(v_cache), reinterpret_cast<const char*>(k_state),
                       reinterpret_cast<const char*>(v_state), cache_head_stride * elem_bytes, k_head_stride * elem_bytes,
                       k_row_stride * elem_bytes, v_head_stride * elem_bytes, v_row_stride * elem_bytes, fill, t, rb);
    KVZ_CHECK_LAUNCH("dense_append_kernel");
    return KVZ_OK;
}


// ---- measurement hook: a plain 16-bytes-per-lane copy kernel (include/kvzip_hip_debug.h) -----------------------------------------------
namespace kvz {
template <bool NT>
__global__ __launch_bounds__(256) void copy16_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t n16) {
    constexpr int U = 4;   // 16-byte moves in flight per lane
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n16; i += stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + (size_t)u * 256 < n16) v[u] = NT ? __builtin_nontemporal_load(src + i + (size_t)u * 256) : src[i + (size_t)u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + (size_t)u * 256 < n16) {
                if (NT) __builtin_nontemporal_store(v[u], dst + i + (size_t)u * 256);
                else dst[i + (size_t)u * 256] = v[u];
            }
    }
}
}  // namespace kvz

extern "C" int kvz_debug_copy_kernel(void* dst, const void* src, size_t nbytes, int variant, kvz_stream_t stream_) {
    KVZ_REQUIRE(dst && src && aligned16(dst) && aligned16(src) && nbytes % 16 == 0, KVZ_EINVAL, "kvz_debug_copy_kernel: 16-byte aligned buffers and size");
    if (nbytes == 0) return KVZ_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const size_t n16 = nbytes / 16;
    size_t blocks = (n16 + 1023) / 1024;
    const size_t cap = (size_t)device_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (variant == 1) hipLaunchKernelGGL((kvz::copy16_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<kvz::u32x4*>(dst), reinterpret_cast<const kvz::u32x4*>(src), n16);
    else hipLaunchKernelGGL((kvz::copy16_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<kvz::u32x4*>(dst), reinterpret_cast<const kvz::u32x4*>(src), n16);
    KVZ_CHECK_LAUNCH("copy16_kernel");
    return KVZ_OK;
}
