// kvz_attn.hip — variable-length (ragged per KV head) post-prune attention for gfx950.
//
// Replaces the reference's call into flash_attn_varlen_func (attention/attn.py:56-73; flash-attn
// 2.7.4.post1, third party, not vendored): every KV head is one ragged "sequence", its G query heads
// are MQA heads, causal mask bottom-right aligned.
//
// Decode (q_len = 1) is HBM-bound: every kept K and V row is read exactly once per generated token.
// Design: split-K "flash decoding" in ONE launch.
//   grid = (work items, row tiles of 16 query rows); a block owns one key range of one head (ranges are cut from the SUM of
//   the ragged head lengths), its 8 waves stride over 32-key tiles of the range.
//   S^T = K.Q^T on v_mfma_f32_16x16x32 with K rows loaded straight from HBM as the A operand (16-byte
//   contiguous per lane, no LDS); softmax state lives in registers (lane = one query row);
//   O^T += V^T.P^T with V staged through a per-wave padded LDS tile and read back transposed by
//   ds_read_b64_tr_b16.  The last block of a head to finish merges the head's partials.
#include "kvz_common.h"


namespace kvz {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct HalfTraits;
template <> struct HalfTraits<_Float16> {
    typedef h8 v8;
    __device__ static inline f4 mfma(v8 a, v8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct HalfTraits<__bf16> {
    typedef b8 v8;
    __device__ static inline f4 mfma(v8 a, v8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

constexpr int AT_THREADS = 512;
constexpr int AT_WAVES = 8;
constexpr int AT_KT = 32;           // keys per wave-tile
constexpr int AT_RT = 16;           // query rows per block (MFMA N)

template <int D> struct AttnCfg {
    static constexpr int ROW_BYTES = D * 2;
    static constexpr int VSTRIDE = ROW_BYTES + 32;          // padded LDS row: conflict-free tr reads
    static constexpr int WAVE_LDS = AT_KT * VSTRIDE;        // bytes of V staging per wave
    static constexpr int KK = D / 32;                       // MFMA k-steps over the head dim
    static constexpr int DB = D / 16;                       // 16-wide output column blocks
    static constexpr int CPR = ROW_BYTES / 16;              // 16-byte chunks per row
    static constexpr int VLOADS = AT_KT * CPR / WAVE;       // 16-byte V loads per lane per tile
};

// APPEND: the decode step's O(1) cache append is done by the attention kernel.  k_len_offset already counts the new token; the
// block that owns the last key range of a head first copies the new K and V row of that head from the state tensors into
// the cache (row k_start[h] + len - 1) and then reads it back like any other key (no other block touches that row).

// ---- round 2: ragged-aware work items -------------------------------------------------------------------------------------
// Work items are cut from the SUM of the head lengths (a head of 120 k keys gets 30 x the items of a head of 4 k keys; a
// dropped head of a head-level cache gets one), not from Hkv x the longest head: every block derives its (head, key range)
// from the lengths (kernel arguments when the host knows them, else scalar loads).  The partials of a head are merged by a
// second, split-parallel kernel.  (Merging them inside this kernel - last block of a head to arrive, write-through partials,
// one device-scope counter, one acquire fence - was built and measured in round 2: correct, but the chain store drain ->
// atomic -> fence -> two dependent rounds of reads costs 10-12 us of serial tail against 5.4 + 2 us for the separate launch;
// see profiles/r2_attn_fused_vs_split.txt.)
constexpr int AT_MAXH = 64;
struct HeadMeta { int32_t start[AT_MAXH]; int32_t len[AT_MAXH]; };

template <typename T, int D, bool APPEND, bool HOSTMETA>
__global__ __launch_bounds__(AT_THREADS) void varlen_attn_split2_kernel(
    const T* __restrict__ q, const T* k, const T* v, const int32_t* __restrict__ k_start, const int32_t* __restrict__ k_len,
    int k_len_offset, const int32_t* __restrict__ k_len_offset_dev, HeadMeta hm, int Hkv, int G, int q_len, int target_items, float scale,
    int causal, float* part_o, float* part_ml, uint32_t* counters, int n_rtiles, T* __restrict__ out, const T* __restrict__ k_new,
    const T* __restrict__ v_new, int64_t kn_head_stride, int64_t vn_head_stride) {
    typedef AttnCfg<D> C;
    typedef typename HalfTraits<T>::v8 v8;
    // (the part of the appended-token count that lives on the DEVICE: a captured generation step is replayed with unchanged arguments)
    if (k_len_offset_dev) k_len_offset += *k_len_offset_dev;
    auto len_of = [&](int hh) -> int { return (HOSTMETA ? hm.len[hh] : k_len[hh]) + k_len_offset; };
    // ---- this block's work item: (head, key range) from the ragged lengths (all wave-uniform scalar work) ----
    int total = 0;
    for (int hh = 0; hh < Hkv; ++hh) total += len_of(hh);
    int chunk = (total + target_items - 1) / target_items;
    chunk = (chunk + 127) / 128 * 128;
    if (chunk < 128) chunk = 128;
    int h = -1, split = 0, first_item = 0;
    {
        int b = blockIdx.x, items = 0;
        for (int hh = 0; hh < Hkv; ++hh) {
            const int n = max(1, (len_of(hh) + chunk - 1) / chunk);  // (an empty head still gets one item: it writes zeros)
            if (h < 0) {
                if (b < n) { h = hh; split = b; first_item = items; }
                else b -= n;
            }
            items += n;
        }
    }
    if (h < 0) return;  // beyond the last item
    const int rt = blockIdx.y;
    const int len = len_of(h);
    const int64_t seg = HOSTMETA ? hm.start[h] : k_start[h];
    const int c0 = split * chunk;
    const int c1 = min(len, c0 + chunk);
    if (APPEND && c1 == len) {
        constexpr int CH = D * 2 / 16;  // 16-byte chunks per row
        if ((int)threadIdx.x < 2 * CH) {
            const bool is_v = (int)threadIdx.x >= CH;
            const int c = (int)threadIdx.x - (is_v ? CH : 0);
            const char* src = reinterpret_cast<const char*>(is_v ? v_new + h * vn_head_stride : k_new + h * kn_head_stride) + c * 16;
            char* dst = const_cast<char*>(reinterpret_cast<const char*>(is_v ? v : k)) + (seg + len - 1) * (D * 2) + c * 16;
            *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
        }
        __syncthreads();  // (workgroup-scope release/acquire: the row is visible to the loads of this block below)
    }
    const int R = q_len * G;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l15 = lane & 15, quad = lane >> 4;

    __shared__ __attribute__((aligned(16))) char lds[AT_WAVES * C::WAVE_LDS];
    char* wl = lds + wave * C::WAVE_LDS;

    // ---- stationary Q fragment (B operand of S^T = K.Q^T): col = query row, k = head-dim slice ----
    const int qrow = rt * AT_RT + l15;
    const bool qvalid = qrow < R;
    v8 qf[C::KK];
    {
        const T* qp = q + ((int64_t)h * R + (qvalid ? qrow : 0)) * D + quad * 8;
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
            u32x4 raw = *reinterpret_cast<const u32x4*>(qp + kk * 32);
            if (!qvalid) raw = u32x4{0, 0, 0, 0};
            qf[kk] = __builtin_bit_cast(v8, raw);
        }
    }
    int limit = causal ? (qrow / G) + len - q_len : len - 1;  // last visible key of this lane's query row (bottom-right aligned)
    if (!qvalid) limit = -1;

    const char* kbase = reinterpret_cast<const char*>(k) + seg * C::ROW_BYTES;
    const char* vbase = reinterpret_cast<const char*>(v) + seg * C::ROW_BYTES;

    float m_run = -INFINITY, l_run = 0.f;
    f4 o[C::DB];
#pragma unroll
    for (int i = 0; i < C::DB; ++i) o[i] = f4{0.f, 0.f, 0.f, 0.f};
    const float sl2 = scale * 1.44269504088896340736f;  // work in the exp2 domain

    auto load_tile = [&](u32x4 (&kr)[2][C::KK], u32x4 (&vr)[C::VLOADS], int t0) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            int key = t0 + sub * 16 + l15;
            key = key < len ? key : len - 1;  // clamp (masked below)
            const char* kp = kbase + (int64_t)key * C::ROW_BYTES + quad * 16;
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) kr[sub][kk] = *reinterpret_cast<const u32x4*>(kp + kk * 64);
        }
#pragma unroll
        for (int it = 0; it < C::VLOADS; ++it) {
            const int c = it * WAVE + lane;
            int key = t0 + c / C::CPR;
            key = key < len ? key : len - 1;
            vr[it] = *reinterpret_cast<const u32x4*>(vbase + (int64_t)key * C::ROW_BYTES + (c % C::CPR) * 16);
        }
    };
    u32x4 kraw[2][C::KK], vraw[C::VLOADS], knext[2][C::KK], vnext[C::VLOADS];
    // one 32-key tile of this wave: S^T = K.Q^T, online softmax, V through the wave's LDS tile, O^T += V^T.P^T
    auto compute_tile = [&](const u32x4 (&kr)[2][C::KK], const u32x4 (&vr)[C::VLOADS], int t0) __attribute__((always_inline)) {
        f4 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            s[sub] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk)
                s[sub] = HalfTraits<T>::mfma(__builtin_bit_cast(v8, kr[sub][kk]), qf[kk], s[sub]);
        }
#pragma unroll
        for (int it = 0; it < C::VLOADS; ++it) {
            const int c = it * WAVE + lane;
            *reinterpret_cast<u32x4*>(wl + (c / C::CPR) * C::VSTRIDE + (c % C::CPR) * 16) = vr[it];
        }
        float sv[8];
        float tmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key = t0 + (j >> 2) * 16 + quad * 4 + (j & 3);
            float x = s[j >> 2][j & 3] * sl2;
            x = (key <= limit && key < c1) ? x : -INFINITY;
            sv[j] = x;
            tmax = fmaxf(tmax, x);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = exp2f(m_run - m_safe);
        float psum = 0.f;
        v8 pb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = exp2f(sv[j] - m_safe);
            psum += p;
            pb[j] = (T)p;
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < C::DB; ++i) {
            o[i][0] *= alpha; o[i][1] *= alpha; o[i][2] *= alpha; o[i][3] *= alpha;
        }
        const char* trp = wl + (quad * 4 + (l15 >> 2)) * C::VSTRIDE + (l15 & 3) * 8;
#pragma unroll
        for (int db = 0; db < C::DB; ++db) {
            s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(trp + db * 32));
            s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(trp + 16 * C::VSTRIDE + db * 32));
            s8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            o[db] = HalfTraits<T>::mfma(__builtin_bit_cast(v8, both), pb, o[db]);
        }
    };
    // The wave's tiles, one requested AHEAD of the one being computed.  Round 4: the steady-state loop issues the next tile's loads
    // UNCONDITIONALLY and the last tile is computed after the loop.  With the request inside `if (tn < c1)` the compiler has to place
    // waits that are right on both paths - s_waitcnt vmcnt(7) .. vmcnt(0) before the V rows of the CURRENT tile go to LDS - which
    // drains the tile that has just been requested in the middle of every iteration: no load was ever in flight while a wave
    // computed, and the kernel ran at 21 us where the bare read stream takes 12.5 (profiles/r4_probe_stream.txt).
    const int t_first = c0 + wave * AT_KT;
    constexpr int T_STEP = AT_WAVES * AT_KT;
    if (t_first < c1) {
        load_tile(kraw, vraw, t_first);
        int t0 = t_first;
#pragma unroll 1
        for (; t0 + T_STEP < c1; t0 += T_STEP) {
            load_tile(knext, vnext, t0 + T_STEP);
            compute_tile(kraw, vraw, t0);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int kk = 0; kk < C::KK; ++kk) kraw[sub][kk] = knext[sub][kk];
#pragma unroll
            for (int it = 0; it < C::VLOADS; ++it) vraw[it] = vnext[it];
        }
        compute_tile(kraw, vraw, t0);
    }

    // ---- merge the 8 waves of the block through LDS (each wave writes only its own region) ----------
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    // Rows of D + 4 floats: with a stride of D floats (512 B) the 16 lanes of a quad - 16 query rows, the same output columns -
    // all hit one LDS bank (round 5: 16-way conflict on every store, ~1 us per block, found by ablation: profiles/r5_decode_ablation.txt);
    // the padded row puts the 16-byte stores of 8 consecutive query rows on 8 different bank groups.  Only valid query rows are
    // written and merged (decode: G of the 16).
    constexpr int WSTR = D + 4;                          // floats per staged row
    static_assert((AT_RT * WSTR + 2 * AT_RT) * 4 <= C::WAVE_LDS, "the wave's V tile holds its merge staging");
    float* wo = reinterpret_cast<float*>(wl);            // [16 q][D + 4]
    float* wm = wo + AT_RT * WSTR;                       // [16] m, [16] l
    const int rows_valid = min(AT_RT, R - rt * AT_RT);
    if (l15 < rows_valid) {
#pragma unroll
        for (int db = 0; db < C::DB; ++db) *reinterpret_cast<f4*>(wo + l15 * WSTR + db * 16 + quad * 4) = o[db];
        if (quad == 0) { wm[l15] = m_run; wm[16 + l15] = l_run; }
    }
    __syncthreads();

    const int64_t pbase = ((int64_t)(first_item + split) * n_rtiles + rt) * AT_RT;  // partial slot of this item
    for (int e = threadIdx.x; e < rows_valid * (D / 4); e += AT_THREADS) {
        const int qq = e / (D / 4), d = (e % (D / 4)) * 4;
        float mw[AT_WAVES], M = -INFINITY;
#pragma unroll
        for (int w = 0; w < AT_WAVES; ++w) {
            mw[w] = reinterpret_cast<const float*>(lds + w * C::WAVE_LDS)[AT_RT * WSTR + qq];
            M = fmaxf(M, mw[w]);
        }
        f4 acc4 = f4{0.f, 0.f, 0.f, 0.f};
        float lsum = 0.f;
#pragma unroll
        for (int w = 0; w < AT_WAVES; ++w) {
            const float* pw = reinterpret_cast<const float*>(lds + w * C::WAVE_LDS);
            const float wgt = (mw[w] == -INFINITY) ? 0.f : exp2f(mw[w] - M);
            const f4 ov = *reinterpret_cast<const f4*>(pw + qq * WSTR + d);
            acc4[0] += wgt * ov[0]; acc4[1] += wgt * ov[1]; acc4[2] += wgt * ov[2]; acc4[3] += wgt * ov[3];
            lsum += wgt * pw[AT_RT * WSTR + 16 + qq];
        }
        *reinterpret_cast<f4*>(part_o + (pbase + qq) * D + d) = acc4;
        if (d == 0) *reinterpret_cast<float2*>(part_ml + (pbase + qq) * 2) = make_float2(M, lsum);
    }
}

// merge the partials of the key ranges.  grid = (Hkv, row tiles, 16 query rows); block = 1024 threads = (1024/D) slices of the
// partials x D columns.  Every load of the main loop is independent (split-parallel): throughput-, not latency-bound.
constexpr int CB_THREADS = 1024;
template <typename T, int D, bool HOSTMETA>
__global__ __launch_bounds__(CB_THREADS) void varlen_attn_combine2_kernel(const float* __restrict__ part_o,
                                                                         const float* __restrict__ part_ml,
                                                                         const int32_t* __restrict__ k_len, int k_len_offset,
                                                                         const int32_t* __restrict__ k_len_offset_dev,
                                                                         HeadMeta hm, int Hkv, int G, int q_len, int target_items,
                                                                         int n_rtiles, T* __restrict__ out) {
    constexpr int SLICES = CB_THREADS / D;
    if (k_len_offset_dev) k_len_offset += *k_len_offset_dev;
    const int h = blockIdx.x, rt = blockIdx.y, qq = blockIdx.z;
    const int R = q_len * G;
    if (rt * AT_RT + qq >= R) return;
    auto len_of = [&](int hh) -> int { return (HOSTMETA ? hm.len[hh] : k_len[hh]) + k_len_offset; };
    int total = 0;
    for (int hh = 0; hh < Hkv; ++hh) total += len_of(hh);
    int chunk = (total + target_items - 1) / target_items;
    chunk = (chunk + 127) / 128 * 128;
    if (chunk < 128) chunk = 128;
    int first_item = 0;
    for (int hh = 0; hh < h; ++hh) first_item += max(1, (len_of(hh) + chunk - 1) / chunk);
    const int nsp = max(1, (len_of(h) + chunk - 1) / chunk);
    const int64_t base = ((int64_t)first_item * n_rtiles + rt) * AT_RT + qq;  // + s * n_rtiles * AT_RT
    const int64_t istride = (int64_t)n_rtiles * AT_RT;
    const int tid = threadIdx.x;
    const int d = tid % D, slice = tid / D;

    __shared__ float s_red[CB_THREADS / WAVE];
    __shared__ float s_acc[SLICES][D];
    __shared__ float s_M, s_L;

    constexpr int PER = 8;  // partials per thread on the fast path
    if (nsp <= PER * SLICES) {
        // ONE round trip to memory: every thread issues the loads of its (up to 8) partials - statistics and its column -
        // before anything depends on them; the row maximum and the denominator are then block reductions over registers.
        float2 ml[PER];
        float ov[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int s2 = slice + i * SLICES;
            const int64_t pi = base + (int64_t)min(s2, nsp - 1) * istride;
            ml[i] = *reinterpret_cast<const float2*>(part_ml + pi * 2);
            ov[i] = part_o[pi * D + d];
            if (s2 >= nsp) ml[i] = make_float2(-INFINITY, 0.f);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < PER; ++i) mx = fmaxf(mx, ml[i].x);
        mx = wave_reduce_max(mx);
        if ((tid & 63) == 0) s_red[tid >> 6] = mx;
        __syncthreads();
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < CB_THREADS / WAVE; ++w) M = fmaxf(M, s_red[w]);
        float acc = 0.f, lp = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const float wgt = (ml[i].x == -INFINITY) ? 0.f : exp2f(ml[i].x - M);
            acc += wgt * ov[i];
            lp += wgt * ml[i].y;
        }
        s_acc[slice][d] = acc;
        __syncthreads();  // (also: everybody has read s_red)
        if (d == 0) s_red[slice] = lp;  // one thread per slice holds that slice's share of the denominator
        __syncthreads();
        if (tid < D) {
            float a = 0.f, l = 0.f;
#pragma unroll
            for (int sl = 0; sl < SLICES; ++sl) { a += s_acc[sl][tid]; l += s_red[sl]; }
            out[((int64_t)h * R + rt * AT_RT + qq) * D + tid] = (T)((l > 0.f) ? a / l : 0.f);
        }
        return;
    }
    // general path (more partials than the fast path holds in registers): three dependent passes
    float mx = -INFINITY;
    for (int s2 = tid; s2 < nsp; s2 += CB_THREADS) mx = fmaxf(mx, part_ml[(base + s2 * istride) * 2]);
    mx = wave_reduce_max(mx);
    if ((tid & 63) == 0) s_red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        float m2 = -INFINITY;
#pragma unroll
        for (int w = 0; w < CB_THREADS / WAVE; ++w) m2 = fmaxf(m2, s_red[w]);
        s_M = m2;
    }
    __syncthreads();
    const float M = s_M;
    float lp = 0.f;
    for (int s2 = tid; s2 < nsp; s2 += CB_THREADS) {
        const float2 ml = *reinterpret_cast<const float2*>(part_ml + (base + s2 * istride) * 2);
        lp += (ml.x == -INFINITY) ? 0.f : exp2f(ml.x - M) * ml.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lp += __shfl_xor(lp, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = lp;
    __syncthreads();
    if (tid == 0) {
        float l2 = 0.f;
#pragma unroll
        for (int w = 0; w < CB_THREADS / WAVE; ++w) l2 += s_red[w];
        s_L = l2;
    }
    float acc = 0.f;
    for (int s2 = slice; s2 < nsp; s2 += SLICES) {
        const int64_t pi = base + s2 * istride;
        const float ms = part_ml[pi * 2];
        const float wgt = (ms == -INFINITY) ? 0.f : exp2f(ms - M);
        acc += wgt * part_o[pi * D + d];
    }
    s_acc[slice][d] = acc;
    __syncthreads();
    if (tid < D) {
        float a = 0.f;
#pragma unroll
        for (int sl = 0; sl < SLICES; ++sl) a += s_acc[sl][tid];
        const float l = s_L;
        out[((int64_t)h * R + rt * AT_RT + qq) * D + tid] = (T)((l > 0.f) ? a / l : 0.f);
    }
}

__global__ void add_i32_kernel(int32_t* p, int delta) { *p += delta; }

// work items a decode call is cut into.  Default (knob 0): CUs - Hkv (256 - Hkv on MI355X), the most that can never exceed one block per CU (every head
// rounds its item count up, so a call has at most target + Hkv items; one item beyond 256 costs a second round of blocks: 25 -> 34 us
// per layer, profiles/r4_decode_cold_probe.txt).  Round 2 had settled on 192 with a probe whose single 80-MB cache lived in the
// infinity cache; streaming a whole model's caches from cold HBM, 252 items are 3 % faster than 192.
static inline int attn_items(int Hkv) {
    const int t = tunable(TUNE_ATTN_ITEMS);
    if (t > 0) return t;
    const int cus = device_cus();
    return Hkv < cus * 3 / 4 ? cus - Hkv : cus / 4;
}
static inline size_t align256a(size_t x) { return (x + 255) & ~(size_t)255; }
struct AttnWs { uint32_t* counters; float* part_ml; float* part_o; size_t bytes; };
static inline AttnWs attn_ws(void* ws, int Hkv, int n_rtiles, int D) {
    const size_t items = (size_t)attn_items(Hkv) + Hkv;
    AttnWs w;
    char* p = reinterpret_cast<char*>(ws);
    w.counters = reinterpret_cast<uint32_t*>(p);
    p += align256a((size_t)Hkv * n_rtiles * sizeof(uint32_t));
    w.part_ml = reinterpret_cast<float*>(p);
    p += align256a(items * n_rtiles * AT_RT * 2 * sizeof(float));
    w.part_o = reinterpret_cast<float*>(p);
    p += align256a(items * n_rtiles * AT_RT * D * sizeof(float));
    w.bytes = (size_t)(p - reinterpret_cast<char*>(ws));
    return w;
}

template <typename T, int D>
static int launch_attn(const void* q, const void* k, const void* v, const int32_t* k_start, const int32_t* k_len,
                       int k_len_offset, const int32_t* off_dev, const int32_t* meta_host, int Hkv, int G, int q_len, float scale, int causal, void* out,
                       void* ws, hipStream_t stream, const void* k_new = nullptr, const void* v_new = nullptr,
                       int64_t kn_stride = 0, int64_t vn_stride = 0) {
    const int n_rtiles = (q_len * G + AT_RT - 1) / AT_RT;
    const AttnWs w = attn_ws(ws, Hkv, n_rtiles, D);
    const int target = attn_items(Hkv);
    HeadMeta hm;
    const bool host = meta_host != nullptr && Hkv <= AT_MAXH;
    if (host)
        for (int h = 0; h < Hkv; ++h) { hm.start[h] = meta_host[h]; hm.len[h] = meta_host[Hkv + h]; }
    const dim3 grid(target + Hkv, n_rtiles), block(AT_THREADS);
    ProfScope ps("varlen_attn", stream);
#define KVZ_ATTN_LAUNCH(APP, HOST)                                                                                                \
    hipLaunchKernelGGL((varlen_attn_split2_kernel<T, D, APP, HOST>), grid, block, 0, stream, reinterpret_cast<const T*>(q),          \
                       reinterpret_cast<const T*>(k), reinterpret_cast<const T*>(v), k_start, k_len, k_len_offset, off_dev, hm, Hkv, G, \
                       q_len, target, scale, causal, w.part_o, w.part_ml, w.counters, n_rtiles, reinterpret_cast<T*>(out),          \
                       reinterpret_cast<const T*>(k_new), reinterpret_cast<const T*>(v_new), kn_stride, vn_stride)
    if (k_new) { if (host) KVZ_ATTN_LAUNCH(true, true); else KVZ_ATTN_LAUNCH(true, false); }
    else { if (host) KVZ_ATTN_LAUNCH(false, true); else KVZ_ATTN_LAUNCH(false, false); }
#undef KVZ_ATTN_LAUNCH
    KVZ_CHECK_LAUNCH("varlen_attn_split2_kernel");
    const dim3 cgrid(Hkv, n_rtiles, AT_RT), cblock(CB_THREADS);
    if (host)
        hipLaunchKernelGGL((varlen_attn_combine2_kernel<T, D, true>), cgrid, cblock, 0, stream, w.part_o, w.part_ml, k_len, k_len_offset,
                           off_dev, hm, Hkv, G, q_len, target, n_rtiles, reinterpret_cast<T*>(out));
    else
        hipLaunchKernelGGL((varlen_attn_combine2_kernel<T, D, false>), cgrid, cblock, 0, stream, w.part_o, w.part_ml, k_len, k_len_offset,
                           off_dev, hm, Hkv, G, q_len, target, n_rtiles, reinterpret_cast<T*>(out));
    KVZ_CHECK_LAUNCH("varlen_attn_combine2_kernel");
    return KVZ_OK;
}

}  // namespace kvz

using namespace kvz;

// query rows per head above which the multi-row kernel takes over from the split-key decode kernel
static int flash_min_rows() { return tunable(TUNE_FLASH_MIN_ROWS); }

extern "C" size_t kvz_varlen_attn_workspace_bytes(int Hkv, int G, int q_len, int D, int max_len_k) {
    if (Hkv <= 0 || G <= 0 || q_len <= 0 || D <= 0 || max_len_k < 0) return 0;
    if (q_len * G > flash_min_rows()) return kvz_flash_workspace_bytes(Hkv, G, q_len, D) + 256;  // (never 0: callers pass a buffer)
    const int n_rtiles = (q_len * G + AT_RT - 1) / AT_RT;
    return attn_ws(nullptr, Hkv, n_rtiles, D).bytes;
}

extern "C" int kvz_varlen_attn(const void* q, const void* k, const void* v, const int32_t* k_start,
                               const int32_t* k_len, int k_len_offset, const int32_t* k_len_offset_dev, const int32_t* k_meta_host, int Hkv, int G, int q_len,
                               int D, int max_len_k, float scale,
                               int causal, int dtype, void* out, void* ws, size_t ws_bytes, kvz_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    KVZ_REQUIRE(q && k && v && k_start && k_len && out && ws, KVZ_EINVAL, "kvz_varlen_attn: null pointer");
    KVZ_REQUIRE(Hkv > 0 && Hkv <= 65535 && G > 0 && q_len > 0 && max_len_k >= 0, KVZ_EINVAL, "kvz_varlen_attn: bad shape");
    KVZ_REQUIRE(D == 64 || D == 128, KVZ_EUNSUPPORTED, "kvz_varlen_attn: head_dim %d unsupported (64 or 128)", D);
    KVZ_REQUIRE(dtype == KVZ_F16 || dtype == KVZ_BF16, KVZ_EINVAL, "kvz_varlen_attn: bad dtype %d", dtype);
    KVZ_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(ws), KVZ_EINVAL,
                "kvz_varlen_attn: q/k/v/ws must be 16-byte aligned");
    if (q_len * G > flash_min_rows()) {  // many query rows: one pass over the keys per 128-row block (kvz_flash.hip)
        KVZ_REQUIRE(k_len_offset_dev == nullptr, KVZ_EUNSUPPORTED, "kvz_varlen_attn: a device-side offset is a decode-step feature (q_len*G <= %d rows)", flash_min_rows());
        KVZ_REQUIRE(ws_bytes >= kvz_varlen_attn_workspace_bytes(Hkv, G, q_len, D, max_len_k), KVZ_EWORKSPACE,
                    "kvz_varlen_attn: workspace too small");
        return kvz_flash_fwd(q, (int64_t)q_len * G * D, D, (int64_t)G * D, k, v, k_start, k_len, k_len_offset, k_meta_host, Hkv, G,
                             q_len, D, scale, causal, dtype, out, (int64_t)q_len * G * D, D, (int64_t)G * D, nullptr, ws, ws_bytes,
                             stream_);
    }
    KVZ_REQUIRE((q_len * G + AT_RT - 1) / AT_RT <= 65535, KVZ_EINVAL, "kvz_varlen_attn: too many query rows");
    KVZ_REQUIRE(ws_bytes >= kvz_varlen_attn_workspace_bytes(Hkv, G, q_len, D, max_len_k), KVZ_EWORKSPACE,
                "kvz_varlen_attn: workspace too small");
    if (dtype == KVZ_F16) {
        if (D == 128) return launch_attn<_Float16, 128>(q, k, v, k_start, k_len, k_len_offset, k_len_offset_dev, k_meta_host, Hkv, G, q_len, scale, causal, out, ws, stream);
        return launch_attn<_Float16, 64>(q, k, v, k_start, k_len, k_len_offset, k_len_offset_dev, k_meta_host, Hkv, G, q_len, scale, causal, out, ws, stream);
    }
    if (D == 128) return launch_attn<__bf16, 128>(q, k, v, k_start, k_len, k_len_offset, k_len_offset_dev, k_meta_host, Hkv, G, q_len, scale, causal, out, ws, stream);
    return launch_attn<__bf16, 64>(q, k, v, k_start, k_len, k_len_offset, k_len_offset_dev, k_meta_host, Hkv, G, q_len, scale, causal, out, ws, stream);
}

extern "C" int kvz_varlen_attn_append(const void* q, void* k_cache, void* v_cache, const void* k_state, const void* v_state,
                                      int64_t k_state_head_stride, int64_t v_state_head_stride, const int32_t* k_start,
                                      const int32_t* k_len, int k_len_offset, const int32_t* k_len_offset_dev,
                                      const int32_t* k_meta_host, int Hkv, int G, int D, int max_len_k, float scale, int dtype, void* out,
                                      void* ws, size_t ws_bytes, kvz_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    KVZ_REQUIRE(q && k_cache && v_cache && k_state && v_state && k_start && k_len && out && ws, KVZ_EINVAL,
                "kvz_varlen_attn_append: null pointer");
    KVZ_REQUIRE(Hkv > 0 && Hkv <= 65535 && G > 0 && max_len_k >= 1 && k_len_offset >= 0, KVZ_EINVAL, "kvz_varlen_attn_append: bad shape");
    KVZ_REQUIRE(D == 64 || D == 128, KVZ_EUNSUPPORTED, "kvz_varlen_attn_append: head_dim %d unsupported (64 or 128)", D);
    KVZ_REQUIRE(dtype == KVZ_F16 || dtype == KVZ_BF16, KVZ_EINVAL, "kvz_varlen_attn_append: bad dtype %d", dtype);
    KVZ_REQUIRE(aligned16(q) && aligned16(k_cache) && aligned16(v_cache) && aligned16(k_state) && aligned16(v_state) && aligned16(ws),
                KVZ_EINVAL, "kvz_varlen_attn_append: pointers must be 16-byte aligned");
    KVZ_REQUIRE((k_state_head_stride * 2) % 16 == 0 && (v_state_head_stride * 2) % 16 == 0, KVZ_EINVAL,
                "kvz_varlen_attn_append: state head strides must be multiples of 8 elements");
    KVZ_REQUIRE(ws_bytes >= kvz_varlen_attn_workspace_bytes(Hkv, G, 1, D, max_len_k), KVZ_EWORKSPACE,
                "kvz_varlen_attn_append: workspace too small");
    const int off = k_len_offset + 1;  // the keys attended to include the token appended by this call
    if (dtype == KVZ_F16) {
        if (D == 128) return launch_attn<_Float16, 128>(q, k_cache, v_cache, k_start, k_len, off, k_len_offset_dev, k_meta_host, Hkv, G, 1, scale, 1, out, ws, stream, k_state, v_state, k_state_head_stride, v_state_head_stride);
        return launch_attn<_Float16, 64>(q, k_cache, v_cache, k_start, k_len, off, k_len_offset_dev, k_meta_host, Hkv, G, 1, scale, 1, out, ws, stream, k_state, v_state, k_state_head_stride, v_state_head_stride);
    }
    if (D == 128) return launch_attn<__bf16, 128>(q, k_cache, v_cache, k_start, k_len, off, k_len_offset_dev, k_meta_host, Hkv, G, 1, scale, 1, out, ws, stream, k_state, v_state, k_state_head_stride, v_state_head_stride);
    return launch_attn<__bf16, 64>(q, k_cache, v_cache, k_start, k_len, off, k_len_offset_dev, k_meta_host, Hkv, G, 1, scale, 1, out, ws, stream, k_state, v_state, k_state_head_stride, v_state_head_stride);
}


// the device-side part of the appended-token count (k_len_offset_dev of the attention calls): += delta, one tiny launch - the last
// node of a captured generation step
extern "C" int kvz_add_i32(int32_t* p, int delta, kvz_stream_t stream_) {
    KVZ_REQUIRE(p && (reinterpret_cast<uintptr_t>(p) & 3u) == 0, KVZ_EINVAL, "kvz_add_i32: bad pointer");
    hipLaunchKernelGGL(add_i32_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, p, delta);
    KVZ_CHECK_LAUNCH("add_i32_kernel");
    return KVZ_OK;
}
