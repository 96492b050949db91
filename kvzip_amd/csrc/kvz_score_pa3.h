// kvz_score_pa3.h — pass A, third generation (included by kvz_score.hip inside namespace kvz, after the shared helpers).
//
// What changed against score_rowstat2_kernel and why (serial-issue model of DESIGN.md 3.1: a SIMD's time is the SUM of the
// issue costs of everything its two waves execute, so every instruction that is not the rounding chain / exponential of a
// logit has to be shared by more logits):
//  * 64 query rows per wave (two 32-row groups, 512 rows per block): one fragment read (8 ds_read_b128), one DMA piece, one
//    barrier and one round of loop control now serve TWO 32x32 result blocks.  The query fragments (64 VGPRs) stay in
//    registers for a whole segment and are loaded straight from global memory (no LDS transit area: the ring is the only LDS).
//  * The two row groups take turns: the pipeline runs over the sequence (block 0, group 0) (0,1) (1,0) (1,1) ..., the chain of
//    the next element inside the epilogue of the current one, so there are still only two accumulator sets (32 VGPRs).
//  * The rounded logits of a block are not kept: the (rare) cold path recomputes them from the accumulators, which are
//    untouched until the next step's chain.
//  * Static, exactly balanced partition: the (head, 512-row tile, key tile) space is laid out as one sequence of key tiles and
//    cut into 256 equal ranges by the host (PaPlan, passed by value).  A block walks its range as at most a few segments
//    (= consecutive key tiles of one row tile); a row tile's statistics come out as one partial per block that touched it.
#pragma once

constexpr int P3_WAVES = 8;
constexpr int P3_RG = 2;
constexpr int P3_ROWS = P3_WAVES * P3_RG * 32;  // 512 query rows per row tile
constexpr int P3_MAX_BLOCKS = PLAN_MAX_BLOCKS;

// (PaPlan, plan_ntiles and make_plan live in kvz_score.hip: the 32-rows-per-wave kernel uses the same partition)
__host__ __device__ static inline int p3_ntiles(int rt, int R, int q_len, int sink, int m) { return plan_ntiles(rt, P3_ROWS, R, q_len, sink, m); }
static bool p3_make_plan(PaPlan& p, int sink, int m, int q_len, int G, int Hkv) { return make_plan(p, P3_ROWS, sink, m, q_len, G, Hkv); }

template <typename T, int D, bool FAST>
__global__ __launch_bounds__(P3_WAVES * 64, 2) void score_rowstat3_kernel(ScoreArgs a, PaPlan plan) {
    constexpr int NWAVES = P3_WAVES;
    constexpr int RG = P3_RG;
    typedef ScoreCfg<D> C;
    typedef typename Mfma32<T>::v8 v8;
    constexpr int RING = 3;
    __shared__ __attribute__((aligned(16))) char lds[RING * C::TILE_BYTES];
    constexpr int PIECES = C::TILE_BYTES / 1024 / NWAVES;
    constexpr float L2E = 1.44269504088896340736f;
    static_assert(SC_TILE / 32 == 4, "four 32-key blocks per tile");

    const int R = a.G * a.q_len;
    const int KT = a.sink + a.m + a.q_len;
    const int Hkv = a.n_kv_heads;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int off_ctx = a.start - a.sink;
    const int off_rep = a.klen - a.q_len - a.sink - a.m;
    const uint32_t lane_off = stage_lane_offset<D, NWAVES>(wave, lane);
    const int b_ = blockIdx.x;

    // ---- the block's range of the tile sequence ----
    const int u_first = plan.unit[b_], t_first = plan.tile[b_];
    const int u_end = plan.unit[b_ + 1], t_end = plan.tile[b_ + 1];  // exclusive: (u_end, t_end)
    // ordinal of the first segment inside its unit = earlier blocks that also started inside this unit (+ the one that opened it)
    int ord0 = 0;
    if (t_first > 0) {
        ord0 = 1;
        for (int bb = b_ - 1; bb > 0 && plan.unit[bb] == u_first && plan.tile[bb] > 0; --bb) ++ord0;
    }
    struct Seg { int u, h, rt, t_lo, t_hi; };
    auto seg_of = [&](int u, int t_lo) __attribute__((always_inline)) -> Seg {
        Seg s;
        s.u = u;
        s.rt = u / Hkv;
        s.h = u - s.rt * Hkv;
        s.t_lo = t_lo;
        const int nt = p3_ntiles(s.rt, R, a.q_len, a.sink, a.m);
        s.t_hi = (u == u_end) ? t_end : nt;
        return s;
    };
    auto has_seg = [&](int u) { return u < u_end || (u == u_end && t_end > 0); };

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
#ifndef P3_NO_GATHER
        else if (t >= ts_hi) linear = false;
#endif
        if (linear) {
            stage_tile_linear_a<D, NWAVES>(dst, kh + (int64_t)(kv0 + off) * C::ROW_BYTES, lane_off, wave);
        } else {
            constexpr int ROWS_PER_INSTR = 1024 / C::ROW_BYTES;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int ci = i * NWAVES + wave;
                const int row = ci * ROWS_PER_INSTR + lane / C::CPR;
                const int pch = lane % C::CPR;
                const int chunk = (D == 128) ? (pch ^ (row & 15)) : (pch ^ ((row >> 1) & 7));
                const int kv = min(kv0 + row, KT - 1);
                const int crow = kv + (kv < a.sink ? 0 : (kv < a.sink + a.m ? off_ctx : off_rep));
                lds_dma16a(kh, (uint32_t)(crow * C::ROW_BYTES + chunk * 16), dst + (uint32_t)(ci * 1024));
            }
        }
    };
    // staging cursor over the block's tile sequence
    Seg sq = seg_of(u_first, t_first);
    int sq_t = t_first;
    bool sq_done = false;
    auto sq_stage = [&](int b) __attribute__((always_inline)) {
        stage(b, sq.h, sq_t);
        ++sq_t;
        if (sq_t >= sq.t_hi) {
            if (has_seg(sq.u + 1) && sq.u < u_end) {
                sq = seg_of(sq.u + 1, 0);
                sq_t = 0;
                if (sq.t_hi == 0) sq_done = true;
            } else {
                sq_done = true;
            }
        }
    };

    FragAddr<D> fa0;
    fa0.init(lds, l31, half);

    Seg cur = seg_of(u_first, t_first);
    int ord = ord0;
    v8 bq[RG][C::KK];
    int row_r[RG], row_limit[RG];
    auto load_q = [&](const Seg& s) __attribute__((always_inline)) {
        // B operand of the 32x32x16 MFMA: lane (row l31, half) holds elements half*8 .. +8 of every 16-wide k-step
        const char* qh = reinterpret_cast<const char*>(a.q) + (int64_t)s.h * a.G * a.q_head_stride * 2;
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const int r = s.rt * P3_ROWS + (wave * RG + g) * 32 + l31;
            const int rc = min(r, R - 1);
            const int gg = rc / a.q_len, qi = rc - gg * a.q_len;
            row_r[g] = r;
            row_limit[g] = a.sink + a.m + qi;  // key j (virtual index) is visible to query i iff j <= sink + m + i  (score.py:67-85)
            const char* qp = qh + ((int64_t)gg * a.q_head_stride + (int64_t)qi * D) * 2 + half * 16;
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) bq[g][kk] = __builtin_bit_cast(v8, *reinterpret_cast<const u32x4*>(qp + kk * 32));
        }
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int kk = 0; kk < C::KK; ++kk) asm volatile("" : "+v"(bq[g][kk]));  // the wait for these loads belongs here
    };

    // ---- start-up: two tiles in flight, the query rows, the first fragments ----
    sq_stage(0);
    int staged = 1;
    if (!sq_done) {
        sq_stage(1);
        staged = 2;
    }
    load_q(cur);  // (its wait drains the two tiles as well)
    stage_wait();
    block_barrier();

    u32x4 fr[2][C::KK];  // fragments of block b of a tile live in set b & 1
    // fragment addresses of the CURRENT tile buffer: moved by one buffer at every hand-over (one v_add per address and tile)
    // instead of one copy of the tile body per ring position
    auto load_frags = [&](u32x4 (&f)[C::KK], int byte_off) __attribute__((always_inline)) { frag_load<D>(f, fa0, byte_off); };

    int pbuf = 0, sp = 0, t = cur.t_lo;
    float m_ref[RG], nml2_ref[RG], l_run[RG];
    int wmin;
    auto start_item = [&]() __attribute__((always_inline)) {
        int lo = min(row_limit[0], row_limit[1]);
#pragma unroll
        for (int g = 0; g < RG; ++g) { m_ref[g] = 0.f; nml2_ref[g] = 0.f; l_run[g] = 0.f; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lo = min(lo, __shfl_xor(lo, o, 64));
        wmin = __builtin_amdgcn_readfirstlane(lo);  // keys <= wmin are visible to every row of the wave
    };
    start_item();

    f16v acc[2];
    acc[0] = acc[1] = f16v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // (tied operands: defined once)

    // One pipeline step = one 32x32 result block: epilogue of the CURRENT block (accc: row group GC, first key k0), matrix chain
    // of the NEXT block in the sequence (b,0) (b,1) (b+1,0) ... (fragments frn x query rows of group 1-GC -> accn).  `hook` runs
    // after the first quarter of the step (fragment prefetch / tile hand-over): by then the last MFMA of the previous step has
    // read the fragment registers that the prefetch overwrites.
    auto step = [&](f16v& accn, const f16v& accc, const u32x4 (&frn)[C::KK], int k0, float ps_low, auto gc_tag, auto mask_tag,
                    auto&& hook) __attribute__((always_inline)) {
        constexpr int GC = decltype(gc_tag)::value, GN = 1 - GC;
        constexpr bool MASK = decltype(mask_tag)::value;
        float ps0 = 0.f, ps1 = 0.f;
        const int rel = row_limit[GC] - (k0 + 4 * half);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            float arg[4];
            uint32_t xa, xb;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < MfmaSched<C::KK>::count(2 * qd); ++c) {
                const int kk = MfmaSched<C::KK>::first(2 * qd) + c;
                if (kk == 0) Mfma32<T>::mfma_first(accn, __builtin_bit_cast(v8, frn[kk]), bq[GN][kk]);
                else accn = Mfma32<T>::mfma(__builtin_bit_cast(v8, frn[kk]), bq[GN][kk], accn);
            }
            {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = qd * 4 + j;
                    v[j] = (!MASK || (i & 3) + 8 * (i >> 2) <= rel) ? accc[i] : -INFINITY;
                }
                quad_args<T, FAST>(v[0], v[1], v[2], v[3], xa, xb, arg, a.c, a.rcp, L2E, nml2_ref[GC]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < MfmaSched<C::KK>::count(2 * qd + 1); ++c) {
                const int kk = MfmaSched<C::KK>::first(2 * qd + 1) + c;
                accn = Mfma32<T>::mfma(__builtin_bit_cast(v8, frn[kk]), bq[GN][kk], accn);
            }
            quad_sum(arg, ps0, ps1);
            if (qd == 0) {
                __builtin_amdgcn_sched_barrier(0);
                hook();
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        float ps = ps0 + ps1;
#ifndef P3_NO_COLD
        if (__builtin_amdgcn_ballot_w64(!(ps <= PA2_SUM_LIMIT) || ps < ps_low) != 0) {  // wave-uniform and rare
            asm volatile("" ::: "memory");
            // cold path: the reference moves (up: a logit far above it; down, first block of a segment only: everything far
            // below it).  The rounded logits are recomputed from the accumulators.
            float tmax = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool vis = !MASK || (i & 3) + 8 * (i >> 2) <= rel;
                if (vis) tmax = fmaxf(tmax, round_chain<T, FAST>(accc[i], a.c, a.rcp));
            }
            if (tmax > m_ref[GC] || (l_run[GC] == 0.f && tmax > -INFINITY)) {
                const float nml2_new = -(tmax * L2E);
                l_run[GC] *= __builtin_amdgcn_exp2f(nml2_new - nml2_ref[GC]);
                m_ref[GC] = tmax;
                nml2_ref[GC] = nml2_new;
            }
            ps = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool vis = !MASK || (i & 3) + 8 * (i >> 2) <= rel;
                if (vis) ps += __builtin_amdgcn_exp2f(__builtin_fmaf(round_chain<T, FAST>(accc[i], a.c, a.rcp), L2E, nml2_ref[GC]));
            }
        }
#endif
        l_run[GC] += ps;
    };
    // matrix chain of block (0, group 0) of a segment's first tile (nothing to overlap it with)
    auto chain0 = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) {
            if (kk == 0) Mfma32<T>::mfma_first(acc[0], __builtin_bit_cast(v8, fr[0][kk]), bq[0][kk]);
            else acc[0] = Mfma32<T>::mfma(__builtin_bit_cast(v8, fr[0][kk]), bq[0][kk], acc[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // hand-over in the third step of a tile (see score_rowstat2_kernel): DMA of the tile two positions ahead BEFORE the barrier
    auto turnover = [&]() __attribute__((always_inline)) {
        const int b1 = (pbuf == RING - 1) ? 0 : pbuf + 1, b2 = (b1 == RING - 1) ? 0 : b1 + 1;
        int newer = 0;
        if (staged < sp + 2 && !sq_done) {  // (only when the stream was a single tile so far)
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
        const uint32_t delta = (uint32_t)((b1 - pbuf) * C::TILE_BYTES);  // wave-uniform
#pragma unroll
        for (int kk = 0; kk < C::KK; ++kk) fa0.a[kk] += delta;
        pbuf = b1;
    };
    auto tile_steps = [&](auto mask_tag) __attribute__((always_inline)) {
        constexpr int BLK = 32 * C::ROW_BYTES;
        const int k0 = t * SC_TILE;
        typedef std::integral_constant<int, 0> G0;
        typedef std::integral_constant<int, 1> G1;
        auto none = []() __attribute__((always_inline)) {};
        const float low = (t == cur.t_lo) ? PA2_SUM_LOW : 0.f;
        // epilogue (block, group) | chain issued                      | prefetch (past the end of the stream: whatever the buffer
        // holds, computed and never used)
        step(acc[1], acc[0], fr[0], k0, low, G0{}, mask_tag, none);                                                               // (0,0) | (0,1)
        step(acc[0], acc[1], fr[1], k0, low, G1{}, mask_tag, [&]() __attribute__((always_inline)) { load_frags(fr[0], 2 * BLK); });   // (0,1) | (1,0)
        step(acc[1], acc[0], fr[1], k0 + 32, 0.f, G0{}, mask_tag, none);                                                          // (1,0) | (1,1)
        step(acc[0], acc[1], fr[0], k0 + 32, 0.f, G1{}, mask_tag, [&]() __attribute__((always_inline)) { load_frags(fr[1], 3 * BLK); });  // (1,1) | (2,0)
        step(acc[1], acc[0], fr[0], k0 + 64, 0.f, G0{}, mask_tag, none);                                                          // (2,0) | (2,1)
        step(acc[0], acc[1], fr[1], k0 + 64, 0.f, G1{}, mask_tag, [&]() __attribute__((always_inline)) {                          // (2,1) | (3,0)
            turnover();              // (from here on the fragment addresses point into the next tile's buffer)
            load_frags(fr[0], 0);
        });
        step(acc[1], acc[0], fr[1], k0 + 96, 0.f, G0{}, mask_tag, none);                                                          // (3,0) | (3,1)
        step(acc[0], acc[1], fr[0], k0 + 96, 0.f, G1{}, mask_tag, [&]() __attribute__((always_inline)) { load_frags(fr[1], BLK); });  // (3,1) | next (0,0)
    };

    load_frags(fr[0], 0);
    load_frags(fr[1], 32 * C::ROW_BYTES);
    chain0();
    while (true) {
        {
#ifdef P3_NO_MASK
            const bool masked = false;
#else
            const bool masked = t * SC_TILE + SC_TILE - 1 > wmin;
#endif
            if (masked) tile_steps(std::true_type{});
            else tile_steps(std::false_type{});
        }
        ++sp;
        ++t;
        if (t < cur.t_hi) continue;

        // ---- segment finished: partial statistics of its keys ----
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const float m_o = __shfl_xor(m_ref[g], 32, 64);
            const float nml2_o = __shfl_xor(nml2_ref[g], 32, 64);
            const float l_o = __shfl_xor(l_run[g], 32, 64);
            const float M = fmaxf(m_ref[g], m_o);
            const float NML2 = (m_ref[g] >= m_o) ? nml2_ref[g] : nml2_o;
            const float Lp = l_run[g] * __builtin_amdgcn_exp2f(NML2 - nml2_ref[g]) + l_o * __builtin_amdgcn_exp2f(NML2 - nml2_o);
            if (half == 0 && row_r[g] < R) {
                float2* dst = a.stats + ((int64_t)ord * Hkv + cur.h) * a.stats_stride + row_r[g];
                const float2 val = make_float2(M, Lp);
                asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(val) : "memory");
            }
        }
        if (!(has_seg(cur.u + 1) && cur.u < u_end)) break;
        cur = seg_of(cur.u + 1, 0);
        if (cur.t_hi == 0) break;
        ord = 0;
        t = 0;
        load_q(cur);   // (drains the DMA queue: the tiles staged ahead have landed for this wave, the barrier of the last
        start_item();  //  hand-over already covered the tile computed next)
        chain0();      // (the fragments of the next tile's first two blocks are in registers; the chain issued in the last step
    }                  //  used the old query rows)
}

// merge of the partial statistics:  stats[0] <- (m_r, log l_r); a row tile has one partial per block that touched it
__global__ void score_merge_stats3_kernel(ScoreArgs a, PaPlan plan, int R, int64_t rows_total, int unit_rows) {
    constexpr float L2E = 1.44269504088896340736f;
    float2* __restrict__ stats = a.stats;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [Hkv, stats_stride]
    if (i >= rows_total) return;
    const int r = (int)(i % a.stats_stride);
    const int h = (int)(i / a.stats_stride);
    if (r >= R) {
        stats[i] = make_float2(INFINITY, 0.f);
        return;
    }
    const int u = (r / unit_rows) * a.n_kv_heads + h;
    // first block whose range reaches into unit u: the smallest b with (unit[b+1], tile[b+1]) > (u, 0)
    int lo = 0, hi = plan.nb - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const bool reaches = plan.unit[mid + 1] > u || (plan.unit[mid + 1] == u && plan.tile[mid + 1] > 0);
        if (reaches) hi = mid;
        else lo = mid + 1;
    }
    int slices = 1;
    while (lo + slices < plan.nb && plan.unit[lo + slices] == u && plan.tile[lo + slices] > 0) ++slices;
    float M = -INFINITY;
    for (int s = 0; s < slices; ++s) M = fmaxf(M, stats[s * rows_total + i].x);
    const float ML2 = M * L2E;
    float Lp = 0.f;
    for (int s = 0; s < slices; ++s) {
        const float2 ps = stats[s * rows_total + i];
        Lp += ps.y * __builtin_amdgcn_exp2f(ps.x * L2E - ML2);
    }
    const float delta = __builtin_fmaf(M, L2E, -ML2);
    stats[i] = make_float2(M, logf(Lp) - delta * 0.69314718055994530942f);
}
