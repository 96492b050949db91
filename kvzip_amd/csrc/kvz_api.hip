// kvz_api.hip — error plumbing, ABI version and the optional per-kernel HIP-event profiler of the
// C ABI (include/kvzip_hip.h).
#include "kvz_common.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

namespace kvz {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- profiler: hipEvent pairs recorded on the launch stream around selected kernels ----------------
struct ProfEntry {
    std::string name;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
    double total_ms = 0.0;
    int64_t count = 0;
    int64_t seen = 0;  // launches since the last reset (sampled or not)
};
static int g_prof_period = 0;  // 0 = off, n = bracket every n-th launch of each kernel
static std::mutex g_prof_mu;
static std::vector<ProfEntry> g_prof;
static std::vector<hipEvent_t> g_free_events;

static hipEvent_t get_event() {
    if (!g_free_events.empty()) {
        hipEvent_t e = g_free_events.back();
        g_free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

bool prof_enabled() { return g_prof_period > 0; }

void prof_begin(const char* name, hipStream_t stream, int* slot, size_t* idx) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int s = -1;
    for (size_t i = 0; i < g_prof.size(); ++i)
        if (g_prof[i].name == name) { s = (int)i; break; }
    if (s < 0) {
        g_prof.push_back(ProfEntry{});
        g_prof.back().name = name;
        s = (int)g_prof.size() - 1;
    }
    if (g_prof[s].seen++ % g_prof_period != 0) {  // not a sampled launch
        *slot = -1;
        return;
    }
    hipEvent_t a = get_event(), b = get_event();
    (void)hipEventRecord(a, stream);
    g_prof[s].spans.emplace_back(a, b);
    *slot = s;
    *idx = g_prof[s].spans.size() - 1;
}
void prof_end(hipStream_t stream, int slot, size_t idx) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof[slot].spans[idx].second, stream);
}

// fold finished spans into totals (synchronises on the recorded events)
static void prof_collect() {
    for (auto& e : g_prof) {
        for (auto& sp : e.spans) {
            (void)hipEventSynchronize(sp.second);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, sp.first, sp.second) == hipSuccess) {
                e.total_ms += ms;
                e.count += 1;
            }
            g_free_events.push_back(sp.first);
            g_free_events.push_back(sp.second);
        }
        e.spans.clear();
    }
}
}  // namespace kvz

// ---- tuning knobs --------------------------------------------------------------------------------------------------------
namespace kvz {
static const char* const g_tune_name[TUNE_COUNT] = {"attn_items", "flash_min_rows", "flash2_min_blocks", "flash2_xcd", "sel_blocks", "emit_blocks", "flash2_split", "score_prune"};
// attn_items: work items the key ranges of a decode call are cut into; 0 (default) = 256 - Hkv, see kvz_attn.hip:attn_items
//   (round 2 measured 128 / 192 / 256 / 384 on an infinity-cache-resident probe, profiles/r2_attn_items.txt; round 4 on cold HBM);
// flash_min_rows: query rows per head above which the multi-row kernels take over from the split-key decode kernel;
// flash2_min_blocks: (head, 256-row tile) blocks from which the 32-row dense forward is used instead of the 16-row one;
// flash2_xcd: 1 = XCD-aware block order of the 32-row dense forward (a head's row tiles on 8 / Hkv XCDs), 0 = head-major grid.
// flash2_split: 1 = the last round of blocks of the 32-row dense forward is split along the keys when that saves a tenth of the rounds
//   (parts merged by a second launch; needs the workspace), 2 = whenever the units do not fill the last round (tests), 0 = never.
// sel_blocks / emit_blocks: 1024-thread blocks of the histogram passes / of the mask pass of the global-threshold selection (default 0 =
//   one per CU, device_cus(): every block costs a histogram flush or a prologue, profiles/r4_select.txt).
// NOTE: debug / measurement hooks - process-wide, set them while no other thread is launching (tests and tools/ only).
// score_prune (both dtypes, deferred-log entry points, chunks of >= 32 query positions; everything else takes the two full passes whatever the knob says): 3 = key-per-lane
//   row statistics + candidate keys per row group + gathered column maxima (kvz_score.hip, round 6: four launches); 6 (default) = the same kernels, the three small launches of
//   the asynchronous calls as ONE launch pipelined over the calls of a side stream (score_tail_kernel; synchronous entry points and workspaces below three sets: 3);
//   5 = candidate (group, key block) pairs (round 5); 0 = two full passes over
//   Q.K^T; 1 / 4 = check variants (key-per-lane statistics with the full column-maximum pass / every (group, key) item through the candidate pass); 16 + mask: launches left out
//   (time measurements only).  KVZIP_SCORE_PRUNE presets it.
static const int g_tune_default[TUNE_COUNT] = {0, 64, 128, 1, 0, 0, 1, 6};   // (0 for attn_items / sel_blocks / emit_blocks: derived from device_cus())
static std::atomic<int> g_tune[TUNE_COUNT] = {{0}, {64}, {128}, {1}, {0}, {0}, {1}, {6}};   // (atomic: a probe may flip a knob while another thread launches)
// (KVZIP_SCORE_PRUNE in the environment presets the score_prune knob when the library is loaded: A/B runs of whole programs)
static const int g_tune_env = [] {
    const char* e = getenv("KVZIP_SCORE_PRUNE");
    if (e && *e) g_tune[TUNE_SCORE_PRUNE].store(atoi(e), std::memory_order_relaxed);
    return 0;
}();
int tunable(Tunable t) { return g_tune[t].load(std::memory_order_relaxed); }
int device_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        return cus > 0 ? cus : 256;
    }();
    return n;
}
}  // namespace kvz
extern "C" int kvz_debug_set_tunable(const char* name, int value) {
    KVZ_REQUIRE(name, KVZ_EINVAL, "kvz_debug_set_tunable: null name");
    for (int i = 0; i < kvz::TUNE_COUNT; ++i)
        if (strcmp(name, kvz::g_tune_name[i]) == 0) {
            // (value <= 0 restores the default; the on / off knobs flash2_xcd, flash2_split and score_prune take 0 as "off" and negative
            // values as "default")
            const bool onoff = i == kvz::TUNE_FLASH2_XCD || i == kvz::TUNE_FLASH2_SPLIT || i == kvz::TUNE_SCORE_PRUNE;
            return kvz::g_tune[i].exchange((value > 0 || (onoff && value == 0)) ? value : kvz::g_tune_default[i]);
        }
    kvz::set_error("kvz_debug_set_tunable: unknown knob '%s'", name);
    return KVZ_EINVAL;
}

extern "C" int kvz_debug_get_tunable(const char* name) {
    KVZ_REQUIRE(name, KVZ_EINVAL, "kvz_debug_get_tunable: null name");
    for (int i = 0; i < kvz::TUNE_COUNT; ++i)
        if (strcmp(name, kvz::g_tune_name[i]) == 0) return kvz::tunable((kvz::Tunable)i);
    kvz::set_error("kvz_debug_get_tunable: unknown knob '%s'", name);
    return KVZ_EINVAL;
}

extern "C" int kvz_abi_version(void) { return KVZ_ABI_VERSION; }
extern "C" const char* kvz_last_error(void) { return kvz::g_err; }

extern "C" void kvz_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(kvz::g_prof_mu);
    kvz::g_prof_period = on > 0 ? on : 0;
}
extern "C" void kvz_prof_reset(void) {
    std::lock_guard<std::mutex> lk(kvz::g_prof_mu);
    kvz::prof_collect();
    for (auto& e : kvz::g_prof) { e.total_ms = 0.0; e.count = 0; e.seen = 0; }
}
extern "C" int kvz_prof_read(const char* name, double* total_ms, int64_t* count) {
    std::lock_guard<std::mutex> lk(kvz::g_prof_mu);
    kvz::prof_collect();
    for (auto& e : kvz::g_prof)
        if (e.name == name) {
            if (total_ms) *total_ms = e.total_ms;
            if (count) *count = e.count;
            return KVZ_OK;
        }
    if (total_ms) *total_ms = 0.0;
    if (count) *count = 0;
    return KVZ_OK;
}

// ---- asynchronous scoring context ---------------------------------------------------------------------------------------
// The scores of a layer are a side product of the scoring forward pass: nothing consumes them before prune().  The context
// keeps, per slot (= layer), one "inputs ready" and one "scores done" event, so that a scoring call can be issued on a SIDE
// stream and ordered against the caller's stream without any host-side event juggling:
//     kvz_score_chunk_async:  record(ready[slot], caller); wait(side, ready[slot]); <kernels on side>; record(done[slot], side)
//     kvz_async_wait:         wait(stream, done[slot])   (slot < 0: every pending slot)
// Host objects only (events); no device memory.
namespace kvz {
struct AsyncCtx {
    std::vector<hipEvent_t> ready, done;
    std::vector<char> pending;
    std::mutex mu;   // the per-slot state (pending flags, event records) is touched under this lock: a context may be driven from
                     // several host threads (ctypes releases the GIL); stream ORDER between their calls stays the callers' business
};
static std::mutex g_async_mu;
static std::vector<AsyncCtx*> g_async;
static AsyncCtx* async_get(int h) {
    std::lock_guard<std::mutex> lk(g_async_mu);
    return (h >= 0 && h < (int)g_async.size()) ? g_async[h] : nullptr;
}
}  // namespace kvz

extern "C" int kvz_async_create(int n_slots) {
    KVZ_REQUIRE(n_slots > 0 && n_slots <= 65536, KVZ_EINVAL, "kvz_async_create: bad slot count %d", n_slots);
    auto* c = new kvz::AsyncCtx();
    c->pending.assign(n_slots, 0);
    c->ready.assign(n_slots, nullptr);
    c->done.assign(n_slots, nullptr);
    for (int i = 0; i < n_slots; ++i) {
        if (hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming) != hipSuccess) {
            for (auto e : c->ready) if (e) (void)hipEventDestroy(e);   // nothing of a half-built context survives
            for (auto e : c->done) if (e) (void)hipEventDestroy(e);
            delete c;
            kvz::set_error("kvz_async_create: hipEventCreate failed");
            return KVZ_ELAUNCH;
        }
    }
    std::lock_guard<std::mutex> lk(kvz::g_async_mu);
    for (size_t i = 0; i < kvz::g_async.size(); ++i)
        if (!kvz::g_async[i]) { kvz::g_async[i] = c; return (int)i; }
    kvz::g_async.push_back(c);
    return (int)kvz::g_async.size() - 1;
}
extern "C" int kvz_async_destroy(int handle) {
    kvz::AsyncCtx* c = nullptr;
    {
        std::lock_guard<std::mutex> lk(kvz::g_async_mu);
        if (handle < 0 || handle >= (int)kvz::g_async.size() || !kvz::g_async[handle]) return KVZ_OK;
        c = kvz::g_async[handle];
        kvz::g_async[handle] = nullptr;
    }
    for (auto e : c->ready) (void)hipEventDestroy(e);
    for (auto e : c->done) (void)hipEventDestroy(e);
    delete c;
    return KVZ_OK;
}
extern "C" int kvz_async_wait(int handle, int slot, kvz_stream_t stream) {
    kvz::AsyncCtx* c = kvz::async_get(handle);
    KVZ_REQUIRE(c, KVZ_EINVAL, "kvz_async_wait: bad handle %d", handle);
    KVZ_REQUIRE(slot < (int)c->pending.size(), KVZ_EINVAL, "kvz_async_wait: bad slot %d", slot);
    const int lo = slot < 0 ? 0 : slot, hi = slot < 0 ? (int)c->pending.size() : slot + 1;
    std::lock_guard<std::mutex> lk(c->mu);
    for (int i = lo; i < hi; ++i)
        if (c->pending[i]) {
            if (hipStreamWaitEvent((hipStream_t)stream, c->done[i], 0) != hipSuccess) {
                kvz::set_error("kvz_async_wait: hipStreamWaitEvent failed");
                return KVZ_ELAUNCH;
            }
            c->pending[i] = 0;
        }
    return KVZ_OK;
}
static int score_chunk_async_impl(int handle, int slot, kvz_stream_t caller, kvz_stream_t side, const void* q,
                                  int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen, int sink, int start,
                                  int end, int q_len, int Hkv, int G, int D, int dtype, void* out, int64_t out_head_stride,
                                  void* ws, size_t ws_bytes, bool log);

extern "C" int kvz_score_chunk_async(int handle, int slot, kvz_stream_t caller, kvz_stream_t side, const void* q,
                                     int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen, int sink, int start,
                                     int end, int q_len, int Hkv, int G, int D, int dtype, void* out, int64_t out_head_stride,
                                     void* ws, size_t ws_bytes) {
    return score_chunk_async_impl(handle, slot, caller, side, q, q_head_stride, k, k_head_stride, klen, sink, start, end, q_len, Hkv, G,
                                  D, dtype, out, out_head_stride, ws, ws_bytes, false);
}
extern "C" int kvz_score_chunk_async_log(int handle, int slot, kvz_stream_t caller, kvz_stream_t side, const void* q,
                                         int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen, int sink, int start,
                                         int end, int q_len, int Hkv, int G, int D, int dtype, uint32_t* log_out,
                                         int64_t log_head_stride, void* ws, size_t ws_bytes) {
    return score_chunk_async_impl(handle, slot, caller, side, q, q_head_stride, k, k_head_stride, klen, sink, start, end, q_len, Hkv, G,
                                  D, dtype, log_out, log_head_stride, ws, ws_bytes, true);
}

static int score_chunk_async_impl(int handle, int slot, kvz_stream_t caller, kvz_stream_t side, const void* q,
                                  int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen, int sink, int start,
                                  int end, int q_len, int Hkv, int G, int D, int dtype, void* out, int64_t out_head_stride,
                                  void* ws, size_t ws_bytes, bool log) {
    kvz::AsyncCtx* c = kvz::async_get(handle);
    KVZ_REQUIRE(c, KVZ_EINVAL, "kvz_score_chunk_async: bad handle %d", handle);
    KVZ_REQUIRE(slot >= 0 && slot < (int)c->pending.size(), KVZ_EINVAL, "kvz_score_chunk_async: bad slot %d", slot);
    std::lock_guard<std::mutex> lk(c->mu);
    if (side != caller) {
        if (hipEventRecord(c->ready[slot], (hipStream_t)caller) != hipSuccess ||
            hipStreamWaitEvent((hipStream_t)side, c->ready[slot], 0) != hipSuccess) {
            kvz::set_error("kvz_score_chunk_async: could not order the side stream behind the caller's stream");
            return KVZ_ELAUNCH;
        }
    }
    const int rc = log ? (side != caller ? kvz::score_chunk_log_deferred(q, q_head_stride, k, k_head_stride, klen, sink, start, end, q_len, Hkv, G, D,
                                                                         dtype, reinterpret_cast<uint32_t*>(out), out_head_stride, ws, ws_bytes, side)
                                         : kvz_score_chunk_log(q, q_head_stride, k, k_head_stride, klen, sink, start, end, q_len, Hkv, G, D, dtype,
                                                               reinterpret_cast<uint32_t*>(out), out_head_stride, ws, ws_bytes, side))
                       : kvz_score_chunk(q, q_head_stride, k, k_head_stride, klen, sink, start, end, q_len, Hkv, G, D, dtype, out,
                                         out_head_stride, ws, ws_bytes, side);
    if (rc != KVZ_OK) return rc;
    if (side != caller) {
        if (hipEventRecord(c->done[slot], (hipStream_t)side) != hipSuccess) {
            kvz::set_error("kvz_score_chunk_async: hipEventRecord failed");
            return KVZ_ELAUNCH;
        }
        c->pending[slot] = 1;
    }
    return KVZ_OK;
}


// the phases earlier calls left pending on `ws` (pipelined tail), then the done-event of `slot` on `side`: kvz_async_wait covers them
extern "C" int kvz_score_tail_flush_async(int handle, int slot, const void* ws, kvz_stream_t side) {
    kvz::AsyncCtx* c = kvz::async_get(handle);
    KVZ_REQUIRE(c, KVZ_EINVAL, "kvz_score_tail_flush_async: bad handle %d", handle);
    KVZ_REQUIRE(slot >= 0 && slot < (int)c->pending.size(), KVZ_EINVAL, "kvz_score_tail_flush_async: bad slot %d", slot);
    const int rc = kvz_score_tail_flush(ws);
    if (rc <= 0) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    if (hipEventRecord(c->done[slot], (hipStream_t)side) != hipSuccess) {
        kvz::set_error("kvz_score_tail_flush_async: hipEventRecord failed");
        return KVZ_ELAUNCH;
    }
    c->pending[slot] = 1;
    return 1;
}

// one host call per layer of a scoring pass: append the repeat chunk's K,V to the dense cache on the caller's stream, then score
extern "C" int kvz_update_score_async_log(int handle, int slot, kvz_stream_t caller, kvz_stream_t side, void* k_cache, void* v_cache,
                                          int64_t cache_head_stride, int fill, const void* k_state, const void* v_state,
                                          int64_t ks_head_stride, int64_t ks_row_stride, int64_t vs_head_stride,
                                          int64_t vs_row_stride, int t, const void* q, int64_t q_head_stride, int sink, int start,
                                          int end, int q_len, int Hkv, int G, int D, int dtype, uint32_t* log_out,
                                          int64_t log_head_stride, void* ws, size_t ws_bytes) {
    // the previous scoring call of this slot read the rows that the append overwrites
    int rc = kvz_async_wait(handle, slot, caller);
    if (rc != KVZ_OK) return rc;
    rc = kvz_dense_append(k_cache, v_cache, cache_head_stride, fill, k_state, v_state, ks_head_stride, ks_row_stride, vs_head_stride,
                          vs_row_stride, Hkv, t, D, 2, caller);
    if (rc != KVZ_OK) return rc;
    return score_chunk_async_impl(handle, slot, caller, side, q, q_head_stride, k_cache, cache_head_stride, fill + t, sink, start, end,
                                  q_len, Hkv, G, D, dtype, log_out, log_head_stride, ws, ws_bytes, true);
}

// pass B only, asynchronously: the row statistics came out of the scoring forward (kvz_flash_fwd_window on the caller's stream)
extern "C" int kvz_score_from_stats_async_log(int handle, int slot, kvz_stream_t caller, kvz_stream_t side, const void* q,
                                             int64_t q_head_stride, const void* k, int64_t k_head_stride, int klen, int sink,
                                             int start, int end, int q_len, int Hkv, int G, int D, int dtype, const float* stats,
                                             int64_t stats_head_stride, uint32_t* log_out, int64_t log_head_stride) {
    kvz::AsyncCtx* c = kvz::async_get(handle);
    KVZ_REQUIRE(c, KVZ_EINVAL, "kvz_score_from_stats_async_log: bad handle %d", handle);
    KVZ_REQUIRE(slot >= 0 && slot < (int)c->pending.size(), KVZ_EINVAL, "kvz_score_from_stats_async_log: bad slot %d", slot);
    std::lock_guard<std::mutex> lk(c->mu);
    if (side != caller) {
        if (hipEventRecord(c->ready[slot], (hipStream_t)caller) != hipSuccess ||
            hipStreamWaitEvent((hipStream_t)side, c->ready[slot], 0) != hipSuccess) {
            kvz::set_error("kvz_score_from_stats_async_log: could not order the side stream behind the caller's stream");
            return KVZ_ELAUNCH;
        }
    }
    const int rc = kvz_score_from_stats_log(q, q_head_stride, k, k_head_stride, klen, sink, start, end, q_len, Hkv, G, D, dtype, stats,
                                            stats_head_stride, log_out, log_head_stride, side);
    if (rc != KVZ_OK) return rc;
    if (side != caller) {
        if (hipEventRecord(c->done[slot], (hipStream_t)side) != hipSuccess) {
            kvz::set_error("kvz_score_from_stats_async_log: hipEventRecord failed");
            return KVZ_ELAUNCH;
        }
        c->pending[slot] = 1;
    }
    return KVZ_OK;
}
