// kvz_api.hip — error plumbing, ABI version and the optional per-kernel HIP-event profiler of the
// C ABI (include/kvzip_hip.h).
#include "kvz_common.h"

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
