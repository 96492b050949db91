// kvz_api.hip — error plumbing and version of the C ABI (include/kvzip_hip.h).
#include "kvz_common.h"

namespace kvz {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace kvz

extern "C" int kvz_abi_version(void) { return KVZ_ABI_VERSION; }
extern "C" const char* kvz_last_error(void) { return kvz::g_err; }
