// Version / error reporting / stream helpers of the C ABI.
#include "dfh_common.h"

#include <climits>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace dfh {

char *last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// The option table: every field -1 until DFH_OPTIONS (read here, once per process -- the library's only look at the
// environment) or dfh_set_option() says otherwise.
static Options g_options;
static std::once_flag g_options_once;

static long *option_slot(const char *name, size_t len) {
#define DFH_X(n) if (len == sizeof(#n) - 1 && strncmp(name, #n, len) == 0) return &g_options.n;
    DFH_OPTION_LIST(DFH_X)
#undef DFH_X
    return nullptr;
}

static void load_options() {
#define DFH_X(n) g_options.n = -1;
    DFH_OPTION_LIST(DFH_X)
#undef DFH_X
    const char *e = getenv("DFH_OPTIONS");
    while (e && *e) {
        const char *end = strchr(e, ',');
        const size_t len = end ? (size_t)(end - e) : strlen(e);
        const char *eq = static_cast<const char *>(memchr(e, '=', len));
        const size_t nlen = eq ? (size_t)(eq - e) : len;
        if (long *slot = option_slot(e, nlen)) *slot = eq ? strtol(eq + 1, nullptr, 10) : 1;
        else fprintf(stderr, "libdfusion_hip: unknown option '%.*s' in DFH_OPTIONS ignored\n", (int)nlen, e);
        e = end ? end + 1 : nullptr;
    }
}

static Options &options_mut() {
    std::call_once(g_options_once, load_options);
    return g_options;
}

const Options &opt() { return options_mut(); }

DeviceInfo &device_info(int device) {
    static DeviceInfo info[64];
    return info[device >= 0 && device < 64 ? device : 0];
}

}  // namespace dfh

extern "C" {

int dfh_set_option(const char *name, long value) {
    DFH_REQUIRE(name, "dfh_set_option: null name");
    dfh::options_mut();
    long *slot = dfh::option_slot(name, strlen(name));
    DFH_REQUIRE(slot, "dfh_set_option: unknown option '%s'", name);
    *slot = value;
    return DFH_OK;
}

long dfh_get_option(const char *name) {
    if (!name) return LONG_MIN;
    dfh::options_mut();
    const long *slot = dfh::option_slot(name, strlen(name));
    return slot ? *slot : LONG_MIN;
}

int dfh_version(void) { return DFH_ABI_VERSION; }

const char *dfh_last_error(void) { return dfh::last_error_buf(); }

int dfh_stream_synchronize(void *stream) {
    DFH_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return DFH_OK;
}

}  // extern "C"
