// Version / error reporting / stream helpers of the C ABI.
#include "dfh_common.h"

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

}  // namespace dfh

extern "C" {

int dfh_version(void) { return DFH_ABI_VERSION; }

const char *dfh_last_error(void) { return dfh::last_error_buf(); }

int dfh_stream_synchronize(void *stream) {
    DFH_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return DFH_OK;
}

}  // extern "C"
