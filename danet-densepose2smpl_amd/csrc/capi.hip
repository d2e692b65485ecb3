// Library-level entry points: version + thread-local error string.
#include "common.h"

namespace danet {
char* last_error_buf() {
    static thread_local char buf[kErrLen] = {0};
    return buf;
}
}  // namespace danet

extern "C" int danet_version(void) { return 100; }
extern "C" const char* danet_last_error(void) { return danet::last_error_buf(); }
