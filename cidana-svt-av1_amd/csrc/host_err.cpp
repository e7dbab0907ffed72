// host_err.cpp — svt_hip_last_error() and the setter behind it.  Plain host C++ (no HIP): part of the sanitizer builds too.
#include <stdarg.h>
#include <stdio.h>

#include "host_err.h"

namespace svthost {
thread_local char g_err[512] = "";
int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace svthost

extern "C" const char* svt_hip_last_error(void) { return svthost::g_err; }
