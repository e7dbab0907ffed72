// host_err.h — the library's thread-local error text, usable from host-only translation units (no HIP headers).
#pragma once
#include "../../include/svt_hip_dsp.h"

namespace svthost {
extern thread_local char g_err[512];
int set_err(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace svthost
