// Error reporting and version of libsvamd.
#include "common.h"

namespace svamd {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

}  // namespace svamd

extern "C" int svamd_version(void) { return 100; }

extern "C" const char *svamd_last_error(void) { return svamd::g_error; }
