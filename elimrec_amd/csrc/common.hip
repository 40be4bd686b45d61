#include "common.h"
#include <cstring>

namespace elimrec {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace elimrec

extern "C" int elimrec_abi_version(void) { return ELIMREC_ABI_VERSION; }
extern "C" const char *elimrec_last_error(void) { return elimrec::g_err; }
