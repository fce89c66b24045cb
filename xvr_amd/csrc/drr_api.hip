// Library-wide pieces of the C ABI of include/xvr_drr.h: version and the thread-local error text.
#include <stdio.h>

#include "xvr_drr.h"

namespace {
thread_local char g_err[512] = "";
}

extern "C" {

int xvr_drr_abi_version(void) { return XVR_DRR_ABI_VERSION; }
const char* xvr_drr_last_error(void) { return g_err; }
// shared by the other translation units of the library (sim_kernels.hip); not part of the public header
void xvr_drr_set_last_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }

}  // extern "C"
