// Library-wide pieces of the C ABI of include/xvr_drr.h: version, the thread-local error text, and the table of A/B
// switches (read from the environment once, when the library is loaded; never per launch).
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "xvr_drr.h"

namespace xvr_detail {
int option(int id);
}

namespace {
thread_local char g_err[512] = "";

struct OptionDef {
    const char* name;
    const char* env;
    int def, lo, hi;
};
// (order = xvr_detail::Option in drr_common.hiph)
const OptionDef OPTIONS[] = {
    {"fwd_lds", "XVR_DRR_FWD_LDS", 0, 0, 1},
    {"tile_shape", "XVR_DRR_TILE_SHAPE", -1, -1, 2},
    {"block_order", "XVR_DRR_BLOCK_ORDER", -1, -1, 4},
    {"order_group", "XVR_DRR_ORDER_GROUP", 0, 0, 0xffff},
    {"fwd_split", "XVR_DRR_FWD_SPLIT", 0, 0, 199},
    {"gather_splat", "XVR_DRR_GATHER_SPLAT", 1, 0, 3},
    {"fwd_slabs", "XVR_DRR_FWD_SLABS", 0, -1, 64},
    {"fwd_slab_axis", "XVR_DRR_FWD_SLAB_AXIS", 1, 0, 2},
    {"tile_geom", "XVR_DRR_TILE_GEOM", 1, 0, 2},
    {"siddon_slab", "XVR_DRR_SIDDON_SLAB", 1, 0, 2},
    {"siddon_gather_fast", "XVR_DRR_SIDDON_GATHER_FAST", 1, 0, 1},
    {"siddon_splat", "XVR_DRR_SIDDON_SPLAT", 1, 0, 2},
    {"gather_slab", "XVR_DRR_GATHER_SLAB", 0, 0, 0xffff},
};
constexpr int N_OPTIONS = sizeof(OPTIONS) / sizeof(OPTIONS[0]);
std::atomic<int> g_opt[N_OPTIONS];

struct OptionInit {
    OptionInit() {
        for (int i = 0; i < N_OPTIONS; ++i) {
            int v = OPTIONS[i].def;
            const char* e = getenv(OPTIONS[i].env);
            if (e && *e) {
                int gx = 0, gy = 0;
                if (i == 3 && sscanf(e, "%dx%d", &gx, &gy) == 2) v = (gx & 0xff) | (gy << 8);   // "<gx>x<gy>" tiles
                else v = atoi(e);
                if (v < OPTIONS[i].lo || v > OPTIONS[i].hi) v = OPTIONS[i].def;
            }
            g_opt[i].store(v, std::memory_order_relaxed);
        }
    }
} g_option_init;

int find_option(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < N_OPTIONS; ++i)
        if (!strcmp(name, OPTIONS[i].name)) return i;
    return -1;
}
}  // namespace

int xvr_detail::option(int id) { return g_opt[id].load(std::memory_order_relaxed); }

extern "C" {

int xvr_drr_abi_version(void) { return XVR_DRR_ABI_VERSION; }
const char* xvr_drr_last_error(void) { return g_err; }
// shared by the other translation units of the library (sim_kernels.hip); not part of the public header
void xvr_drr_set_last_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }

int xvr_drr_set_option(const char* name, int value) {
    const int i = find_option(name);
    if (i < 0) { xvr_drr_set_last_error("unknown option"); return XVR_DRR_E_ARG; }
    if (value < OPTIONS[i].lo || value > OPTIONS[i].hi) { xvr_drr_set_last_error("option value out of range"); return XVR_DRR_E_ARG; }
    g_opt[i].store(value, std::memory_order_relaxed);
    return XVR_DRR_OK;
}

int xvr_drr_get_option(const char* name, int* value) {
    const int i = find_option(name);
    if (i < 0 || !value) { xvr_drr_set_last_error("unknown option"); return XVR_DRR_E_ARG; }
    *value = g_opt[i].load(std::memory_order_relaxed);
    return XVR_DRR_OK;
}

}  // extern "C"
