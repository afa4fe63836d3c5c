// Runtime glue of libreidgan_hip.so: error reporting and the optional per-launch event profiler.
#include "rg_common.h"

#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>

namespace rg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return RG_ERR_LAUNCH;
    }
    return RG_OK;
}

// ---- profiler -------------------------------------------------------------------------------
// When enabled every ProfScope brackets its launches with a pair of HIP events recorded on the
// launch stream.  rg_profile_collect() synchronises the events (the only synchronising entry
// point of the library; never called on the training path) and accumulates per family.
struct Rec {
    int fam;
    double flops;
    double bytes;
    hipEvent_t e0, e1;
};
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<Rec> g_recs;
static double g_ms[FAM_COUNT], g_flops[FAM_COUNT], g_bytes[FAM_COUNT];
static long long g_calls[FAM_COUNT];

ProfScope::ProfScope(int fam_, hipStream_t stream_, double flops_, double bytes_)
    : fam(fam_), stream(stream_), slot(-1), flops(flops_) {
    if (!g_prof_on) return;
    Rec r;
    r.fam = fam;
    r.flops = flops_;
    r.bytes = bytes_;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    hipEventRecord(r.e0, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_recs.push_back(r);
    slot = (int)g_recs.size() - 1;
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEventRecord(g_recs[slot].e1, stream);
}

}  // namespace rg

extern "C" const char* rg_last_error(void) { return rg::g_err; }

extern "C" int rg_version(void) { return 1; }

extern "C" int rg_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(rg::g_prof_mu);
    rg::g_prof_on = on != 0;
    return RG_OK;
}

extern "C" int rg_profile_reset(void) {
    std::lock_guard<std::mutex> lk(rg::g_prof_mu);
    for (auto& r : rg::g_recs) {
        hipEventDestroy(r.e0);
        hipEventDestroy(r.e1);
    }
    rg::g_recs.clear();
    memset(rg::g_ms, 0, sizeof(rg::g_ms));
    memset(rg::g_flops, 0, sizeof(rg::g_flops));
    memset(rg::g_bytes, 0, sizeof(rg::g_bytes));
    memset(rg::g_calls, 0, sizeof(rg::g_calls));
    return RG_OK;
}

// Drains pending event pairs into the per-family accumulators.  out_ms / out_flops / out_bytes /
// out_calls are arrays of RG_FAMILY_COUNT entries (any may be NULL).
extern "C" int rg_profile_collect(double* out_ms, double* out_flops, double* out_bytes, long long* out_calls) {
    std::lock_guard<std::mutex> lk(rg::g_prof_mu);
    for (auto& r : rg::g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            rg::g_ms[r.fam] += ms;
            rg::g_flops[r.fam] += r.flops;
            rg::g_bytes[r.fam] += r.bytes;
            rg::g_calls[r.fam] += 1;
        }
        hipEventDestroy(r.e0);
        hipEventDestroy(r.e1);
    }
    rg::g_recs.clear();
    for (int i = 0; i < rg::FAM_COUNT; ++i) {
        if (out_ms) out_ms[i] = rg::g_ms[i];
        if (out_flops) out_flops[i] = rg::g_flops[i];
        if (out_bytes) out_bytes[i] = rg::g_bytes[i];
        if (out_calls) out_calls[i] = rg::g_calls[i];
    }
    return RG_OK;
}

extern "C" int rg_family_count(void) { return rg::FAM_COUNT; }
