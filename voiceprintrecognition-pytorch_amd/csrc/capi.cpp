// Error plumbing and ABI version of libmvector_hip.so.
#include "common.h"

#include <vector>

namespace mv {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MV_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return MV_OK;
}

// ---- optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline legs) ----
// Off by default: the launchers call prof_begin / prof_end, which do nothing until mv_profile_enable(1).
namespace {
struct ProfRecord {
    int cls;
    hipEvent_t start, stop;
    double work;
};
bool g_prof_on = false;
std::vector<ProfRecord> g_prof;
std::vector<hipEvent_t> g_prof_free;

hipEvent_t prof_event() {
    if (!g_prof_free.empty()) {
        hipEvent_t e = g_prof_free.back();
        g_prof_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}
}  // namespace

int prof_begin(int cls, double work, hipStream_t stream) {
    if (!g_prof_on) return -1;
    ProfRecord r{cls, prof_event(), prof_event(), work};
    hipEventRecord(r.start, stream);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}

void prof_end(int token, hipStream_t stream) {
    if (token >= 0 && token < (int)g_prof.size()) hipEventRecord(g_prof[token].stop, stream);
}

}  // namespace mv

extern "C" {

const char* mv_last_error(void) { return mv::g_last_error.c_str(); }

int mv_profile_enable(int32_t on) {
    mv::g_prof_on = on != 0;
    return MV_OK;
}

int mv_profile_read(int32_t kernel_class, int32_t* calls, double* total_ms, double* total_work, int32_t reset) {
    MV_REQUIRE(calls != nullptr && total_ms != nullptr && total_work != nullptr, "mv_profile_read: null argument");
    *calls = 0;
    *total_ms = 0.0;
    *total_work = 0.0;
    for (const auto& r : mv::g_prof) {
        if (r.cls != kernel_class && !(kernel_class == MV_PROF_CONV1D && r.cls == MV_PROF_CONV1D_RING)) continue;   // (the ring GEMM's launches are conv1d launches)
        MV_HIP_OK(hipEventSynchronize(r.stop));
        float ms = 0.0f;
        MV_HIP_OK(hipEventElapsedTime(&ms, r.start, r.stop));
        *calls += 1;
        *total_ms += ms;
        *total_work += r.work;
    }
    if (reset) {
        for (const auto& r : mv::g_prof) {
            mv::g_prof_free.push_back(r.start);
            mv::g_prof_free.push_back(r.stop);
        }
        mv::g_prof.clear();
    }
    return MV_OK;
}

int mv_abi_version(void) { return MV_ABI_VERSION; }

}
