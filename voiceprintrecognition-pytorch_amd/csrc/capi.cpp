// Error plumbing and ABI version of libmvector_hip.so.
#include "common.h"

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

}  // namespace mv

extern "C" {

const char* mv_last_error(void) { return mv::g_last_error.c_str(); }

int mv_abi_version(void) { return MV_ABI_VERSION; }

}
