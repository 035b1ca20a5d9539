// Error plumbing and version for libshgvqa.so.
#include <mutex>
#include <string>

#include "common.h"

namespace shg {
static std::mutex g_err_mu;
static std::string g_err = "";

void set_error(const char* msg) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err = msg ? msg : "";
}
int fail_arg(const char* msg) {
    set_error(msg);
    return SHG_ERR_INVALID;
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        std::string m = std::string(what) + ": " + hipGetErrorString(e);
        set_error(m.c_str());
        return (int)e;
    }
    return 0;
}
}  // namespace shg

extern "C" int shg_version(void) { return 100; }

extern "C" const char* shg_last_error_string(void) {
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(shg::g_err_mu);
    copy = shg::g_err;
    return copy.c_str();
}
