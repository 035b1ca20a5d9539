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
// ---- tuning switches (common.h enum Tune): name, measured-best default
struct TuneEntry { const char* name; int64_t def; };
static const TuneEntry g_tune_table[TUNE_COUNT] = {
    {"attn_nb", 2}, {"attn_nb_dq", 1}, {"attn_nb_dkv", 1}, {"tile_order", 0}, {"gemm4_max_tiles", 256},
    {"streamk_sigma", 112}, {"streamk", 1}, {"gemm8", 1}, {"gemm8_min_tiles", 120}, {"splitk_target", 384},
    {"splitk_min_steps", 8}, {"large_min_k", 128}, {"wgrad_group", 7}, {"conv_wgrad_remainder", 1},
    {"bertadam_mode", 3}, {"bertadam_blocks", 65536}, {"gemm8_tile_m", 256}, {"attn_bwd_fused", 1}, {"epilogue_side", 1}, {"conv_k_order", 126}, {"ln_half_vec", 1}, {"decoder_kseg", 1}, {"wgrad_group_cap", 256}, {"wgrad_group_split", 1},
    {"repeat_family", 0},
};
thread_local int g_in_repeat = 0;
static std::atomic<int64_t> g_tune[TUNE_COUNT];
static std::once_flag g_tune_once;
static void tune_init() {
    for (int i = 0; i < TUNE_COUNT; ++i) g_tune[i].store(g_tune_table[i].def, std::memory_order_relaxed);
}
int64_t tuning(int key) {
    std::call_once(g_tune_once, tune_init);
    return g_tune[key].load(std::memory_order_relaxed);
}
static int tune_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < TUNE_COUNT; ++i)
        if (std::string(name) == g_tune_table[i].name) return i;
    return -1;
}
}  // namespace shg

extern "C" int shg_version(void) { return 101; }

extern "C" int shg_set_tuning(const char* name, int64_t value) {
    const int i = shg::tune_index(name);
    if (i < 0) return shg::fail_arg("set_tuning: unknown switch");
    std::call_once(shg::g_tune_once, shg::tune_init);
    shg::g_tune[i].store(value, std::memory_order_relaxed);
    return 0;
}
extern "C" int64_t shg_get_tuning(const char* name) {
    const int i = shg::tune_index(name);
    if (i < 0) { shg::set_error("get_tuning: unknown switch"); return INT64_MIN; }
    return shg::tuning(i);
}
extern "C" const char* shg_tuning_name(int index) {
    return (index >= 0 && index < shg::TUNE_COUNT) ? shg::g_tune_table[index].name : nullptr;
}

extern "C" const char* shg_last_error_string(void) {
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(shg::g_err_mu);
    copy = shg::g_err;
    return copy.c_str();
}
