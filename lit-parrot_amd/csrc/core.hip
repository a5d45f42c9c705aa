// Library core: error channel, profiling sink, version.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>

#include "parrot_common.h"

namespace parrot {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return PARROT_EHIP;
}

ProfSink& prof_sink() {
    static ProfSink s;
    return s;
}

static const char* const kNames[K_COUNT] = {
    "w4_gemv",      "w4_gemv_dual", "w4_repack",  "bf16_gemv",   "bf16_gemv_dual", "w8_quantize_rows",
    "w8_prep_act",  "w8_gemv",      "rmsnorm",    "layernorm",   "rope_kvappend",  "attn_decode",
    "attn_combine", "embedding",    "argmax_advance", "w4_gemm", "bf16_gemm", "attn_fused_decode", "eng_token", "e4_repack", "stop_check", "gptq_block",
    "w4c_gemv", "w4c_gemv_dual", "w4c_dequant", "gemm_xsum", "gemm_splitk_epilogue", "w4c_gemm", "attn_prefill_vt", "attn_prefill", "w8_gemm", "w8_dequant_epilogue", "w8_outlier", "topk_sample"};

}  // namespace parrot

using namespace parrot;

extern "C" {

int parrot_version(void) { return PARROT_ABI_VERSION; }

const char* parrot_last_error(void) { return g_err; }

const char* parrot_kernel_name(int kernel_id) {
    if (kernel_id < 0 || kernel_id >= K_COUNT) return "?";
    return kNames[kernel_id];
}

int parrot_prof_begin(void) {
    ProfSink& ps = prof_sink();
    for (auto& r : ps.records) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    ps.records.clear();
    ps.enabled = true;
    return PARROT_OK;
}

int parrot_prof_end(int cap, int* kernel_ids_host, double* total_ms_host, int64_t* launches_host) {
    ProfSink& ps = prof_sink();
    ps.enabled = false;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "hipDeviceSynchronize");
    std::map<int, std::pair<double, int64_t>> agg;
    for (auto& r : ps.records) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.start, r.stop);
        if (e != hipSuccess) return hip_fail(e, "hipEventElapsedTime");
        auto& a = agg[r.kid];
        a.first += ms;
        a.second += 1;
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    ps.records.clear();
    int n = 0;
    for (auto& kv : agg) {
        if (n < cap) {
            kernel_ids_host[n] = kv.first;
            total_ms_host[n] = kv.second.first;
            launches_host[n] = kv.second.second;
        }
        ++n;
    }
    return n;
}

}  // extern "C"
