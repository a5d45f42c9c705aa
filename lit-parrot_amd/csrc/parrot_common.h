// Shared device/host helpers for libparrot_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

#include "../../include/parrot_hip.h"

namespace parrot {

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define PARROT_REQUIRE(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            ::parrot::set_error(__VA_ARGS__);     \
            return PARROT_EINVAL;                 \
        }                                         \
    } while (0)

#define PARROT_UNSUPPORTED(cond, ...)             \
    do {                                          \
        if (!(cond)) {                            \
            ::parrot::set_error(__VA_ARGS__);     \
            return PARROT_EUNSUPPORTED;           \
        }                                         \
    } while (0)

// The A/B switches behind the measurements of DESIGN.md exist only in a diagnostic build (hipcc -DPARROT_DIAG, `python
// lit-parrot_amd/_build.py --diag`): the shipped library reads no environment variable and keeps no tuning state.
#ifdef PARROT_DIAG
inline int tune_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
inline int tune_env(const char*, int dflt) { return dflt; }
#endif

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// argument checks shared by every Linear-shaped entry point
inline int check_linear_args(const char* who, const void* W, const void* W2, const void* x, int ldx, int M,
                             const void* residual, int ldr, const void* out, int ldo, int N, int K, int epi) {
    PARROT_REQUIRE(W && x && out, "%s: null pointer", who);
    PARROT_REQUIRE(M >= 1 && N >= 1 && K >= 1, "%s: bad shape M=%d N=%d K=%d", who, M, N, K);
    PARROT_REQUIRE(epi >= PARROT_EPI_NONE && epi <= PARROT_EPI_SWIGLU, "%s: unknown epilogue %d", who, epi);
    PARROT_REQUIRE((epi == PARROT_EPI_SWIGLU) == (W2 != nullptr), "%s: second weight iff SWIGLU", who);
    PARROT_REQUIRE((epi == PARROT_EPI_RESIDUAL) == (residual != nullptr), "%s: residual iff RESIDUAL epilogue", who);
    PARROT_REQUIRE(ldx >= K && ldo >= N && (!residual || ldr >= N), "%s: leading dimension too small", who);
    PARROT_REQUIRE(aligned16(W) && aligned16(x) && (ldx % 8 == 0) && (!W2 || aligned16(W2)),
                   "%s: W and x must be 16-byte aligned, ldx a multiple of 8", who);
    return PARROT_OK;
}

// ---------------------------------------------------------------- kernel ids (profiling sink)
enum KernelId : int {
    K_W4_GEMV = 0,
    K_W4_GEMV_DUAL,
    K_W4_REPACK,
    K_BF16_GEMV,
    K_BF16_GEMV_DUAL,
    K_W8_QUANT_ROWS,
    K_W8_PREP_ACT,
    K_W8_GEMV,
    K_RMSNORM,
    K_LAYERNORM,
    K_ROPE_KVAPPEND,
    K_ATTN_DECODE,
    K_ATTN_COMBINE,
    K_EMBEDDING,
    K_ARGMAX,
    K_W4_GEMM,
    K_BF16_GEMM,
    K_ATTN_FUSED,
    K_ENG_TOKEN,
    K_E4_REPACK,
    K_STOP_CHECK,
    K_GPTQ_BLOCK,
    K_W4C_GEMV,
    K_W4C_GEMV_DUAL,
    K_W4C_DEQUANT,
    K_GEMM_XSUM,
    K_GEMM_SPLITK,
    K_W4C_GEMM,
    K_ATTN_PREFILL_VT,
    K_ATTN_PREFILL,
    K_W8_GEMM,
    K_W8_DEQUANT,
    K_W8_OUTLIER,
    K_TOPK_SAMPLE,
    K_COUNT
};

struct ProfRecord {
    int kid;
    hipEvent_t start, stop;
};
struct ProfSink {
    bool enabled = false;
    std::vector<ProfRecord> records;
};
ProfSink& prof_sink();

// Launch through one place so that the profiling sink can bracket each launch with
// HIP events taken from the dispatch packet itself (hipExtLaunchKernelGGL).
template <typename... KArgs, typename... Args>
inline int launch(int kid, void (*kern)(KArgs...), dim3 grid, dim3 block, size_t shmem,
                  hipStream_t st, Args... args) {
    ProfSink& ps = prof_sink();
    if (ps.enabled) {
        ProfRecord r;
        r.kid = kid;
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess)
            return hip_fail(hipGetLastError(), "hipEventCreate");
        hipExtLaunchKernelGGL(kern, grid, block, shmem, st, r.start, r.stop, 0, static_cast<KArgs>(args)...);
        ps.records.push_back(r);
    } else {
        hipLaunchKernelGGL(kern, grid, block, shmem, st, static_cast<KArgs>(args)...);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return PARROT_OK;
}

// ---------------------------------------------------------------- device helpers
typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(static_cast<uint32_t>(v) << 16); }
__device__ __forceinline__ float bflo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bfhi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
// round-to-nearest-even, NaN preserving (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = static_cast<__bf16>(f);
    return __builtin_bit_cast(bf16_t, b);
}
// round a float to bf16 precision and come back (the reference's per-op bf16 rounding points)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// streamed-once data (weights): non-temporal 16-B / 4-B loads
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_nt16(const uint4* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t load_nt4(const uint32_t* p) { return __builtin_nontemporal_load(p); }

__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b),
                                           acc, false);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// DPP move helper: lanes whose row is masked out, or whose source lane is out of range, read 0
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}
// Sum over the 64 lanes with DPP only (no LDS crossbar, no waits); the total is valid in LANE 63 only.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp0<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp0<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp0<0x141>(v);       // row_half_mirror
    v += dpp0<0x140>(v);       // row_mirror: every lane now holds its 16-lane row total
    v += dpp0<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    v += dpp0<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
    return v;
}
// Eight wave sums at once: p[g] summed over the 64 lanes, the total of p[g] valid in the 8 lanes 8g .. 8g+7.
// gfx950 lane-swap instructions pack two half-reduced values into one register per step (v_permlane32_swap: a's upper 32
// lanes <-> b's lower 32; v_permlane16_swap: odd 16-lane rows of a <-> even rows of b), so the whole thing is 18
// instructions instead of 8 x 13 for eight separate wave_sum_to_lane63 calls.
__device__ __forceinline__ float swap_add32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);  // [a.lo + a.hi | b.lo + b.hi]
}
__device__ __forceinline__ float swap_add16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);  // rows: [a0 + a1 | b0 + b1 | a2 + a3 | b2 + b3]
}
__device__ __forceinline__ float wave_sum8(const float (&p)[8]) {
    const float s0 = swap_add32(p[0], p[4]), s1 = swap_add32(p[1], p[5]);
    const float s2 = swap_add32(p[2], p[6]), s3 = swap_add32(p[3], p[7]);
    float z0 = swap_add16(s0, s2);  // 16-lane rows: p0 | p2 | p4 | p6
    float z1 = swap_add16(s1, s3);  //               p1 | p3 | p5 | p7
    z0 += dpp0<0x128>(z0);          // row_ror:8 - both halves of a row hold the 8 pair sums
    z1 += dpp0<0x128>(z1);
    float w = (threadIdx.x & 8) ? z1 : z0;  // 8-lane group g = 2 * row + half holds p[g]
    w += dpp0<0x141>(w);                    // row_half_mirror
    w += dpp0<0xB1>(w);                     // quad_perm [1,0,3,2]
    w += dpp0<0x4E>(w);                     // quad_perm [2,3,0,1]
    return w;
}
__device__ __forceinline__ int wave_sum_i32_to_lane63(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// gelu(x) = 0.5 x (1 + erf(x / sqrt(2))) — torch.nn.functional.gelu default ("none" approximation)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

// ---------------------------------------------------------------- fused norm prologue of the GEMV kernels
// The Linear's input rows can be normalised on the fly (norm_1 / norm_2 / ln_f fused into the following Linear):
// every workgroup recomputes the row statistics from the raw activations it loads anyway (8-64 KB, L2 resident).
struct NormArgs {
    int kind;  // 0 none, 1 RMSNorm (lit_gpt/rmsnorm.py:17-21), 2 LayerNorm (torch.nn.LayerNorm)
    const bf16_t* weight;
    const bf16_t* bias;  // LayerNorm only, may be null
    float eps;
    int rsqrt_mode;  // RMSNorm: see parrot_rmsnorm
    int d;           // normalised width (= K of the Linear)
};

inline int make_norm_args(const parrot_norm_t* norm, int K, NormArgs* na) {
    na->kind = 0;
    na->weight = nullptr;
    na->bias = nullptr;
    na->eps = 0.f;
    na->rsqrt_mode = 0;
    na->d = K;
    if (norm == nullptr || norm->kind == 0) return PARROT_OK;
    PARROT_REQUIRE(norm->kind == 1 || norm->kind == 2, "norm prologue: kind must be 0, 1 (RMSNorm) or 2 (LayerNorm)");
    PARROT_REQUIRE(norm->weight != nullptr && aligned16(norm->weight) && (!norm->bias || aligned16(norm->bias)),
                   "norm prologue: weight/bias must be 16-byte aligned device pointers");
    na->kind = norm->kind;
    na->weight = (const bf16_t*)norm->weight;
    na->bias = (const bf16_t*)norm->bias;
    na->eps = norm->eps;
    na->rsqrt_mode = norm->rsqrt_mode;
    return PARROT_OK;
}

// sum over all waves of the workgroup (nwaves <= 16); every thread gets the total.  sh: >= 16 floats.
__device__ __forceinline__ float block_sum_waves(float v, float* sh, int nwaves) {
    v = wave_sum_to_lane63(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nwaves; ++i) t += sh[i];
    return t;
}

// lane-partial of the first statistic: sum of bf16(x*x) (RMSNorm) or sum of x (LayerNorm)
__device__ __forceinline__ float norm_stat1(uint32_t packed, int kind) {
    const float a = bflo(packed), b = bfhi(packed);
    return kind == 1 ? rbf(a * a) + rbf(b * b) : a + b;
}
__device__ __forceinline__ float norm_stat2(uint32_t packed, float mean) {
    const float a = bflo(packed) - mean, b = bfhi(packed) - mean;
    return a * a + b * b;
}
// finish the statistics: returns the scale r (and mean through *mean for LayerNorm)
__device__ __forceinline__ float norm_scale(const NormArgs& na, float stat) {
    if (na.kind == 1) {
        const float ms = rbf(stat / (float)na.d);
        const float t = rbf(ms + na.eps);
        return na.rsqrt_mode ? rbf(__fdiv_rn(1.0f, rbf(sqrtf(t)))) : rbf(__fdiv_rn(1.0f, sqrtf(t)));
    }
    return 1.0f / sqrtf(stat / (float)na.d + na.eps);
}
// normalise one packed pair with its weight / bias pairs; same rounding points as norm.hip
__device__ __forceinline__ uint32_t norm_apply(uint32_t x, uint32_t w, uint32_t b, int kind, float mean, float r) {
    float lo, hi;
    if (kind == 1) {
        lo = bflo(w) * rbf(bflo(x) * r);
        hi = bfhi(w) * rbf(bfhi(x) * r);
    } else {
        lo = (bflo(x) - mean) * r * bflo(w) + bflo(b);
        hi = (bfhi(x) - mean) * r * bfhi(w) + bfhi(b);
    }
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// shared GEMV epilogue: acc is the fp32 dot product (bias not yet added); res is the residual element (RESIDUAL only)
__device__ __forceinline__ bf16_t apply_epilogue_v(int epi, float acc, float acc2, const bf16_t* bias, float res, int col) {
    float v = acc;
    if (bias != nullptr) v += bf2f(bias[col]);
    v = rbf(v);
    if (epi == PARROT_EPI_RESIDUAL) {
        v = res + v;
    } else if (epi == PARROT_EPI_GELU) {
        v = gelu_erf(v);
    } else if (epi == PARROT_EPI_SWIGLU) {
        v = rbf(silu(v)) * rbf(acc2);
    }
    return f2bf(v);
}
// the same with the bias and residual ELEMENTS already in registers (requested early, off the kernel's tail)
__device__ __forceinline__ bf16_t apply_epilogue_vals(int epi, float acc, float acc2, bool has_bias, float bias_v, float res) {
    float v = acc;
    if (has_bias) v += bias_v;
    v = rbf(v);
    if (epi == PARROT_EPI_RESIDUAL) {
        v = res + v;
    } else if (epi == PARROT_EPI_GELU) {
        v = gelu_erf(v);
    } else if (epi == PARROT_EPI_SWIGLU) {
        v = rbf(silu(v)) * rbf(acc2);
    }
    return f2bf(v);
}
__device__ __forceinline__ bf16_t apply_epilogue(int epi, float acc, float acc2, const bf16_t* bias,
                                                 const bf16_t* residual, int col) {
    return apply_epilogue_v(epi, acc, acc2, bias, epi == PARROT_EPI_RESIDUAL ? bf2f(residual[col]) : 0.f, col);
}

// ---------------------------------------------------------------- agent-scope accessors
// For data that another workgroup of the same launch produces or consumes (attention partials, the persistent step):
// write-through stores / L2-bypassing loads; pair with s_waitcnt vmcnt(0) + barrier + one relaxed agent atomic.
__device__ __forceinline__ uint64_t ld_agent64(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_agent32(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace parrot
