// RMSNorm (lit_gpt/rmsnorm.py:17-21) and LayerNorm (torch.nn.LayerNorm, lit_gpt/config.py:86-92), bf16 rows.
//
// One workgroup per token row; 16-B vector loads; the row stays in registers between the statistics
// pass and the scaling pass (d <= 256 threads * 8 * kNormV elements).
//
// RMSNorm follows the reference's bf16 choreography op by op (it computes in the input dtype, no fp32
// upcast): sq = bf16(x*x); ms = bf16(mean_fp32(sq)); r = bf16(rsqrt(bf16(ms + eps)));
// out = bf16(w * bf16(x * r)).  LayerNorm is torch's: fp32 statistics, one rounding at the output.
#include "parrot_common.h"

namespace parrot {

constexpr int kNormThreads = 256;
constexpr int kNormV = 8;  // 16-B chunks per thread: d <= 256*8*8 = 16384

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // protect sh from the previous use
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < kNormThreads / 64; ++i) t += sh[i];
    return t;
}

// XS: also write the per-group sums of the OUTPUT row (the rounded bf16 values) the int4 prompt GEMM folds its zero points with
// (gemm2.hip: sum_g(x), layout [group][Mpad]) - the same 16-lane butterfly as gemm2_xsum_kernel, bit for bit, without its
// launch.  xs_lanes = group / 8 (8 or 16 lanes of consecutive 16-byte chunks share a group); rows [M, Mpad) get zeros.
template <bool RMS, bool XS>
__global__ void __launch_bounds__(kNormThreads)
norm_kernel(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ weight, const bf16_t* __restrict__ bias,
            bf16_t* __restrict__ out, int ldo, int d, float eps, int rsqrt_mode, int M, int Mpad, int xs_lanes, float* __restrict__ xs) {
    __shared__ float sh[kNormThreads / 64];
    const int chunks = d >> 3;
    if (XS && (int)blockIdx.x >= M) {  // padding row of the last 128-row tile
        for (int g = threadIdx.x; g < chunks / xs_lanes; g += kNormThreads) xs[(int64_t)g * Mpad + blockIdx.x] = 0.f;
        return;
    }
    const uint4* xp = reinterpret_cast<const uint4*>(x + (int64_t)blockIdx.x * ldx);
    uint4 v[kNormV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kNormV; ++i) {
        const int c = threadIdx.x + i * kNormThreads;
        v[i] = (c < chunks) ? xp[c] : make_uint4(0, 0, 0, 0);
        const uint32_t dw[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = bflo(dw[j]), b = bfhi(dw[j]);
            if (RMS)
                s += rbf(a * a) + rbf(b * b);
            else
                s += a + b;
        }
    }
    s = block_sum(s, sh);
    float mean = 0.f, r;
    if (RMS) {
        const float ms = rbf(s / (float)d);
        const float t = rbf(ms + eps);
        // rsqrt_mode 0: one rounding (torch's GPU kernel).  1: torch's CPU scalar path for < 16-element tensors
        // (one value per token row): sqrt rounded to bf16, then the reciprocal rounded again.
        r = rsqrt_mode ? rbf(__fdiv_rn(1.0f, rbf(sqrtf(t)))) : rbf(__fdiv_rn(1.0f, sqrtf(t)));
    } else {
        mean = s / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < kNormV; ++i) {
            const int c = threadIdx.x + i * kNormThreads;
            if (c < chunks) {
                const uint32_t dw[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = bflo(dw[j]) - mean, b = bfhi(dw[j]) - mean;
                    q += a * a + b * b;
                }
            }
        }
        q = block_sum(q, sh);
        r = 1.0f / sqrtf(q / (float)d + eps);
    }
    const uint4* wp = reinterpret_cast<const uint4*>(weight);
    const uint4* bp = reinterpret_cast<const uint4*>(bias);
    uint4* op = reinterpret_cast<uint4*>(out + (int64_t)blockIdx.x * ldo);
#pragma unroll
    for (int i = 0; i < kNormV; ++i) {
        const int c = threadIdx.x + i * kNormThreads;
        float part = 0.f;
        if (XS && i * kNormThreads >= chunks) break;  // (uniform: the butterfly below runs on whole waves)
        if (c < chunks) {
            const uint4 w4 = wp[c];
            const uint32_t dw[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            const uint32_t ww[4] = {w4.x, w4.y, w4.z, w4.w};
            uint32_t bb[4] = {0, 0, 0, 0};
            if (!RMS && bias != nullptr) {
                const uint4 b4 = bp[c];
                bb[0] = b4.x; bb[1] = b4.y; bb[2] = b4.z; bb[3] = b4.w;
            }
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float lo, hi;
                if (RMS) {
                    lo = bflo(ww[j]) * rbf(bflo(dw[j]) * r);
                    hi = bfhi(ww[j]) * rbf(bfhi(dw[j]) * r);
                } else {
                    lo = (bflo(dw[j]) - mean) * r * bflo(ww[j]) + bflo(bb[j]);
                    hi = (bfhi(dw[j]) - mean) * r * bfhi(ww[j]) + bfhi(bb[j]);
                }
                o[j] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
            }
            op[c] = make_uint4(o[0], o[1], o[2], o[3]);
            if (XS) part = (bflo(o[0]) + bfhi(o[0])) + (bflo(o[1]) + bfhi(o[1])) + (bflo(o[2]) + bfhi(o[2])) + (bflo(o[3]) + bfhi(o[3]));
        }
        if (XS) {
            part += __shfl_xor(part, 1);
            part += __shfl_xor(part, 2);
            part += __shfl_xor(part, 4);
            if (xs_lanes == 16) part += __shfl_xor(part, 8);
            if (c < chunks && (c & (xs_lanes - 1)) == 0) xs[(int64_t)(c / xs_lanes) * Mpad + blockIdx.x] = part;
        }
    }
}

static int norm_check(const char* who, const void* x, int ldx, const void* w, const void* out, int ldo, int M, int d) {
    PARROT_REQUIRE(x && w && out, "%s: null pointer", who);
    PARROT_REQUIRE(M >= 1 && d >= 8, "%s: bad shape M=%d d=%d", who, M, d);
    PARROT_UNSUPPORTED(d % 8 == 0 && d <= kNormThreads * 8 * kNormV, "%s: d=%d must be a multiple of 8 and <= %d", who, d,
                       kNormThreads * 8 * kNormV);
    PARROT_REQUIRE(ldx >= d && ldo >= d && ldx % 8 == 0 && ldo % 8 == 0, "%s: leading dims must be >= d and multiples of 8", who);
    PARROT_REQUIRE(aligned16(x) && aligned16(w) && aligned16(out), "%s: pointers must be 16-byte aligned", who);
    return PARROT_OK;
}

// the norm of a prompt's rows + their per-group sums in one launch (parrot_w4_gemm with a norm argument)
bool norm_xsum_takes(const NormArgs& na, int d, int G) { return (na.kind == 1 || na.kind == 2) && (G == 64 || G == 128) && d % G == 0 && d <= kNormThreads * 8 * kNormV; }
int norm_xsum_launch(const NormArgs& na, const void* x, int ldx, void* xn, int ldo, int M, int Mpad, int d, int G, float* xs, hipStream_t st) {
    const int rc = norm_check("w4_gemm (norm)", x, ldx, na.weight, xn, ldo, M, d);
    if (rc != PARROT_OK) return rc;
    if (na.kind == 1)
        return launch(K_RMSNORM, norm_kernel<true, true>, dim3(Mpad), dim3(kNormThreads), 0, st, (const bf16_t*)x, ldx, na.weight,
                      (const bf16_t*)nullptr, (bf16_t*)xn, ldo, d, na.eps, na.rsqrt_mode, M, Mpad, G / 8, xs);
    return launch(K_LAYERNORM, norm_kernel<false, true>, dim3(Mpad), dim3(kNormThreads), 0, st, (const bf16_t*)x, ldx, na.weight, na.bias,
                  (bf16_t*)xn, ldo, d, na.eps, 0, M, Mpad, G / 8, xs);
}

}  // namespace parrot

using namespace parrot;

extern "C" {

int parrot_rmsnorm(const void* x, int ldx, const void* weight, void* out, int ldo, int M, int d, float eps,
                   int rsqrt_mode, void* stream) {
    const int rc = norm_check("rmsnorm", x, ldx, weight, out, ldo, M, d);
    if (rc != PARROT_OK) return rc;
    return launch(K_RMSNORM, norm_kernel<true, false>, dim3(M), dim3(kNormThreads), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                  (const bf16_t*)weight, (const bf16_t*)nullptr, (bf16_t*)out, ldo, d, eps, rsqrt_mode, M, M, 16, (float*)nullptr);
}

int parrot_layernorm(const void* x, int ldx, const void* weight, const void* bias, void* out, int ldo, int M, int d,
                     float eps, void* stream) {
    const int rc = norm_check("layernorm", x, ldx, weight, out, ldo, M, d);
    if (rc != PARROT_OK) return rc;
    PARROT_REQUIRE(!bias || aligned16(bias), "layernorm: bias must be 16-byte aligned");
    return launch(K_LAYERNORM, norm_kernel<false, false>, dim3(M), dim3(kNormThreads), 0, (hipStream_t)stream, (const bf16_t*)x,
                  ldx, (const bf16_t*)weight, (const bf16_t*)bias, (bf16_t*)out, ldo, d, eps, 0, M, M, 16, (float*)nullptr);
}

}  // extern "C"
