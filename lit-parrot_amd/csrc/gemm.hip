// Prefill Linears on the matrix cores: out[M, N] = epilogue(x[M, K] @ W[N, K]^T + b) for many token rows.
//
// Reference: torch.nn.Linear on bf16 (lit_gpt/model.py:29,188,190,281-295) and ColBlockQuantizedLinear.forward /
// qlinear_4bit_weight for int4 (quantize/gptq.py:156-201, :254-264; the Triton kernel there is per-channel only and
// pads M to 256).  Both operands are K-contiguous, which is exactly the A/B fragment shape of
// v_mfma_f32_32x32x16_bf16 (lane l holds 8 consecutive k of row/column l & 31), so tiles go global -> registers -> LDS
// in 16-byte pieces and come back as ds_read_b128 fragments.
//
//   tile 128 (M) x 128 (N) x 32 (K) per 256-thread workgroup, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles;
//   LDS rows padded to 80 B so that the 16 lanes of a ds_read_b128 group hit 16 different 16-byte slots.
//
// int4: a W4K slice (16 B = 32 k of one output row) is exactly one row of the K-tile.  It is expanded to the bf16 values
// 128 + q (exact in bf16, two at a time: ((dword >> 4i) & 0x000F000F) | 0x43004300) and multiplied on the MFMA like any
// bf16 operand; at every quantisation-group boundary the group's partial product is folded into the result with the
// group's scale / zero and the row's activation sum:  acc += scale * (partial - (128 + zero) * sum_g(x)).  These are
// the numerics of the decode GEMV (w4.hip), so prefill and decode agree.  sum_g(x) comes from a small pre-pass.
#include "parrot_common.h"
#include "w4_plan.h"

namespace parrot {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int GBM = 128, GBN = 128, GBK = 32;
constexpr int GLD = 40;  // LDS row stride in bf16 elements (80 B)

// sum of the activations of every (row, quantisation group): xsum[m][g] = sum_{k in group g} x[m][k]
__global__ void __launch_bounds__(256)
gemm_xsum_kernel(const bf16_t* __restrict__ x, int ldx, int M, int K, int G, int ngroups, float* __restrict__ xsum) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)M * ngroups) return;
    const int m = (int)(t / ngroups), g = (int)(t % ngroups);
    const int k0 = g * G, k1 = min(K, k0 + G);
    const uint4* p = reinterpret_cast<const uint4*>(x + (int64_t)m * ldx + k0);
    float s = 0.f;
    for (int c = 0; c < (k1 - k0) / 8; ++c) {
        const uint4 v = p[c];
        s += (bflo(v.x) + bfhi(v.x)) + (bflo(v.y) + bfhi(v.y)) + (bflo(v.z) + bfhi(v.z)) + (bflo(v.w) + bfhi(v.w));
    }
    xsum[t] = s;
}

template <bool W4>
__global__ void __launch_bounds__(256)
gemm_kernel(const bf16_t* __restrict__ A, int lda, int M, const void* __restrict__ Wv, const void* __restrict__ W2v, int N,
            int K, const float* __restrict__ xsum, const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr,
            bf16_t* out, int ldo, int epi, W4Plan plan) {
    __shared__ __attribute__((aligned(16))) bf16_t As[GBM * GLD];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[GBN * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    const int lr = lane & 31, lh = lane >> 5;
    const int ktiles = K / GBK;
    const int Gs = W4 ? plan.Gs : ktiles;  // K-tiles per quantisation group
    const int ngroups = W4 ? plan.ngroups : 1;

    f32x16_t total[2][2];
    uint32_t gate[2][2][8];  // SwiGLU: bf16(silu(bf16(fc_1))) of the first pass, packed
    const int npass = (epi == PARROT_EPI_SWIGLU) ? 2 : 1;

    for (int pass = 0; pass < npass; ++pass) {
        const void* Wp = pass ? W2v : Wv;
        f32x16_t acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[i][jn][r] = 0.f;
                    total[i][jn][r] = 0.f;
                }
        int slab = 0;
        for (int kt = 0; kt < ktiles; ++kt) {
            // ---- stage the K-tile: A rows (activations) and B rows (weights), 16-byte pieces
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = tid + it * 256;
                const int row = idx >> 2, c = idx & 3;
                const int64_t gm = min(m0 + row, M - 1);
                const uint4 v = reinterpret_cast<const uint4*>(A + gm * lda + (int64_t)kt * GBK)[c];
                *reinterpret_cast<uint4*>(&As[row * GLD + c * 8]) = v;
            }
            if (W4) {
                while (slab + 1 < plan.nslabs && kt >= plan.slab[slab + 1].slice0) ++slab;
                if (tid < GBN) {
                    const int64_t gn = min(n0 + tid, N - 1);
                    const uint4* rec = reinterpret_cast<const uint4*>(Wp) + gn * plan.row16;
                    const uint4 q = rec[plan.slab[slab].w_off16 + (kt - plan.slab[slab].slice0)];
                    const uint32_t dw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        uint32_t o[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = ((dw[d] >> (4 * i)) & 0x000F000Fu) | 0x43004300u;
                        *reinterpret_cast<uint4*>(&Bs[tid * GLD + d * 8]) = make_uint4(o[0], o[1], o[2], o[3]);
                    }
                }
            } else {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int idx = tid + it * 256;
                    const int row = idx >> 2, c = idx & 3;
                    const int64_t gn = min(n0 + row, N - 1);
                    const uint4 v = reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(Wp) + gn * K + (int64_t)kt * GBK)[c];
                    *reinterpret_cast<uint4*>(&Bs[row * GLD + c * 8]) = v;
                }
            }
            __syncthreads();
            // ---- 2 k-steps x (2 x 2) MFMA 32x32x16
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8_t af[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint4 va = *reinterpret_cast<const uint4*>(&As[(wm * 64 + i * 32 + lr) * GLD + s * 16 + lh * 8]);
                    af[i] = __builtin_bit_cast(bf16x8_t, va);
                    const uint4 vb = *reinterpret_cast<const uint4*>(&Bs[(wn * 64 + i * 32 + lr) * GLD + s * 16 + lh * 8]);
                    bfr[i] = __builtin_bit_cast(bf16x8_t, vb);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[jn], acc[i][jn], 0, 0, 0);
            }
            __syncthreads();
            if (W4 && ((kt + 1) % Gs == 0 || kt + 1 == ktiles)) {
                // ---- quantisation-group boundary: fold the group's partial product into the result
                const int g = kt / Gs;
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    const int64_t gn = min(n0 + wn * 64 + jn * 32 + lr, N - 1);
                    const uint4* rec = reinterpret_cast<const uint4*>(Wp) + gn * plan.row16;
                    const uint32_t mt = reinterpret_cast<const uint32_t*>(rec + plan.slab[slab].meta_off16)[g - plan.slab[slab].g0];
                    const float sc = bflo(mt), zz = 128.0f + bfhi(mt);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int64_t gm = min(m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1);
                            total[i][jn][r] += sc * (acc[i][jn][r] - zz * xsum[gm * ngroups + g]);
                            acc[i][jn][r] = 0.f;
                        }
                }
            }
        }
        if (!W4) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) total[i][jn] = acc[i][jn];
        }
        if (npass == 2 && pass == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const bf16_t g0 = f2bf(silu(rbf(total[i][jn][r]))), g1 = f2bf(silu(rbf(total[i][jn][r + 1])));
                        gate[i][jn][r >> 1] = (uint32_t)g0 | ((uint32_t)g1 << 16);
                    }
        }
    }
    // ---- epilogue: C layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int col = n0 + wn * 64 + jn * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M && col < N) {
                    bf16_t o;
                    if (epi == PARROT_EPI_SWIGLU) {
                        const uint32_t gp = gate[i][jn][r >> 1];
                        const float gv = (r & 1) ? bfhi(gp) : bflo(gp);
                        o = f2bf(gv * rbf(total[i][jn][r]));
                    } else {
                        o = apply_epilogue(epi, total[i][jn][r], 0.f, bias, residual ? residual + (int64_t)row * ldr : nullptr, col);
                    }
                    out[(int64_t)row * ldo + col] = o;
                }
            }
        }
}

}  // namespace parrot

using namespace parrot;

extern "C" {

int64_t parrot_gemm_workspace_floats(int M, int K, int group) {
    if (group <= 0 || group > K) group = K;
    return (int64_t)M * ((K + group - 1) / group);
}

int parrot_bf16_gemm(const void* W, const void* W2, const void* x, int ldx, int M, const void* bias, const void* residual,
                     int ldr, void* out, int ldo, int N, int K, int epilogue, const parrot_norm_t* norm, void* stream) {
    if (M <= 8) return parrot_bf16_gemv(W, W2, x, ldx, M, bias, residual, ldr, out, ldo, N, K, epilogue, norm, stream);
    int rc = check_linear_args("bf16_gemm", W, W2, x, ldx, M, residual, ldr, out, ldo, N, K, epilogue);
    if (rc != PARROT_OK) return rc;
    PARROT_UNSUPPORTED(norm == nullptr || norm->kind == 0, "bf16_gemm: apply the norm to the rows first (parrot_rmsnorm / parrot_layernorm)");
    PARROT_UNSUPPORTED(K % GBK == 0, "bf16_gemm: K=%d must be a multiple of %d", K, GBK);
    PARROT_UNSUPPORTED(!(epilogue == PARROT_EPI_SWIGLU && bias), "bf16_gemm: SWIGLU epilogue takes no bias");
    PARROT_REQUIRE(M <= 65535 * GBM, "bf16_gemm: M too large");
    W4Plan plan = {};
    const dim3 grid((N + GBN - 1) / GBN, (M + GBM - 1) / GBM);
    return launch(K_BF16_GEMM, gemm_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, M, W, W2, N, K,
                  (const float*)nullptr, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, epilogue, plan);
}

int parrot_w4_gemm(const void* packed, const void* packed2, const void* x, int ldx, int M, const void* bias,
                   const void* residual, int ldr, void* out, int ldo, int N, int K, int group, int epilogue,
                   const parrot_norm_t* norm, void* workspace, void* stream) {
    if (M <= 8)
        return parrot_w4_gemv(packed, packed2, x, ldx, M, bias, residual, ldr, out, ldo, N, K, group, epilogue, norm, stream);
    int rc = check_linear_args("w4_gemm", packed, packed2, x, ldx, M, residual, ldr, out, ldo, N, K, epilogue);
    if (rc != PARROT_OK) return rc;
    PARROT_UNSUPPORTED(norm == nullptr || norm->kind == 0, "w4_gemm: apply the norm to the rows first (parrot_rmsnorm / parrot_layernorm)");
    PARROT_REQUIRE(workspace != nullptr, "w4_gemm: workspace of parrot_gemm_workspace_floats(M, K, group) floats required");
    PARROT_UNSUPPORTED(!(epilogue == PARROT_EPI_SWIGLU && bias), "w4_gemm: SWIGLU epilogue takes no bias");
    W4Plan plan;
    rc = w4_make_plan(N, K, group, &plan);
    if (rc != PARROT_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int G = plan.Gs * 32;
    const int64_t n = (int64_t)M * plan.ngroups;
    rc = launch(K_W4_GEMM, gemm_xsum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const bf16_t*)x, ldx, M, K, G,
                plan.ngroups, (float*)workspace);
    if (rc != PARROT_OK) return rc;
    const dim3 grid((N + GBN - 1) / GBN, (M + GBM - 1) / GBM);
    return launch(K_W4_GEMM, gemm_kernel<true>, grid, dim3(256), 0, st, (const bf16_t*)x, ldx, M, packed, packed2, N, K,
                  (const float*)workspace, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, epilogue, plan);
}

}  // extern "C"
