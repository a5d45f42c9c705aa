// Dense bf16 Linear for the decode step: y = epilogue(x @ W^T + b), W (N, K) row-major bf16.
// Reference: torch.nn.Linear on the bf16-true path (lit_gpt/model.py:29,188,190,281-282,293-295).
//
// Memory-bound GEMV with the structure of the int4 kernel (w4.hip), which is where it was measured:
//   * workgroup = nslabs x wps waves; wave (slab c, j) owns the chunks of slab c of K (16 B = 8 bf16 per lane and load,
//     J loads per row, 64 lanes = 1 KiB contiguous each) of RU consecutive rows of row group j;
//   * the activations - and the optional RMSNorm / LayerNorm of them - are prepared ONCE per workgroup through LDS
//     (the first version normalised all of K in every 4-row wave: more VALU work than the dot products);
//   * weight rows go through a rolling window (PRIME rows requested ahead) so that a wave never sits in the issue of
//     loads the CU cannot accept yet; every load is unconditional (clamped), or the compiler loses its vmcnt bookkeeping;
//   * the grid is what is resident at once and every workgroup walks `iters` batches of rows;
//   * fp32 accumulation through v_dot2c_f32_bf16, wave reduction (DPP), cross-slab reduction in LDS, fused epilogue
//     whose bias / residual elements are requested at batch start.
#include "parrot_common.h"

namespace parrot {

constexpr int kDenseJ = 4;         // 16-B loads per lane per row (max)
constexpr int kDenseMaxSlabs = 16;  // K <= 32768

template <int M, bool DUAL, int J, int RU, int MAXW>
__global__ void __launch_bounds__(MAXW * 64)
bf16_gemv_kernel(const uint4* __restrict__ W, const uint4* __restrict__ W2, const bf16_t* __restrict__ x, int ldx,
                 const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr, bf16_t* out, int ldo, int N, int K,
                 int wps, int nslabs, int epi, int iters, NormArgs na) {
    constexpr int NW = DUAL ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char dense_smem[];  // normalised activations [M][K] bf16 (norm only)
    __shared__ float red[2][MAXW][RU * M * NW];
    __shared__ float stat[16];
    __builtin_amdgcn_s_setprio(3);  // prologue at raised priority, see w4.hip
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = nslabs * wps;
    const int slab = wave / wps, j = wave % wps;
    const int chunks = K >> 3;  // 16-B units per row
    const int c0 = (int)((int64_t)slab * chunks / nslabs), c1 = (int)((int64_t)(slab + 1) * chunks / nslabs);
    const int R = wps * RU;
    const int nthreads = nwaves * 64;

    int cidx[J];
    bool cok[J];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
        const int c = c0 + jj * 64 + lane;
        cok[jj] = c < c1;
        cidx[jj] = cok[jj] ? c : c1 - 1;
    }
    uint32_t xr[M][J][4];
    constexpr int kMaxChunkIt = 4;  // K <= 4 * 8 * nthreads is checked on the host for the norm path
    uint4 cx[M][kMaxChunkIt], cw[kMaxChunkIt], cb[kMaxChunkIt];
    if (na.kind == 0) {
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int jj = 0; jj < J; ++jj) {
                uint4 v = reinterpret_cast<const uint4*>(x + (int64_t)m * ldx)[cidx[jj]];
                if (!cok[jj]) v = make_uint4(0, 0, 0, 0);
                xr[m][jj][0] = v.x;
                xr[m][jj][1] = v.y;
                xr[m][jj][2] = v.z;
                xr[m][jj][3] = v.w;
            }
    } else {
#pragma unroll
        for (int it = 0; it < kMaxChunkIt; ++it) {
            cw[it] = cb[it] = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int m = 0; m < M; ++m) cx[m][it] = make_uint4(0, 0, 0, 0);
            if (it * nthreads < chunks) {  // workgroup-uniform: rounds past the row are skipped, not computed on zeros
                const int c = threadIdx.x + it * nthreads;
                const int cc = c < chunks ? c : chunks - 1;
                cw[it] = reinterpret_cast<const uint4*>(na.weight)[cc];
                if (na.kind == 2 && na.bias != nullptr) cb[it] = reinterpret_cast<const uint4*>(na.bias)[cc];
#pragma unroll
                for (int m = 0; m < M; ++m) cx[m][it] = reinterpret_cast<const uint4*>(x + (int64_t)m * ldx)[cc];
            }
        }
    }

    uint4 w[NW][RU][J];
    constexpr int PRIME = RU >= 4 ? 2 : 1;  // rows requested ahead (J x 1 KiB per wave each)
#define DENSE_ROW0(T) (((int)blockIdx.x + (T) * (int)gridDim.x) * R + j * RU)
#define DENSE_ISSUE_ROW(T, U)                                                       \
    {                                                                               \
        const int64_t row_ = min(DENSE_ROW0(T) + (U), N - 1);                       \
        _Pragma("unroll") for (int jj = 0; jj < J; ++jj) {                          \
            w[0][U][jj] = load_nt16(W + row_ * chunks + cidx[jj]);                  \
            if (DUAL) w[1][U][jj] = load_nt16(W2 + row_ * chunks + cidx[jj]);       \
        }                                                                           \
    }
#pragma unroll
    for (int u = 0; u < PRIME; ++u) DENSE_ISSUE_ROW(0, u)
    asm volatile("" ::: "memory");  // keep the remaining requests below the prologue

    if (na.kind != 0) {  // fused RMSNorm / LayerNorm of the input rows, once per workgroup, through LDS
        uint4* xn = reinterpret_cast<uint4*>(dense_smem);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float s1 = 0.f;
#pragma unroll
            for (int it = 0; it < kMaxChunkIt; ++it) {
                if (it * nthreads < chunks) {
                    const uint32_t dw[4] = {cx[m][it].x, cx[m][it].y, cx[m][it].z, cx[m][it].w};
                    float t = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) t += norm_stat1(dw[i], na.kind);
                    s1 += (threadIdx.x + it * nthreads < chunks) ? t : 0.f;
                }
            }
            s1 = block_sum_waves(s1, stat, nwaves);
            float mean = 0.f, r;
            if (na.kind == 2) {
                mean = s1 / (float)na.d;
                float s2 = 0.f;
#pragma unroll
                for (int it = 0; it < kMaxChunkIt; ++it) {
                    if (threadIdx.x + it * nthreads < chunks) {
                        const uint32_t dw[4] = {cx[m][it].x, cx[m][it].y, cx[m][it].z, cx[m][it].w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) s2 += norm_stat2(dw[i], mean);
                    }
                }
                r = norm_scale(na, block_sum_waves(s2, stat, nwaves));
            } else {
                r = norm_scale(na, s1);
            }
#pragma unroll
            for (int it = 0; it < kMaxChunkIt; ++it) {
                const int c = threadIdx.x + it * nthreads;
                if (c < chunks) {
                    const uint32_t dx[4] = {cx[m][it].x, cx[m][it].y, cx[m][it].z, cx[m][it].w};
                    const uint32_t dwt[4] = {cw[it].x, cw[it].y, cw[it].z, cw[it].w};
                    const uint32_t dbs[4] = {cb[it].x, cb[it].y, cb[it].z, cb[it].w};
                    uint32_t o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = norm_apply(dx[i], dwt[i], dbs[i], na.kind, mean, r);
                    xn[(int64_t)m * chunks + c] = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int jj = 0; jj < J; ++jj) {
                uint4 v = xn[(int64_t)m * chunks + cidx[jj]];
                if (!cok[jj]) v = make_uint4(0, 0, 0, 0);
                xr[m][jj][0] = v.x;
                xr[m][jj][1] = v.y;
                xr[m][jj][2] = v.z;
                xr[m][jj][3] = v.w;
            }
    }

    __builtin_amdgcn_s_setprio(0);
    const bf16_t* res_p = residual != nullptr ? residual : reinterpret_cast<const bf16_t*>(W);
    const bf16_t* bias_p = bias != nullptr ? bias : reinterpret_cast<const bf16_t*>(W);
    const int e_m = threadIdx.x % M, e_ur = threadIdx.x / M;
    for (int t = 0; t < iters; ++t) {
        float(*rd)[RU * M * NW] = red[t & 1];
        const int e_col = min(((int)blockIdx.x + t * (int)gridDim.x) * R + e_ur, N - 1);
        const bf16_t e_res = res_p[residual != nullptr ? (int64_t)e_m * ldr + e_col : 0];
        const bf16_t e_bias = bias_p[bias != nullptr ? e_col : 0];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            if (u + PRIME < RU) {
                DENSE_ISSUE_ROW(t, u + PRIME)
            } else if (t + 1 < iters) {
                DENSE_ISSUE_ROW(t + 1, u + PRIME - RU)
            }
            asm volatile("" ::: "memory");
            if (RU >= 4) {  // waves that are behind run at higher priority than waves that are ahead
                if (u == 0) __builtin_amdgcn_s_setprio(2);
                if (u == RU / 4) __builtin_amdgcn_s_setprio(1);
                if (u == RU / 2) __builtin_amdgcn_s_setprio(0);
            }
            float part[NW][M];
#pragma unroll
            for (int q = 0; q < NW; ++q)
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    float p0 = 0.f, p1 = 0.f;
#pragma unroll
                    for (int jj = 0; jj < J; ++jj) {
                        const uint4 ww = w[q][u][jj];
                        p0 = dot2_bf16(ww.x, xr[m][jj][0], p0);
                        p1 = dot2_bf16(ww.y, xr[m][jj][1], p1);
                        p0 = dot2_bf16(ww.z, xr[m][jj][2], p0);
                        p1 = dot2_bf16(ww.w, xr[m][jj][3], p1);
                    }
                    part[q][m] = wave_sum_to_lane63(p0 + p1);
                }
#pragma unroll
            for (int q = 0; q < NW; ++q)
#pragma unroll
                for (int m = 0; m < M; ++m)
                    if (lane == 63) rd[wave][(u * M + m) * NW + q] = part[q][m];
        }
        __syncthreads();
        if ((int)threadIdx.x < R * M) {
            const int m = threadIdx.x % M, ur = threadIdx.x / M;
            const int jj = ur / RU, u = ur % RU;
            const int col = ((int)blockIdx.x + t * (int)gridDim.x) * R + ur;
            if (col < N) {
                float a0 = 0.f, a1 = 0.f;
                for (int c = 0; c < nslabs; ++c) {
                    a0 += rd[c * wps + jj][(u * M + m) * NW];
                    if (DUAL) a1 += rd[c * wps + jj][(u * M + m) * NW + 1];
                }
                out[(int64_t)m * ldo + col] = apply_epilogue_vals(epi, a0, a1, bias != nullptr, bf2f(e_bias), bf2f(e_res));
            }
        }
    }
#undef DENSE_ISSUE_ROW
#undef DENSE_ROW0
}

template <int M, int J, int MAXW>
static int bf16_gemv_launch_jw(const void* W, const void* W2, const void* x, int ldx, const void* bias,
                               const void* residual, int ldr, void* out, int ldo, int N, int K, int epi, const NormArgs& na,
                               int nslabs, hipStream_t st) {
    constexpr int RU = 4, RUD = 2;  // rows per wave and batch (single weight / SwiGLU pair)
    const bool dual = epi == PARROT_EPI_SWIGLU;
    const int ru = dual ? RUD : RU;
    int wps = MAXW / nslabs;
    if (wps > 4) wps = 4;
    while (wps > 1 && (int64_t)wps * ru * 256 > N) --wps;  // small N: keep at least ~256 workgroups
    if (na.kind != 0)  // the cooperative norm covers 4 x 8 x nthreads elements of K
        while (wps * nslabs < MAXW && (K >> 3) > 4 * 64 * nslabs * wps) ++wps;
    const int nthreads = 64 * nslabs * wps;
    size_t lds = 0;
    if (na.kind != 0) {
        PARROT_UNSUPPORTED((K >> 3) <= 4 * nthreads, "bf16_gemv: fused norm needs K <= %d with this workgroup shape", 32 * nthreads);
        lds = (size_t)M * K * 2;
        PARROT_UNSUPPORTED(lds <= 64 * 1024, "bf16_gemv: fused norm needs %zu B of LDS", lds);
    }
    const int R = wps * ru;
    const int batches = (N + R - 1) / R;
    const int per_simd = J <= 5 ? 4 : (J <= 6 ? 3 : 2);  // waves per SIMD the build's VGPR count allows (<= 128 / 170 / 256)
    const int resident = 256 * (4 * per_simd / (nslabs * wps) > 0 ? 4 * per_simd / (nslabs * wps) : 1);
    const int iters = (batches + resident - 1) / resident;
    const dim3 grid((batches + iters - 1) / iters), block(nthreads);
    if (dual)
        return launch(K_BF16_GEMV_DUAL, bf16_gemv_kernel<M, true, J, RUD, MAXW>, grid, block, lds, st, (const uint4*)W,
                      (const uint4*)W2, (const bf16_t*)x, ldx, (const bf16_t*)bias, (const bf16_t*)residual, ldr,
                      (bf16_t*)out, ldo, N, K, wps, nslabs, epi, iters, na);
    return launch(K_BF16_GEMV, bf16_gemv_kernel<M, false, J, RU, MAXW>, grid, block, lds, st, (const uint4*)W,
                  (const uint4*)W2, (const bf16_t*)x, ldx, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out,
                  ldo, N, K, wps, nslabs, epi, iters, na);
}

template <int M, int J>
static int bf16_gemv_launch_j(const void* W, const void* W2, const void* x, int ldx, const void* bias,
                              const void* residual, int ldr, void* out, int ldo, int N, int K, int epi, const NormArgs& na,
                              int nslabs, hipStream_t st) {
    if (nslabs <= 8) return bf16_gemv_launch_jw<M, J, 8>(W, W2, x, ldx, bias, residual, ldr, out, ldo, N, K, epi, na, nslabs, st);
    if constexpr (J <= 4) return bf16_gemv_launch_jw<M, J, 16>(W, W2, x, ldx, bias, residual, ldr, out, ldo, N, K, epi, na, nslabs, st);
    set_error("bf16_gemv: K=%d needs more than 8 slabs of %d loads", K, J);
    return PARROT_EUNSUPPORTED;
}

template <int M>
static int bf16_gemv_launch(const void* W, const void* W2, const void* x, int ldx, const void* bias,
                            const void* residual, int ldr, void* out, int ldo, int N, int K, int epi, const NormArgs& na,
                            hipStream_t st) {
    const int chunks = K / 8;
    const int nslabs = (chunks + 64 * kDenseJ - 1) / (64 * kDenseJ);
    const int per_slab = (chunks + nslabs - 1) / nslabs;  // 16-B units of the largest slab
    const int jn = (per_slab + 63) / 64;
#define PARROT_DENSE_J(JV) \
    if (jn <= JV) return bf16_gemv_launch_j<M, JV>(W, W2, x, ldx, bias, residual, ldr, out, ldo, N, K, epi, na, nslabs, st)
    PARROT_DENSE_J(1);
    PARROT_DENSE_J(2);
    PARROT_DENSE_J(3);
    PARROT_DENSE_J(4);
#undef PARROT_DENSE_J
    set_error("bf16_gemv: K=%d too large", K);
    return PARROT_EUNSUPPORTED;
}

}  // namespace parrot

using namespace parrot;

extern "C" {

int parrot_bf16_gemv(const void* W, const void* W2, const void* x, int ldx, int M, const void* bias,
                     const void* residual, int ldr, void* out, int ldo, int N, int K, int epilogue,
                     const parrot_norm_t* norm, void* stream) {
    int rc = check_linear_args("bf16_gemv", W, W2, x, ldx, M, residual, ldr, out, ldo, N, K, epilogue);
    if (rc != PARROT_OK) return rc;
    PARROT_UNSUPPORTED(K % 8 == 0, "bf16_gemv: K=%d must be a multiple of 8", K);
    PARROT_UNSUPPORTED(K <= 64 * 8 * kDenseJ * kDenseMaxSlabs, "bf16_gemv: K=%d too large", K);
    PARROT_UNSUPPORTED(!(epilogue == PARROT_EPI_SWIGLU && bias), "bf16_gemv: SWIGLU epilogue takes no bias");
    NormArgs na;
    rc = make_norm_args(norm, K, &na);
    if (rc != PARROT_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* xb = (const bf16_t*)x;
    const bf16_t* rb = (const bf16_t*)residual;
    bf16_t* ob = (bf16_t*)out;
    for (int m0 = 0; m0 < M; m0 += 2) {
        const void* xm = xb + (int64_t)m0 * ldx;
        const void* rm = rb ? rb + (int64_t)m0 * ldr : nullptr;
        void* om = ob + (int64_t)m0 * ldo;
        if (M - m0 >= 2)
            rc = bf16_gemv_launch<2>(W, W2, xm, ldx, bias, rm, ldr, om, ldo, N, K, epilogue, na, st);
        else
            rc = bf16_gemv_launch<1>(W, W2, xm, ldx, bias, rm, ldr, om, ldo, N, K, epilogue, na, st);
        if (rc != PARROT_OK) return rc;
    }
    return PARROT_OK;
}

}  // extern "C"
