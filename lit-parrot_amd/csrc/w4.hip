// int4 (GPTQ ColBlockQuantizedLinear format) weight path: W4K repack + dequant-into-GEMV.
//
// Reference semantics: quantize/gptq.py:205-264 (ColBlockQuantizedLinear), :243-252 (get_weight),
// Triton kernel :63-153 (per-channel only, no bias).  This file computes
//     y[m, o] = sum_k x[m, k] * (q[o, k] - zero[o, k/G]) * scale[o, k/G]      (+ bias, epilogue)
// with fp32 accumulation, for any group size G that is a multiple of 32.
//
// W4K layout.  A *slice* is 32 consecutive k of one output row = 16 bytes.
// The K axis is cut into <=16 *slabs* of <=64 slices (one slab per wavefront, one slice per lane).
// Row record (row-major over output rows):
//     for each slab c:  [nslices_c x 16 B weight slices][ngroups_c x 4 B {scale bf16, zero bf16}, padded to 16 B]
// Inside a slice, dword d holds k = 8d .. 8d+7; nibble at bits 4i (i<4) is k = 8d+2i and the nibble at bits
// 16+4i is k = 8d+2i+1, so that ((dword >> 4i) & 0x000F000F) | 0x43004300 is the bf16 pair
// {128+q[8d+2i], 128+q[8d+2i+1]}, the operand of v_dot2c_f32_bf16 against the natural bf16 pair of x.
// The +128 bias is removed with the per-lane sum of x: s * (sum x*(128+q) - (128+z) * sum x).
#include "parrot_common.h"
#include "w4_plan.h"

namespace parrot {

// ------------------------------------------------------------------------------------------ repack
// direction 0: reference -> W4K, 1: W4K -> reference.  One thread per (slice-or-meta unit, row);
// rows are the fast index so that the reference side ([K/2][N] bytes) is read/written coalesced.
__global__ void __launch_bounds__(256)
w4_repack_kernel(uint8_t* __restrict__ qref, bf16_t* __restrict__ scales, bf16_t* __restrict__ zeros,
                 uint4* __restrict__ packed, int N, int direction, W4Plan plan) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int o = (int)(tid % N);
    const int unit = (int)(tid / N);
    if (unit >= plan.row16) return;
    // which slab / which part of the record is this 16-B unit?
    int c = 0;
    while (c + 1 < plan.nslabs && unit >= plan.slab[c + 1].w_off16) ++c;
    const W4Slab sl = plan.slab[c];
    uint4* rec = packed + (int64_t)o * plan.row16 + unit;
    if (unit < sl.meta_off16) {
        const int t = sl.slice0 + (unit - sl.w_off16);  // global slice index
        if (direction == 0) {
            uint32_t dw[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint32_t v = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t b = qref[(int64_t)(t * 16 + d * 4 + i) * N + o];
                    v |= (b & 0xFu) << (4 * i);
                    v |= (b >> 4) << (16 + 4 * i);
                }
                dw[d] = v;
            }
            *rec = make_uint4(dw[0], dw[1], dw[2], dw[3]);
        } else {
            const uint4 v4 = *rec;
            const uint32_t dw[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t lo = (dw[d] >> (4 * i)) & 0xFu, hi = (dw[d] >> (16 + 4 * i)) & 0xFu;
                    qref[(int64_t)(t * 16 + d * 4 + i) * N + o] = (uint8_t)(lo | (hi << 4));
                }
        }
    } else {
        const int mu = unit - sl.meta_off16;  // 16-B unit inside the meta block: 4 groups
        if (direction == 0) {
            uint32_t dw[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gi = mu * 4 + i;
                if (gi < sl.ngroups) {
                    const int64_t idx = (int64_t)o * plan.ngroups + sl.g0 + gi;
                    dw[i] = (uint32_t)scales[idx] | ((uint32_t)zeros[idx] << 16);
                }
            }
            *rec = make_uint4(dw[0], dw[1], dw[2], dw[3]);
        } else {
            const uint4 v4 = *rec;
            const uint32_t dw[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gi = mu * 4 + i;
                if (gi < sl.ngroups) {  // a group spanning several slabs is written by each of them (same value)
                    const int64_t idx = (int64_t)o * plan.ngroups + sl.g0 + gi;
                    scales[idx] = (bf16_t)(dw[i] & 0xffffu);
                    zeros[idx] = (bf16_t)(dw[i] >> 16);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ GEMV
constexpr int kMaxRows = 16;  // rows per workgroup

// RU = rows whose loads are in flight together per wave; MAXW = waves per workgroup the build allows (the register budget
// follows from it: 8 waves -> 256 VGPRs, 16 waves -> 128)
template <int M, bool DUAL, int RU, int MAXW>
__global__ void __launch_bounds__(MAXW * 64)
w4_gemv_kernel(const uint4* __restrict__ W, const uint4* __restrict__ W2, const bf16_t* __restrict__ x, int ldx,
               const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr, bf16_t* out, int ldo, int N,
               int rows_per_wg, int epi, NormArgs na, W4Plan plan) {
    constexpr int NW = DUAL ? 2 : 1;
    __shared__ float red[kMaxSlabs][kMaxRows * M * NW];
    __shared__ float stat[kMaxSlabs];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const W4Slab sl = plan.slab[wave];
    const bool active = lane < sl.nslices;
    const int lslice = active ? lane : sl.nslices - 1;
    const int gslice = sl.slice0 + lslice;
    const int gl = gslice / plan.Gs - sl.g0;
    const int64_t row16 = plan.row16;
    const int r_begin = blockIdx.x * rows_per_wg;
    const int r_end = min(N, r_begin + rows_per_wg);

    constexpr int kU = RU;
    uint4 w[NW][kU];
    uint32_t mt[NW][kU];
    // All kU row loads of a batch are issued back to back before anything waits.  Weights are read exactly once per
    // token: non-temporal loads keep them from displacing the activations in L2.
#define W4_LOAD_BATCH(R0)                                                                            \
    _Pragma("unroll") for (int u = 0; u < kU; ++u) {                                                 \
        const int64_t row = min((R0) + u, N - 1);                                                    \
        const uint4* rec = W + row * row16;                                                          \
        w[0][u] = load_nt16(rec + sl.w_off16 + lslice);                                              \
        mt[0][u] = load_nt4(reinterpret_cast<const uint32_t*>(rec + sl.meta_off16) + gl);           \
        if (DUAL) {                                                                                  \
            const uint4* rec2 = W2 + row * row16;                                                    \
            w[1][u] = load_nt16(rec2 + sl.w_off16 + lslice);                                         \
            mt[1][u] = load_nt4(reinterpret_cast<const uint32_t*>(rec2 + sl.meta_off16) + gl);       \
        }                                                                                            \
    }
    // Load order matters because vmcnt retires in order: first the small L2-resident operands (activations, norm
    // parameters), then the first batch of weights.  The norm prologue then only waits for the former while the
    // HBM stream of the latter is already running.
    // this lane's 32 activations per row of x, as 16 packed bf16 pairs
    uint32_t xr[M][16];
    float xs[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const uint4* xp = reinterpret_cast<const uint4*>(x + (int64_t)m * ldx + (int64_t)gslice * 32);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint4 v = xp[j];
            if (!active) v = make_uint4(0, 0, 0, 0);
            xr[m][4 * j + 0] = v.x;
            xr[m][4 * j + 1] = v.y;
            xr[m][4 * j + 2] = v.z;
            xr[m][4 * j + 3] = v.w;
        }
    }
    uint32_t nw[16], nb[16];
    if (na.kind != 0) {
        const uint4* wp = reinterpret_cast<const uint4*>(na.weight + (int64_t)gslice * 32);
        const uint4* bp = reinterpret_cast<const uint4*>(na.bias + (int64_t)gslice * 32);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 v = wp[j];
            nw[4 * j] = v.x; nw[4 * j + 1] = v.y; nw[4 * j + 2] = v.z; nw[4 * j + 3] = v.w;
            uint4 b = make_uint4(0, 0, 0, 0);
            if (na.kind == 2 && na.bias != nullptr) b = bp[j];
            nb[4 * j] = b.x; nb[4 * j + 1] = b.y; nb[4 * j + 2] = b.z; nb[4 * j + 3] = b.w;
        }
    }
    W4_LOAD_BATCH(r_begin)

    if (na.kind != 0) {  // fused RMSNorm / LayerNorm of the input rows (wave-uniform branch)
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s1 += norm_stat1(xr[m][i], na.kind);
            s1 = block_sum_waves(s1, stat, plan.nslabs);
            float mean = 0.f, r;
            if (na.kind == 2) {
                mean = s1 / (float)na.d;
                float s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) s2 += active ? norm_stat2(xr[m][i], mean) : 0.f;
                r = norm_scale(na, block_sum_waves(s2, stat, plan.nslabs));
            } else {
                r = norm_scale(na, s1);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) xr[m][i] = active ? norm_apply(xr[m][i], nw[i], nb[i], na.kind, mean, r) : 0u;
        }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += bflo(xr[m][i]) + bfhi(xr[m][i]);
        xs[m] = s;
    }

    for (int r0 = r_begin; r0 < r_end; r0 += kU) {
        if (r0 != r_begin) {
            W4_LOAD_BATCH(r0)
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                const float s = bflo(mt[q][u]);
                const float zz = 128.0f + bfhi(mt[q][u]);
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const float p = w4_slice_dot(w[q][u], xr[m]);
                    float v = s * (p - zz * xs[m]);
                    v = wave_sum_to_lane63(v);
                    if (lane == 63) red[wave][((r0 - r_begin + u) * M + m) * NW + q] = v;
                }
            }
        }
    }
#undef W4_LOAD_BATCH
    __syncthreads();
    const int nrows = r_end - r_begin;
    if ((int)threadIdx.x < nrows * M) {
        const int ur = threadIdx.x / M, m = threadIdx.x % M;
        float a0 = 0.f, a1 = 0.f;
        for (int c = 0; c < plan.nslabs; ++c) {
            a0 += red[c][(ur * M + m) * NW];
            if (DUAL) a1 += red[c][(ur * M + m) * NW + 1];
        }
        const int col = r_begin + ur;
        out[(int64_t)m * ldo + col] =
            apply_epilogue(epi, a0, a1, bias, residual ? residual + (int64_t)m * ldr : nullptr, col);
    }
}

static int g_rows_per_wg_override = 0;  // tuning hook (tools/microbench.py), 0 = heuristic

static int pick_rows_per_wg(int N) {
    if (g_rows_per_wg_override > 0) return g_rows_per_wg_override;
    if (N >= 16 * 2048) return 16;
    return 8;
}

template <int M, bool DUAL, int RU, int MAXW>
static int w4_gemv_launch_v(const void* packed, const void* packed2, const void* x, int ldx, const void* bias,
                            const void* residual, int ldr, void* out, int ldo, int N, int epi, const NormArgs& na,
                            const W4Plan& plan, hipStream_t st) {
    int R = pick_rows_per_wg(N);
    if (R < RU) R = RU;
    const dim3 grid((N + R - 1) / R), block(64 * plan.nslabs);
    return launch(DUAL ? K_W4_GEMV_DUAL : K_W4_GEMV, w4_gemv_kernel<M, DUAL, RU, MAXW>, grid, block, 0, st,
                  (const uint4*)packed, (const uint4*)packed2, (const bf16_t*)x, ldx, (const bf16_t*)bias,
                  (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, N, R, epi, na, plan);
}

template <int M>
static int w4_gemv_launch(const void* packed, const void* packed2, const void* x, int ldx, const void* bias,
                          const void* residual, int ldr, void* out, int ldo, int N, int epi, const NormArgs& na,
                          const W4Plan& plan, hipStream_t st) {
    // rows in flight: 8 for the single-row decode kernel, 4 when a second weight or more rows share the registers
    constexpr int RU1 = (M == 1) ? 8 : 4;
    constexpr int RU2 = (M <= 2) ? 4 : 2;
#define PARROT_W4_GO(DUALV, RUV, MAXWV) \
    return w4_gemv_launch_v<M, DUALV, RUV, MAXWV>(packed, packed2, x, ldx, bias, residual, ldr, out, ldo, N, epi, na, plan, st)
    if (plan.nslabs <= 8) {
        if (epi == PARROT_EPI_SWIGLU) PARROT_W4_GO(true, RU2, 8);
        PARROT_W4_GO(false, RU1, 8);
    }
    if (epi == PARROT_EPI_SWIGLU) PARROT_W4_GO(true, 2, 16);
    PARROT_W4_GO(false, 4, 16);
#undef PARROT_W4_GO
}

}  // namespace parrot

using namespace parrot;

extern "C" {

// tuning hook, not part of the public header: rows of output per workgroup (0 = heuristic)
int parrot_tune_w4_rows_per_wg(int rows) {
    g_rows_per_wg_override = (rows == 4 || rows == 8 || rows == 16) ? rows : 0;
    return PARROT_OK;
}

int64_t parrot_w4_packed_bytes(int N, int K, int group) {
    W4Plan plan;
    const int rc = w4_make_plan(N, K, group, &plan);
    if (rc != PARROT_OK) return rc;
    return (int64_t)N * plan.row16 * 16;
}

int parrot_w4_repack(void* quant_weight_ref, void* scales, void* zeros, int N, int K, int group, void* packed,
                     int direction, void* stream) {
    PARROT_REQUIRE(quant_weight_ref && scales && zeros && packed, "w4_repack: null pointer");
    PARROT_REQUIRE(direction == 0 || direction == 1, "w4_repack: direction must be 0 or 1");
    PARROT_REQUIRE(aligned16(packed), "w4_repack: packed buffer must be 16-byte aligned");
    W4Plan plan;
    const int rc = w4_make_plan(N, K, group, &plan);
    if (rc != PARROT_OK) return rc;
    const int64_t total = (int64_t)N * plan.row16;
    const int64_t blocks = (total + 255) / 256;
    PARROT_UNSUPPORTED(blocks < (1ll << 31), "w4_repack: matrix too large");
    return launch(K_W4_REPACK, w4_repack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                  (uint8_t*)quant_weight_ref, (bf16_t*)scales, (bf16_t*)zeros, (uint4*)packed, N, direction, plan);
}

int parrot_w4_gemv(const void* packed, const void* packed2, const void* x, int ldx, int M, const void* bias,
                   const void* residual, int ldr, void* out, int ldo, int N, int K, int group, int epilogue,
                   const parrot_norm_t* norm, void* stream) {
    int rc = check_linear_args("w4_gemv", packed, packed2, x, ldx, M, residual, ldr, out, ldo, N, K, epilogue);
    if (rc != PARROT_OK) return rc;
    PARROT_UNSUPPORTED(!(epilogue == PARROT_EPI_SWIGLU && bias), "w4_gemv: SWIGLU epilogue takes no bias");
    W4Plan plan;
    rc = w4_make_plan(N, K, group, &plan);
    if (rc != PARROT_OK) return rc;
    NormArgs na;
    rc = make_norm_args(norm, K, &na);
    if (rc != PARROT_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* xb = (const bf16_t*)x;
    const bf16_t* rb = (const bf16_t*)residual;
    bf16_t* ob = (bf16_t*)out;
    for (int m0 = 0; m0 < M; m0 += 4) {
        const int mm = (M - m0 < 4) ? M - m0 : 4;
        const void* xm = xb + (int64_t)m0 * ldx;
        const void* rm = rb ? rb + (int64_t)m0 * ldr : nullptr;
        void* om = ob + (int64_t)m0 * ldo;
        switch (mm) {
            case 1: rc = w4_gemv_launch<1>(packed, packed2, xm, ldx, bias, rm, ldr, om, ldo, N, epilogue, na, plan, st); break;
            case 2: rc = w4_gemv_launch<2>(packed, packed2, xm, ldx, bias, rm, ldr, om, ldo, N, epilogue, na, plan, st); break;
            case 3: rc = w4_gemv_launch<3>(packed, packed2, xm, ldx, bias, rm, ldr, om, ldo, N, epilogue, na, plan, st); break;
            default: rc = w4_gemv_launch<4>(packed, packed2, xm, ldx, bias, rm, ldr, om, ldo, N, epilogue, na, plan, st); break;
        }
        if (rc != PARROT_OK) return rc;
    }
    return PARROT_OK;
}

// Prefill entry point.  Round 1: row blocks of 4 through the GEMV kernel (weights re-read per block);
// the MFMA dequant-to-LDS kernel replaces this body without changing the contract.
int parrot_w4_gemm(const void* packed, const void* packed2, const void* x, int ldx, int M, const void* bias,
                   const void* residual, int ldr, void* out, int ldo, int N, int K, int group, int epilogue,
                   const parrot_norm_t* norm, void* stream) {
    return parrot_w4_gemv(packed, packed2, x, ldx, M, bias, residual, ldr, out, ldo, N, K, group, epilogue, norm, stream);
}

}  // extern "C"
