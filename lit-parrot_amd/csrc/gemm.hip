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
#include <stdlib.h>

#include "parrot_common.h"
#include "w4_plan.h"

namespace parrot {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int GBK = 32;  // K-tile; the M x N tile is a template parameter (128 x 128 for long prompts, 64 x 64 to fill the chip on short ones)
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

// SPLIT = the split-K instantiation (raw fp32 partials out, no epilogue); kept apart from the one-pass kernel so that the
// latter's register allocation is not disturbed (as a run-time switch it cost the bf16 kernel 56 more VGPRs and 30 %).
template <bool W4, int GBM, int GBN, bool SPLIT, bool SWI>
__global__ void __launch_bounds__(256)
gemm_kernel(const bf16_t* __restrict__ A, int lda, int M, const void* __restrict__ Wv, const void* __restrict__ W2v, int N,
            int K, const float* __restrict__ xsum, const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr,
            bf16_t* out, int ldo, int epi, int xs_lds, W4Plan plan, int ksplit, float* __restrict__ part,
            float* __restrict__ part2) {
    // two LDS stages: the global loads of K-tile t+1 are in flight (registers) while tile t is multiplied, one barrier per tile
    __shared__ __attribute__((aligned(16))) bf16_t As[2][GBM * GLD];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[2][GBN * GLD];
    extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];  // int4: xsum of this tile's rows [GBM][ngroups]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    const int lr = lane & 31, lh = lane >> 5;
    const int ktiles = K / GBK;
    const int Gs = W4 ? plan.Gs : ktiles;  // K-tiles per quantisation group
    // split-K (short prompts: too few tiles to fill the chip): blockIdx.z owns K-tiles [kt_begin, kt_end), whole groups,
    // and writes raw fp32 partial results; gemm_splitk_epilogue_kernel sums them in a fixed order and applies the epilogue
    const int kt_per = SPLIT ? ktiles / ksplit : ktiles;
    const int kt_begin = SPLIT ? (int)blockIdx.z * kt_per : 0, kt_end = kt_begin + kt_per;
    const int ngroups = W4 ? plan.ngroups : 1;
    constexpr int IM = GBM / 64, JN = GBN / 64;  // 32x32 MFMA tiles per wave: each wave owns (GBM/2) x (GBN/2)
    constexpr int AIT = GBM / 64;                // 16-byte A pieces per thread per K-tile
    constexpr int BIT = W4 ? (GBN / 64) : (GBN / 64);  // bf16: 16-byte pieces; int4: one dword (8 weights) per piece

    float* xs_l = reinterpret_cast<float*>(gemm_smem);
    if (W4 && xs_lds) {
        for (int i = tid; i < GBM * ngroups; i += 256) {
            const int64_t gm = min(m0 + i / ngroups, M - 1);
            xs_l[i] = xsum[gm * ngroups + i % ngroups];
        }
    }

    // `total` (int4: the result across quantisation groups) and `gate` (SwiGLU) only exist in the instantiations that
    // need them: the bf16 kernel reads its accumulators directly in the epilogue
    f32x16_t total[W4 ? IM : 1][W4 ? JN : 1];
    uint32_t gate[SWI ? IM : 1][SWI ? JN : 1][8];  // SwiGLU: bf16(silu(bf16(fc_1))) of the first pass, packed
    const int npass = SWI ? 2 : 1;

    f32x16_t acc[IM][JN];
    for (int pass = 0; pass < npass; ++pass) {
        const void* Wp = pass ? W2v : Wv;
#pragma unroll
        for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int jn = 0; jn < JN; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[i][jn][r] = 0.f;
                    if (W4) total[W4 ? i : 0][W4 ? jn : 0][r] = 0.f;
                }
        uint4 ra[AIT];
        uint4 rb[BIT];    // bf16 weights
        uint32_t rq[BIT];  // int4 weights: 8 nibbles each
        int slab = 0;
        // global -> registers for K-tile kt
        auto fetch = [&](int kt) {
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                const int idx = tid + it * 256;
                const int64_t gm = min(m0 + (idx >> 2), M - 1);
                ra[it] = reinterpret_cast<const uint4*>(A + gm * lda + (int64_t)kt * GBK)[idx & 3];
            }
            if (W4) {
                while (slab + 1 < plan.nslabs && kt >= plan.slab[slab + 1].slice0) ++slab;
                const int soff = plan.slab[slab].w_off16 + (kt - plan.slab[slab].slice0);
#pragma unroll
                for (int it = 0; it < BIT; ++it) {
                    const int idx = tid + it * 256;  // (row, dword of the slice)
                    const int64_t gn = min(n0 + (idx >> 2), N - 1);
                    rq[it] = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint4*>(Wp) + gn * plan.row16 + soff)[idx & 3];
                }
            } else {
#pragma unroll
                for (int it = 0; it < BIT; ++it) {
                    const int idx = tid + it * 256;
                    const int64_t gn = min(n0 + (idx >> 2), N - 1);
                    rb[it] = reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(Wp) + gn * K + (int64_t)kt * GBK)[idx & 3];
                }
            }
        };
        // registers -> LDS stage st (int4: expand 8 nibbles to the bf16 values 128 + q)
        auto stage = [&](int st) {
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                const int idx = tid + it * 256;
                *reinterpret_cast<uint4*>(&As[st][(idx >> 2) * GLD + (idx & 3) * 8]) = ra[it];
            }
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int idx = tid + it * 256;
                uint4 v;
                if (W4) {
                    uint32_t o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = ((rq[it] >> (4 * i)) & 0x000F000Fu) | 0x43004300u;
                    v = make_uint4(o[0], o[1], o[2], o[3]);
                } else {
                    v = rb[it];
                }
                *reinterpret_cast<uint4*>(&Bs[st][(idx >> 2) * GLD + (idx & 3) * 8]) = v;
            }
        };
        fetch(kt_begin);
        __syncthreads();  // previous pass finished with the LDS stages; xs_l is filled
        stage(0);
        __syncthreads();
        uint32_t mtg[JN];  // int4: {scale, zero} of the current quantisation group for this lane's output columns
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int st = (kt - kt_begin) & 1;
            const int slab_now = slab;  // slab of tile kt (fetch() below may advance it)
            if (W4 && kt % Gs == 0) {
                // requested at the START of the group and used at its end (Gs tiles later): the first version loaded it at
                // the fold, a full memory latency on the critical path of every group
                const int g = kt / Gs;
                const int gs = slab_now;  // the slab of the group's first slice stores the group's metadata
#pragma unroll
                for (int jn = 0; jn < JN; ++jn) {
                    const int64_t gn = min(n0 + wn * (GBN / 2) + jn * 32 + lr, N - 1);
                    const uint4* rec = reinterpret_cast<const uint4*>(Wp) + gn * plan.row16;
                    mtg[jn] = reinterpret_cast<const uint32_t*>(rec + plan.slab[gs].meta_off16)[g - plan.slab[gs].g0];
                }
            }
            if (kt + 1 < kt_end) fetch(kt + 1);
            // ---- 2 k-steps of MFMA 32x32x16 on stage st
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8_t af[IM], bfr[JN];
#pragma unroll
                for (int i = 0; i < IM; ++i) {
                    const uint4 va = *reinterpret_cast<const uint4*>(&As[st][(wm * (GBM / 2) + i * 32 + lr) * GLD + s * 16 + lh * 8]);
                    af[i] = __builtin_bit_cast(bf16x8_t, va);
                }
#pragma unroll
                for (int jn = 0; jn < JN; ++jn) {
                    const uint4 vb = *reinterpret_cast<const uint4*>(&Bs[st][(wn * (GBN / 2) + jn * 32 + lr) * GLD + s * 16 + lh * 8]);
                    bfr[jn] = __builtin_bit_cast(bf16x8_t, vb);
                }
#pragma unroll
                for (int i = 0; i < IM; ++i)
#pragma unroll
                    for (int jn = 0; jn < JN; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[jn], acc[i][jn], 0, 0, 0);
            }
            if (W4 && ((kt + 1) % Gs == 0 || kt + 1 == kt_end)) {
                // ---- quantisation-group boundary: fold the group's partial product into the result
                const int g = kt / Gs;
#pragma unroll
                for (int jn = 0; jn < JN; ++jn) {
                    const uint32_t mt = mtg[jn];
                    const float sc = bflo(mt), zz = 128.0f + bfhi(mt);
#pragma unroll
                    for (int i = 0; i < IM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int lm = wm * (GBM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            const float xs = xs_lds ? xs_l[lm * ngroups + g] : xsum[(int64_t)min(m0 + lm, M - 1) * ngroups + g];
                            total[W4 ? i : 0][W4 ? jn : 0][r] += sc * (acc[i][jn][r] - zz * xs);
                            acc[i][jn][r] = 0.f;
                        }
                }
            }
            if (kt + 1 < kt_end) stage(st ^ 1);  // stage st^1 was last read in iteration kt-1, before the barrier below it
            __syncthreads();
        }
#define GEMM_RESULT(I, JNN, R) (W4 ? total[W4 ? (I) : 0][W4 ? (JNN) : 0][R] : acc[I][JNN][R])
        if constexpr (SPLIT) {  // raw partial result of this K range (C layout of the 32x32 MFMA, see the epilogue below)
            float* dst = (pass ? part2 : part) + (int64_t)blockIdx.z * M * N;
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int jn = 0; jn < JN; ++jn) {
                    const int col = n0 + wn * (GBN / 2) + jn * 32 + lr;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * (GBM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (row < M && col < N) dst[(int64_t)row * N + col] = GEMM_RESULT(i, jn, r);
                    }
                }
            continue;
        }
        if (SWI && pass == 0) {
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int jn = 0; jn < JN; ++jn)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const bf16_t g0 = f2bf(silu(rbf(GEMM_RESULT(i, jn, r)))), g1 = f2bf(silu(rbf(GEMM_RESULT(i, jn, r + 1))));
                        gate[SWI ? i : 0][SWI ? jn : 0][r >> 1] = (uint32_t)g0 | ((uint32_t)g1 << 16);
                    }
        }
    }
    if constexpr (SPLIT) return;
    // ---- epilogue: C layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int jn = 0; jn < JN; ++jn) {
            const int col = n0 + wn * (GBN / 2) + jn * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (GBM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M && col < N) {
                    bf16_t o;
                    if (SWI) {
                        const uint32_t gp = gate[SWI ? i : 0][SWI ? jn : 0][r >> 1];
                        const float gv = (r & 1) ? bfhi(gp) : bflo(gp);
                        o = f2bf(gv * rbf(GEMM_RESULT(i, jn, r)));
                    } else {
                        o = apply_epilogue(epi, GEMM_RESULT(i, jn, r), 0.f, bias, residual ? residual + (int64_t)row * ldr : nullptr, col);
                    }
                    out[(int64_t)row * ldo + col] = o;
                }
            }
        }
}

#undef GEMM_RESULT

// split-K second stage: sum the partials in a fixed order and apply the epilogue (same rounding points as the one-pass kernel)
__global__ void __launch_bounds__(256)
gemm_splitk_epilogue_kernel(const float* __restrict__ part, const float* __restrict__ part2, int ksplit, int M, int N,
                            const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr, bf16_t* out, int ldo, int epi) {
    // one row per ceil(N / 256) workgroups (a flat element index cost a 64-bit division per element)
    const unsigned bpr = (unsigned)(N + 255) / 256u;
    const int row = (int)(blockIdx.x / bpr), col = (int)(blockIdx.x % bpr) * 256 + (int)threadIdx.x;
    if (row >= M || col >= N) return;
    float a = 0.f, b = 0.f;
    for (int z = 0; z < ksplit; ++z) {
        a += part[((int64_t)z * M + row) * N + col];
        if (epi == PARROT_EPI_SWIGLU) b += part2[((int64_t)z * M + row) * N + col];
    }
    bf16_t o;
    if (epi == PARROT_EPI_SWIGLU)
        o = f2bf(bf2f(f2bf(silu(rbf(a)))) * rbf(b));
    else
        o = apply_epilogue(epi, a, 0.f, bias, residual ? residual + (int64_t)row * ldr : nullptr, col);
    out[(int64_t)row * ldo + col] = o;
}

// the same with four columns per thread (16-byte loads of the partials, one 8-byte store): the launch is bound by its slab traffic
// (ks x M x N x 4 bytes - 31 MB for the QKV Linear of Llama-2-7B at 128 rows), which one-float-per-lane loads move at half the rate
__global__ void __launch_bounds__(256)
gemm_splitk_epilogue4_kernel(const float* __restrict__ part, const float* __restrict__ part2, int ksplit, int M, int N,
                             const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr, bf16_t* out, int ldo, int epi) {
    const unsigned bpr = (unsigned)(N + 1023) / 1024u;
    const int row = (int)(blockIdx.x / bpr), col = ((int)(blockIdx.x % bpr) * 256 + (int)threadIdx.x) * 4;
    if (row >= M || col >= N) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    for (int z = 0; z < ksplit; ++z) {  // (the same fixed order of the sum as the one-column kernel)
        const float4 p = *reinterpret_cast<const float4*>(part + ((int64_t)z * M + row) * N + col);
        a.x += p.x, a.y += p.y, a.z += p.z, a.w += p.w;
        if (epi == PARROT_EPI_SWIGLU) {
            const float4 q = *reinterpret_cast<const float4*>(part2 + ((int64_t)z * M + row) * N + col);
            b.x += q.x, b.y += q.y, b.z += q.z, b.w += q.w;
        }
    }
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
    bf16_t o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (epi == PARROT_EPI_SWIGLU)
            o[c] = f2bf(bf2f(f2bf(silu(rbf(av[c])))) * rbf(bv[c]));
        else
            o[c] = apply_epilogue(epi, av[c], 0.f, bias, residual ? residual + (int64_t)row * ldr : nullptr, col + c);
    }
    *reinterpret_cast<uint2*>(out + (int64_t)row * ldo + col) = make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
}

int launch_splitk_epilogue(const float* part, const float* part2, int ksplit, int M, int N, const bf16_t* bias, const bf16_t* residual,
                           int ldr, bf16_t* out, int ldo, int epi, hipStream_t st) {
    const bool vec = N % 4 == 0 && ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0 && aligned16(part) && (!part2 || aligned16(part2));
    if (vec)
        return launch(K_GEMM_SPLITK, gemm_splitk_epilogue4_kernel, dim3((unsigned)((int64_t)M * ((N + 1023) / 1024))), dim3(256), 0, st, part, part2,
                      ksplit, M, N, bias, residual, ldr, out, ldo, epi);
    return launch(K_GEMM_SPLITK, gemm_splitk_epilogue_kernel, dim3((unsigned)((int64_t)M * ((N + 255) / 256))), dim3(256), 0, st, part, part2, ksplit,
                  M, N, bias, residual, ldr, out, ldo, epi);
}

// 128 x 128 tiles only when they alone give the chip >= 2 workgroups per CU; otherwise 64 x 64 (4x the workgroups)
static bool gemm_big_tiles(int M, int N) {
    const int env = tune_env("PARROT_GEMM_BIG_MIN", 512);  // PARROT_GEMM_BIG_MIN: minimum number of 128 x 128 tiles for the big-tile kernel (experiment hook)
    return (int64_t)((M + 127) / 128) * ((N + 127) / 128) >= env;
}

// K splits for launches with too few tiles to fill the chip (short prompts): equal ranges of whole quantisation groups,
// at least 8 K-tiles each, at most 8 splits, aiming at >= ~1536 workgroups
static int gemm_ksplit(int M, int N, int K, int gs_tiles) {
    const int env = tune_env("PARROT_GEMM_KSPLIT", -2);  // PARROT_GEMM_KSPLIT: 0 = never split (A/B), n = force
    if (env == 0) return 1;
    const int64_t tiles = gemm_big_tiles(M, N) ? (int64_t)((M + 127) / 128) * ((N + 127) / 128) : (int64_t)((M + 63) / 64) * ((N + 63) / 64);
    const int ktiles = K / GBK;
    int ks = env > 0 ? env : (int)(1536 / (tiles > 0 ? tiles : 1));
    if (ks > 8) ks = 8;
    while (ks > 1 && (ktiles % (ks * gs_tiles) != 0 || ktiles / ks < 8)) --ks;
    return ks < 1 ? 1 : ks;
}

// launch the tiles (x ksplit) and, when K was split, the second stage
template <bool W4>
static int gemm_launch(int kid, const void* Wp, const void* W2p, const void* x, int ldx, int M, const void* bias, const void* residual,
                       int ldr, void* out, int ldo, int N, int K, int epilogue, const float* xsum, float* part_ws, const W4Plan& plan,
                       int gs_tiles, hipStream_t st) {
    const int ksplit = gemm_ksplit(M, N, K, gs_tiles);
    float* part = ksplit > 1 ? part_ws : nullptr;
    float* part2 = (ksplit > 1 && epilogue == PARROT_EPI_SWIGLU) ? part_ws + (int64_t)ksplit * M * N : nullptr;
    PARROT_REQUIRE(ksplit == 1 || part_ws != nullptr, "gemm: this shape splits K %d ways and needs the workspace of parrot_gemm_workspace_floats", ksplit);
    int rc;
#define PARROT_GEMM_GO2(BMV, BNV, SPLITV, SWIV)                                                                                \
    rc = launch(kid, gemm_kernel<W4, BMV, BNV, SPLITV, SWIV>, grid, dim3(256), in_lds ? xb : 0, st, (const bf16_t*)x, ldx, M, Wp, W2p, N, \
                K, xsum, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, epilogue, in_lds, plan, ksplit, part, part2)
#define PARROT_GEMM_GO(BMV, BNV, SPLITV)                       \
    if (epilogue == PARROT_EPI_SWIGLU)                         \
        PARROT_GEMM_GO2(BMV, BNV, SPLITV, true);               \
    else                                                       \
        PARROT_GEMM_GO2(BMV, BNV, SPLITV, false)
    if (gemm_big_tiles(M, N)) {
        const size_t xb = W4 ? (size_t)128 * plan.ngroups * 4 : 0;
        const int in_lds = W4 && xb <= 20 * 1024;
        const dim3 grid((N + 127) / 128, (M + 127) / 128, ksplit);
        if (ksplit > 1)
            PARROT_GEMM_GO(128, 128, true);
        else
            PARROT_GEMM_GO(128, 128, false);
    } else {
        const size_t xb = W4 ? (size_t)64 * plan.ngroups * 4 : 0;
        const int in_lds = W4 && xb <= 40 * 1024;
        const dim3 grid((N + 63) / 64, (M + 63) / 64, ksplit);
        if (ksplit > 1)
            PARROT_GEMM_GO(64, 64, true);
        else
            PARROT_GEMM_GO(64, 64, false);
    }
#undef PARROT_GEMM_GO
#undef PARROT_GEMM_GO2
    if (rc != PARROT_OK || ksplit == 1) return rc;
    const int64_t n = (int64_t)M * N;
    return launch_splitk_epilogue((const float*)part, (const float*)part2, ksplit, M, N, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, epilogue, st);
}

// norm + per-group sums of a prompt's rows in one launch (norm.hip)
bool norm_xsum_takes(const NormArgs& na, int d, int G);
int norm_xsum_launch(const NormArgs& na, const void* x, int ldx, void* xn, int ldo, int M, int Mpad, int d, int G, float* xs, hipStream_t st);

// second-generation bf16 kernel (gemm2.hip)
int gemm2_ksplit(int M, int N, int K);
bool gemm2_enabled();
int gemm2_launch(const void* W, const void* x, int ldx, int M, const void* bias, const void* residual, int ldr, void* out, int ldo,
                 int N, int K, int epilogue, float* part, hipStream_t st, int* ksplit_out);
bool gemm2_w4_takes(const W4Plan& plan, int K);
int gemm2_w4_ksplit(int M, int N, int K, const W4Plan& plan);
int64_t gemm2_w4_xs_floats(int M, const W4Plan& plan);
int gemm2_w4_launch(const void* Wq, const void* Wq2, const void* x, int ldx, int M, const void* bias, const void* residual, int ldr,
                    void* out, int ldo, int N, int K, int epilogue, float* workspace, const W4Plan& plan, hipStream_t st, int* ksplit_out,
                    float** part_out, float** part2_out, const void* code, bool have_xs);
static bool gemm2_takes(int K, int ldx, int epilogue, const void* W, const void* x) {
    return gemm2_enabled() && epilogue != PARROT_EPI_SWIGLU && K % 64 == 0 && ldx % 8 == 0 && aligned16(W) && aligned16(x);
}

}  // namespace parrot

using namespace parrot;

extern "C" {

// floats of workspace a GEMM call needs: the per-group activation sums (int4: group != 0) + the split-K partial results
int64_t parrot_gemm_workspace_floats(int M, int N, int K, int group, int epilogue) {
    int64_t n = 0;
    int gs_tiles = 1;  // bf16: any K-tile boundary will do
    if (group != 0) {
        if (group < 0 || group > K) group = K;
        n += (int64_t)M * ((K + group - 1) / group);
        gs_tiles = group / GBK > 0 ? group / GBK : 1;
    }
    if (K % GBK == 0) {
        int ks = gemm_ksplit(M, N, K, gs_tiles);
        if (group == 0 && K % 64 == 0 && epilogue != PARROT_EPI_SWIGLU && gemm2_enabled()) {  // whichever bf16 kernel ends up running
            const int ks2 = gemm2_ksplit(M, N, K);
            if (ks2 > ks) ks = ks2;
        }
        if (ks > 1) n += (int64_t)ks * M * N * (epilogue == PARROT_EPI_SWIGLU ? 2 : 1);
    }
    if (group != 0 && K % 32 == 0) {  // the second-generation int4 kernel lays the workspace out differently: cover both
        W4Plan plan;
        if (w4_make_plan(N, K, group, &plan) == PARROT_OK && gemm2_w4_takes(plan, K)) {
            const int ks2 = gemm2_w4_ksplit(M, N, K, plan);
            int64_t n2 = gemm2_w4_xs_floats(M, plan) + (ks2 > 1 ? (int64_t)ks2 * M * N * (epilogue == PARROT_EPI_SWIGLU ? 2 : 1) : 0);
            if (plan.Gs == 2 || plan.Gs == 4) n2 += ((int64_t)M * K + 1) / 2;  // room for the normalised rows of a fused norm (parrot_w4_gemm)
            if (n2 > n) n = n2;
        }
    }
    return n;
}

int parrot_bf16_gemm(const void* W, const void* W2, const void* x, int ldx, int M, const void* bias, const void* residual,
                     int ldr, void* out, int ldo, int N, int K, int epilogue, const parrot_norm_t* norm, void* workspace,
                     void* stream) {
    if (M <= 8) return parrot_bf16_gemv(W, W2, x, ldx, M, bias, residual, ldr, out, ldo, N, K, epilogue, norm, stream);
    int rc = check_linear_args("bf16_gemm", W, W2, x, ldx, M, residual, ldr, out, ldo, N, K, epilogue);
    if (rc != PARROT_OK) return rc;
    PARROT_UNSUPPORTED(norm == nullptr || norm->kind == 0, "bf16_gemm: apply the norm to the rows first (parrot_rmsnorm / parrot_layernorm)");
    PARROT_UNSUPPORTED(K % GBK == 0, "bf16_gemm: K=%d must be a multiple of %d", K, GBK);
    PARROT_UNSUPPORTED(!(epilogue == PARROT_EPI_SWIGLU && bias), "bf16_gemm: SWIGLU epilogue takes no bias");
    PARROT_REQUIRE(M <= 65535 * 64, "bf16_gemm: M too large");
    if (gemm2_takes(K, ldx, epilogue, W, x)) {
        int ks = 1;
        rc = gemm2_launch(W, x, ldx, M, bias, residual, ldr, out, ldo, N, K, epilogue, (float*)workspace, (hipStream_t)stream, &ks);
        if (rc != PARROT_OK || ks == 1) return rc;
        const int64_t n = (int64_t)M * N;
        return launch_splitk_epilogue((const float*)workspace, (const float*)nullptr, ks, M, N, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, epilogue, (hipStream_t)stream);
    }
    W4Plan plan = {};
    return gemm_launch<false>(K_BF16_GEMM, W, W2, x, ldx, M, bias, residual, ldr, out, ldo, N, K, epilogue, nullptr, (float*)workspace,
                              plan, 1, (hipStream_t)stream);
}

int parrot_w4_gemm(const void* packed, const void* packed2, const void* x, int ldx, int M, const void* bias,
                   const void* residual, int ldr, void* out, int ldo, int N, int K, int group, int epilogue,
                   const parrot_norm_t* norm, void* workspace, void* stream) {
    if (M <= 8)
        return parrot_w4_gemv(packed, packed2, x, ldx, M, bias, residual, ldr, out, ldo, N, K, group, epilogue, norm, stream);
    int rc = check_linear_args("w4_gemm", packed, packed2, x, ldx, M, residual, ldr, out, ldo, N, K, epilogue);
    if (rc != PARROT_OK) return rc;
    PARROT_REQUIRE(workspace != nullptr, "w4_gemm: workspace of parrot_gemm_workspace_floats(M, N, K, group, epilogue) floats required");
    PARROT_UNSUPPORTED(!(epilogue == PARROT_EPI_SWIGLU && bias), "w4_gemm: SWIGLU epilogue takes no bias");
    W4Plan plan;
    rc = w4_make_plan(N, K, group, &plan);
    if (rc != PARROT_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    NormArgs na;
    rc = make_norm_args(norm, K, &na);
    if (rc != PARROT_OK) return rc;
    const bool fused_norm = na.kind != 0;
    if (fused_norm) {
        // the norm of the prompt's rows AND the per-group sums of its output in ONE launch in front of the GEMM (instead of
        // parrot_rmsnorm + the activation-sum pre-pass): the normalised rows go to the head of the workspace
        PARROT_UNSUPPORTED(gemm2_w4_takes(plan, K) && ldx % 8 == 0 && norm_xsum_takes(na, K, plan.Gs * 32),
                           "w4_gemm: the fused norm takes groups of 64 / 128 on the LDS-DMA kernel; apply the norm to the rows first "
                           "(parrot_rmsnorm / parrot_layernorm) for K=%d group=%d", K, group);
        const int Mpad = (M + 127) / 128 * 128;
        bf16_t* xn = reinterpret_cast<bf16_t*>(workspace);
        float* ws2 = reinterpret_cast<float*>(workspace) + ((int64_t)M * K + 1) / 2;
        rc = norm_xsum_launch(na, x, ldx, xn, K, M, Mpad, K, plan.Gs * 32, ws2, st);
        if (rc != PARROT_OK) return rc;
        x = xn, ldx = K, workspace = ws2;
    }
    if (gemm2_w4_takes(plan, K) && ldx % 8 == 0) {
        int ks = 1;
        float *part = nullptr, *part2 = nullptr;
        rc = gemm2_w4_launch(packed, packed2, x, ldx, M, bias, residual, ldr, out, ldo, N, K, epilogue, (float*)workspace, plan, st, &ks,
                             &part, &part2, nullptr, fused_norm);
        if (rc != PARROT_OK || ks == 1) return rc;
        const int64_t mn = (int64_t)M * N;
        return launch_splitk_epilogue((const float*)part, (const float*)part2, ks, M, N, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, epilogue, st);
    }
    const int G = plan.Gs * 32;
    const int64_t n = (int64_t)M * plan.ngroups;
    rc = launch(K_GEMM_XSUM, gemm_xsum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const bf16_t*)x, ldx, M, K, G,
                plan.ngroups, (float*)workspace);
    if (rc != PARROT_OK) return rc;
    // the split-K partials follow the activation sums in the workspace
    return gemm_launch<true>(K_W4_GEMM, packed, packed2, x, ldx, M, bias, residual, ldr, out, ldo, N, K, epilogue, (const float*)workspace,
                             (float*)workspace + n, plan, plan.Gs, st);
}

// NF4 / FP4 (bitsandbytes 4-bit) prompt rows on the matrix cores: the codebook variant of the int4 LDS-DMA kernel.  Needs
// K % 64 == 0 and blocks of 64; `workspace` = parrot_gemm_workspace_floats(M, N, K, block, epilogue) floats (split-K partials).
int parrot_w4c_gemm(const void* packed, const void* packed2, const void* code16_bf16, const void* x, int ldx, int M, const void* bias,
                    const void* residual, int ldr, void* out, int ldo, int N, int K, int block, int epilogue,
                    const parrot_norm_t* norm, void* workspace, void* stream) {
    if (M <= 8)
        return parrot_w4c_gemv(packed, packed2, code16_bf16, x, ldx, M, bias, residual, ldr, out, ldo, N, K, block, epilogue, norm, stream);
    int rc = check_linear_args("w4c_gemm", packed, packed2, x, ldx, M, residual, ldr, out, ldo, N, K, epilogue);
    if (rc != PARROT_OK) return rc;
    PARROT_REQUIRE(code16_bf16 != nullptr, "w4c_gemm: codebook pointer is null");
    PARROT_UNSUPPORTED(norm == nullptr || norm->kind == 0, "w4c_gemm: apply the norm to the rows first (parrot_rmsnorm / parrot_layernorm)");
    PARROT_REQUIRE(workspace != nullptr, "w4c_gemm: workspace of parrot_gemm_workspace_floats(M, N, K, block, epilogue) floats required");
    PARROT_UNSUPPORTED(!(epilogue == PARROT_EPI_SWIGLU && bias), "w4c_gemm: SWIGLU epilogue takes no bias");
    W4Plan plan;
    rc = w4_make_plan(N, K, block, &plan);
    if (rc != PARROT_OK) return rc;
    PARROT_UNSUPPORTED(gemm2_w4_takes(plan, K) && ldx % 8 == 0, "w4c_gemm: needs K %% 64 == 0, blocks that are multiples of 64 and 16-byte rows (K=%d block=%d)", K, block);
    hipStream_t st = (hipStream_t)stream;
    int ks = 1;
    float *part = nullptr, *part2 = nullptr;
    rc = gemm2_w4_launch(packed, packed2, x, ldx, M, bias, residual, ldr, out, ldo, N, K, epilogue, (float*)workspace, plan, st, &ks, &part,
                         &part2, code16_bf16, false);
    if (rc != PARROT_OK || ks == 1) return rc;
    const int64_t mn = (int64_t)M * N;
    return launch_splitk_epilogue((const float*)part, (const float*)part2, ks, M, N, (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, epilogue, st);
}

}  // extern "C"
