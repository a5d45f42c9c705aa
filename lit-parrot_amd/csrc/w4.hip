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
#include <stdlib.h>

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

// diagnostic stamps (tools/microbench.py --stamps): 100 MHz clock of a few workgroups at phase boundaries.  The buffer
// pointer is a KERNEL ARGUMENT (scalar load): an earlier version read it from a __device__ global, and the
// s_waitcnt vmcnt(0) behind that vector load also waited for every weight load in flight - a diagnostic switch in
// front of the dot loop serialised "all weights arrived" -> "first dot product" in every launch (+2.5 us each).
__device__ __forceinline__ void w4_stamp(unsigned long long* dbg, int i) {
    if (dbg != nullptr && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1)) {
        const int b = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 2 : 1);
        dbg[b * 8 + i] = __builtin_amdgcn_s_memrealtime();
    }
}
#ifdef PARROT_DIAG
static unsigned long long* g_w4_dbg_host = nullptr;  // diagnostic build only: set by parrot_tune_w4_stamps
#else
static constexpr unsigned long long* g_w4_dbg_host = nullptr;
#endif

// Workgroup shape: nslabs x wps waves.  Wave (slab c, j) streams slab c of row group j: RU consecutive rows.  The
// activations (and the optional norm of them) are prepared ONCE per workgroup and shared by its row groups and by the
// `iters` batches of rows it walks (batch t of workgroup b = rows of workgroup-sized block b + t * gridDim.x): the host
// sizes the grid to what is resident at once, so that no workgroup starts late and pays the activation / norm chain
// behind everybody else's weight stream (measured on lm_head: a 4th-round workgroup entered at +11 us and had its norm
// ready 6.5 us later).  MAXW = waves the build allows (8 -> 256 VGPRs, 16 -> 128).
//
// CB = true: the nibbles index a 16-entry codebook (bitsandbytes NF4 / FP4, quantize/bnb.py:62-75) and the group metadata
// word is the block's fp32 absmax: y = sum over blocks of absmax * sum_k x[k] * code[q[k]].  The codebook (bf16) is
// spread over LDS as one private column per lane (entry e of lane l at byte e * 256 + l * 4: every lane stays in its own
// bank, no conflicts whatever the indices are); a weight pair is fetched with two 16-bit LDS reads joined by one v_lshl_or,
// and is the operand of the same v_dot2 as the affine path.
template <int M, bool DUAL, int RU, int MAXW, bool CB>
__global__ void __launch_bounds__(MAXW * 64)
w4_gemv_kernel(const uint4* __restrict__ W, const uint4* __restrict__ W2, const bf16_t* __restrict__ x, int ldx,
               const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr, bf16_t* out, int ldo, int N, int K,
               int wps, int epi, int iters, NormArgs na, W4Plan plan, unsigned long long* dbg, const uint32_t* __restrict__ code) {
    constexpr int NW = DUAL ? 2 : 1;
    __shared__ __attribute__((aligned(4096))) uint32_t cbt[CB ? 16 * 64 : 1];
    extern __shared__ __attribute__((aligned(16))) unsigned char w4_smem[];  // normalised activations [M][K] bf16 (norm only)
    __shared__ float red[2][MAXW][RU * M * NW];  // double-buffered over the batches: one barrier per batch
    __shared__ float stat[16];

    w4_stamp(dbg, 0);
    // The prologue (activations, fused norm) runs at raised wave priority: the SIMDs arbitrate oldest-first, so without it
    // the 3rd / 4th workgroup of a CU gets its norm instructions in only when the older workgroups' dot loops stall
    // (measured: norm ready at +3.4 us in the first workgroup of a CU, +7.5 .. +10 us in the last), and the launch ends
    // when the slowest workgroup does.  Priority drops back before the dot loop.
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = plan.nslabs * wps;
    const int slab = wave / wps, j = wave % wps;
    const W4Slab sl = plan.slab[slab];
    const bool active = lane < sl.nslices;
    const int lslice = active ? lane : sl.nslices - 1;
    const int gslice = sl.slice0 + lslice;
    const int gl = gslice / plan.Gs - sl.g0;
    const int64_t row16 = plan.row16;
    const int R = wps * RU;  // rows per workgroup and batch

    // Load order matters because vmcnt retires in order: first the small L2-resident operands (activations, norm
    // parameters), then the weights.  The prologue then only waits for the former while the HBM stream is running.
    uint32_t xr[M][16];
    const int nthreads = nwaves * 64;
    const int chunks = K >> 3;  // 16-byte units of a row of x
    constexpr int kMaxChunkIt = 4;  // K <= 4 * 8 * nthreads is checked on the host for the norm path
    uint4 cx[M][kMaxChunkIt], cw[kMaxChunkIt], cb[kMaxChunkIt];
    if (na.kind == 0) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const uint4* xp = reinterpret_cast<const uint4*>(x + (int64_t)m * ldx + (int64_t)gslice * 32);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint4 v = xp[q];
                if (!active) v = make_uint4(0, 0, 0, 0);
                xr[m][4 * q + 0] = v.x;
                xr[m][4 * q + 1] = v.y;
                xr[m][4 * q + 2] = v.z;
                xr[m][4 * q + 3] = v.w;
            }
        }
    } else {
        // cooperative: thread t owns chunks t, t + nthreads, ... of every row (and of the norm parameters);
        // rounds past the row (a workgroup-uniform condition) are skipped altogether, not computed on zeros
#pragma unroll
        for (int it = 0; it < kMaxChunkIt; ++it) {
            cw[it] = cb[it] = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int m = 0; m < M; ++m) cx[m][it] = make_uint4(0, 0, 0, 0);
            if (it * nthreads < chunks) {
                const int c = threadIdx.x + it * nthreads;
                const int cc = c < chunks ? c : chunks - 1;
                cw[it] = reinterpret_cast<const uint4*>(na.weight)[cc];
                if (na.kind == 2 && na.bias != nullptr) cb[it] = reinterpret_cast<const uint4*>(na.bias)[cc];
#pragma unroll
                for (int m = 0; m < M; ++m) cx[m][it] = reinterpret_cast<const uint4*>(x + (int64_t)m * ldx)[cc];
                // (a thread past the row inside a live round holds a clamped copy: masked where it is used, so that
                //  the loads of all rounds are in flight together instead of load -> wait -> select per round)
            }
        }
    }

    if constexpr (CB) {  // entry e is wave-uniform: scalar loads (their own counter, nothing waits for vector memory here)
        for (int e = wave; e < 16; e += nwaves) cbt[e * 64 + lane] = code[e];
    }
    uint4 w[NW][RU];
    uint32_t mt[NW][RU];
    // Rolling window of weight loads.  A CU accepts only so many vector-memory instructions in flight; a wave that issues
    // all its rows at once sits in the ISSUE of those loads until earlier ones (its own and the other workgroups' of the
    // CU) have returned at HBM speed, and everything behind them in program order - the fused norm above all - waits
    // too.  About 3 KB per wave in flight already covers the HBM latency-bandwidth product, so PRIME rows are requested
    // up front and row u + PRIME (of this batch, or of the next one) is requested right before row u is consumed.
    // (Measured alternatives: PRIME 5 - no gain; requesting the whole batch at once for small launches behind a run-time
    //  flag - 6 % SLOWER overall, because loads under a run-time condition defeat the compiler's static vmcnt
    //  bookkeeping and every wait becomes vmcnt(0).  Keep every load of this kernel unconditional.)
#ifndef W4_PRIME_SMALL
#define W4_PRIME_SMALL 2
#endif
    constexpr int PRIME = (RU * NW >= 8) ? 3 : (RU == 4 && M == 1 && !DUAL ? W4_PRIME_SMALL : 2);
    static_assert(PRIME <= RU, "rolling window longer than a batch");
    // first row of this wave in batch T; batches past the matrix are clamped to the last row (loaded, never stored)
#define W4_ROW0(T) (((int)blockIdx.x + (T) * (int)gridDim.x) * R + j * RU)
#define W4_ISSUE_ROW(T, U)                                                                            \
    {                                                                                                 \
        const int64_t row_ = min(W4_ROW0(T) + (U), N - 1);                                            \
        const uint4* rec_ = W + row_ * row16;                                                         \
        /* weights are read exactly once per token: non-temporal loads keep them out of the way of the activations in L2 */ \
        w[0][U] = load_nt16(rec_ + sl.w_off16 + lslice);                                              \
        mt[0][U] = load_nt4(reinterpret_cast<const uint32_t*>(rec_ + sl.meta_off16) + gl);            \
        if (DUAL) {                                                                                   \
            const uint4* rec2_ = W2 + row_ * row16;                                                   \
            w[1][U] = load_nt16(rec2_ + sl.w_off16 + lslice);                                         \
            mt[1][U] = load_nt4(reinterpret_cast<const uint32_t*>(rec2_ + sl.meta_off16) + gl);       \
        }                                                                                             \
    }
#pragma unroll
    for (int u = 0; u < PRIME; ++u) W4_ISSUE_ROW(0, u)
    asm volatile("" ::: "memory");  // keep the remaining requests below the prologue

    if (na.kind != 0) {  // fused RMSNorm / LayerNorm of the input rows, once per workgroup, through LDS
        uint4* xn = reinterpret_cast<uint4*>(w4_smem);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float s1 = 0.f;
#pragma unroll
            for (int it = 0; it < kMaxChunkIt; ++it) {
                if (it * nthreads < chunks) {  // uniform
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
        for (int m = 0; m < M; ++m) {
            const uint4* xl = xn + (int64_t)m * chunks + (int64_t)gslice * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint4 v = xl[q];
                if (!active) v = make_uint4(0, 0, 0, 0);
                xr[m][4 * q + 0] = v.x;
                xr[m][4 * q + 1] = v.y;
                xr[m][4 * q + 2] = v.z;
                xr[m][4 * q + 3] = v.w;
            }
        }
    }
    float xs[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        float s = 0.f;
        if constexpr (!CB) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s += bflo(xr[m][i]) + bfhi(xr[m][i]);
        }
        xs[m] = s;
    }
    if constexpr (CB) __syncthreads();  // codebook columns written (the norm path's barrier lies before on its own branch only)
    // LDS byte address of this lane's codebook column; the table must start on a 4-KB boundary so that the lookups can
    // write "nibble" into byte 1 of the address: the table must lie at LDS offset 0 (it is the most-aligned LDS variable of the kernel)
    const uint32_t cb_col = (uint32_t)(uintptr_t)cbt + lane * 4;
    if (CB && ((uint32_t)(uintptr_t)cbt & 0xFFFFu) != 0) __builtin_trap();  // folded away: the most-aligned LDS variable is placed at 0
    w4_stamp(dbg, 1);
    __builtin_amdgcn_s_setprio(0);

    // bias / residual elements of the epilogue threads: requested at the START of a batch (every thread, clamped, always:
    // a load behind a condition would cost the compiler its static vmcnt bookkeeping), so that the kernel's tail does not
    // end with "load residual -> wait a full memory latency -> add -> store"
    const bf16_t* res_p = residual != nullptr ? residual : reinterpret_cast<const bf16_t*>(W);
    const bf16_t* bias_p = bias != nullptr ? bias : reinterpret_cast<const bf16_t*>(W);
    const int e_m = threadIdx.x % M, e_ur = threadIdx.x / M;
    constexpr bool kSum8 = (M == 1 && RU == 8);  // eight row sums in one packed reduction (wave_sum8)
    float lanep[NW][8];
    for (int t = 0; t < iters; ++t) {
        float(*rd)[RU * M * NW] = red[t & 1];
        const int e_col = min(((int)blockIdx.x + t * (int)gridDim.x) * R + e_ur, N - 1);
        const bf16_t e_res = res_p[residual != nullptr ? (int64_t)e_m * ldr + e_col : 0];
        const bf16_t e_bias = bias_p[bias != nullptr ? e_col : 0];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            // request row u + PRIME: of this batch, or - across the batch boundary - of the next one
            if (u + PRIME < RU) {
                W4_ISSUE_ROW(t, u + PRIME)
            } else if (t + 1 < iters) {
                W4_ISSUE_ROW(t + 1, u + PRIME - RU)
            }
            asm volatile("" ::: "memory");
            // waves that are behind run at higher priority than waves that are ahead (the SIMD's default is oldest first,
            // which lets the first workgroup of a CU finish early and leaves its share of the memory queue idle)
            if (RU >= 4) {
                if (u == 0) __builtin_amdgcn_s_setprio(3);
                if (u == RU / 4) __builtin_amdgcn_s_setprio(2);
                if (u == RU / 2) __builtin_amdgcn_s_setprio(1);
                if (u == (3 * RU) / 4) __builtin_amdgcn_s_setprio(0);
            }
            // consume row u (its registers are re-filled PRIME steps later at the earliest ... by row u + PRIME - RU of the
            // next batch, which is only requested after this use)
            float part[NW][M];
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                const float s = CB ? __uint_as_float(mt[q][u]) : bflo(mt[q][u]);
                const float zz = CB ? 0.f : 128.0f + bfhi(mt[q][u]);
                uint32_t wcb[16];
                if constexpr (CB) w4c_slice_lookup(w[q][u], cb_col, wcb);  // once per weight row, shared by the M input rows
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    float v;
                    if constexpr (CB)
                        v = s * w4c_pairs_dot(wcb, xr[m]);
                    else
                        v = s * (w4_slice_dot(w[q][u], xr[m]) - zz * xs[m]);
                    if constexpr (kSum8)
                        lanep[q][u & 7] = v;  // all eight rows are reduced together after the batch
                    else
                        part[q][m] = wave_sum_to_lane63(v);
                }
            }
            if constexpr (!kSum8) {
#pragma unroll
                for (int q = 0; q < NW; ++q)
#pragma unroll
                    for (int m = 0; m < M; ++m)
                        if (lane == 63) rd[wave][(u * M + m) * NW + q] = part[q][m];
            } else {
                // without the per-row reduction (and its LDS store) nothing anchors a row's arithmetic between the load
                // requests any more: the compiler sinks all eight dot products below the last request - every row's
                // weights live at once (195 - 256 VGPRs, spills) and no compute under the loads.  Pin the row's result
                // to this point of the request sequence.
#pragma unroll
                for (int q = 0; q < NW; ++q) asm volatile("" : "+v"(lanep[q][u & 7])::"memory");
            }
        }
        if constexpr (kSum8) {
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                const float tot = wave_sum8(lanep[q]);
                if ((lane & 7) == 0) rd[wave][(lane >> 3) * NW + q] = tot;  // row u = lane / 8 (M = 1)
            }
        }
        if (t == 0) w4_stamp(dbg, 2);
        __syncthreads();
        if (t == 0) w4_stamp(dbg, 3);
        // epilogue: one thread per (row of the batch, m); sums the slabs in a fixed order
        if ((int)threadIdx.x < R * M) {
            const int m = threadIdx.x % M, ur = threadIdx.x / M;
            const int jj = ur / RU, u = ur % RU;
            const int col = ((int)blockIdx.x + t * (int)gridDim.x) * R + ur;
            if (col < N) {
                float a0 = 0.f, a1 = 0.f;
                for (int c = 0; c < plan.nslabs; ++c) {
                    a0 += rd[c * wps + jj][(u * M + m) * NW];
                    if (DUAL) a1 += rd[c * wps + jj][(u * M + m) * NW + 1];
                }
                out[(int64_t)m * ldo + col] = apply_epilogue_vals(epi, a0, a1, bias != nullptr, bf2f(e_bias), bf2f(e_res));
            }
        }
    }
#undef W4_ISSUE_ROW
#undef W4_ROW0
    w4_stamp(dbg, 4);
}

#ifdef PARROT_DIAG  // tuning hooks of the diagnostic build (tools/microbench.py)
static int g_resident_override = 0;  // workgroups per launch before a workgroup walks several batches
static int g_wps_override = 0;       // row groups per workgroup, 0 = heuristic
#else
static constexpr int g_resident_override = 0, g_wps_override = 0;
#endif

// row groups per workgroup: aim at ~640 workgroups per launch, bounded by the waves the build allows
static int pick_wps(int N, int RU, int nslabs, int maxw, bool norm) {
    const int env_wps = tune_env("PARROT_W4_WPS", 0);  // experiment hook
    if (env_wps > 0 && g_wps_override == 0) {
        int w = env_wps;
        if (w * nslabs > maxw) w = maxw / nslabs;
        return w < 1 ? 1 : w;
    }
    int cap = maxw / nslabs;
    if (cap < 1) cap = 1;
    int wps = g_wps_override > 0 ? g_wps_override : (int)((N + (int64_t)RU * 320) / ((int64_t)RU * 640));
    if (norm && wps < 2 && cap >= 2 && g_wps_override == 0) wps = 2;  // share the norm over at least two row groups
    if (wps > 2 && g_wps_override == 0) wps = 2;  // measured in the decode pipeline: 2 row groups per workgroup is the sweet spot
    if (wps < 1) wps = 1;
    if (wps > cap) wps = cap;
    return wps;
}

template <int M, bool DUAL, int RU, int MAXW, bool CB>
static int w4_gemv_launch_v(const void* packed, const void* packed2, const void* x, int ldx, const void* bias,
                            const void* residual, int ldr, void* out, int ldo, int N, int K, int epi, const NormArgs& na,
                            const W4Plan& plan, hipStream_t st, const void* code) {
    const int wps = pick_wps(N, RU, plan.nslabs, MAXW, na.kind != 0);
    const int R = wps * RU;
    const int nthreads = 64 * plan.nslabs * wps;
    size_t lds = 0;
    if (na.kind != 0) {
        PARROT_UNSUPPORTED((K >> 3) <= 4 * nthreads, "w4_gemv: fused norm needs K <= %d with this workgroup shape", 32 * nthreads);
        lds = (size_t)M * K * 2;
        PARROT_UNSUPPORTED(lds <= 64 * 1024, "w4_gemv: fused norm needs %zu B of LDS", lds);
    }
    // grid = what is resident at once (4 waves per SIMD with either build), every workgroup walks `iters` batches of rows
    const int batches = (N + R - 1) / R;
    int resident = 256 * (16 / (plan.nslabs * wps) > 0 ? 16 / (plan.nslabs * wps) : 1);
    if (g_resident_override > 0) resident = g_resident_override;
    const int iters = (batches + resident - 1) / resident;
    const dim3 grid((batches + iters - 1) / iters), block(nthreads);
    return launch(CB ? (DUAL ? K_W4C_GEMV_DUAL : K_W4C_GEMV) : (DUAL ? K_W4_GEMV_DUAL : K_W4_GEMV),
                  w4_gemv_kernel<M, DUAL, RU, MAXW, CB>, grid, block, lds, st,
                  (const uint4*)packed, (const uint4*)packed2, (const bf16_t*)x, ldx, (const bf16_t*)bias,
                  (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, N, K, wps, epi, iters, na, plan, g_w4_dbg_host,
                  (const uint32_t*)code);
}

template <int M, bool CB>
static int w4_gemv_launch(const void* packed, const void* packed2, const void* x, int ldx, const void* bias,
                          const void* residual, int ldr, void* out, int ldo, int N, int K, int epi, const NormArgs& na,
                          const W4Plan& plan, hipStream_t st, const void* code) {
    // rows in flight per wave: 8 for the single-row decode kernel, fewer when a second weight or more rows share the registers
    constexpr int RU1 = (M == 1) ? 8 : 4;  // (16 / 8 rows in flight measured slower: occupancy drops to 3 waves per SIMD)
    // (codebook weights: the 32 raw lookups of a slice live beside the weights: 8 x 2 rows would take 168 VGPRs, 3 waves per SIMD)
    constexpr int RU2 = (M == 1) ? (CB ? 4 : 8) : ((M <= 2) ? 4 : 2);  // M = 1: 8 rows x 2 weights per wave measured 3 % faster end to end than 4
#define PARROT_W4_GO(DUALV, RUV, MAXWV) \
    return w4_gemv_launch_v<M, DUALV, RUV, MAXWV, CB>(packed, packed2, x, ldx, bias, residual, ldr, out, ldo, N, K, epi, na, plan, st, code)
    if (plan.nslabs <= 8) {
        if (epi == PARROT_EPI_SWIGLU) PARROT_W4_GO(true, RU2, 8);
        // small single-row launches (the attention out-projection: N x slabs <= 16 K): 4 rows per wave, twice the waves -
        // measured +1 % end to end; for the MLP down-projection (N = 4096 but 6 slabs) the same change costs 2.5 %
        if (M == 1 && (int64_t)N * plan.nslabs <= 8192 * 2) PARROT_W4_GO(false, 4, 8);
        PARROT_W4_GO(false, RU1, 8);
    }
    if (epi == PARROT_EPI_SWIGLU) PARROT_W4_GO(true, 2, 16);
    PARROT_W4_GO(false, 4, 16);
#undef PARROT_W4_GO
}

template <bool CB>
static int w4_gemv_entry(const char* who, const void* packed, const void* packed2, const void* code, const void* x, int ldx, int M,
                         const void* bias, const void* residual, int ldr, void* out, int ldo, int N, int K, int group,
                         int epilogue, const parrot_norm_t* norm, void* stream) {
    int rc = check_linear_args(who, packed, packed2, x, ldx, M, residual, ldr, out, ldo, N, K, epilogue);
    if (rc != PARROT_OK) return rc;
    PARROT_UNSUPPORTED(!(epilogue == PARROT_EPI_SWIGLU && bias), "%s: SWIGLU epilogue takes no bias", who);
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
            case 1: rc = w4_gemv_launch<1, CB>(packed, packed2, xm, ldx, bias, rm, ldr, om, ldo, N, K, epilogue, na, plan, st, code); break;
            case 2: rc = w4_gemv_launch<2, CB>(packed, packed2, xm, ldx, bias, rm, ldr, om, ldo, N, K, epilogue, na, plan, st, code); break;
            case 3: rc = w4_gemv_launch<3, CB>(packed, packed2, xm, ldx, bias, rm, ldr, om, ldo, N, K, epilogue, na, plan, st, code); break;
            default: rc = w4_gemv_launch<4, CB>(packed, packed2, xm, ldx, bias, rm, ldr, om, ldo, N, K, epilogue, na, plan, st, code); break;
        }
        if (rc != PARROT_OK) return rc;
    }
    return PARROT_OK;
}

// Codebook weights -> dense bf16 (N, K): w = bf16(code[q] * absmax), the rounding bitsandbytes' dequantize_4bit applies before
// the matmul (the prefill path multiplies this on the matrix cores, as the reference does: bnb MatMul4Bit = dequantise + F.linear).
// One thread per 16-byte slice (32 weights -> 64 bytes of output).
__global__ void __launch_bounds__(256)
w4c_dequant_kernel(const uint4* __restrict__ W, const float* __restrict__ code, bf16_t* __restrict__ out, int ldo, int N,
                   int slices_per_row, W4Plan plan) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (int64_t)N * slices_per_row) return;
    const int o = (int)(tid / slices_per_row), t = (int)(tid % slices_per_row);
    int c = 0;
    while (c + 1 < plan.nslabs && t >= plan.slab[c + 1].slice0) ++c;
    const W4Slab sl = plan.slab[c];
    const uint4* rec = W + (int64_t)o * plan.row16;
    const uint4 v4 = rec[sl.w_off16 + (t - sl.slice0)];
    const float absmax = __uint_as_float(reinterpret_cast<const uint32_t*>(rec + sl.meta_off16)[t / plan.Gs - sl.g0]);
    const uint32_t dw[4] = {v4.x, v4.y, v4.z, v4.w};
    uint32_t o32[16];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float lo = code[(dw[d] >> (4 * i)) & 0xFu] * absmax, hi = code[(dw[d] >> (16 + 4 * i)) & 0xFu] * absmax;
            o32[4 * d + i] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
        }
    uint4* dst = reinterpret_cast<uint4*>(out + (int64_t)o * ldo + (int64_t)t * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = make_uint4(o32[4 * q], o32[4 * q + 1], o32[4 * q + 2], o32[4 * q + 3]);
}

}  // namespace parrot

using namespace parrot;

extern "C" {

#ifdef PARROT_DIAG  // diagnostic build: tuning / stamp hooks, not part of the public header
int parrot_tune_w4_stamps(void* dbg24_u64) {  // device buffer of 24 uint64, or NULL to switch off
    g_w4_dbg_host = (unsigned long long*)dbg24_u64;
    return PARROT_OK;
}
int parrot_tune_w4_resident(int workgroups) {  // 0 = derive from the workgroup shape
    g_resident_override = workgroups > 0 ? workgroups : 0;
    return PARROT_OK;
}
int parrot_tune_w4_rows_per_wg(int wps) {  // (name kept) row groups per workgroup, 0 = heuristic
    g_wps_override = (wps >= 1 && wps <= 8) ? wps : 0;
    return PARROT_OK;
}
#endif

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
    return w4_gemv_entry<false>("w4_gemv", packed, packed2, nullptr, x, ldx, M, bias, residual, ldr, out, ldo, N, K, group, epilogue,
                                norm, stream);
}

int parrot_w4c_gemv(const void* packed, const void* packed2, const void* code16_bf16, const void* x, int ldx, int M,
                    const void* bias, const void* residual, int ldr, void* out, int ldo, int N, int K, int block,
                    int epilogue, const parrot_norm_t* norm, void* stream) {
    PARROT_REQUIRE(code16_bf16 != nullptr, "w4c_gemv: codebook pointer is null");
    return w4_gemv_entry<true>("w4c_gemv", packed, packed2, code16_bf16, x, ldx, M, bias, residual, ldr, out, ldo, N, K, block,
                               epilogue, norm, stream);
}

int parrot_w4c_dequant(const void* packed, const void* code16_f32, void* out, int ldo, int N, int K, int block, void* stream) {
    PARROT_REQUIRE(packed && code16_f32 && out, "w4c_dequant: null pointer");
    PARROT_REQUIRE(ldo >= K && ldo % 8 == 0 && aligned16(out) && aligned16(packed), "w4c_dequant: out must be 16-byte aligned rows, ldo >= K");
    W4Plan plan;
    const int rc = w4_make_plan(N, K, block, &plan);
    if (rc != PARROT_OK) return rc;
    const int64_t total = (int64_t)N * (K / 32);
    const int64_t blocks = (total + 255) / 256;
    PARROT_UNSUPPORTED(blocks < (1ll << 31), "w4c_dequant: matrix too large");
    return launch(K_W4C_DEQUANT, w4c_dequant_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)packed,
                  (const float*)code16_f32, (bf16_t*)out, ldo, N, K / 32, plan);
}

}  // extern "C"
