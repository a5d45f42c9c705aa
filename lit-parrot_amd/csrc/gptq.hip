// GPTQ column loop on the device (reference quantize/gptq.py:397-431; Frantar et al., arXiv:2210.17323).
//
// One block of BS = 128 columns of the weight matrix: column by column  q = grid(w),  err = (w - q) / Hinv_ii,
// w[i:] -= err * Hinv[i, i:]  inside the block.  The rows are independent, the columns are a serial chain - so ONE WAVE
// OWNS ONE ROW and its 64 lanes hold the 128 columns (two registers per lane): a step broadcasts column i from its lane,
// every lane computes the (uniform) quantised value and error, and updates its own two columns with the Hinv row from
// LDS.  The inverse-Hessian block (128 x 128 fp32 = 64 KB) is staged once per workgroup of 16 rows.
// Outputs per block: the quantised (dequantised-value) columns, the error columns for the trailing update
// W[:, behind] -= Err @ Hinv[block, behind] (a GEMM, done by the caller), per-group grid parameters when a group starts
// inside the block, and the row's contribution to the loss.
#include "parrot_common.h"

namespace parrot {

constexpr int kGptqBS = 128;
constexpr int kGptqRows = 16;  // rows (waves) per workgroup

__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// groupsize: 0 = the grid parameters in scales/zeros[row * ngroups + group0] are given (per-channel: computed from the
// original weights by the caller); else a divisor of 128: parameters are recomputed at every group start from the current
// (error-compensated) columns of the group and written to scales/zeros[row * ngroups + (col0 + i) / groupsize].
__global__ void __launch_bounds__(kGptqRows * 64)
gptq_block_kernel(float* __restrict__ W, int ldw, int rows, int col0, int ncols, const float* __restrict__ Hinv, int ldh,
                  float* __restrict__ Q, int ldq, float* __restrict__ Err, float* __restrict__ scales, float* __restrict__ zeros,
                  int ngroups, int groupsize, float maxq, int round_bf16, float* __restrict__ loss_rows) {
    extern __shared__ float hb[];  // [ncols][kGptqBS] the block of the upper Cholesky factor of H^-1
    for (int idx = threadIdx.x; idx < ncols * kGptqBS; idx += kGptqRows * 64) {
        const int i = idx / kGptqBS, jj = idx % kGptqBS;
        hb[idx] = jj < ncols ? Hinv[(int64_t)(col0 + i) * ldh + col0 + jj] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * kGptqRows + wave;
    if (row >= rows) return;  // (no barrier below)
    float* wrow = W + (int64_t)row * ldw + col0;
    float w[2], qv[2] = {0.f, 0.f}, ev[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2; ++r) w[r] = (r * 64 + lane < ncols) ? wrow[r * 64 + lane] : 0.f;
    int g = groupsize > 0 ? col0 / groupsize : 0;
    float scale = scales[(int64_t)row * ngroups + g], zero = zeros[(int64_t)row * ngroups + g];
    float loss = 0.f;
    for (int i = 0; i < ncols; ++i) {
        if (groupsize > 0 && (col0 + i) % groupsize == 0) {
            // find_params_weight (gptq.py:317-347) over the group's current values: range always contains 0
            float lo = 0.f, hi = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int c = r * 64 + lane;
                if (c >= i && c < i + groupsize && c < ncols) {
                    lo = fminf(lo, w[r]);
                    hi = fmaxf(hi, w[r]);
                }
            }
            lo = wave_min_f(lo);
            hi = wave_max_f(hi);
            if (lo == 0.f && hi == 0.f) {
                lo = -1.f;
                hi = 1.f;
            }
            scale = __fdiv_rn(__fsub_rn(hi, lo), maxq);
            zero = rintf(__fdiv_rn(-lo, scale));
            if (round_bf16) scale = rbf(scale);  // the grid that will be STORED (bf16 checkpoint): quantise onto exactly that one
            g = (col0 + i) / groupsize;
            if (lane == 0) {
                scales[(int64_t)row * ngroups + g] = scale;
                zeros[(int64_t)row * ngroups + g] = zero;
            }
        }
        const float wi = __shfl(i < 64 ? w[0] : w[1], i & 63, 64);
        const float d = hb[i * kGptqBS + i];
        const float q = __fmul_rn(scale, __fsub_rn(fminf(fmaxf(__fadd_rn(rintf(__fdiv_rn(wi, scale)), zero), 0.f), maxq), zero));
        const float diff = __fsub_rn(wi, q);
        const float e = __fdiv_rn(diff, d);
        loss += __fdiv_rn(__fmul_rn(diff, diff), __fmul_rn(d, d));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int c = r * 64 + lane;
            if (c == i) {
                qv[r] = q;
                ev[r] = e;
            }
            if (c >= i) w[r] = __fsub_rn(w[r], __fmul_rn(e, hb[i * kGptqBS + c]));  // product and difference rounded separately, as the reference's outer product + subtraction
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int c = r * 64 + lane;
        if (c < ncols) {
            Q[(int64_t)row * ldq + col0 + c] = qv[r];
            Err[(int64_t)row * kGptqBS + c] = ev[r];
        } else {
            Err[(int64_t)row * kGptqBS + c] = 0.f;
        }
    }
    if (lane == 0) loss_rows[row] += 0.5f * loss;
}

}  // namespace parrot

using namespace parrot;

extern "C" {

int parrot_gptq_block(void* W, int ldw, int rows, int col0, int ncols, const void* Hinv, int ldh, void* Q, int ldq, void* Err,
                      void* scales, void* zeros, int ngroups, int groupsize, int maxq, int round_bf16, void* loss_rows, void* stream) {
    PARROT_REQUIRE(W && Hinv && Q && Err && scales && zeros && loss_rows, "gptq_block: null pointer");
    PARROT_REQUIRE(rows >= 1 && col0 >= 0 && ncols >= 1 && ncols <= kGptqBS, "gptq_block: bad block rows=%d col0=%d ncols=%d", rows, col0, ncols);
    PARROT_UNSUPPORTED(groupsize == 0 || (groupsize > 0 && kGptqBS % groupsize == 0 && col0 % groupsize == 0),
                       "gptq_block: group size %d must divide the block of %d columns (or 0 for given parameters)", groupsize, kGptqBS);
    PARROT_REQUIRE(ngroups >= 1 && maxq >= 1, "gptq_block: bad ngroups / maxq");
    const size_t lds = (size_t)ncols * kGptqBS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {  // 64 KB of dynamic LDS needs the opt-in
        hipError_t e = hipFuncSetAttribute((const void*)gptq_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kGptqBS * kGptqBS * 4);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");
        attr_set = true;
    }
    return launch(K_GPTQ_BLOCK, gptq_block_kernel, dim3((rows + kGptqRows - 1) / kGptqRows), dim3(kGptqRows * 64), lds, (hipStream_t)stream,
                  (float*)W, ldw, rows, col0, ncols, (const float*)Hinv, ldh, (float*)Q, ldq, (float*)Err, (float*)scales, (float*)zeros,
                  ngroups, groupsize, (float)maxq, round_bf16, (float*)loss_rows);
}

}  // extern "C"
