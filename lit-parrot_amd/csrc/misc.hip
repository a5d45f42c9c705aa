// Small ops of the decode step: token embedding gather and the greedy sampling step.
#include "parrot_common.h"

namespace parrot {

// x[m] = wte[tokens[base + m]]  (lit_gpt/model.py:99); 16-B copies
__global__ void __launch_bounds__(256)
embedding_kernel(const uint4* __restrict__ wte, int d16, const int64_t* __restrict__ tokens,
                 const int32_t* __restrict__ pos_ptr, uint4* __restrict__ out, int ldo16) {
    const int m = blockIdx.y;
    const int64_t base = pos_ptr ? (int64_t)pos_ptr[0] : 0;
    const int64_t tok = tokens[base + m];
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < d16; c += gridDim.x * blockDim.x)
        out[(int64_t)m * ldo16 + c] = wte[tok * d16 + c];
}

// generate/base.py:136-153 with temperature > 0 and top_k = 1: the sampled token is the arg-max of the logits.
// (The reference draws from a one-hot multinomial; with tied maxima it picks one of them at random, here the
// lowest index wins.)  Single workgroup; then the loop state advances: tokens[pos+1] = best, pos += 1.
constexpr int kArgmaxThreads = 1024;
__global__ void __launch_bounds__(kArgmaxThreads)
argmax_advance_kernel(const bf16_t* __restrict__ logits, int V, int vec, int64_t* __restrict__ tokens, int32_t* __restrict__ pos_ptr) {
    __shared__ float sv[kArgmaxThreads / 64];
    __shared__ int si[kArgmaxThreads / 64];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    // 16-byte chunks (8 logits), thread t owns chunks t, t + 1024, ...: kArgmaxIt of them are requested together
    // (clamped, unconditional) before the first compare; indices ascend per thread, so ties keep the lowest.
    // (The row must be 16-byte aligned: checked on the host; a ragged tail is masked by index.)
    constexpr int kArgmaxIt = 4;
    const int chunks = (V + 7) >> 3;
    const uint4* lg = reinterpret_cast<const uint4*>(logits);
    if (!vec) {  // fewer than 8 logits or an unaligned row: element loads
        for (int i = threadIdx.x; i < V; i += kArgmaxThreads) {
            float v = bf2f(logits[i]);
            if (v != v) v = -INFINITY;
            if (bi == 0x7fffffff || v > best) {
                best = v;
                bi = i;
            }
        }
    }
    for (int c0 = threadIdx.x; vec && c0 < chunks; c0 += kArgmaxThreads * kArgmaxIt) {
        uint4 v4[kArgmaxIt];
#pragma unroll
        for (int k = 0; k < kArgmaxIt; ++k) v4[k] = lg[min(c0 + k * kArgmaxThreads, (V >> 3) - 1)];
#pragma unroll
        for (int k = 0; k < kArgmaxIt; ++k) {
            const int c = c0 + k * kArgmaxThreads;
            const uint32_t dw[4] = {v4[k].x, v4[k].y, v4[k].z, v4[k].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = c * 8 + e;
                float v = (e & 1) ? bfhi(dw[e >> 1]) : bflo(dw[e >> 1]);
                if (c >= (V >> 3)) v = (i < V) ? bf2f(logits[i < V ? i : 0]) : -INFINITY;  // ragged tail chunk: element loads
                if (v != v) v = -INFINITY;  // a NaN logit never wins
                if (i < V && (bi == 0x7fffffff || v > best)) {
                    best = v;
                    bi = i;
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sv[wave] = best;
        si[wave] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kArgmaxThreads / 64; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
                best = sv[w];
                bi = si[w];
            }
        const int pos = pos_ptr[0];
        tokens[pos + 1] = (bi == 0x7fffffff) ? 0 : bi;
        pos_ptr[0] = pos + 1;
    }
}

// Device-side stop-sequence check of the chat loop (chat/base.py:80-87 restated on the token buffer): after the sampling
// step has written tokens[pos] (generated token number t = pos - first_gen), a stop sequence of n tokens matches iff
// t >= L - 1 (L = longest stop sequence: the reference's look-back buffer is still filling before that and its tail holds
// the filler) and tokens[pos-n+1 .. pos] equals it; sequences are tried in list order, the first hit is latched:
// flag[0] = t of the hit (-1: none so far), flag[1] = n.  One wave; the host reads the flag every few tokens instead of
// comparing on the host after every token.
__global__ void __launch_bounds__(64)
stop_check_kernel(const int64_t* __restrict__ tokens, const int32_t* __restrict__ pos_ptr, const int32_t* __restrict__ first_gen,
                  const int64_t* __restrict__ stop_flat, const int32_t* __restrict__ stop_off, int n_stop, int L,
                  int32_t* __restrict__ flag) {
    if (flag[0] >= 0) return;  // latched
    const int pos = pos_ptr[0];
    const int t = pos - first_gen[0];
    if (t < L - 1) return;
    for (int s = 0; s < n_stop; ++s) {
        const int o = stop_off[s], n = stop_off[s + 1] - o;
        bool eq = true;
        for (int i = threadIdx.x; i < n; i += 64) eq = eq && (tokens[pos - n + 1 + i] == stop_flat[o + i]);
        if (__all(eq)) {  // uniform
            if (threadIdx.x == 0) {
                flag[0] = t;
                flag[1] = n;
            }
            return;
        }
    }
}

}  // namespace parrot

using namespace parrot;

extern "C" {

int parrot_embedding(const void* wte, int d, const int64_t* tokens, const int32_t* pos, int M, void* out, int ldo,
                     void* stream) {
    PARROT_REQUIRE(wte && tokens && out, "embedding: null pointer");
    PARROT_REQUIRE(M >= 1 && M <= 65535 && d >= 8 && d % 8 == 0 && ldo % 8 == 0 && ldo >= d,
                   "embedding: d and ldo must be multiples of 8 (d=%d ldo=%d M=%d)", d, ldo, M);
    PARROT_REQUIRE(aligned16(wte) && aligned16(out), "embedding: pointers must be 16-byte aligned");
    const int d16 = d / 8;
    return launch(K_EMBEDDING, embedding_kernel, dim3((d16 + 255) / 256, M), dim3(256), 0, (hipStream_t)stream,
                  (const uint4*)wte, d16, tokens, pos, (uint4*)out, ldo / 8);
}

int parrot_stop_check(const int64_t* tokens, const int32_t* pos, const int32_t* first_gen, const int64_t* stop_flat,
                      const int32_t* stop_off, int n_stop, int longest, int32_t* flag, void* stream) {
    PARROT_REQUIRE(tokens && pos && first_gen && flag, "stop_check: null pointer");
    PARROT_REQUIRE(n_stop >= 0 && longest >= 1, "stop_check: bad arguments n_stop=%d longest=%d", n_stop, longest);
    PARROT_REQUIRE(n_stop == 0 || (stop_flat && stop_off), "stop_check: stop sequences missing");
    return launch(K_STOP_CHECK, stop_check_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tokens, pos, first_gen, stop_flat,
                  stop_off, n_stop, longest, flag);
}

int parrot_argmax_advance(const void* logits, int V, int64_t* tokens, int32_t* pos, void* stream) {
    PARROT_REQUIRE(logits && tokens && pos, "argmax_advance: null pointer");
    PARROT_REQUIRE(V >= 1, "argmax_advance: V=%d", V);
    const int vec = (V >= 8 && aligned16(logits)) ? 1 : 0;
    return launch(K_ARGMAX, argmax_advance_kernel, dim3(1), dim3(kArgmaxThreads), 0, (hipStream_t)stream,
                  (const bf16_t*)logits, V, vec, tokens, pos);
}

}  // extern "C"
