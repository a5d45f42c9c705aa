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

// generate/base.py:136-153 with top_k != 1 (the reference's default call is temperature 0.8, top_k 200): the whole sampling
// step in one launch, with the arithmetic of the torch device ops the reference runs there, so that the same generator state
// draws the same token (tools/probes/torch_sampling_probe.py checks each equivalence on the GPU):
//     logits = logits / temperature                      bf16(float(l) * (1.0f / T))  (bf16 tensor / python scalar)
//     v, _ = topk(logits, k); logits = where(logits < v[[-1]], -inf, logits)    kept: the values >= the k-th largest
//     probs = softmax(logits)                            bf16(exp(x - max) / sum), fp32 inside
//     idx = multinomial(probs, 1)                        = argmax(probs / q), q = empty_like(probs).exponential_(1): the caller
//                                                        draws q with that very torch call (same generator consumption, graph
//                                                        safe) and hands it over; ties: the lowest index
//     tokens[pos + 1] = idx; pos += 1
// torch.multinomial's validity checks (two host syncs per token) are not reproduced.  One workgroup: the k-th largest value by a
// two-pass radix select over the 16-bit keys (256-bin histograms in LDS), then the maximum / sum, then the arg-max.
constexpr int kSampleThreads = 1024;
__device__ __forceinline__ uint32_t bf16_key(bf16_t b) {  // monotone: a < b (as numbers) <=> key(a) < key(b); -0 < +0
    return (b & 0x8000u) ? (uint32_t)(uint16_t)~b : (uint32_t)b | 0x8000u;
}
// bin (from the top) in which the running count reaches k: wave 0 scans the 256-bin histogram, 4 bins per lane from the top;
// returns the bin and, through *before, how many elements lie in the bins above it
__device__ __forceinline__ int select_bin(const uint32_t* hist, uint32_t k, uint32_t* before) {
    const int lane = threadIdx.x & 63;
    uint32_t h[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = hist[255 - (4 * lane + i)];
        s += h[i];
    }
    uint32_t incl = s;  // inclusive prefix over the lanes (lane 0 = the top bins)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += o;
    }
    const uint32_t excl = incl - s;
    const bool mine = excl < k && k <= incl;
    int bin = -1;
    uint32_t bef = 0;
    if (mine) {
        uint32_t c = excl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (bin < 0 && c + h[i] >= k) {
                bin = 255 - (4 * lane + i);
                bef = c;
            }
            c += h[i];
        }
    }
    const uint64_t who = __ballot(mine);
    const int src = who ? __ffsll((long long)who) - 1 : 0;
    *before = (uint32_t)__shfl((int)bef, src, 64);
    return __shfl(bin, src, 64);
}
// The scaled logits of a thread: NCH > 0: chunks of 8 consecutive elements t, t + 1024, ... held in registers (one 16-byte load
// each; the row 16-byte aligned, V a multiple of 8, V <= 8192 * NCH); NCH = 0: any V, re-read element by element in every pass.
// each(f) calls f(index, scaled bf16) for the thread's elements in ascending index order.
template <int NCH>
struct SampleElems {
    uint32_t tv[NCH > 0 ? NCH : 1][4];
    const bf16_t* logits;
    int V;
    float inv_t;
    __device__ __forceinline__ bf16_t scaled(bf16_t l) const { return f2bf(__fmul_rn(bf2f(l), inv_t)); }
    __device__ __forceinline__ void load() {
        if constexpr (NCH > 0) {
            const uint4* lg = reinterpret_cast<const uint4*>(logits);
            const int chunks = V >> 3;
            uint4 v4[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) v4[j] = lg[min((int)threadIdx.x + j * kSampleThreads, chunks - 1)];
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const uint32_t dw[4] = {v4[j].x, v4[j].y, v4[j].z, v4[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    tv[j][e] = (uint32_t)scaled((bf16_t)(dw[e] & 0xffffu)) | ((uint32_t)scaled((bf16_t)(dw[e] >> 16)) << 16);
            }
        }
    }
    template <class F>
    __device__ __forceinline__ void each(F f) const {
        if constexpr (NCH > 0) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c = (int)threadIdx.x + j * kSampleThreads;
                if (c < (V >> 3)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f(c * 8 + e, (bf16_t)((e & 1) ? (tv[j][e >> 1] >> 16) : (tv[j][e >> 1] & 0xffffu)));
                }
            }
        } else {
            for (int i = threadIdx.x; i < V; i += kSampleThreads) f(i, scaled(logits[i]));
        }
    }
};

template <int NCH>
__global__ void __launch_bounds__(kSampleThreads)
topk_sample_kernel(const bf16_t* __restrict__ logits, int V, float inv_temperature, int top_k, const bf16_t* __restrict__ noise,
                   bf16_t* __restrict__ probs_out, int64_t* __restrict__ tokens, int32_t* __restrict__ pos_ptr) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel[2];
    __shared__ float redf[kSampleThreads / 64];
    __shared__ int redi[kSampleThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    SampleElems<NCH> el;
    el.logits = logits;
    el.V = V;
    el.inv_t = inv_temperature;
    el.load();
    // ---- the k-th largest scaled logit (as a key); top_k <= 0 or >= V: everything is kept
    uint32_t kth_key = 0u;
    if (top_k > 0 && top_k < V) {
        uint32_t k = (uint32_t)top_k, hi = 0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = threadIdx.x; i < 256; i += kSampleThreads) hist[i] = 0u;
            __syncthreads();
            el.each([&](int, bf16_t t) {
                const uint32_t key = bf16_key(t);
                if (pass == 0)
                    atomicAdd(&hist[key >> 8], 1u);
                else if ((key >> 8) == hi)
                    atomicAdd(&hist[key & 0xffu], 1u);
            });
            __syncthreads();
            if (wave == 0) {
                uint32_t before;
                const int bin = select_bin(hist, k, &before);
                if (lane == 0) {
                    sel[0] = (uint32_t)bin;
                    sel[1] = before;
                }
            }
            __syncthreads();
            if (pass == 0) {
                hi = sel[0];
                k -= sel[1];
            } else {
                kth_key = (hi << 8) | sel[0];
            }
            __syncthreads();
        }
    }
    // ---- maximum and sum of exp over the kept values (fp32; the maximum is always kept)
    float mx = -INFINITY;
    el.each([&](int, bf16_t t) { mx = fmaxf(mx, bf2f(t)); });
    mx = wave_max(mx);
    if (lane == 0) redf[wave] = mx;
    __syncthreads();
    for (int w = 0; w < kSampleThreads / 64; ++w) mx = fmaxf(mx, redf[w]);
    __syncthreads();
    float sum = 0.f;
    el.each([&](int, bf16_t t) {
        if (bf16_key(t) >= kth_key) sum += expf(bf2f(t) - mx);
    });
    sum = wave_sum(sum);
    if (lane == 0) redf[wave] = sum;
    __syncthreads();
    sum = 0.f;
    for (int w = 0; w < kSampleThreads / 64; ++w) sum += redf[w];
    // ---- arg-max of probs / q (both bf16, the quotient rounded to bf16), lowest index on ties
    float best = -INFINITY;
    int bi = 0x7fffffff;
    el.each([&](int i, bf16_t t) {
        const bf16_t pb = bf16_key(t) >= kth_key ? f2bf(__fdiv_rn(expf(bf2f(t) - mx), sum)) : (bf16_t)0;
        if (probs_out != nullptr) probs_out[i] = pb;
        // (a cropped element has probability 0 and never beats a kept one: its noise is not even read)
        float r = pb != 0 ? bf2f(f2bf(__fdiv_rn(bf2f(pb), bf2f(noise[i])))) : 0.f;
        if (r != r) r = -INFINITY;
        if (bi == 0x7fffffff || r > best) {  // (indices ascend per thread: a tie keeps the lower one)
            best = r;
            bi = i;
        }
    });
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) {
            best = ov;
            bi = oi;
        }
    }
    __syncthreads();
    if (lane == 0) {
        redf[wave] = best;
        redi[wave] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kSampleThreads / 64; ++w)
            if (redi[w] != 0x7fffffff && (bi == 0x7fffffff || redf[w] > best || (redf[w] == best && redi[w] < bi))) {
                best = redf[w];
                bi = redi[w];
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

int parrot_topk_sample(const void* logits, int V, float temperature, int top_k, const void* noise_exp1, void* probs_out,
                       int64_t* tokens, int32_t* pos, void* stream) {
    PARROT_REQUIRE(logits && noise_exp1 && tokens && pos, "topk_sample: null pointer");
    PARROT_REQUIRE(V >= 1, "topk_sample: V=%d", V);
    PARROT_REQUIRE(temperature > 0.f, "topk_sample: temperature must be positive (got %g)", (double)temperature);
    // the row in registers when it is 16-byte aligned, a multiple of 8 long and at most 32768 elements; else re-read per pass
    const bool vec = V % 8 == 0 && aligned16(logits);
#define PARROT_SAMPLE_GO(NCHV)                                                                                                    \
    return launch(K_TOPK_SAMPLE, topk_sample_kernel<NCHV>, dim3(1), dim3(kSampleThreads), 0, (hipStream_t)stream, (const bf16_t*)logits, \
                  V, 1.0f / temperature, top_k, (const bf16_t*)noise_exp1, (bf16_t*)probs_out, tokens, pos)
    if (vec && V <= 8192 * 4) PARROT_SAMPLE_GO(4);
    PARROT_SAMPLE_GO(0);
#undef PARROT_SAMPLE_GO
}

int parrot_argmax_advance(const void* logits, int V, int64_t* tokens, int32_t* pos, void* stream) {
    PARROT_REQUIRE(logits && tokens && pos, "argmax_advance: null pointer");
    PARROT_REQUIRE(V >= 1, "argmax_advance: V=%d", V);
    const int vec = (V >= 8 && aligned16(logits)) ? 1 : 0;
    return launch(K_ARGMAX, argmax_advance_kernel, dim3(1), dim3(kArgmaxThreads), 0, (hipStream_t)stream,
                  (const bf16_t*)logits, V, vec, tokens, pos);
}

}  // extern "C"
