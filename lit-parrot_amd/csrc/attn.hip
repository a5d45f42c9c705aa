// CausalSelfAttention without the two Linears (lit_gpt/model.py:194-275):
//   qkv_rope_kvappend : per-group q/k/v split (:208-214), partial RoPE (:226-232, apply_rope :330-336),
//                       KV-cache write (:243-245).  GQA is stored natively: one K/V head per query group
//                       (the reference expands K/V to n_head with repeat_interleave, :217-220).
//   attn_decode       : softmax(q k^T / sqrt(hs)) v over the cached slots the causal mask admits
//                       (:256-275 with the mask rows built at :88-92), flash-decoding style split over the
//                       sequence + a combine pass.
//
// Cache layout: [n_groups][S][hs] bf16.  Slot of position p is p % S: when p >= S this overwrites the oldest
// entry, which is what the reference's roll-left + write-last does (:238-245) up to slot order.
//
// Decode attention is memory-bound on K/V rows.  A K (or V) row of hs bf16 is read by hs/8 lanes with one
// 16-B load each, so one wave instruction covers 64*8/hs rows (1 KiB contiguous); scores are reduced over
// those lanes with shuffles; every (wave, row-slot) keeps its own online-softmax state (m, l, acc[8]) in
// registers and the states are merged once at the end through LDS.
#include <hip/hip_fp16.h>

#include "parrot_common.h"

namespace parrot {

// ------------------------------------------------------------------------------------------ rope + kv append
// one thread per (row m, head-slot j in [0, n_groups*(q_per_kv+2)), pair index i in [0, hs/2))
__global__ void __launch_bounds__(256)
rope_kvappend_kernel(const bf16_t* __restrict__ qkv, int ldqkv, int M, const __half* __restrict__ rope_cos,
                     const __half* __restrict__ rope_sin, int n_elem, int rope_local, const int32_t* __restrict__ pos_ptr, int n_groups,
                     int q_per_kv, int hs, int S, bf16_t* __restrict__ q_out, bf16_t* __restrict__ k_cache,
                     bf16_t* __restrict__ v_cache) {
    const int half_hs = hs >> 1;
    const int per_group = q_per_kv + 2;
    const int slots = n_groups * per_group;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(tid % half_hs);
    const int j = (int)((tid / half_hs) % slots);
    const int m = (int)(tid / ((int64_t)half_hs * slots));
    if (m >= M) return;
    const int g = j / per_group, t = j % per_group;  // t < q_per_kv: query head; == q_per_kv: key; else value
    const int pos = pos_ptr[0] + m;
    const bf16_t* src = qkv + (int64_t)m * ldqkv + (int64_t)j * hs;
    bf16_t* dst;
    if (t < q_per_kv)
        dst = q_out + ((int64_t)m * n_groups * q_per_kv + (int64_t)g * q_per_kv + t) * hs;
    else if (t == q_per_kv)
        dst = k_cache + ((int64_t)g * S + (pos % S)) * hs;
    else
        dst = v_cache + ((int64_t)g * S + (pos % S)) * hs;

    const int half_n = n_elem >> 1;
    // element pair handled by this thread: (i, i + half_n) inside the rotary part, or two pass-through dims
    if (t <= q_per_kv && i < half_n) {
        const float x1 = bf2f(src[i]), x2 = bf2f(src[i + half_n]);
        const int64_t rrow = rope_local ? m : pos;  // tables pre-indexed per row (Block.forward callers) or absolute
        const float c1 = __half2float(rope_cos[rrow * n_elem + i]);
        const float s1 = __half2float(rope_sin[rrow * n_elem + i]);
        const float c2 = __half2float(rope_cos[rrow * n_elem + i + half_n]);
        const float s2 = __half2float(rope_sin[rrow * n_elem + i + half_n]);
        // roped = x*cos + rotate_half(x)*sin, every product and the sum rounded to fp32 separately (no FMA),
        // as the reference's promoted bf16*fp16 tensor ops do (model.py:330-336)
        const float o1 = __fadd_rn(__fmul_rn(x1, c1), __fmul_rn(-x2, s1));
        const float o2 = __fadd_rn(__fmul_rn(x2, c2), __fmul_rn(x1, s2));
        dst[i] = f2bf(o1);
        dst[i + half_n] = f2bf(o2);
    } else {
        // pass-through: value heads entirely, and dims >= n_elem of q/k.  Thread i covers 2 dims.
        const int base = (t <= q_per_kv) ? n_elem : 0;
        const int idx = (t <= q_per_kv) ? (i - half_n) * 2 : i * 2;
        dst[base + idx] = src[base + idx];
        dst[base + idx + 1] = src[base + idx + 1];
    }
}

// The same, eight dims per thread (16-byte loads / stores): one thread per (row, head slot, 8-dim chunk).  Chunk u of a query / key head
// inside the first half of the rotary part handles dims 8u .. 8u+7 AND their partners 8u + n_elem/2 ..; the chunks of the second half
// idle (their partner thread wrote them); everything else is a 16-byte copy.  Needs hs % 8 == 0, n_elem % 16 == 0 and 16-byte rows
// (every BASELINE config); the pair kernel above stays for the rest.  StableLM-3B, 512 rows x 12288 columns: 16.7 -> 9.3 us per launch.
__global__ void __launch_bounds__(256)
rope_kvappend8_kernel(const bf16_t* __restrict__ qkv, int ldqkv, int M, const __half* __restrict__ rope_cos,
                      const __half* __restrict__ rope_sin, int n_elem, int rope_local, const int32_t* __restrict__ pos_ptr, int n_groups,
                      int q_per_kv, int hs, int S, bf16_t* __restrict__ q_out, bf16_t* __restrict__ k_cache,
                      bf16_t* __restrict__ v_cache) {
    const unsigned U = (unsigned)hs >> 3;  // chunks per head
    const unsigned per_group = (unsigned)q_per_kv + 2, slots = (unsigned)n_groups * per_group;
    const unsigned per_row = slots * U;
    const int m = (int)blockIdx.y;
    const unsigned w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= per_row) return;
    const unsigned j = w / U, u = w - j * U;
    const unsigned g = j / per_group, t = j - g * per_group;  // t < q_per_kv: query head; == q_per_kv: key; else value
    const int pos = pos_ptr[0] + m;
    const bf16_t* src = qkv + (int64_t)m * ldqkv + (int64_t)j * hs;
    bf16_t* dst;
    if (t < (unsigned)q_per_kv)
        dst = q_out + ((int64_t)m * n_groups * q_per_kv + (int64_t)g * q_per_kv + t) * hs;
    else if (t == (unsigned)q_per_kv)
        dst = k_cache + ((int64_t)g * S + (pos % S)) * hs;
    else
        dst = v_cache + ((int64_t)g * S + (pos % S)) * hs;
    const unsigned half_n = (unsigned)n_elem >> 1, uh = half_n >> 3, un = (unsigned)n_elem >> 3;
    if (t <= (unsigned)q_per_kv && u < un) {
        if (u >= uh) return;  // second half of the rotary part: written by the thread of chunk u - uh
        const uint4 a = *reinterpret_cast<const uint4*>(src + 8 * u), b = *reinterpret_cast<const uint4*>(src + 8 * u + half_n);
        const int64_t rrow = rope_local ? m : pos;
        const __half* cr = rope_cos + rrow * n_elem + 8 * u;
        const __half* sr = rope_sin + rrow * n_elem + 8 * u;
        const uint4 c1 = *reinterpret_cast<const uint4*>(cr), s1 = *reinterpret_cast<const uint4*>(sr);
        const uint4 c2 = *reinterpret_cast<const uint4*>(cr + half_n), s2 = *reinterpret_cast<const uint4*>(sr + half_n);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
        const uint32_t c1w[4] = {c1.x, c1.y, c1.z, c1.w}, s1w[4] = {s1.x, s1.y, s1.z, s1.w};
        const uint32_t c2w[4] = {c2.x, c2.y, c2.z, c2.w}, s2w[4] = {s2.x, s2.y, s2.z, s2.w};
        uint32_t o1[4], o2[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            bf16_t r1[2], r2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float x1 = h ? bfhi(aw[d]) : bflo(aw[d]), x2 = h ? bfhi(bw[d]) : bflo(bw[d]);
                const float fc1 = __half2float(__ushort_as_half((unsigned short)(c1w[d] >> (16 * h))));
                const float fs1 = __half2float(__ushort_as_half((unsigned short)(s1w[d] >> (16 * h))));
                const float fc2 = __half2float(__ushort_as_half((unsigned short)(c2w[d] >> (16 * h))));
                const float fs2 = __half2float(__ushort_as_half((unsigned short)(s2w[d] >> (16 * h))));
                // (the pair kernel's arithmetic: every product and the sum rounded to fp32 separately, no FMA)
                r1[h] = f2bf(__fadd_rn(__fmul_rn(x1, fc1), __fmul_rn(-x2, fs1)));
                r2[h] = f2bf(__fadd_rn(__fmul_rn(x2, fc2), __fmul_rn(x1, fs2)));
            }
            o1[d] = (uint32_t)r1[0] | ((uint32_t)r1[1] << 16);
            o2[d] = (uint32_t)r2[0] | ((uint32_t)r2[1] << 16);
        }
        *reinterpret_cast<uint4*>(dst + 8 * u) = make_uint4(o1[0], o1[1], o1[2], o1[3]);
        *reinterpret_cast<uint4*>(dst + 8 * u + half_n) = make_uint4(o2[0], o2[1], o2[2], o2[3]);
    } else {
        *reinterpret_cast<uint4*>(dst + 8 * u) = *reinterpret_cast<const uint4*>(src + 8 * u);
    }
}

// ------------------------------------------------------------------------------------------ decode attention
constexpr int kAttnWaves = 4;

// sum over the LPR lanes that share a key row (LPR = 4, 8 or 16 consecutive lanes); every lane of the group gets the
// total.  DPP only: no LDS crossbar (ds_bpermute) and no lgkmcnt waits in the key loop's dependent chain.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    v += dpp0<0xB1>(v);                    // quad_perm [1,0,3,2]
    v += dpp0<0x4E>(v);                    // quad_perm [2,3,0,1]
    if (LPR >= 8) v += dpp0<0x141>(v);     // row_half_mirror: the two quads of an 8-lane group
    if (LPR >= 16) v += dpp0<0x128>(v);    // row_ror:8: the two halves of a 16-lane row
    return v;
}

// all-reduce over the 64 / LPR row slots of a wave (lanes with the same lane % LPR): DPP rotations inside the 16-lane rows,
// then the gfx950 lane swaps across rows (v_permlane16_swap / v_permlane32_swap with both operands the same register
// return the two halves, which are then combined) - no ds_bpermute.
template <int LPR, bool MAX>
__device__ __forceinline__ float slot_allreduce(float v) {
    auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
    if (LPR <= 4) v = op(v, dpp0<0x124>(v));  // row_ror:4
    if (LPR <= 8) v = op(v, dpp0<0x128>(v));  // row_ror:8 - every lane now holds its 16-lane row's result
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));  // rows 0+1 and 2+3
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));  // both halves
    }
    return v;
}


// partial state layout in the workspace: [M][n_head][nsplit][hs + 2] floats: acc[hs], m, l
// HQ = query heads of the group processed in one pass over the K/V rows (1 for MHA, up to 4 for GQA/MQA)
//
// PM ("softmax_mode 1", parity runs only): the softmax as torch's CPU flash-attention kernel computes it for the bf16 reference
// (lit_gpt/model.py:256-275 -> scaled_dot_product_attention): the probabilities p = exp(s - max) are ROUNDED TO bf16 before
// they multiply V, with the row maximum of the whole key block as the reference point (torch walks the keys in blocks of 512:
// for windows up to 512 keys that is the global maximum, and this mode is refused beyond), while the denominator sums the
// unrounded p; accurate expf.  It needs the maximum before the first product, hence a pass over the K rows (scores only) in
// front of the normal one.  The default (PM = false) keeps p in fp32: closer to the exact result, further from the
// reference's bits.
template <int HS, int HQ, bool PM = false>
__global__ void __launch_bounds__(kAttnWaves * 64)
attn_decode_kernel(const bf16_t* __restrict__ q, const int32_t* __restrict__ pos_ptr, const bf16_t* __restrict__ k_cache,
                   const bf16_t* __restrict__ v_cache, int n_groups, int q_per_kv, int S, int nsplit,
                   float* __restrict__ ws, bf16_t* __restrict__ y, int ldy) {
    constexpr int LPR = HS / 8;    // lanes per K/V row
    constexpr int RPW = 64 / LPR;  // rows per wave instruction
    constexpr int STRIDE = kAttnWaves * RPW;
    __shared__ float sh_acc[HQ][kAttnWaves][HS];  // one merged state per wave
    __shared__ float sh_m[HQ][kAttnWaves], sh_l[HQ][kAttnWaves];

    const int g = blockIdx.x, split = blockIdx.y, m = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, dl = lane % LPR;
    const int n_head = n_groups * q_per_kv;
    const int pos = pos_ptr[0] + m;
    const int n_valid = min(pos + 1, S);  // slots 0..n_valid-1 hold the admitted keys
    const int per = (S + nsplit - 1) / nsplit;
    const int s_begin = split * per, s_end = min(n_valid, s_begin + per);
    const int s_last = max(s_end - 1, 0);
    const float scale = 1.0f / sqrtf((float)HS);
    const uint4* kc = reinterpret_cast<const uint4*>(k_cache + (int64_t)g * S * HS);
    const uint4* vc = reinterpret_cast<const uint4*>(v_cache + (int64_t)g * S * HS);
    const int s_first = s_begin + wave * RPW;

    for (int h0 = 0; h0 < q_per_kv; h0 += HQ) {
        // the same key loop as the fused single-token kernel below: packed-bf16 dot2 scores, DPP sums over the lanes of a
        // key, two K/V register sets in ping-pong with unconditional (clamped) loads, slots merged in-wave
        uint32_t qp[HQ][4];
        float mrun[HQ], lrun[HQ], acc[HQ][8];
#pragma unroll
        for (int hh = 0; hh < HQ; ++hh) {
            const int head = g * q_per_kv + min(h0 + hh, q_per_kv - 1);
            const uint4 qv = reinterpret_cast<const uint4*>(q + ((int64_t)m * n_head + head) * HS)[dl];
            qp[hh][0] = qv.x;
            qp[hh][1] = qv.y;
            qp[hh][2] = qv.z;
            qp[hh][3] = qv.w;
            mrun[hh] = -INFINITY;
            lrun[hh] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[hh][e] = 0.f;
        }
        float gmax[HQ];  // PM: the maximum score over all the keys of this row and head
        if constexpr (PM) {
#pragma unroll
            for (int hh = 0; hh < HQ; ++hh) gmax[hh] = -INFINITY;
            for (int s0 = s_first; s0 < s_end; s0 += STRIDE) {
                const uint4 kv = kc[(int64_t)min(s0 + sub, s_last) * LPR + dl];
#pragma unroll
                for (int hh = 0; hh < HQ; ++hh) {
                    float p0 = dot2_bf16(kv.x, qp[hh][0], 0.f), p1 = dot2_bf16(kv.y, qp[hh][1], 0.f);
                    p0 = dot2_bf16(kv.z, qp[hh][2], p0);
                    p1 = dot2_bf16(kv.w, qp[hh][3], p1);
                    if (s0 + sub < s_end) gmax[hh] = fmaxf(gmax[hh], group_sum<LPR>(p0 + p1) * scale);
                    else (void)group_sum<LPR>(p0 + p1);
                }
            }
            __syncthreads();  // (the previous head chunk's merge has finished reading sh_m)
#pragma unroll
            for (int hh = 0; hh < HQ; ++hh) {
                gmax[hh] = slot_allreduce<LPR, true>(gmax[hh]);
                if (lane == 0) sh_m[hh][wave] = gmax[hh];
            }
            __syncthreads();
#pragma unroll
            for (int hh = 0; hh < HQ; ++hh) {
                for (int t = 0; t < kAttnWaves; ++t) gmax[hh] = fmaxf(gmax[hh], sh_m[hh][t]);
                mrun[hh] = gmax[hh];
            }
        }
        auto step = [&](const uint4 kv, const uint4 vv, int s) {
            const bool ok = s < s_end;
            const uint32_t vd[4] = {vv.x, vv.y, vv.z, vv.w};
            float vf[8];
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) {
                vf[2 * jq] = bflo(vd[jq]);
                vf[2 * jq + 1] = bfhi(vd[jq]);
            }
#pragma unroll
            for (int hh = 0; hh < HQ; ++hh) {
                float p0 = dot2_bf16(kv.x, qp[hh][0], 0.f), p1 = dot2_bf16(kv.y, qp[hh][1], 0.f);
                p0 = dot2_bf16(kv.z, qp[hh][2], p0);
                p1 = dot2_bf16(kv.w, qp[hh][3], p1);
                const float sc_ = ok ? group_sum<LPR>(p0 + p1) * scale : -INFINITY;
                if constexpr (PM) {
                    const float p = ok ? expf(sc_ - gmax[hh]) : 0.f;
                    const float pr = rbf(p);  // the probability as the reference's P.V product sees it
                    lrun[hh] += p;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[hh][e] = fmaf(pr, vf[e], acc[hh][e]);
                    continue;
                }
                const float mn = fmaxf(mrun[hh], sc_);
                const float corr = (mn == -INFINITY) ? 1.f : __expf(mrun[hh] - mn);
                const float p = (mn == -INFINITY) ? 0.f : __expf(sc_ - mn);
                lrun[hh] = lrun[hh] * corr + p;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[hh][e] = acc[hh][e] * corr + p * vf[e];
                mrun[hh] = mn;
            }
        };
        const int sc0 = min(s_first + sub, s_last);
        uint4 k0 = kc[(int64_t)sc0 * LPR + dl], v0 = vc[(int64_t)sc0 * LPR + dl];
        for (int s0 = s_first; s0 < s_end; s0 += 2 * STRIDE) {
            const int sn1 = min(s0 + STRIDE + sub, s_last);
            const uint4 k1 = kc[(int64_t)sn1 * LPR + dl], v1 = vc[(int64_t)sn1 * LPR + dl];
            step(k0, v0, s0 + sub);
            const int sn2 = min(s0 + 2 * STRIDE + sub, s_last);
            k0 = kc[(int64_t)sn2 * LPR + dl];
            v0 = vc[(int64_t)sn2 * LPR + dl];
            if (s0 + STRIDE < s_end) step(k1, v1, s0 + STRIDE + sub);  // wave-uniform
        }
        __syncthreads();  // the previous head chunk's merge has finished reading the LDS states
#pragma unroll
        for (int hh = 0; hh < HQ; ++hh) {
            const float mw = slot_allreduce<LPR, true>(mrun[hh]);
            const float c = (mrun[hh] == -INFINITY) ? 0.f : __expf(mrun[hh] - mw);
            lrun[hh] = slot_allreduce<LPR, false>(lrun[hh] * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[hh][e] = slot_allreduce<LPR, false>(acc[hh][e] * c);
            mrun[hh] = mw;
            if (sub == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sh_acc[hh][wave][dl * 8 + e] = acc[hh][e];
                if (dl == 0) {
                    sh_m[hh][wave] = mrun[hh];
                    sh_l[hh][wave] = lrun[hh];
                }
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < HQ * HS; idx += kAttnWaves * 64) {
            const int hh = idx / HS, d = idx % HS;
            if (h0 + hh < q_per_kv) {
                float mx = -INFINITY;
                for (int t = 0; t < kAttnWaves; ++t) mx = fmaxf(mx, sh_m[hh][t]);
                float l = 0.f, a = 0.f;
                for (int t = 0; t < kAttnWaves; ++t) {
                    const float mt = sh_m[hh][t];
                    const float w = (mt == -INFINITY) ? 0.f : __expf(mt - mx);
                    l += sh_l[hh][t] * w;
                    a += sh_acc[hh][t][d] * w;
                }
                const int head = g * q_per_kv + h0 + hh;
                if (nsplit == 1) {
                    y[(int64_t)m * ldy + (int64_t)head * HS + d] = f2bf(a / l);
                } else {
                    float* p = ws + (((int64_t)m * n_head + head) * nsplit + split) * (HS + 2);
                    p[d] = a;
                    if (d == 0) {
                        p[HS] = mx;
                        p[HS + 1] = l;
                    }
                }
            }
        }
    }
}

// merge the nsplit partial states of one (row, head); one thread per output dim
template <int HS>
__global__ void __launch_bounds__(HS)
attn_combine_kernel(const float* __restrict__ ws, int nsplit, bf16_t* __restrict__ y, int ldy, int n_head) {
    const int head = blockIdx.x, m = blockIdx.y, d = threadIdx.x;
    const float* p = ws + ((int64_t)m * n_head + head) * nsplit * (HS + 2);
    float mx = -INFINITY;
    for (int t = 0; t < nsplit; ++t) mx = fmaxf(mx, p[t * (HS + 2) + HS]);
    float l = 0.f, a = 0.f;
    for (int t = 0; t < nsplit; ++t) {
        const float mt = p[t * (HS + 2) + HS];
        const float w = (mt == -INFINITY) ? 0.f : __expf(mt - mx);
        l += p[t * (HS + 2) + HS + 1] * w;
        a += p[t * (HS + 2) + d] * w;
    }
    y[(int64_t)m * ldy + (int64_t)head * HS + d] = f2bf(a / l);
}

// ------------------------------------------------------------------------------------------ fused decode step
// One launch per layer for a single new token (M = 1): q/k/v split + RoPE + KV append + attention over the cache +
// the cross-split combine.  Replaces rope_kvappend + attn_decode + attn_combine (3 launches) on the decode path.
//   * every workgroup (group g, split j) ropes q, k_new, v_new of its group itself (a few hundred elements);
//   * the workgroup whose slot range contains slot(pos) writes the new K/V row to the cache; in the loop that row is
//     taken from registers, never from the global copy that is being written;
//   * partial softmax states go to the workspace with write-through (agent-scope) stores; an arrival ticket per group
//     elects the last workgroup, which merges the splits (guide recipe: sc1 payload -> every wave vmcnt(0) -> barrier
//     -> one relaxed agent atomic; reducer reads with agent-scope loads).  The ticket is reset by the reducer.
constexpr int kFusedMaxQ = 16;  // query heads per group held in LDS

__device__ __forceinline__ void store_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// diagnostic stamps (tools/microbench.py --attn): 100 MHz clock of workgroup (0,0,0) at phase boundaries; the buffer
// pointer is a kernel argument (a scalar load - see w4.hip for why it must not be a global)
__device__ __forceinline__ void attn_stamp(unsigned long long* dbg, int i) {
    if (dbg != nullptr && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        dbg[i] = __builtin_amdgcn_s_memrealtime();
}
#ifdef PARROT_DIAG
static unsigned long long* g_attn_dbg_host = nullptr;  // diagnostic build only: set by parrot_tune_attn_stamps
#else
static constexpr unsigned long long* g_attn_dbg_host = nullptr;
#endif

template <int HS, int HQ, int WAVES, bool PM = false>  // PM: "softmax_mode 1", see attn_decode_kernel
__global__ void __launch_bounds__(WAVES * 64)
attn_fused_decode_kernel(const bf16_t* __restrict__ qkv, const __half* __restrict__ rope_cos,
                         const __half* __restrict__ rope_sin, int n_elem, const int32_t* __restrict__ pos_ptr,
                         bf16_t* __restrict__ k_cache, bf16_t* __restrict__ v_cache, int n_groups, int q_per_kv, int S,
                         int nsplit, float* __restrict__ ws, unsigned int* __restrict__ tickets, bf16_t* __restrict__ y,
                         unsigned long long* dbg) {
    constexpr int LPR = HS / 8;
    constexpr int RPW = 64 / LPR;
    attn_stamp(dbg, 0);
    __shared__ float sh_acc[HQ][WAVES][HS];  // one merged state per wave
    __shared__ float sh_m[HQ][WAVES], sh_l[HQ][WAVES];
    __shared__ __attribute__((aligned(16))) bf16_t sh_q[kFusedMaxQ][HS];  // roped q rows as bf16 (the scale is applied to the score)
    __shared__ __attribute__((aligned(16))) bf16_t sh_kv[2][HS];  // roped k_new, v_new as stored in the cache
    __shared__ int sh_last;

    const int g = blockIdx.x, split = blockIdx.y;
    const int h0 = blockIdx.z * HQ;  // this workgroup's chunk of the group's query heads (GQA: chunks run in parallel)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, dl = lane % LPR;
    const int n_head = n_groups * q_per_kv;
    const int pos = pos_ptr[0];
    const int n_valid = min(pos + 1, S);
    const int slot_new = pos % S;
    const int per = (S + nsplit - 1) / nsplit;
    const int s_begin = split * per, s_end = min(n_valid, s_begin + per);
    const float scale = 1.0f / sqrtf((float)HS);
    const int half_n = n_elem >> 1;

    const uint4* kc = reinterpret_cast<const uint4*>(k_cache + (int64_t)g * S * HS);
    const uint4* vc = reinterpret_cast<const uint4*>(v_cache + (int64_t)g * S * HS);
    constexpr int STRIDE = WAVES * RPW;
    // Load order matters (vmcnt retires in order): first the small L2-resident operands of the RoPE phase (this
    // group's q/k/v rows, cos/sin), then the first K/V rows of this wave, which do not depend on q — their HBM latency
    // overlaps the split + RoPE phase.  (Rows past the range are clamped; the appended row is replaced later.)
    constexpr int ROPE_IT = ((kFusedMaxQ + 2) * HS + WAVES * 64 - 1) / (WAVES * 64);
    const bf16_t* grp = qkv + (int64_t)g * (q_per_kv + 2) * HS;
    const int n_rope_elems = (q_per_kv + 2) * HS;
    float rx[ROPE_IT], ro[ROPE_IT], rc[ROPE_IT], rs[ROPE_IT];
#pragma unroll
    for (int it = 0; it < ROPE_IT; ++it) {
        const int idx = threadIdx.x + it * WAVES * 64;
        rx[it] = ro[it] = rs[it] = 0.f;
        rc[it] = 1.f;
        if (idx < n_rope_elems) {
            const int t = idx / HS, d = idx % HS;
            rx[it] = bf2f(grp[idx]);
            if (t <= q_per_kv && d < n_elem) {
                rc[it] = __half2float(rope_cos[(int64_t)pos * n_elem + d]);
                rs[it] = __half2float(rope_sin[(int64_t)pos * n_elem + d]);
                ro[it] = d < half_n ? -bf2f(grp[idx + half_n]) : bf2f(grp[idx - half_n]);
            }
        }
    }
    const int s_first = s_begin + wave * RPW;
    // first K/V rows of this wave: requested here, before the RoPE phase (clamped, unconditional)
    const int sc_first = min(s_first + sub, max(s_end - 1, 0));
    const uint4 kv_cur = kc[(int64_t)sc_first * LPR + dl];
    const uint4 vv_cur = vc[(int64_t)sc_first * LPR + dl];

    // ---- split + RoPE of this group's rows (reference model.py:208-232): x*cos + rotate_half(x)*sin, each product and
    // the sum rounded to fp32 separately; elements outside the rotary part pass through (cos = 1, sin = 0 is exact)
#pragma unroll
    for (int it = 0; it < ROPE_IT; ++it) {
        const int idx = threadIdx.x + it * WAVES * 64;
        if (idx < n_rope_elems) {
            const int t = idx / HS, d = idx % HS;
            const float v = __fadd_rn(__fmul_rn(rx[it], rc[it]), __fmul_rn(ro[it], rs[it]));
            const bf16_t vb = f2bf(v);
            if (t < q_per_kv)
                sh_q[t][d] = vb;
            else
                sh_kv[t - q_per_kv][d] = vb;
        }
    }
    __syncthreads();
    attn_stamp(dbg, 1);
    // ---- KV append by the workgroup that owns the new slot
    if (blockIdx.z == 0 && slot_new >= s_begin && slot_new < s_begin + per && threadIdx.x < 2 * LPR) {
        const int which = threadIdx.x / LPR, c = threadIdx.x % LPR;
        bf16_t* dst = (which ? v_cache : k_cache) + ((int64_t)g * S + slot_new) * HS;
        reinterpret_cast<uint4*>(dst)[c] = reinterpret_cast<const uint4*>(sh_kv[which])[c];
    }
    const uint4 knew = reinterpret_cast<const uint4*>(sh_kv[0])[dl];
    const uint4 vnew = reinterpret_cast<const uint4*>(sh_kv[1])[dl];

    {
        uint32_t qp[HQ][4];  // this lane's 8 dims of every query head, packed bf16 pairs (operand of v_dot2_f32_bf16)
        float mrun[HQ], lrun[HQ], acc[HQ][8];
#pragma unroll
        for (int hh = 0; hh < HQ; ++hh) {
            const int hq = min(h0 + hh, q_per_kv - 1);
            const uint4 qv = reinterpret_cast<const uint4*>(sh_q[hq])[dl];
            qp[hh][0] = qv.x;
            qp[hh][1] = qv.y;
            qp[hh][2] = qv.z;
            qp[hh][3] = qv.w;
            mrun[hh] = -INFINITY;
            lrun[hh] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[hh][e] = 0.f;
        }
        const int s_last = max(s_end - 1, 0);
        float gmax[HQ];  // PM: the maximum score over all the keys of this head
        if constexpr (PM) {
#pragma unroll
            for (int hh = 0; hh < HQ; ++hh) gmax[hh] = -INFINITY;
            for (int s0 = s_first; s0 < s_end; s0 += STRIDE) {
                const int sk = min(s0 + sub, s_last);
                uint4 kv = kc[(int64_t)sk * LPR + dl];
                if (sk == slot_new) kv = knew;  // the row being appended by this launch
#pragma unroll
                for (int hh = 0; hh < HQ; ++hh) {
                    float p0 = dot2_bf16(kv.x, qp[hh][0], 0.f), p1 = dot2_bf16(kv.y, qp[hh][1], 0.f);
                    p0 = dot2_bf16(kv.z, qp[hh][2], p0);
                    p1 = dot2_bf16(kv.w, qp[hh][3], p1);
                    const float sc_ = group_sum<LPR>(p0 + p1) * scale;
                    if (s0 + sub < s_end) gmax[hh] = fmaxf(gmax[hh], sc_);
                }
            }
#pragma unroll
            for (int hh = 0; hh < HQ; ++hh) {
                gmax[hh] = slot_allreduce<LPR, true>(gmax[hh]);
                if (lane == 0) sh_m[hh][wave] = gmax[hh];
            }
            __syncthreads();
#pragma unroll
            for (int hh = 0; hh < HQ; ++hh) {
                for (int t = 0; t < WAVES; ++t) gmax[hh] = fmaxf(gmax[hh], sh_m[hh][t]);
                mrun[hh] = gmax[hh];
            }
            __syncthreads();  // (sh_m is written again by the wave merge below)
        }
        // one step = the STRIDE keys of the workgroup; this lane: key s, dims 8*dl .. 8*dl+7
        auto step = [&](uint4 kv, uint4 vv, int s) {
            const bool ok = s < s_end;
            if ((ok ? s : s_last) == slot_new) {  // the row being appended by this launch: use the in-register copy
                kv = knew;
                vv = vnew;
            }
            const uint32_t vd[4] = {vv.x, vv.y, vv.z, vv.w};
            float vf[8];
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) {
                vf[2 * jq] = bflo(vd[jq]);
                vf[2 * jq + 1] = bfhi(vd[jq]);
            }
#pragma unroll
            for (int hh = 0; hh < HQ; ++hh) {
                // q.k over this lane's 8 dims straight from the packed bf16 pairs, then over the LPR lanes of the key
                float p0 = dot2_bf16(kv.x, qp[hh][0], 0.f), p1 = dot2_bf16(kv.y, qp[hh][1], 0.f);
                p0 = dot2_bf16(kv.z, qp[hh][2], p0);
                p1 = dot2_bf16(kv.w, qp[hh][3], p1);
                const float sc_ = ok ? group_sum<LPR>(p0 + p1) * scale : -INFINITY;  // a masked key never raises the maximum
                if constexpr (PM) {
                    const float p = ok ? expf(sc_ - gmax[hh]) : 0.f;
                    const float pr = rbf(p);  // the probability as the reference's P.V product sees it
                    lrun[hh] += p;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[hh][e] = fmaf(pr, vf[e], acc[hh][e]);
                    continue;
                }
                const float mn = fmaxf(mrun[hh], sc_);
                const float corr = (mn == -INFINITY) ? 1.f : __expf(mrun[hh] - mn);
                const float p = (mn == -INFINITY) ? 0.f : __expf(sc_ - mn);
                lrun[hh] = lrun[hh] * corr + p;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[hh][e] = acc[hh][e] * corr + p * vf[e];
                mrun[hh] = mn;
            }
        };
        // two register sets in ping-pong; every load is unconditional (rows past the range are clamped and masked in
        // step()): loads behind run-time conditions cost the compiler its vmcnt bookkeeping
        uint4 k0 = kv_cur, v0 = vv_cur;
        for (int s0 = s_first; s0 < s_end; s0 += 2 * STRIDE) {
            const int sn1 = min(s0 + STRIDE + sub, s_last);
            const uint4 k1 = kc[(int64_t)sn1 * LPR + dl], v1 = vc[(int64_t)sn1 * LPR + dl];
            step(k0, v0, s0 + sub);
            const int sn2 = min(s0 + 2 * STRIDE + sub, s_last);
            k0 = kc[(int64_t)sn2 * LPR + dl];
            v0 = vc[(int64_t)sn2 * LPR + dl];
            if (s0 + STRIDE < s_end) step(k1, v1, s0 + STRIDE + sub);  // wave-uniform
        }
        attn_stamp(dbg, 2);
        // merge the RPW row slots of this wave: first the slots' common maximum, then every lane rescales its own state ONCE
        // and the rest is plain sums (the pairwise merge re-evaluated two exponentials per step and value); one state per
        // wave goes to LDS
#pragma unroll
        for (int hh = 0; hh < HQ; ++hh) {
            const float mw = slot_allreduce<LPR, true>(mrun[hh]);
            const float c = (mrun[hh] == -INFINITY) ? 0.f : __expf(mrun[hh] - mw);
            lrun[hh] = slot_allreduce<LPR, false>(lrun[hh] * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[hh][e] = slot_allreduce<LPR, false>(acc[hh][e] * c);
            mrun[hh] = mw;
            if (sub == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sh_acc[hh][wave][dl * 8 + e] = acc[hh][e];
                if (dl == 0) {
                    sh_m[hh][wave] = mrun[hh];
                    sh_l[hh][wave] = lrun[hh];
                }
            }
        }
        __syncthreads();
        attn_stamp(dbg, 3);
        for (int idx = threadIdx.x; idx < HQ * HS; idx += WAVES * 64) {
            const int hh = idx / HS, d = idx % HS;
            if (h0 + hh < q_per_kv) {
                constexpr int NSLOT = WAVES;
                float mx = -INFINITY;
                for (int t = 0; t < NSLOT; ++t) mx = fmaxf(mx, sh_m[hh][t]);
                float l = 0.f, a = 0.f;
                for (int t = 0; t < NSLOT; ++t) {
                    const float mt = sh_m[hh][t];
                    const float wgt = (mt == -INFINITY) ? 0.f : __expf(mt - mx);
                    l += sh_l[hh][t] * wgt;
                    a += sh_acc[hh][t][d] * wgt;
                }
                const int head = g * q_per_kv + h0 + hh;
                if (nsplit == 1) {
                    y[(int64_t)head * HS + d] = f2bf(a / l);
                } else {
                    float* p = ws + ((int64_t)head * nsplit + split) * (HS + 2);
                    store_agent(p + d, a);
                    if (d == 0) {
                        store_agent(p + HS, mx);
                        store_agent(p + HS + 1, l);
                    }
                }
            }
        }
    }
    attn_stamp(dbg, 4);
    if (nsplit == 1) return;
    // ---- arrival ticket: the last workgroup of this group merges the splits
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = __hip_atomic_fetch_add(&tickets[g * gridDim.z + blockIdx.z], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_last = (t == (unsigned int)(nsplit - 1));
        if (sh_last) __hip_atomic_store(&tickets[g * gridDim.z + blockIdx.z], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
    }
    __syncthreads();
    if (!sh_last) return;
    for (int idx = threadIdx.x; idx < HQ * HS; idx += WAVES * 64) {
        const int hq = h0 + idx / HS, d = idx % HS;
        if (hq >= q_per_kv) continue;
        const int head = g * q_per_kv + hq;
        const float* p = ws + (int64_t)head * nsplit * (HS + 2);
        float mx = -INFINITY;
        for (int t = 0; t < nsplit; ++t) mx = fmaxf(mx, load_agent(p + t * (HS + 2) + HS));
        float l = 0.f, a = 0.f;
        for (int t = 0; t < nsplit; ++t) {
            const float mt = load_agent(p + t * (HS + 2) + HS);
            const float wgt = (mt == -INFINITY) ? 0.f : __expf(mt - mx);
            l += load_agent(p + t * (HS + 2) + HS + 1) * wgt;
            a += load_agent(p + t * (HS + 2) + d) * wgt;
        }
        y[(int64_t)head * HS + d] = f2bf(a / l);
    }
}

template <int HS>
static int attn_fused_launch(const void* qkv, const void* cosp, const void* sinp, int n_elem, const int32_t* pos,
                             void* k_cache, void* v_cache, int n_groups, int q_per_kv, int S, int nsplit, void* ws,
                             void* tickets, void* y, int softmax_mode, hipStream_t st) {
    const int hq = q_per_kv == 1 ? 1 : (q_per_kv == 2 ? 2 : 4);
    // 16 waves when a split holds more keys than 4 waves cover in two steps
    const int per = (S + nsplit - 1) / nsplit;
    const bool wide = per > 2 * kAttnWaves * (64 / (HS / 8));
    const dim3 grid(n_groups, nsplit, (q_per_kv + hq - 1) / hq), block((wide ? 16 : kAttnWaves) * 64);
#define PARROT_FUSED_GO(HQV, WV)                                                                                            \
    do {                                                                                                                    \
        if (softmax_mode == 1)                                                                                              \
            return launch(K_ATTN_FUSED, attn_fused_decode_kernel<HS, HQV, 16, true>, grid, dim3(16 * 64), 0, st, (const bf16_t*)qkv, \
                          (const __half*)cosp, (const __half*)sinp, n_elem, pos, (bf16_t*)k_cache, (bf16_t*)v_cache, n_groups, \
                          q_per_kv, S, nsplit, (float*)ws, (unsigned int*)tickets, (bf16_t*)y, g_attn_dbg_host);             \
        return launch(K_ATTN_FUSED, attn_fused_decode_kernel<HS, HQV, WV>, grid, block, 0, st, (const bf16_t*)qkv,          \
                      (const __half*)cosp, (const __half*)sinp, n_elem, pos, (bf16_t*)k_cache, (bf16_t*)v_cache, n_groups,   \
                      q_per_kv, S, nsplit, (float*)ws, (unsigned int*)tickets, (bf16_t*)y, g_attn_dbg_host);                 \
    } while (0)
    if (wide) {
        if (q_per_kv == 1) PARROT_FUSED_GO(1, 16);
        if (q_per_kv == 2) PARROT_FUSED_GO(2, 16);
        PARROT_FUSED_GO(4, 16);
    }
    if (q_per_kv == 1) PARROT_FUSED_GO(1, kAttnWaves);
    if (q_per_kv == 2) PARROT_FUSED_GO(2, kAttnWaves);
    PARROT_FUSED_GO(4, kAttnWaves);
#undef PARROT_FUSED_GO
}

template <int HS>
static int attn_launch(const void* q, int M, const int32_t* pos, const void* k_cache, const void* v_cache, int n_groups,
                       int q_per_kv, int S, int nsplit, void* ws, void* y, int ldy, int softmax_mode, hipStream_t st) {
    const dim3 grid(n_groups, nsplit, M), block(kAttnWaves * 64);
    int rc;
#define PARROT_ATTN_GO(HQV)                                                                                          \
    rc = softmax_mode == 1                                                                                            \
        ? launch(K_ATTN_DECODE, attn_decode_kernel<HS, HQV, true>, grid, block, 0, st, (const bf16_t*)q, pos,         \
                 (const bf16_t*)k_cache, (const bf16_t*)v_cache, n_groups, q_per_kv, S, nsplit, (float*)ws, (bf16_t*)y, ldy) \
        : launch(K_ATTN_DECODE, attn_decode_kernel<HS, HQV>, grid, block, 0, st, (const bf16_t*)q, pos,                \
                (const bf16_t*)k_cache, (const bf16_t*)v_cache, n_groups, q_per_kv, S, nsplit, (float*)ws, (bf16_t*)y, ldy)
    if (q_per_kv == 1) {
        PARROT_ATTN_GO(1);
    } else if (q_per_kv == 2) {
        PARROT_ATTN_GO(2);
    } else {
        PARROT_ATTN_GO(4);
    }
#undef PARROT_ATTN_GO
    if (rc != PARROT_OK || nsplit == 1) return rc;
    return launch(K_ATTN_COMBINE, attn_combine_kernel<HS>, dim3(n_groups * q_per_kv, M), dim3(HS), 0, st, (const float*)ws,
                  nsplit, (bf16_t*)y, ldy, n_groups * q_per_kv);
}

}  // namespace parrot

using namespace parrot;

extern "C" {

int parrot_qkv_rope_kvappend(const void* qkv, int ldqkv, int M, const void* rope_cos, const void* rope_sin, int n_elem,
                             int rope_local, const int32_t* pos, int n_groups, int q_per_kv, int hs, int S, void* q_out, void* k_cache,
                             void* v_cache, void* stream) {
    PARROT_REQUIRE(qkv && pos && q_out && k_cache && v_cache, "qkv_rope_kvappend: null pointer");
    PARROT_REQUIRE(M >= 1 && n_groups >= 1 && q_per_kv >= 1 && hs >= 2 && S >= 1, "qkv_rope_kvappend: bad shape");
    PARROT_REQUIRE(hs % 2 == 0 && n_elem % 2 == 0 && n_elem >= 0 && n_elem <= hs,
                   "qkv_rope_kvappend: hs=%d and n_elem=%d must be even, n_elem <= hs", hs, n_elem);
    PARROT_REQUIRE(n_elem == 0 || (rope_cos && rope_sin), "qkv_rope_kvappend: rope tables missing");
    PARROT_REQUIRE(ldqkv >= n_groups * (q_per_kv + 2) * hs, "qkv_rope_kvappend: ldqkv too small");
    PARROT_REQUIRE(M <= S, "qkv_rope_kvappend: M=%d rows do not fit a cache of %d slots", M, S);
    const bool tables_ok = n_elem == 0 || (aligned16(rope_cos) && aligned16(rope_sin));
    if (hs % 8 == 0 && n_elem % 16 == 0 && ldqkv % 8 == 0 && aligned16(qkv) && aligned16(q_out) && aligned16(k_cache) && aligned16(v_cache) &&
        tables_ok && M <= 65535) {
        const unsigned per_row = (unsigned)(n_groups * (q_per_kv + 2) * (hs / 8));
        return launch(K_ROPE_KVAPPEND, rope_kvappend8_kernel, dim3((per_row + 255) / 256, (unsigned)M), dim3(256), 0, (hipStream_t)stream,
                      (const bf16_t*)qkv, ldqkv, M, (const __half*)rope_cos, (const __half*)rope_sin, n_elem, rope_local, pos, n_groups, q_per_kv,
                      hs, S, (bf16_t*)q_out, (bf16_t*)k_cache, (bf16_t*)v_cache);
    }
    const int64_t total = (int64_t)M * n_groups * (q_per_kv + 2) * (hs / 2);
    return launch(K_ROPE_KVAPPEND, rope_kvappend_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                  (hipStream_t)stream, (const bf16_t*)qkv, ldqkv, M, (const __half*)rope_cos, (const __half*)rope_sin,
                  n_elem, rope_local, pos, n_groups, q_per_kv, hs, S, (bf16_t*)q_out, (bf16_t*)k_cache, (bf16_t*)v_cache);
}

#ifdef PARROT_DIAG
int parrot_tune_attn_stamps(void* dbg8_u64) {  // diagnostic build: device buffer of 8 uint64, or NULL to switch off
    g_attn_dbg_host = (unsigned long long*)dbg8_u64;
    return PARROT_OK;
}
#endif

// softmax_mode 1 needs the whole window in one key block of the reference's kernel: one split, at most 512 slots
static int check_softmax_mode(const char* who, int softmax_mode, int S, int nsplit) {
    PARROT_REQUIRE(softmax_mode == 0 || softmax_mode == 1, "%s: softmax_mode must be 0 or 1", who);
    PARROT_UNSUPPORTED(softmax_mode == 0 || (nsplit == 1 && S <= 512),
                       "%s: softmax_mode 1 (the reference's bf16 probabilities) is built for one split and windows up to 512 slots (S=%d nsplit=%d)", who, S, nsplit);
    return PARROT_OK;
}

int parrot_attn_fused_decode(const void* qkv, const void* rope_cos, const void* rope_sin, int n_elem, const int32_t* pos,
                             int n_groups, int q_per_kv, int hs, int S, int nsplit, void* workspace, void* tickets,
                             void* k_cache, void* v_cache, void* y, int softmax_mode, void* stream) {
    {
        const int rc = check_softmax_mode("attn_fused_decode", softmax_mode, S, nsplit);
        if (rc != PARROT_OK) return rc;
    }
    PARROT_REQUIRE(qkv && pos && k_cache && v_cache && y, "attn_fused_decode: null pointer");
    PARROT_REQUIRE(n_groups >= 1 && q_per_kv >= 1 && S >= 1, "attn_fused_decode: bad shape");
    PARROT_UNSUPPORTED(q_per_kv <= kFusedMaxQ, "attn_fused_decode: at most %d query heads per group (got %d)", kFusedMaxQ, q_per_kv);
    PARROT_REQUIRE(n_elem % 2 == 0 && n_elem >= 0 && n_elem <= hs, "attn_fused_decode: bad n_elem=%d", n_elem);
    PARROT_REQUIRE(n_elem == 0 || (rope_cos && rope_sin), "attn_fused_decode: rope tables missing");
    PARROT_REQUIRE(nsplit >= 1 && nsplit <= 65535, "attn_fused_decode: nsplit out of range");
    PARROT_REQUIRE(nsplit == 1 || (workspace && tickets), "attn_fused_decode: workspace and tickets required when nsplit > 1");
    PARROT_REQUIRE(aligned16(qkv) && aligned16(k_cache) && aligned16(v_cache), "attn_fused_decode: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    switch (hs) {
        case 32: return attn_fused_launch<32>(qkv, rope_cos, rope_sin, n_elem, pos, k_cache, v_cache, n_groups, q_per_kv, S, nsplit, workspace, tickets, y, softmax_mode, st);
        case 64: return attn_fused_launch<64>(qkv, rope_cos, rope_sin, n_elem, pos, k_cache, v_cache, n_groups, q_per_kv, S, nsplit, workspace, tickets, y, softmax_mode, st);
        case 128: return attn_fused_launch<128>(qkv, rope_cos, rope_sin, n_elem, pos, k_cache, v_cache, n_groups, q_per_kv, S, nsplit, workspace, tickets, y, softmax_mode, st);
        default: break;
    }
    set_error("attn_fused_decode: head size %d not built (32, 64, 128)", hs);
    return PARROT_EUNSUPPORTED;
}

int64_t parrot_attn_workspace_floats(int M, int n_head, int hs, int nsplit) {
    return (int64_t)M * n_head * nsplit * (hs + 2);
}

int parrot_attn_decode(const void* q, int M, const int32_t* pos, const void* k_cache, const void* v_cache, int n_groups,
                       int q_per_kv, int hs, int S, int nsplit, void* workspace, void* y, int ldy, int softmax_mode, void* stream) {
    {
        const int rc = check_softmax_mode("attn_decode", softmax_mode, S, nsplit);
        if (rc != PARROT_OK) return rc;
    }
    PARROT_REQUIRE(q && pos && k_cache && v_cache && y, "attn_decode: null pointer");
    PARROT_REQUIRE(M >= 1 && n_groups >= 1 && q_per_kv >= 1 && S >= 1, "attn_decode: bad shape");
    PARROT_REQUIRE(nsplit >= 1 && nsplit <= 65535 && M <= 65535, "attn_decode: nsplit/M out of range");
    PARROT_REQUIRE(nsplit == 1 || workspace, "attn_decode: workspace required when nsplit > 1");
    PARROT_REQUIRE(ldy >= n_groups * q_per_kv * hs, "attn_decode: ldy too small");
    PARROT_REQUIRE(aligned16(q) && aligned16(k_cache) && aligned16(v_cache), "attn_decode: q/k/v must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    switch (hs) {
        case 32: return attn_launch<32>(q, M, pos, k_cache, v_cache, n_groups, q_per_kv, S, nsplit, workspace, y, ldy, softmax_mode, st);
        case 64: return attn_launch<64>(q, M, pos, k_cache, v_cache, n_groups, q_per_kv, S, nsplit, workspace, y, ldy, softmax_mode, st);
        case 128: return attn_launch<128>(q, M, pos, k_cache, v_cache, n_groups, q_per_kv, S, nsplit, workspace, y, ldy, softmax_mode, st);
        default: break;
    }
    set_error("attn_decode: head size %d not built (32, 64, 128)", hs);
    return PARROT_EUNSUPPORTED;
}

}  // extern "C"
