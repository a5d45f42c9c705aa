// Causal attention over the T prompt rows on the matrix cores (reference lit_gpt/model.py:256-275 scaled_dot_product_attention
// with the tril mask of :126-128, for the rows of one prefill call; q is already split + roped, K / V are in the cache).
//
// One wave owns 32 consecutive query rows of one head and walks the key blocks of 32 up to its diagonal (flash attention:
// running max / sum, no T x T matrix).  Everything is arranged so that a LANE belongs to ONE query:
//   S^T = K_blk . Q^T        v_mfma_f32_32x32x16_bf16 with A = K rows (keys x dims, as they lie in the cache), B = the wave's Q rows:
//                            lane l holds the scores of query l & 31 for 16 of the block's 32 keys (the other 16 sit in lane l ^ 32);
//   softmax                  max / sum over the lane's 16 registers + one exchange with lane l ^ 32; the rescale factor is per lane;
//   O^T += V^T_blk . P^T     A = V^T (dims x keys), B = P^T straight from the score registers (rounded to bf16): the MFMA's K index
//                            may be any permutation as long as A and B agree, so registers 8 ks .. 8 ks + 7 of a lane ARE its B
//                            fragment of step ks, and A takes the matching keys: two runs of four consecutive keys per lane.
// V^T needs the keys contiguous per dim: a small pre-pass transposes the call's V rows into a scratch [group][dim][slot]
// (zero beyond the last key, so that masked columns multiply finite numbers).  Requires *pos + M <= S (no ring wrap inside the
// call): the prompt of generate().  P is rounded to bf16 before P.V (torch's flash kernels do the same; the decode kernels keep
// P in fp32): results agree with the multi-row decode path to bf16 rounding.
#include <stdlib.h>

#include <type_traits>

#include "parrot_common.h"

namespace parrot {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// vT[g][d][s] = V[g][s][d] for s < n_keys, 0 for n_keys <= s < Spad.  One thread per (g, s, 8 dims).
__global__ void __launch_bounds__(256)
attn_vt_kernel(const bf16_t* __restrict__ v_cache, const int32_t* __restrict__ pos_ptr, int M, int n_groups, int hs, int S, int Spad,
               bf16_t* __restrict__ vT) {
    const int n_keys = min(pos_ptr[0] + M, S);
    const int n_fill = min((n_keys + 31) & ~31, Spad);
    const int d8 = hs >> 3;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = (int)(t % Spad);
    const int c = (int)((t / Spad) % d8);
    const int g = (int)(t / ((int64_t)Spad * d8));
    if (g >= n_groups || s >= n_fill) return;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (s < n_keys) v = *reinterpret_cast<const uint4*>(v_cache + ((int64_t)g * S + s) * hs + c * 8);
    const uint32_t dw[4] = {v.x, v.y, v.z, v.w};
    bf16_t* dst = vT + ((int64_t)g * hs + c * 8) * Spad + s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dst[(int64_t)(2 * i) * Spad] = (bf16_t)(dw[i] & 0xffffu);
        dst[(int64_t)(2 * i + 1) * Spad] = (bf16_t)(dw[i] >> 16);
    }
}

template <int HS>
__global__ void __launch_bounds__(256)
attn_prefill_kernel(const bf16_t* __restrict__ q, int ldq, int M, const int32_t* __restrict__ pos_ptr, const bf16_t* __restrict__ k_cache,
                    const bf16_t* __restrict__ vT, int n_groups, int q_per_kv, int S, int Spad, bf16_t* __restrict__ y, int ldy) {
    constexpr int KS = HS / 16;  // MFMA k-steps of Q.K^T
    constexpr int DT = HS / 32;  // 32-dim tiles of the output
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, g = h / q_per_kv;
    // heavy (late) query blocks first: the grid's x index counts down
    const int qb = ((int)gridDim.x - 1 - (int)blockIdx.x) * 4 + wave;
    const int q0 = qb * 32;
    if (q0 >= M) return;  // (no barriers in this kernel)
    const int pos0 = pos_ptr[0];
    const int qrow = min(q0 + lr, M - 1);
    const int qpos = pos0 + q0 + lr;  // this lane's query position (rows past M are computed and dropped)
    const float scale = 1.0f / sqrtf((float)HS);

    bf16x8_t qf[KS];
    const bf16_t* qp = q + (int64_t)qrow * ldq + (int64_t)h * HS + lh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(qp + ks * 16));

    f32x16_t o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const bf16_t* kc = k_cache + (int64_t)g * S * HS;
    const bf16_t* vg = vT + (int64_t)g * HS * Spad;
    const int last_pos = pos0 + min(q0 + 31, M - 1);
    const int n_kb = last_pos / 32 + 1;
    for (int kb = 0; kb < n_kb; ++kb) {
        // ---- scores of 32 keys x 32 queries
        f32x16_t sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        const bf16_t* kp = kc + (int64_t)min(kb * 32 + lr, S - 1) * HS + lh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t kf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(kp + ks * 16));
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc, 0, 0, 0);
        }
        // ---- causal mask + online softmax (this lane: query lr, keys kb*32 + (r&3) + 8 (r>>2) + 4 lh)
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float s = key <= qpos ? sacc[r] * scale : -INFINITY;
            sacc[r] = s;
            mx = fmaxf(mx, s);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);  // finite: key 0 is visible to every query
        const float alpha = __expf(m_run - m_new);
        float lsum = 0.f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = __expf(sacc[r] - m_new);
            lsum += p[r];
        }
        lsum += __shfl_xor(lsum, 32);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        // ---- O^T += V^T . P^T
        bf16x8_t pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint32_t w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f2bf(p[8 * ks + 2 * i]) | ((uint32_t)f2bf(p[8 * ks + 2 * i + 1]) << 16);
            pf[ks] = __builtin_bit_cast(bf16x8_t, make_uint4(w[0], w[1], w[2], w[3]));
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16_t* vp = vg + (int64_t)(dt * 32 + lr) * Spad + kb * 32 + 4 * lh;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint2 v0 = *reinterpret_cast<const uint2*>(vp + 16 * ks);      // keys 16 ks + 4 lh + 0..3
                const uint2 v1 = *reinterpret_cast<const uint2*>(vp + 16 * ks + 8);  // keys 16 ks + 8 + 4 lh + 0..3
                const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, make_uint4(v0.x, v0.y, v1.x, v1.y));
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ks], o[dt], 0, 0, 0);
            }
        }
    }
    // ---- this lane: query lr, dims dt*32 + (r&3) + 8 (r>>2) + 4 lh
    if (q0 + lr < M) {
        const float inv = 1.0f / l_run;
        bf16_t* yp = y + (int64_t)(q0 + lr) * ldy + (int64_t)h * HS + 4 * lh;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t w0 = (uint32_t)f2bf(o[dt][4 * c] * inv) | ((uint32_t)f2bf(o[dt][4 * c + 1] * inv) << 16);
                const uint32_t w1 = (uint32_t)f2bf(o[dt][4 * c + 2] * inv) | ((uint32_t)f2bf(o[dt][4 * c + 3] * inv) << 16);
                *reinterpret_cast<uint2*>(yp + dt * 32 + 8 * c) = make_uint2(w0, w1);
            }
    }
}

// The same computation with the key / value blocks shared by the four waves of a workgroup through LDS (four consecutive query
// blocks of one head): each wave of the kernel above streams its own copy of K and V^T from L2 - 16 KB per 16 MFMAs, four times
// what a CU's vector-memory path sustains - here a block is fetched once per workgroup.  Rows are padded so that the fragment
// reads are conflict-free: K rows by 16 B (a ds_read_b128 lane group reads 16 consecutive-ish rows: slot = row mod 16), V^T rows
// (32 keys = 64 B) to 72 B (a ds_read_b64 half-wave reads 32 dims: 18 * dim mod 64 are 32 distinct even banks).
// The next block travels global -> registers while the current one is multiplied, and is written behind a barrier.
template <int HS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
attn_prefill_lds_kernel(const bf16_t* __restrict__ q, int ldq, int M, const int32_t* __restrict__ pos_ptr, const bf16_t* __restrict__ k_cache,
                        const bf16_t* __restrict__ vT, int n_groups, int q_per_kv, int S, int Spad, bf16_t* __restrict__ y, int ldy) {
    // An LDS image holds TWO key blocks (64 keys): one pair of barriers and one exposed global -> LDS hand-over per 64 keys
    // instead of per 32 (the first version staged 32 keys at a time and ran 2.1 us per key block on StableLM-3B's 512-row prompt).
    constexpr int KS = HS / 16, DT = HS / 32;
    constexpr int KLD = HS + 8;   // K row stride in elements (16-byte pad)
    constexpr int VLD = 68;       // V^T row stride in elements: 64 keys + 8 B (34 dwords: 32 dims land on 32 distinct even banks)
    constexpr int KPT = HS / 32;  // 16-byte pieces of the K image per thread (64 rows x HS*2 B / 256 threads / 16 B)
    constexpr int VPT = HS / 32;  // 16-byte pieces of the V^T image per thread (HS rows x 128 B / 256 threads / 16 B)
    static_assert(HS == 64 || HS == 128, "the LDS kernel is built for head sizes 64 and 128");
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * KLD];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[HS * VLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, g = h / q_per_kv;
    const int wg = (int)gridDim.x - 1 - (int)blockIdx.x;  // late (long) query blocks first
    const int q0 = (wg * 4 + wave) * 32;
    const bool live = q0 < M;
    const int pos0 = pos_ptr[0];
    const int qrow = min(q0 + lr, M - 1);
    const int qpos = pos0 + q0 + lr;
    const float scale = 1.0f / sqrtf((float)HS);

    bf16x8_t qf[KS];
    const bf16_t* qp = q + (int64_t)qrow * ldq + (int64_t)h * HS + lh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(qp + ks * 16));
    f32x16_t o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const bf16_t* kc = k_cache + (int64_t)g * S * HS;
    const bf16_t* vg = vT + (int64_t)g * HS * Spad;
    const int my_kb = live ? (pos0 + min(q0 + 31, M - 1)) / 32 + 1 : 0;          // key blocks (of 32) this wave needs
    const int wg_last_q = min(wg * 128 + 127, M - 1);
    const int n_kb = (pos0 + wg_last_q) / 32 + 1;                                   // ... and the workgroup (wave-uniform)
    const int n_img = (n_kb + 1) / 2;                                               // LDS images of two key blocks
    // staging assignment: K piece p of thread t: row (t * KPT + p) / (HS / 8), 16-byte column (t * KPT + p) % (HS / 8);
    //                     V^T piece: dim (t * VPT + p) / 8, 16-byte column (t * VPT + p) % 8 (8 keys)
    // (named registers: as arrays filled in one lambda and read in another the staged pieces went through scratch memory)
    uint4 rk0, rk1, rk2 = make_uint4(0, 0, 0, 0), rk3 = rk2, rv0, rv1, rv2 = rk2, rv3 = rk2;
    auto kload = [&](int im, int p) {
        const int idx = tid * KPT + p, row = idx / (HS / 8), c = idx % (HS / 8);
        return *reinterpret_cast<const uint4*>(kc + (int64_t)min(im * 64 + row, S - 1) * HS + c * 8);
    };
    auto vload = [&](int im, int p) {
        const int idx = tid * VPT + p, dim = idx / 8, c = idx % 8;
        return *reinterpret_cast<const uint4*>(vg + (int64_t)dim * Spad + im * 64 + c * 8);  // (Spad is a multiple of 64)
    };
    auto kstore = [&](int p, const uint4 v) {
        const int idx = tid * KPT + p, row = idx / (HS / 8), c = idx % (HS / 8);
        *reinterpret_cast<uint4*>(&Ks[row * KLD + c * 8]) = v;
    };
    auto vstore = [&](int p, const uint4 v) {
        const int idx = tid * VPT + p, dim = idx / 8, c = idx % 8;
        uint2* dst = reinterpret_cast<uint2*>(&Vs[dim * VLD + c * 8]);  // 136-byte rows: 8-byte aligned
        dst[0] = make_uint2(v.x, v.y);
        dst[1] = make_uint2(v.z, v.w);
    };
    auto fetch = [&](int im) {
        rk0 = kload(im, 0), rk1 = kload(im, 1);
        rv0 = vload(im, 0), rv1 = vload(im, 1);
        if constexpr (KPT == 4) {
            rk2 = kload(im, 2), rk3 = kload(im, 3);
            rv2 = vload(im, 2), rv3 = vload(im, 3);
        }
    };
    auto stage = [&]() {
        kstore(0, rk0), kstore(1, rk1);
        vstore(0, rv0), vstore(1, rv1);
        if constexpr (KPT == 4) {
            kstore(2, rk2), kstore(3, rk3);
            vstore(2, rv2), vstore(3, rv3);
        }
    };
    fetch(0);
    stage();
    __syncthreads();
    // One softmax step over NB key blocks of the image (NB = 2: both; 1: only the first - the second lies past this wave's diagonal,
    // and past the zero-filled part of V^T).  Two blocks per step: their S accumulators are independent (no MFMA waits on the one in
    // front of it), one rescale of the running output per 64 keys, 16 P.V MFMAs in four independent chains.
    auto process = [&](auto nb_tag, int kb0) {
        constexpr int NB = decltype(nb_tag)::value;
        f32x16_t sacc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[b][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const bf16x8_t kf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(&Ks[(b * 32 + lr) * KLD + ks * 16 + lh * 8]));
                sacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[b], 0, 0, 0);
            }
        float mx = -INFINITY;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = (kb0 + b) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float sv = key <= qpos ? sacc[b][r] * scale : -INFINITY;
                sacc[b][r] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float lsum = 0.f, p[NB][16];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[b][r] = __expf(sacc[b][r] - m_new);
                lsum += p[b][r];
            }
        lsum += __shfl_xor(lsum, 32);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        bf16x8_t pf[NB][2];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint32_t w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f2bf(p[b][8 * ks + 2 * i]) | ((uint32_t)f2bf(p[b][8 * ks + 2 * i + 1]) << 16);
                pf[b][ks] = __builtin_bit_cast(bf16x8_t, make_uint4(w[0], w[1], w[2], w[3]));
            }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16_t* vp = &Vs[(dt * 32 + lr) * VLD + b * 32 + 16 * ks + 4 * lh];
                    const uint2 v0 = *reinterpret_cast<const uint2*>(vp);
                    const uint2 v1 = *reinterpret_cast<const uint2*>(vp + 8);
                    const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, make_uint4(v0.x, v0.y, v1.x, v1.y));
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[b][ks], o[dt], 0, 0, 0);
                }
    };
    for (int im = 0; im < n_img; ++im) {
        if (im + 1 < n_img) fetch(im + 1);
        if (2 * im + 1 < my_kb)
            process(std::integral_constant<int, 2>{}, 2 * im);
        else if (2 * im < my_kb)
            process(std::integral_constant<int, 1>{}, 2 * im);
        __syncthreads();  // everybody is done with this LDS image
        if (im + 1 < n_img) {
            stage();
            __syncthreads();
        }
    }
    if (live && q0 + lr < M) {
        const float inv = 1.0f / l_run;
        bf16_t* yp = y + (int64_t)(q0 + lr) * ldy + (int64_t)h * HS + 4 * lh;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t w0 = (uint32_t)f2bf(o[dt][4 * c] * inv) | ((uint32_t)f2bf(o[dt][4 * c + 1] * inv) << 16);
                const uint32_t w1 = (uint32_t)f2bf(o[dt][4 * c + 2] * inv) | ((uint32_t)f2bf(o[dt][4 * c + 3] * inv) << 16);
                *reinterpret_cast<uint2*>(yp + dt * 32 + 8 * c) = make_uint2(w0, w1);
            }
    }
}

}  // namespace parrot

using namespace parrot;

extern "C" {

int64_t parrot_attn_prefill_scratch_elems(int n_groups, int hs, int S) { return (int64_t)n_groups * hs * ((S + 63) / 64 * 64); }

int parrot_attn_prefill(const void* q, int M, const int32_t* pos, const void* k_cache, const void* v_cache, void* vT_scratch, int n_groups,
                        int q_per_kv, int hs, int S, void* y, int ldy, void* stream) {
    PARROT_REQUIRE(q && pos && k_cache && v_cache && vT_scratch && y, "attn_prefill: null pointer");
    PARROT_REQUIRE(M >= 1 && n_groups >= 1 && q_per_kv >= 1 && S >= 1 && M <= S, "attn_prefill: bad shape (M=%d S=%d)", M, S);
    PARROT_REQUIRE(ldy >= n_groups * q_per_kv * hs && ldy % 4 == 0, "attn_prefill: ldy too small or not a multiple of 4");
    PARROT_REQUIRE(aligned16(q) && aligned16(k_cache) && aligned16(v_cache) && aligned16(vT_scratch) && ((uintptr_t)y % 8 == 0),
                   "attn_prefill: q / caches / scratch must be 16-byte aligned, y 8-byte aligned");
    PARROT_UNSUPPORTED(hs == 32 || hs == 64 || hs == 128, "attn_prefill: head size %d not built (32, 64, 128)", hs);
    hipStream_t st = (hipStream_t)stream;
    const int Spad = (S + 63) / 64 * 64;
    const int64_t nthreads = (int64_t)n_groups * (hs / 8) * Spad;
    int rc = launch(K_ATTN_PREFILL_VT, attn_vt_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, (const bf16_t*)v_cache, pos, M,
                    n_groups, hs, S, Spad, (bf16_t*)vT_scratch);
    if (rc != PARROT_OK) return rc;
    const int n_head = n_groups * q_per_kv, ldq = n_head * hs;
    const dim3 grid((M + 127) / 128, n_head);
    const int use_lds = tune_env("PARROT_ATTN_PREFILL_LDS", 1);  // PARROT_ATTN_PREFILL_LDS=0: every wave streams its own K / V^T (A/B)
#define PARROT_PF_LDS_GO(HSV)                                                                                                              \
    return launch(K_ATTN_PREFILL, attn_prefill_lds_kernel<HSV>, grid, dim3(256), 0, st, (const bf16_t*)q, ldq, M, pos, (const bf16_t*)k_cache, \
                  (const bf16_t*)vT_scratch, n_groups, q_per_kv, S, Spad, (bf16_t*)y, ldy)
    if (use_lds && M > 32) {  // (a single query block has nothing to share)
        if (hs == 64) PARROT_PF_LDS_GO(64);
        if (hs == 128) PARROT_PF_LDS_GO(128);
    }
#undef PARROT_PF_LDS_GO
#define PARROT_PF_GO(HSV)                                                                                                              \
    return launch(K_ATTN_PREFILL, attn_prefill_kernel<HSV>, grid, dim3(256), 0, st, (const bf16_t*)q, ldq, M, pos, (const bf16_t*)k_cache, \
                  (const bf16_t*)vT_scratch, n_groups, q_per_kv, S, Spad, (bf16_t*)y, ldy)
    if (hs == 32) PARROT_PF_GO(32);
    if (hs == 64) PARROT_PF_GO(64);
    PARROT_PF_GO(128);
#undef PARROT_PF_GO
}

}  // extern "C"
