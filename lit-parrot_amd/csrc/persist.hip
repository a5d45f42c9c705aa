// Persistent decode step: ONE launch per token.
//
// The multi-launch step (w4.hip + attn.hip) spends ~3-4.5 us of fixed cost per kernel (dispatch, one HBM round trip,
// reduction tail) 161 times per token, with the HBM idle at every boundary.  Here the whole token is a static *program*
// of ops interpreted by 256 resident workgroups (one per CU, 12 waves each):
//
//   * every (workgroup, wave) owns fixed output rows of every Linear (row groups of 8, dealt cyclically), so its weight
//     addresses are known ahead of time;
//   * ops are separated by an arrival-counter grid barrier (8 counters, one per `blockIdx % 8` shard, on separate
//     128-B lines; polled by 8 lanes of one wave with agent-scope loads);
//   * hand-off data (activation vectors, attention partials) is written with write-through agent-scope stores and read
//     with agent-scope loads (guide: "sc1 payload -> every storing wave waits vmcnt -> barrier -> one relaxed agent
//     atomic"; consumers load sc1) - no fences, no L2 write-back;
//   * the weights of op k+1 are requested right AFTER the result stores of op k and BEFORE its arrival: the counted
//     `s_waitcnt vmcnt(N)` that drains the stores leaves those N loads in flight, so the HBM stream of the next op runs
//     underneath the barrier, the activation broadcast and the norm prologue.
//
// Numerics are those of the multi-launch kernels (same per-lane slices, same DPP/LDS reduction order, same epilogues).
// Every spin is bounded: on a timeout the error word is set, every later wait falls through and the launch ends.
#include <hip/hip_fp16.h>

#include "parrot_common.h"
#include "w4_plan.h"

namespace parrot {

constexpr int PK_WAVES = 12;
constexpr int PK_THREADS = PK_WAVES * 64;
constexpr int PK_WGS = 256;
constexpr int PK_RU = 8;          // rows per group = rows in flight per wave
constexpr int PK_SHARDS = 8;      // arrival counter shards
constexpr int PK_SHARD_STRIDE = 32;  // uint32 per shard line (128 B)
constexpr unsigned PK_MAX_SPINS = 400000;

typedef parrot_pk_op_t PkOp;
typedef parrot_pk_state_t PkState;

// ---- agent-scope (write-through / L1-bypassing) accessors for data that other workgroups produce or consume
// (ld_agent64 / ld_agent32 live in parrot_common.h)
__device__ __forceinline__ float ld_agentf(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent32(void* p, uint32_t v) {
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agentf(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// apply_epilogue of parrot_common.h with the residual passed by value
__device__ __forceinline__ bf16_t pk_epilogue(int epi, float acc, float acc2, const bf16_t* bias, bf16_t res, int col) {
    float v = acc;
    if (bias != nullptr) v += bf2f(bias[col]);
    v = rbf(v);
    if (epi == PARROT_EPI_RESIDUAL) {
        v = bf2f(res) + v;
    } else if (epi == PARROT_EPI_GELU) {
        v = gelu_erf(v);
    } else if (epi == PARROT_EPI_SWIGLU) {
        v = rbf(silu(v)) * rbf(acc2);
    }
    return f2bf(v);
}

// diagnostic stamps: workgroup 0, thread 0 only, into a buffer nothing else reads
__device__ __forceinline__ void pk_stamp(const PkState& st, int k, int i) {
    if (st.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) st.dbg[k * 8 + i] = __builtin_amdgcn_s_memrealtime();
}

struct PkPrefetch {  // the registers that carry the next op's first loads across the barrier
    uint4 w[2][PK_RU];
    uint32_t m[2][PK_RU];
    uint4 k, v;
};

// ------------------------------------------------------------------------------------------ grid barrier
// arrive: after this workgroup's hand-off stores have drained.  wait: all 256 workgroups arrived `epoch` times.
__device__ __forceinline__ void pk_arrive(const PkState& st) {
    __syncthreads();  // every wave of the workgroup has drained its stores (each did its own s_waitcnt)
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add(st.counters + (blockIdx.x % PK_SHARDS) * PK_SHARD_STRIDE, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool pk_wait(const PkState& st, unsigned epoch, int* sh_flag) {
    if (threadIdx.x < 64) {
        const unsigned target = epoch * (PK_WGS / PK_SHARDS);
        const int lane = threadIdx.x;
        bool ok = false;
        unsigned spins = 0;
        for (;;) {
            unsigned v = target, e = 0;
            if (lane < PK_SHARDS) v = ld_agent32(st.counters + lane * PK_SHARD_STRIDE);
            if (lane == PK_SHARDS) e = ld_agent32(st.err);
            const bool done = __all(v >= target);
            const bool failed = __any(e != 0);
            if (done && !failed) {
                ok = true;
                break;
            }
            if (failed || ++spins > PK_MAX_SPINS) {
                if (lane == 0 && !failed) st_agent32(st.err, 0x80000000u | epoch);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (lane == 0) *sh_flag = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = *sh_flag != 0;
    __syncthreads();
    return ok;
}

// ------------------------------------------------------------------------------------------ int4 GEMV op
struct PkLane {  // where this lane sits in the current GEMV op
    int slab, j, wps;
    int lslice, gslice, gl;
    bool active;
    int w_off16, meta_off16;
};

__device__ __forceinline__ PkLane pk_lane(const PkOp* op, int wave, int lane) {
    PkLane L;
    L.wps = PK_WAVES / op->nslabs;
    L.slab = wave / L.wps;
    L.j = wave % L.wps;
    const parrot_pk_slab_t sl = op->slab[L.slab];
    L.active = lane < sl.nslices;
    L.lslice = L.active ? lane : sl.nslices - 1;
    L.gslice = sl.slice0 + L.lslice;
    L.gl = L.gslice / op->Gs - sl.g0;
    L.w_off16 = sl.w_off16;
    L.meta_off16 = sl.meta_off16;
    return L;
}

// first row of the row group that (workgroup, wave j) owns in round rd; groups are dealt cyclically over workgroups first
__device__ __forceinline__ int pk_group_row(const PkLane& L, int rd) { return ((rd * L.wps + L.j) * PK_WGS + (int)blockIdx.x) * PK_RU; }

__device__ __forceinline__ void pk_load_rows(const PkOp* op, const PkLane& L, int r0, PkPrefetch& pf) {
    const uint4* W = reinterpret_cast<const uint4*>(op->W);
    const uint4* W2 = reinterpret_cast<const uint4*>(op->W2);
    const int64_t row16 = op->row16;
    const bool dual = op->W2 != nullptr;
#pragma unroll
    for (int u = 0; u < PK_RU; ++u) {
        const int64_t row = min(r0 + u, op->N - 1);
        const uint4* rec = W + row * row16;
        pf.w[0][u] = load_nt16(rec + L.w_off16 + L.lslice);
        pf.m[0][u] = load_nt4(reinterpret_cast<const uint32_t*>(rec + L.meta_off16) + L.gl);
        if (dual) {
            const uint4* rec2 = W2 + row * row16;
            pf.w[1][u] = load_nt16(rec2 + L.w_off16 + L.lslice);
            pf.m[1][u] = load_nt4(reinterpret_cast<const uint32_t*>(rec2 + L.meta_off16) + L.gl);
        }
    }
}

// LDS carve-up (one dynamic array, every offset a multiple of 16)
struct PkLds {
    unsigned char* x;  // activation vector of the current op (K bf16)
    float* red;        // [PK_WAVES][2 * PK_RU]
    float* stat;       // [16]
    int* flag;         // [4]
    float* best;       // [PK_WAVES][2] arg-max scratch
    unsigned char* attn;
};

__device__ void pk_gemv(const PkState& st, const PkOp* op, PkPrefetch& pf, const PkLds& lds, const bf16_t* emb_row, int k) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = op->K, N = op->N;
    const bool dual = op->W2 != nullptr;
    const int NW = dual ? 2 : 1;

    // ---- 1. broadcast of the input vector: agent-scope loads (another workgroup wrote it) into LDS, once per workgroup
    const bf16_t* xsrc = op->x_from_embedding ? emb_row : reinterpret_cast<const bf16_t*>(op->x);
    uint64_t* x64 = reinterpret_cast<uint64_t*>(lds.x);
    for (int i = threadIdx.x; i < (K >> 2); i += PK_THREADS) x64[i] = ld_agent64(xsrc + 4 * i);
    __syncthreads();
    pk_stamp(st, k, 1);

    // ---- 2. norm statistics over the whole vector
    float mean = 0.f, rscale = 1.f;
    NormArgs na;
    na.kind = op->norm_kind;
    na.eps = op->norm_eps;
    na.rsqrt_mode = st.rsqrt_mode;
    na.d = K;
    if (na.kind != 0) {
        const uint32_t* x32 = reinterpret_cast<const uint32_t*>(lds.x);
        float s1 = 0.f;
        for (int i = threadIdx.x; i < (K >> 1); i += PK_THREADS) s1 += norm_stat1(x32[i], na.kind);
        s1 = block_sum_waves(s1, lds.stat, PK_WAVES);
        if (na.kind == 2) {
            mean = s1 / (float)K;
            float s2 = 0.f;
            for (int i = threadIdx.x; i < (K >> 1); i += PK_THREADS) s2 += norm_stat2(x32[i], mean);
            rscale = norm_scale(na, block_sum_waves(s2, lds.stat, PK_WAVES));
        } else {
            rscale = norm_scale(na, s1);
        }
    }

    // ---- 3. this lane's 32 activations (normalised), as 16 packed bf16 pairs, and their sum
    const PkLane L = pk_lane(op, wave, lane);
    uint32_t xr[16];
    {
        const uint4* xl = reinterpret_cast<const uint4*>(lds.x) + (int64_t)L.gslice * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 v = xl[q];
            if (!L.active) v = make_uint4(0, 0, 0, 0);
            xr[4 * q] = v.x; xr[4 * q + 1] = v.y; xr[4 * q + 2] = v.z; xr[4 * q + 3] = v.w;
        }
        if (na.kind != 0) {
            const uint4* wp = reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(op->norm_w) + (int64_t)L.gslice * 32);
            const uint4* bp = reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(op->norm_b) + (int64_t)L.gslice * 32);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 wv = wp[q];
                uint4 bv = make_uint4(0, 0, 0, 0);
                if (na.kind == 2 && op->norm_b != nullptr) bv = bp[q];
                const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    xr[4 * q + i] = L.active ? norm_apply(xr[4 * q + i], ww[i], bb[i], na.kind, mean, rscale) : 0u;
            }
        }
    }
    float xs = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) xs += bflo(xr[i]) + bfhi(xr[i]);
    pk_stamp(st, k, 2);

    // ---- 4. rounds of row groups; round 0's weights were requested before the barrier
    const int ngroups = (N + PK_RU - 1) / PK_RU;
    const int nrounds = (ngroups + L.wps * PK_WGS - 1) / (L.wps * PK_WGS);
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int rd = 0; rd < nrounds; ++rd) {
        const int r0 = pk_group_row(L, rd);
        if (rd != 0) pk_load_rows(op, L, r0, pf);
#pragma unroll
        for (int u = 0; u < PK_RU; ++u) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q < NW) {
                    const float s = bflo(pf.m[q][u]);
                    const float zz = 128.0f + bfhi(pf.m[q][u]);
                    const float p = w4_slice_dot(pf.w[q][u], xr);
                    const float v = wave_sum_to_lane63(s * (p - zz * xs));
                    if (lane == 63) lds.red[wave * (2 * PK_RU) + u * 2 + q] = v;
                }
            }
        }
        __syncthreads();
        if (rd == 0) pk_stamp(st, k, 3);
        // epilogue: one thread per pair of rows of each group owned by this workgroup in this round
        if ((int)threadIdx.x < L.wps * (PK_RU / 2)) {
            const int jj = threadIdx.x / (PK_RU / 2), up = threadIdx.x % (PK_RU / 2);
            const int row = ((rd * L.wps + jj) * PK_WGS + (int)blockIdx.x) * PK_RU + 2 * up;
            if (row < N) {
                uint32_t packed = 0;
                uint32_t res2 = 0;
                const bf16_t* rsrc = op->res_from_embedding ? emb_row : reinterpret_cast<const bf16_t*>(op->residual);
                if (op->epilogue == PARROT_EPI_RESIDUAL) res2 = ld_agent32(rsrc + row);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float a0 = 0.f, a1 = 0.f;
                    for (int c = 0; c < op->nslabs; ++c) {
                        a0 += lds.red[(c * L.wps + jj) * (2 * PK_RU) + (2 * up + h) * 2];
                        if (dual) a1 += lds.red[(c * L.wps + jj) * (2 * PK_RU) + (2 * up + h) * 2 + 1];
                    }
                    const bf16_t rb = (bf16_t)(h ? (res2 >> 16) : (res2 & 0xffffu));
                    const bf16_t o = pk_epilogue(op->epilogue, a0, a1, reinterpret_cast<const bf16_t*>(op->bias), rb, row + h);
                    packed |= (uint32_t)o << (16 * h);
                    if (op->track_argmax && row + h < st.V) {
                        float v = bf2f(o);
                        if (v != v) v = -INFINITY;
                        if (best_i == 0x7fffffff || v > best) {
                            best = v;
                            best_i = row + h;
                        }
                    }
                }
                st_agent32(reinterpret_cast<bf16_t*>(op->out) + row, packed);
            }
        }
        __syncthreads();  // red is reused by the next round
    }
    if (op->track_argmax) {  // this workgroup's (max, lowest index) -> scratch; ties resolved by index like argmax_advance
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(best_i, off, 64);
            if (ov > best || (ov == best && oi < best_i)) {
                best = ov;
                best_i = oi;
            }
        }
        if (threadIdx.x == 0) {  // epilogue threads all live in wave 0 (wps * 4 <= 48)
            st_agentf(st.argmax_val + blockIdx.x, best);
            st_agent32(st.argmax_idx + blockIdx.x, (uint32_t)best_i);
        }
    }
}

// ------------------------------------------------------------------------------------------ attention op
template <int HS>
__device__ __forceinline__ void pk_attn_prefetch(const PkState& st, const PkOp* op, int pos, PkPrefetch& pf) {
    constexpr int LPR = HS / 8, RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, dl = lane % LPR;
    const int nsplit = PK_WGS / st.n_groups;
    const int g = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
    const int n_valid = min(pos + 1, st.S);
    const int per = (st.S + nsplit - 1) / nsplit;
    const int s_begin = split * per, s_end = min(n_valid, s_begin + per);
    const int s_first = s_begin + wave * RPW;
    pf.k = make_uint4(0, 0, 0, 0);
    pf.v = pf.k;
    if (s_first < s_end) {
        const int sc = min(s_first + sub, s_end - 1);
        pf.k = reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(op->k_cache) + (int64_t)g * st.S * HS)[(int64_t)sc * LPR + dl];
        pf.v = reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(op->v_cache) + (int64_t)g * st.S * HS)[(int64_t)sc * LPR + dl];
    }
}

template <int HS, int HQ>
__device__ void pk_attn(const PkState& st, const PkOp* op, int pos, PkPrefetch& pf, const PkLds& lds) {
    constexpr int LPR = HS / 8, RPW = 64 / LPR, NSLOT = PK_WAVES * RPW, STRIDE = PK_WAVES * RPW;
    float* sh_acc = reinterpret_cast<float*>(lds.attn);                 // [HQ][NSLOT][HS]
    float* sh_m = sh_acc + HQ * NSLOT * HS;                              // [HQ][NSLOT]
    float* sh_l = sh_m + HQ * NSLOT;                                     // [HQ][NSLOT]
    float* sh_q = sh_l + HQ * NSLOT;                                     // [q_per_kv][HS]
    bf16_t* sh_kv = reinterpret_cast<bf16_t*>(sh_q + 16 * HS);           // [2][HS]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, dl = lane % LPR;
    const int q_per_kv = st.q_per_kv, n_elem = st.n_elem, half_n = st.n_elem >> 1, S = st.S;
    const int nsplit = PK_WGS / st.n_groups;
    const int g = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
    const int n_valid = min(pos + 1, S);
    const int slot_new = pos % S;
    const int per = (S + nsplit - 1) / nsplit;
    const int s_begin = split * per, s_end = min(n_valid, s_begin + per);
    const float scale = 1.0f / sqrtf((float)HS);
    bf16_t* k_cache = reinterpret_cast<bf16_t*>(op->k_cache);
    bf16_t* v_cache = reinterpret_cast<bf16_t*>(op->v_cache);
    const uint4* kc = reinterpret_cast<const uint4*>(k_cache + (int64_t)g * S * HS);
    const uint4* vc = reinterpret_cast<const uint4*>(v_cache + (int64_t)g * S * HS);

    // ---- split + RoPE of this group's rows of the QKV vector (agent-scope loads: other workgroups produced it)
    const bf16_t* grp = reinterpret_cast<const bf16_t*>(op->x) + (int64_t)g * (q_per_kv + 2) * HS;
    for (int idx = threadIdx.x; idx < (q_per_kv + 2) * (HS / 2); idx += PK_THREADS) {
        const int t = idx / (HS / 2), dp = idx % (HS / 2);  // element pair (2*dp, 2*dp+1) of row t
        const uint32_t pr = ld_agent32(grp + t * HS + 2 * dp);
        float vals[2] = {bflo(pr), bfhi(pr)};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d = 2 * dp + h;
            float v = vals[h];
            if (t <= q_per_kv && d < n_elem) {
                const float c = __half2float(reinterpret_cast<const __half*>(st.rope_cos)[(int64_t)pos * n_elem + d]);
                const float sn = __half2float(reinterpret_cast<const __half*>(st.rope_sin)[(int64_t)pos * n_elem + d]);
                const int dpart = d < half_n ? d + half_n : d - half_n;
                const uint32_t pp = ld_agent32(grp + t * HS + (dpart & ~1));
                const float other = (dpart & 1) ? bfhi(pp) : bflo(pp);
                v = __fadd_rn(__fmul_rn(v, c), __fmul_rn(d < half_n ? -other : other, sn));
            }
            const bf16_t vb = f2bf(v);
            if (t < q_per_kv)
                sh_q[t * HS + d] = bf2f(vb) * scale;
            else
                sh_kv[(t - q_per_kv) * HS + d] = vb;
        }
    }
    __syncthreads();
    if (slot_new >= s_begin && slot_new < s_begin + per && threadIdx.x < 2 * LPR) {  // KV append
        const int which = threadIdx.x / LPR, c = threadIdx.x % LPR;
        bf16_t* dst = (which ? v_cache : k_cache) + ((int64_t)g * S + slot_new) * HS;
        reinterpret_cast<uint4*>(dst)[c] = reinterpret_cast<const uint4*>(sh_kv + which * HS)[c];
    }
    const uint4 knew = reinterpret_cast<const uint4*>(sh_kv)[dl];
    const uint4 vnew = reinterpret_cast<const uint4*>(sh_kv + HS)[dl];
    const int slot = wave * RPW + sub;
    const int s_first = s_begin + wave * RPW;
    uint4 kv_cur = pf.k, vv_cur = pf.v;

    for (int h0 = 0; h0 < q_per_kv; h0 += HQ) {
        float qf[HQ][8], mrun[HQ], lrun[HQ], acc[HQ][8];
#pragma unroll
        for (int hh = 0; hh < HQ; ++hh) {
            const int hq = min(h0 + hh, q_per_kv - 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[hh][e] = sh_q[hq * HS + dl * 8 + e];
            mrun[hh] = -INFINITY;
            lrun[hh] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[hh][e] = 0.f;
        }
        if (h0 != 0 && s_first < s_end) {
            const int sc = min(s_first + sub, s_end - 1);
            kv_cur = kc[(int64_t)sc * LPR + dl];
            vv_cur = vc[(int64_t)sc * LPR + dl];
        }
        for (int s0 = s_first; s0 < s_end; s0 += STRIDE) {
            uint4 kv_nxt = kv_cur, vv_nxt = vv_cur;
            if (s0 + STRIDE < s_end) {
                const int sn = min(s0 + STRIDE + sub, s_end - 1);
                kv_nxt = kc[(int64_t)sn * LPR + dl];
                vv_nxt = vc[(int64_t)sn * LPR + dl];
            }
            const int s = s0 + sub;
            const bool ok = s < s_end;
            const int sc = ok ? s : s_end - 1;
            uint4 kv = kv_cur, vv = vv_cur;
            if (sc == slot_new) {
                kv = knew;
                vv = vnew;
            }
            const uint32_t kd[4] = {kv.x, kv.y, kv.z, kv.w};
            const uint32_t vd[4] = {vv.x, vv.y, vv.z, vv.w};
            float kf[8], vf[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                kf[2 * j] = bflo(kd[j]);
                kf[2 * j + 1] = bfhi(kd[j]);
                vf[2 * j] = bflo(vd[j]);
                vf[2 * j + 1] = bfhi(vd[j]);
            }
#pragma unroll
            for (int hh = 0; hh < HQ; ++hh) {
                float sc_ = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sc_ = fmaf(qf[hh][e], kf[e], sc_);
#pragma unroll
                for (int off = LPR / 2; off >= 1; off >>= 1) sc_ += __shfl_xor(sc_, off, 64);
                if (ok) {
                    const float mn = fmaxf(mrun[hh], sc_);
                    const float corr = __expf(mrun[hh] - mn);
                    const float p = __expf(sc_ - mn);
                    lrun[hh] = lrun[hh] * corr + p;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[hh][e] = acc[hh][e] * corr + p * vf[e];
                    mrun[hh] = mn;
                }
            }
            kv_cur = kv_nxt;
            vv_cur = vv_nxt;
        }
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < HQ; ++hh) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sh_acc[(hh * NSLOT + slot) * HS + dl * 8 + e] = acc[hh][e];
            if (dl == 0) {
                sh_m[hh * NSLOT + slot] = mrun[hh];
                sh_l[hh * NSLOT + slot] = lrun[hh];
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < HQ * HS; idx += PK_THREADS) {
            const int hh = idx / HS, d = idx % HS;
            if (h0 + hh < q_per_kv) {
                float mx = -INFINITY;
                for (int t = 0; t < NSLOT; ++t) mx = fmaxf(mx, sh_m[hh * NSLOT + t]);
                float l = 0.f, a = 0.f;
                for (int t = 0; t < NSLOT; ++t) {
                    const float mt = sh_m[hh * NSLOT + t];
                    const float wgt = (mt == -INFINITY) ? 0.f : __expf(mt - mx);
                    l += sh_l[hh * NSLOT + t] * wgt;
                    a += sh_acc[(hh * NSLOT + t) * HS + d] * wgt;
                }
                const int head = g * q_per_kv + h0 + hh;
                float* p = st.attn_ws + ((int64_t)head * nsplit + split) * (HS + 2);
                st_agentf(p + d, a);
                if (d == 0) {
                    st_agentf(p + HS, mx);
                    st_agentf(p + HS + 1, l);
                }
            }
        }
    }
    // ---- arrival ticket of this group: the last of its nsplit workgroups merges the partial states into y
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(st.tickets + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)(nsplit - 1));
        if (last) __hip_atomic_store(st.tickets + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds.flag[1] = last;
    }
    __syncthreads();
    if (lds.flag[1]) {
        for (int idx = threadIdx.x; idx < q_per_kv * (HS / 2); idx += PK_THREADS) {
            const int hq = idx / (HS / 2), dp = idx % (HS / 2);
            const int head = g * q_per_kv + hq;
            const float* p = st.attn_ws + (int64_t)head * nsplit * (HS + 2);
            float mx = -INFINITY;
            for (int t = 0; t < nsplit; ++t) mx = fmaxf(mx, ld_agentf(p + t * (HS + 2) + HS));
            float l = 0.f, a0 = 0.f, a1 = 0.f;
            for (int t = 0; t < nsplit; ++t) {
                const float mt = ld_agentf(p + t * (HS + 2) + HS);
                const float wgt = (mt == -INFINITY) ? 0.f : __expf(mt - mx);
                l += ld_agentf(p + t * (HS + 2) + HS + 1) * wgt;
                a0 += ld_agentf(p + t * (HS + 2) + 2 * dp) * wgt;
                a1 += ld_agentf(p + t * (HS + 2) + 2 * dp + 1) * wgt;
            }
            const uint32_t packed = (uint32_t)f2bf(a0 / l) | ((uint32_t)f2bf(a1 / l) << 16);
            st_agent32(reinterpret_cast<bf16_t*>(op->out) + (int64_t)head * HS + 2 * dp, packed);
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------ the token program
template <int HS, int HQ>
__global__ void __launch_bounds__(PK_THREADS)
pk_token_kernel(PkState st) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pk_smem[];
    PkLds lds;
    lds.x = pk_smem;
    lds.red = reinterpret_cast<float*>(pk_smem + st.lds_x_bytes);
    lds.stat = lds.red + PK_WAVES * 2 * PK_RU;
    lds.flag = reinterpret_cast<int*>(lds.stat + 16);
    lds.best = reinterpret_cast<float*>(lds.flag + 4);
    lds.attn = reinterpret_cast<unsigned char*>(lds.best + 28);  // 192 + 16 + 4 + 28 words = 960 B: stays 16-B aligned

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int pos = st.pos[0];
    const int64_t tok = st.tokens[pos];
    const bf16_t* emb_row = reinterpret_cast<const bf16_t*>(st.wte) + tok * st.d;
    const PkOp* ops = st.ops;

    PkPrefetch pf;
    // prefetch for op 0
    if (ops[0].type == PARROT_PK_GEMV) {
        const PkLane L = pk_lane(&ops[0], wave, lane);
        pk_load_rows(&ops[0], L, pk_group_row(L, 0), pf);
    }
    for (int k = 0; k < st.nops; ++k) {
        const PkOp* op = ops + k;
        pk_stamp(st, k, 7);
        if (k > 0 && !pk_wait(st, (unsigned)k, lds.flag)) return;  // all 256 workgroups finished op k-1 (or timeout)
        pk_stamp(st, k, 0);
        if (op->type == PARROT_PK_GEMV) {
            pk_gemv(st, op, pf, lds, emb_row, k);
        } else if (op->type == PARROT_PK_ATTN) {
            pk_attn<HS, HQ>(st, op, pos, pf, lds);
        } else {  // PARROT_PK_ARGMAX: workgroup 0 merges the per-workgroup maxima, writes the token and advances pos
            if (blockIdx.x == 0 && threadIdx.x < 64) {
                float best = -INFINITY;
                int bi = 0x7fffffff;
                for (int i = lane; i < PK_WGS; i += 64) {
                    const float v = ld_agentf(st.argmax_val + i);
                    const int ix = (int)ld_agent32(st.argmax_idx + i);
                    if (v > best || (v == best && ix < bi)) {
                        best = v;
                        bi = ix;
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
                if (lane == 0) {
                    st.tokens[pos + 1] = (bi == 0x7fffffff) ? 0 : bi;
                    st.pos[0] = pos + 1;
                }
            }
            return;
        }
        if (k + 1 >= st.nops) return;
        // ---- drain this op's hand-off stores, then request the next op's first loads, then arrive: those loads stream
        // from HBM underneath the barrier, the activation broadcast and the norm prologue of the next op
        pk_stamp(st, k, 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pk_stamp(st, k, 5);
        const PkOp* nx = op + 1;
        if (nx->type == PARROT_PK_GEMV) {
            const PkLane L = pk_lane(nx, wave, lane);
            pk_load_rows(nx, L, pk_group_row(L, 0), pf);
        } else if (nx->type == PARROT_PK_ATTN) {
            pk_attn_prefetch<HS>(st, nx, pos, pf);
        }
        pk_arrive(st);
        pk_stamp(st, k, 6);
    }
}

static size_t pk_lds_bytes(const PkState& st, int HS, int HQ) {
    const int RPW = 64 / (HS / 8), NSLOT = PK_WAVES * RPW;
    const size_t fixed = (size_t)(PK_WAVES * 2 * PK_RU + 16 + 4 + 28) * 4;
    const size_t attn = (size_t)(HQ * NSLOT * HS + 2 * HQ * NSLOT + 16 * HS) * 4 + 2 * HS * 2;
    return (size_t)st.lds_x_bytes + fixed + attn + 16;
}

}  // namespace parrot

using namespace parrot;

extern "C" {

int parrot_pk_fill_w4(parrot_pk_op_t* op_host, int N, int K, int group) {
    PARROT_REQUIRE(op_host != nullptr, "pk_fill_w4: null op");
    W4Plan plan;
    const int rc = w4_make_plan(N, K, group, &plan);
    if (rc != PARROT_OK) return rc;
    PARROT_UNSUPPORTED(plan.nslabs <= PARROT_PK_MAX_SLABS && PK_WAVES % plan.nslabs == 0,
                       "persistent step: K=%d needs %d slabs, which does not divide the %d waves of a workgroup", K,
                       plan.nslabs, PK_WAVES);
    PARROT_UNSUPPORTED(N % 2 == 0, "persistent step: N=%d must be even", N);
    op_host->N = N;
    op_host->K = K;
    op_host->nslabs = plan.nslabs;
    op_host->row16 = plan.row16;
    op_host->Gs = plan.Gs;
    for (int c = 0; c < PARROT_PK_MAX_SLABS; ++c) {
        const W4Slab& s = plan.slab[c < plan.nslabs ? c : 0];
        op_host->slab[c].slice0 = s.slice0;
        op_host->slab[c].nslices = s.nslices;
        op_host->slab[c].g0 = s.g0;
        op_host->slab[c].w_off16 = s.w_off16;
        op_host->slab[c].meta_off16 = s.meta_off16;
    }
    return PARROT_OK;
}

int parrot_pk_step(const parrot_pk_state_t* state_host, void* stream) {
    PARROT_REQUIRE(state_host != nullptr, "pk_step: null state");
    PkState st = *state_host;
    PARROT_REQUIRE(st.ops && st.nops >= 1 && st.tokens && st.pos && st.wte && st.counters && st.err && st.tickets &&
                       st.attn_ws && st.argmax_val && st.argmax_idx,
                   "pk_step: null pointer in state");
    PARROT_UNSUPPORTED(st.n_groups >= 1 && PK_WGS % st.n_groups == 0, "persistent step: n_query_groups=%d must divide %d",
                       st.n_groups, PK_WGS);
    PARROT_UNSUPPORTED(st.q_per_kv >= 1 && st.q_per_kv <= 16, "persistent step: q_per_kv=%d out of range", st.q_per_kv);
    PARROT_REQUIRE(st.lds_x_bytes > 0 && st.lds_x_bytes % 16 == 0, "pk_step: lds_x_bytes must be a positive multiple of 16");
    PARROT_REQUIRE(st.n_elem % 2 == 0 && st.n_elem <= st.hs, "pk_step: bad n_elem");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(st.counters, 0, PK_SHARDS * PK_SHARD_STRIDE * sizeof(unsigned), s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(counters)");
    const int hq = st.q_per_kv == 1 ? 1 : (st.q_per_kv == 2 ? 2 : 4);
#define PARROT_PK_GO(HSV, HQV)                                                                                   \
    do {                                                                                                         \
        const size_t lds = pk_lds_bytes(st, HSV, HQV);                                                           \
        PARROT_UNSUPPORTED(lds <= 160 * 1024, "persistent step: needs %zu B of LDS", lds);                        \
        static bool attr_set = false;                                                                            \
        if (!attr_set) {                                                                                         \
            e = hipFuncSetAttribute((const void*)pk_token_kernel<HSV, HQV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    160 * 1024);                                                                 \
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");                                       \
            attr_set = true;                                                                                     \
        }                                                                                                        \
        return launch(K_PK_TOKEN, pk_token_kernel<HSV, HQV>, dim3(PK_WGS), dim3(PK_THREADS), lds, s, st);        \
    } while (0)
    if (st.hs == 128) {
        if (hq == 1) PARROT_PK_GO(128, 1);
        if (hq == 2) PARROT_PK_GO(128, 2);
        PARROT_PK_GO(128, 4);
    }
    if (st.hs == 64) {
        if (hq == 1) PARROT_PK_GO(64, 1);
        if (hq == 2) PARROT_PK_GO(64, 2);
        PARROT_PK_GO(64, 4);
    }
#undef PARROT_PK_GO
    set_error("persistent step: head size %d not built (64, 128)", st.hs);
    return PARROT_EUNSUPPORTED;
}

}  // extern "C"
