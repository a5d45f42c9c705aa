// Stream engine: ONE launch per decode token (include/parrot_hip.h, "stream engine").
//
// Reference path: one iteration of generate() (generate/base.py:131-153) = GPT.forward on one token
// (lit_gpt/model.py:63-111): embedding, per Block (:158-180) norm -> fused QKV Linear -> RoPE + KV append + attention
// (:194-275) -> out-projection + residual, norm -> MLP (:278-301) + residual (sequential, or both branches on the block's
// input: parallel residual, :166-171), then ln_f, lm_head, arg-max.
//
// Why one launch: the multi-launch step pays a ramp, a tail and a kernel boundary (~4 us in all) 161 times per token with
// the HBM idle in between.  Here 256 workgroups (one per CU, 16 waves) stay resident for the whole token:
//   * one LOADER wave (two with bf16 weights) walks the CU's share of the token's byte stream - for every Linear the CU's
//     blocks of 8 output rows in the E4 (int4) or E16 (bf16) layout, in front of them the norm's weights, for every
//     attention op the K/V rows of the CU's key range - and moves it into a ring of 6 or 7 LDS slots (17 KiB each) by
//     LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction).  It never waits for a data dependency, only for a
//     free slot, so the weights of the ops behind an activation hand-off are already on chip when the hand-off completes;
//   * the other 15 (14) waves are CONSUMERS.  Per op they gather the input vector into LDS (normalised, bf16), then take
//     the op's work units round-robin as their slots land.  int4: a QUAD = 8 rows x 1024 columns (four 1-KiB pieces + one
//     metadata word per lane); lane l of a piece holds the 32-column slice of row l % 8 in quantisation group
//     8 * quad + l / 8, so a lane accumulates whole groups and the only cross-lane step is one sum over the 8 lanes of a
//     row per quad.  bf16: a unit is a whole slot, 8 rows x 1024 columns.  Attention: a K piece + a V piece.  The wave that
//     finishes a block's last unit sums the units in a fixed order, applies the epilogue and publishes the 8 outputs;
//   * activations pass between CUs as 8-byte GRANULES {data, tag}: a write-through (sc1) store by the producer, an
//     L1-bypassing (sc1) load by the consumer, valid when tag == the launch's epoch.  No flags, no fences, no grid
//     barrier: a consumer simply re-reads a granule until its tag matches.  Every buffer is written once per launch.
// Every wait is bounded: on a time-out the error word is set and every later wait falls through, so the grid drains.
// The kernel's text is felt (14 - 15 waves share four SIMDs and a 64-KB instruction cache serves two CUs): one weight
// format per build, and several "obvious" generalisations were measured and undone for that reason.
#include <hip/hip_fp16.h>

#include "parrot_common.h"
#include "w4_plan.h"

namespace parrot {

// experiment builds (tools/ab_engine.sh) override these
#ifndef ENG_THIN_PIECES_V
#define ENG_THIN_PIECES_V (-1)  // -1: by weight format (EngCfg::THIN)
#endif
#ifndef ENG_MAXFLY_V
#define ENG_MAXFLY_V (-1)  // -1: by weight format (EngCfg::MAXFLY)
#endif
#ifndef ENG_KEYS_PER_SPLIT_V
#define ENG_KEYS_PER_SPLIT_V 32
#endif
#ifndef ENG_POLL_SLEEP
#define ENG_POLL_SLEEP 2
#endif
#ifndef ENG_GATE_DEEP
#define ENG_GATE_DEEP 0
#endif
#ifndef ENG_ATTN_GATE
#define ENG_ATTN_GATE 0
#endif
#ifndef ENG_STAMPS
#ifdef PARROT_DIAG
#define ENG_STAMPS 1
#else
#define ENG_STAMPS 0
#endif
#endif
#ifndef ENG_SPIN_MODE
#define ENG_SPIN_MODE (-1)  // -1: by weight format (bf16: waiting waves drop their priority, +0.8 %; int4: plain spin, +0.7 %)
#endif
constexpr int ENG_WGS = PARROT_ENG_WGS;
constexpr int ENG_KEYS_PER_SPLIT = ENG_KEYS_PER_SPLIT_V;  // keys of a head that one CU attends over before a second CU joins
constexpr int ENG_NC = 15;                // consumer waves at most (EngCfg::NC)
constexpr int ENG_THREADS = 16 * 64;
constexpr int ENG_SLOT_BYTES = 17 * 1024;
constexpr int ENG_META_OFF = 16 * 1024;   // the metadata piece of a slot, in LDS
constexpr int ENG_RED = 16;               // block result buffers in flight
// Two builds of the kernel.  BIG = 0: inputs of up to 11264 elements, a ring of 7 slots (the Llama-2-7B family);
// BIG = 1: inputs of up to 16384 elements (StableLM's MLP), whose LDS image leaves room for 6 slots.
#ifndef ENG_NLOAD_MULTI
#define ENG_NLOAD_MULTI 2  // loader waves of the stream-bound builds (the ring's 6 slots alternate between them).  Measured with 3:
                           // Llama-2-7B int8 477 -> 466, StableLM-3B 832 -> 818, Falcon-40B int4 191 -> 176 tok/s; with 1 (int8): 382
#endif
template <int BIG, int WFMT>
struct EngCfg {
    static constexpr int WF = WFMT & 3;         // PARROT_ENG_W_E4 / _E16 / _E8: one weight format per launch
    // bf16 weights: the arithmetic per byte is a quarter of int4's, the stream is the limit - two loader waves (ring slots
    // alternate between them: twice the LDS-DMA in flight, vmcnt counts per wave) and 14 consumer waves
    // bf16 / int8 weights: a quarter of int4's arithmetic per byte, the stream is the limit - two loader waves.  int4: one
    // loader and 15 consumers for the sequential-residual block (Llama-2-7B: 745 tok/s; 709 with two), two loaders (flag
    // PARROT_ENG_W_TWO_LOADERS) for the parallel-residual block, where weights stream across the hand-offs (Falcon-40B 173
    // -> 190, Falcon-7B gptq.int4 585 -> 615)
    static constexpr int NLOAD = (WF != PARROT_ENG_W_E4 || (WFMT & PARROT_ENG_W_TWO_LOADERS)) ? ENG_NLOAD_MULTI : 1;
    static constexpr int SPIN = ENG_SPIN_MODE >= 0 ? ENG_SPIN_MODE : (WF != PARROT_ENG_W_E4 ? 1 : 0);  // consumer barrier: how waiting waves wait
    static constexpr int NC = 16 - NLOAD;       // consumer waves
    // ring slots with LDS-DMA in flight per loader (vmcnt counts at most 63 operations), and how many pieces may be
    // outstanding when a prefetch slot (an op still behind a hand-off) is issued.  Measured per format (tools/ab_engine.sh):
    // int4, one loader: 3 / 4 (round 2a); bf16, two loaders: 2 / 0 (StableLM-3B 832 -> 844 tok/s: less in flight while the
    // hand-off's stores and polls share the CU's memory pipeline)
    static constexpr int MAXFLY = ENG_MAXFLY_V >= 0 ? ENG_MAXFLY_V : (NLOAD == 2 ? 2 : 3);
    static constexpr int THIN = ENG_THIN_PIECES_V >= 0 ? ENG_THIN_PIECES_V : (NLOAD == 2 ? 0 : 4);
    static constexpr int NSLOT = (BIG || NLOAD >= 2) ? 6 : 7;  // ring slots (a multiple of the loader count: a slot keeps its loader)
    // input groups (128 elements) per consumer wave: K <= NC * MAXG * 128 (int8, wide build: 11 units of 2048 columns)
    static constexpr int MAXG = (WF == PARROT_ENG_W_E8 && BIG) ? 14 : (BIG ? 9 : 6) + (NLOAD - 1);
    static constexpr int MAXQ = BIG ? 16 : 11;  // units (1024 input columns) per block: K <= 1024 * MAXQ
};
constexpr int ENG_MAXQ_BIG = EngCfg<1, 0>::MAXQ, ENG_MAXQ_STD = EngCfg<0, 0>::MAXQ, ENG_MAXG_BIG = EngCfg<1, 0>::MAXG;
constexpr int ENG_NSLOT_BIG = EngCfg<1, 0>::NSLOT, ENG_NSLOT_STD = EngCfg<0, 0>::NSLOT;
constexpr int ENG_MAXG_E8 = EngCfg<0, PARROT_ENG_W_E8>::MAXG, ENG_MAXG_E8_BIG = EngCfg<1, PARROT_ENG_W_E8>::MAXG;
constexpr int ENG_GROUP_STRIDE = 272;     // LDS bytes per 128-element group of an activation buffer (256 + 16: bank spread)
// Every wait is bounded in TIME (the chip-wide 100 MHz clock, s_memrealtime), not in spins: a healthy launch slowed down by
// a profiler or by a second replica on the same device polls more slowly, it does not wait longer.  The clock is first read
// at a wait's first checkpoint (64 / 16 polls in), so the waits that pass at once never touch it.
#ifndef ENG_WAIT_TICKS_V
#define ENG_WAIT_TICKS_V 50000000ull  // 0.5 s
#endif
constexpr uint64_t ENG_WAIT_TICKS = ENG_WAIT_TICKS_V;
// Diagnostic builds that MEASURE the two halves of the token ("the engine's ceiling"; results are garbage):
//   ENG_STUB_UNITS = 1    the Linears' units do no arithmetic (slots are awaited and released, blocks published): what is
//                         left is the stream, the gathers and the hand-offs;
//   ENG_STUB_HANDOFF = 1  every hand-off is taken as satisfied (whatever the granules hold, from an earlier launch): what is
//                         left is the stream, the arithmetic and the CU's own barriers.
#ifndef ENG_STUB_UNITS
#define ENG_STUB_UNITS 0
#endif
#ifndef ENG_STUB_HANDOFF
#define ENG_STUB_HANDOFF 0
#endif
//   ENG_STUB_Q8 = 1       LLM.int8: the gather's activation quantiser does no per-element work (zeros, no outliers; its barrier and
//                         LDS traffic stay): what the ~450 vector instructions per wave of the quantiser cost the token
#ifndef ENG_STUB_Q8
#define ENG_STUB_Q8 0
#endif

typedef parrot_eng_op_t EngOp;
typedef parrot_eng_state_t EngState;

// fixed LDS area behind the ring and the two activation buffers (byte offsets inside it)
constexpr int EF_XS = 0;                                  // [2][128] float: per-group sums of the activations
constexpr int EF_ROPE = EF_XS + 2 * 128 * 4;              // [2][128] float: cos / sin row of this position
constexpr int EF_RESID = EF_ROPE + 2 * 128 * 4;           // [2][64] float: the CU's own rows of the residual stream (x; x + attention)
constexpr int EF_REDB = EF_RESID + 2 * 64 * 4;            // [ENG_RED][8] float: the Linear's bias of a block's rows
constexpr int EF_STAT = EF_REDB + ENG_RED * 8 * 4;        // [2][16] float: the waves' partial norm statistics
constexpr int EF_BESTV = EF_STAT + 128;                   // [16] float
constexpr int EF_BESTI = EF_BESTV + 64;                   // [16] int
constexpr int EF_FULL = EF_BESTI + 64;                    // [8] u32: sequence number + 1 of the slot's landed contents
constexpr int EF_CONS = EF_FULL + 32;                     // [8] u32: units consumed from the ring slot, cumulative
constexpr int EF_EXP = EF_CONS + 32;                      // [8] u32 (loader): units issued into the ring slot, cumulative
constexpr int EF_NPQ = EF_EXP + 32;                       // [8] u32 (loader): pieces of the issued, unpublished slots
constexpr int EF_DONE = EF_NPQ + 32;                      // [ENG_RED] u32: units finished of a block
constexpr int EF_CB = EF_DONE + ENG_RED * 4;              // consumer barrier counter
constexpr int EF_ABORT = EF_CB + 4;
constexpr int EF_GATE = EF_ABORT + 4;                     // op index + 1 whose input producers were seen to be done (one poller per CU)
constexpr int EF_GATE2 = EF_GATE + 4;                     // the same for the attention leader's wait for the partial states
constexpr int EF_CUR = EF_GATE2 + 4;                      // index of the op whose input the consumers have (loader: how urgent a slot is)
constexpr int EF_RED = EF_CUR + 16;                       // [ENG_RED][MAXQ][8] float: the units' partial sums of a block's rows
template <class CF>
constexpr int ef_red2() { return EF_RED + ENG_RED * CF::MAXQ * 8 * 4; }  // E8: [ENG_RED][MAXQ][8] float, the units' outlier sums
template <class CF>
constexpr int ef_q8() { return ef_red2<CF>() + ENG_RED * CF::MAXQ * 8 * 4; }  // E8: activation quantiser state (EQ_*)
constexpr int ENG_Q8_STATE = 384;  // bytes of the E8 quantiser state
template <class CF>
constexpr int ef_bytes() { return CF::WF == PARROT_ENG_W_E8 ? ef_q8<CF>() + ENG_Q8_STATE : ef_red2<CF>(); }
// LLM.int8 (E8) activation quantiser state, byte offsets from ef_q8(): the waves' |x| maxima and outlier counts of the
// vector being gathered - one set per gather round (a second norm of the same input is a second round, and a fast wave may be
// in it while a slow one still reads the first round's) -, then per LDS buffer the row scale (absmax) and the outlier count
constexpr int EQ_MAX = 0, EQ_CNT = 64, EQ_SA = 128, EQ_NO = 136, EQ_ROUND2 = 192;  // second round: EQ_MAX / EQ_CNT + EQ_ROUND2
constexpr int ENG_Q8_CAP = 1024;  // outlier list entries per buffer ({column, fp16 value} in 4 bytes)
constexpr float ENG_MM_DEQUANT = 6.200012e-05f;  // 1 / (127 * 127), the constant of w8.hip / bitsandbytes' mm_dequant
__device__ __forceinline__ float eng_rhalf(float v) { return __half2float(__float2half(v)); }

typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef const __attribute__((address_space(4))) uint32_t* cst_cu32_t;
// the op table is constant for the launch: read it through the constant address space, so that the (wave-uniform) reads
// are scalar loads - a vector load in the loader wave would sit in the same vmcnt queue as its LDS-DMA stream
__device__ __forceinline__ parrot_eng_op_t eng_fetch_op(const parrot_eng_op_t* ops, int k) {
    parrot_eng_op_t o;
    cst_cu32_t src = (cst_cu32_t)(ops + k);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(parrot_eng_op_t) / 4); ++i) dst[i] = src[i];
    return o;
}
typedef const __attribute__((address_space(1))) uint32_t* glb_cu32_t;
typedef const __attribute__((address_space(1))) uint16_t* glb_cu16_t;
typedef __attribute__((address_space(1))) uint64_t glb_u64_t;
typedef __attribute__((address_space(1))) uint32_t glb_u32_t;

__device__ __forceinline__ uint32_t lds_ld(unsigned char* p) {
    return __hip_atomic_load((lds_u32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(unsigned char* p, uint32_t v) {
    __hip_atomic_store((lds_u32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_add(unsigned char* p, uint32_t v) {
    return __hip_atomic_fetch_add((lds_u32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// all of this wave's LDS operations so far have completed (reads returned, writes performed)
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void st_gran(uint64_t* p, uint32_t data, uint32_t tag) {
    // global_store_dwordx2 ... sc1 (explicitly global, never flat: flat operations count in lgkmcnt too)
    __hip_atomic_store((glb_u64_t*)p, (uint64_t)data | ((uint64_t)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_gran(const uint64_t* p) {
    return __hip_atomic_load((glb_u64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_err(const EngState& st) {
    return __hip_atomic_load((glb_u32_t*)st.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- LDS-DMA: one wave instruction moves 64 x 16 B from per-lane global addresses to lds_dst .. lds_dst + 1023.
// Issued from inline asm: the compiler neither counts nor waits for it; the loader counts vmcnt itself (wait_vmcnt).
template <bool NT>
__device__ __forceinline__ void eng_dma(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

// v[lane] = value (both wave-uniform; the lane select goes through m0: one SGPR operand per VALU instruction)
__device__ __forceinline__ void eng_writelane(int& v, int value, int lane) {
    unsigned keep;
    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                 : "+v"(v), "=&s"(keep) : "s"(value), "s"(lane));
}

// wait until at most n of this wave's vector-memory operations are outstanding (n is wave-uniform, 0..63)
// (Inlined at each of its four call sites: as a function its callee prologue waits for vmcnt(0) - the pipeline of slots in
// flight collapsed, StableLM-3B 701 -> 510 tok/s.)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define ENG_VM(N) case N: asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); break;
#define ENG_VM4(N) ENG_VM(N) ENG_VM(N + 1) ENG_VM(N + 2) ENG_VM(N + 3)
#define ENG_VM16(N) ENG_VM4(N) ENG_VM4(N + 4) ENG_VM4(N + 8) ENG_VM4(N + 12)
    switch (n) {
        ENG_VM16(0) ENG_VM16(16) ENG_VM16(32)
        ENG_VM4(48) ENG_VM4(52) ENG_VM4(56) ENG_VM(60) ENG_VM(61) ENG_VM(62)
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
#undef ENG_VM16
#undef ENG_VM4
#undef ENG_VM
}

// sum over the 8 lanes that share a row (same lane % 8); every lane gets the total (DPP + gfx950 lane swaps, no LDS)
__device__ __forceinline__ float row8_allsum(float v) {
    v += dpp0<0x128>(v);  // row_ror:8
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v;
}
// attention: sum over the LPR lanes of a key row / all-reduce over the key rows of a piece (as attn.hip)
template <int LPR>
__device__ __forceinline__ float eng_group_sum(float v) {
    v += dpp0<0xB1>(v);
    v += dpp0<0x4E>(v);
    if (LPR >= 8) v += dpp0<0x141>(v);
    if (LPR >= 16) v += dpp0<0x128>(v);
    return v;
}
template <int LPR, bool MAX>
__device__ __forceinline__ float eng_slot_allreduce(float v) {
    auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
    if (LPR <= 8) v = op(v, dpp0<0x128>(v));
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    return v;
}

// what every wave of the workgroup knows
struct EngCtx {
    unsigned char* ring;
    unsigned char* buf0;  // the two activation buffers (no array: a run-time index would put the struct in scratch)
    unsigned char* buf1;
    unsigned char* fx;  // fixed area
    uint32_t epoch;
    int pos, cu, lane;
    const bf16_t* emb;  // wte row of the token being decoded
};

__device__ __forceinline__ bool eng_aborted(const EngCtx& c) { return lds_ld(c.fx + EF_ABORT) != 0; }
// A wait gave up (or a check failed): this CU's later waits fall through (EF_ABORT) and the launch's error word takes the
// code - the FIRST one only (compare-and-swap from 0: what went wrong first is what the host must see; the CUs that give up
// later, because the first one never published, would overwrite it with their own symptom).  The winner also leaves the code
// in the host-visible words (pinned host memory, read by the watchdog of bench.py without a HIP call).
__device__ __forceinline__ void eng_fail(const EngState& st, const EngCtx& c, uint32_t code) {
    lds_st(c.fx + EF_ABORT, 1u);
    uint32_t expected = 0u;
    if (__hip_atomic_compare_exchange_strong((glb_u32_t*)st.err, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &&
        st.host_words != nullptr)
        __hip_atomic_store((glb_u32_t*)st.host_words, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// another CU has failed: stop waiting, keep its code
__device__ __forceinline__ void eng_abort_local(const EngCtx& c) { lds_st(c.fx + EF_ABORT, 1u); }
// true once a wait that started polling at (lazily read) t0 has used up its time
__device__ __forceinline__ bool eng_timed_out(uint64_t& t0) {
    const uint64_t now = __builtin_amdgcn_s_memrealtime();
    if (t0 == 0) t0 = now;
    return now - t0 > ENG_WAIT_TICKS;
}
// spin (LDS word >= target), bounded
__device__ __forceinline__ void eng_wait_lds_ge(const EngState& st, const EngCtx& c, int off, uint32_t target, uint32_t code) {
    unsigned spins = 0;
    uint64_t t0 = 0;
    while ((int32_t)(lds_ld(c.fx + off) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0) {
            if (eng_aborted(c)) return;
            if (eng_timed_out(t0)) {
                eng_fail(st, c, code);
                return;
            }
        }
    }
}

// the same without sleeping between polls: for waits that are a few hundred cycles long (the consumer barrier)
template <int MODE>
__device__ __forceinline__ void eng_wait_lds_ge_tight(const EngState& st, const EngCtx& c, int off, uint32_t target, uint32_t code) {
    unsigned spins = 0;
    uint64_t t0 = 0;
    if (MODE == 1) __builtin_amdgcn_s_setprio(0);  // the waves still working share this SIMD: they go first
    while ((int32_t)(lds_ld(c.fx + off) - target) < 0) {
        if (MODE == 2) __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0) {
            if (eng_aborted(c)) break;
            if (eng_timed_out(t0)) {
                eng_fail(st, c, code);
                break;
            }
        }
    }
    if (MODE == 1) __builtin_amdgcn_s_setprio(1);
}

// blocks of an op that CU c owns: local block bl = 0 .. nb - 1 is block bs + bl * bstep of the matrix.
// Contiguous ranges.  (ENG_CYCLIC = 1, an experiment: block b belongs to CU b % 256, so that at any moment the 256 loaders
// read neighbouring pieces of the matrix.  Measured: +0.5 % on StableLM-3B bf16, -5 % on Llama-2-7B int4 - the 8 outputs
// of a block are a quarter of a cache line, and neighbouring blocks now publish from different XCDs.)
#ifndef ENG_CYCLIC
#define ENG_CYCLIC 0
#endif
__device__ __forceinline__ void eng_block_range(const EngOp* op, int cu, int& bs, int& nb, int& bstep) {
    const int nblocks = op->nblocks;
    if (ENG_CYCLIC) {
        bs = cu;
        nb = (nblocks - cu + ENG_WGS - 1) / ENG_WGS;
        if (nb < 0) nb = 0;
        bstep = ENG_WGS;
    } else {
        bs = (int)(((int64_t)cu * nblocks) / ENG_WGS);
        nb = (int)(((int64_t)(cu + 1) * nblocks) / ENG_WGS) - bs;
        bstep = 1;
    }
    if (op->blk_parts > 1) {  // this op is part blk_part of blk_parts of the Linear: that share of the CU's blocks
        const int l0 = nb * op->blk_part / op->blk_parts, l1 = nb * (op->blk_part + 1) / op->blk_parts;
        bs += l0 * bstep;
        nb = l1 - l0;
    }
}
// key range of CU c in an attention op: group g, split s; keys [kb, ke) of the n_valid admitted slots
struct EngKeys {
    bool part;  // this CU takes part in the attention op
    int g, s, ns, kb, ke, nunits;  // (virtual) group, split, splits in use at this position, key range, K/V units
    int gr, jv;                    // the K/V group the virtual group belongs to, and which of its st.vper parts it is
};
template <int HS>
__device__ __forceinline__ EngKeys eng_keys(const EngState& st, int cu, int pos) {
    constexpr int KPP = 512 / HS;  // keys per 1-KiB piece
    EngKeys k;
    const int n_valid = min(pos + 1, st.S);
    // splits in use: one CU per ENG_KEYS_PER_SPLIT keys (a power of two, at most the nsplit CUs reserved per group).  A head
    // with few keys stays on ONE CU: its partial state then needs no trip through memory to a leader (3.2 us per layer)
    int ns = 1;
    while (ns < st.nsplit && ns * ENG_KEYS_PER_SPLIT < n_valid) ns *= 2;
    ns = min(ns, st.nsplit);
    k.ns = ns;
    k.g = cu / st.nsplit;
    k.s = cu % st.nsplit;
    // A K/V group with more than 2 query heads is attended as st.vper VIRTUAL groups of 1 or 2 heads each (the kernel's HQ):
    // every one of them streams the group's K/V rows for its own heads, so GQA / MQA models fill the 256 CUs too
    k.gr = k.g / st.vper;
    k.jv = k.g - k.gr * st.vper;
    k.part = cu < st.n_groups * st.vper * st.nsplit && k.s < ns;
    int per = (n_valid + ns - 1) / ns;
    per = (per + KPP - 1) / KPP * KPP;
    k.kb = min(k.s * per, n_valid);
    k.ke = min(k.kb + per, n_valid);
    k.nunits = k.part ? (k.ke - k.kb + KPP - 1) / KPP : 0;
    return k;
}

// ------------------------------------------------------------------------------------------ the loader wave(s)
// Loader li of CF::NLOAD walks every slot of the CU's stream and issues the ones with sequence number % NLOAD == li.
template <class CF, int HS>
__device__ __forceinline__ void eng_loader(const EngState& st, const EngCtx& c, int li) {
    constexpr int KPP = 512 / HS;
    constexpr int NSLOT = CF::NSLOT, NLOAD = CF::NLOAD;
    static_assert(NLOAD == 1 || NSLOT % NLOAD == 0, "a ring slot must keep its loader");
    const unsigned ring_lds = (unsigned)(uintptr_t)c.ring;
    int seq = 0;                  // sequence number of the next slot of the stream (all loaders count alike)
    int mine = 0, pub = 0;        // own slots issued / announced
    int inflight = 0;             // own pieces in flight
    uint64_t stalled = 0;         // diagnostic: 100 MHz ticks spent waiting for a free ring slot while issuing the current op
    __builtin_amdgcn_s_setprio(3);  // a handful of instructions per slot: never behind the consumers' arithmetic

    // the loader's own bookkeeping lives in two registers, one lane per entry (an LDS round trip per look-up was a fifth
    // of the time the loader has per slot at full stream rate): exp_v[ring slot] = units issued into it so far;
    // npq_v[own slot number % 8] = pieces of the issued, unannounced slots
    int exp_v = 0, npq_v = 0;
    auto publish_oldest = [&]() {
        const int np = __builtin_amdgcn_readlane(npq_v, pub & 7);
        wait_vmcnt(inflight - np);
        inflight -= np;
        const int g = li + pub * NLOAD;  // its sequence number
        lds_st(c.fx + EF_FULL + (g % NSLOT) * 4, (uint32_t)(g + 1));
        ++pub;
    };
    // wait for ring slot seq % NSLOT to be free; returns the cumulative unit count it had.
    // A hand-off's stores and polls queue behind whatever this CU has in flight (measured: one granule load took 0.5 us
    // with the loader idle, 1.6 - 3 us behind three slots of LDS-DMA).  Slots of the op the consumers are computing are
    // urgent and go out three deep; slots of ops whose input is still outstanding are prefetch: one slot in flight, so that
    // the epilogue stores of the current op and the polls of the next one pass quickly.
    auto acquire = [&](int np, int k) -> uint32_t {
        const int r = seq % NSLOT;
        const uint32_t target = (uint32_t)__builtin_amdgcn_readlane(exp_v, r);
        // (one publish loop for the three reasons to announce older slots first - the vmcnt switch inside is 64 cases of
        // code per copy and the kernel's text is felt: never sleep on a busy slot with landed data unannounced; at most
        // MAXFLY slots / 60 pieces in flight; prefetch thin)
        const bool busy = lds_ld(c.fx + EF_CONS + r * 4) != target;
        const bool thin = k > (int)lds_ld(c.fx + EF_CUR);
        while (pub < mine && (busy || mine - pub >= CF::MAXFLY || inflight + np > 60 || (thin && inflight > CF::THIN))) publish_oldest();
        if (busy) {
            const uint64_t t0 = (ENG_STAMPS && st.dbg != nullptr) ? __builtin_amdgcn_s_memrealtime() : 0;
            eng_wait_lds_ge(st, c, EF_CONS + r * 4, target, 0x10000000u | (uint32_t)seq);
            if (ENG_STAMPS && st.dbg != nullptr) stalled += __builtin_amdgcn_s_memrealtime() - t0;  // diagnostic: ring full
        }
        return target;
    };
    auto commit = [&](uint32_t target, int np, int nunits) {
        eng_writelane(exp_v, __builtin_amdgcn_readfirstlane((int)(target + (uint32_t)nunits)), seq % NSLOT);
        eng_writelane(npq_v, np, mine & 7);
        inflight += np;
        ++mine;
        ++seq;
    };
    auto own = [&]() -> bool { return NLOAD == 1 || seq % NLOAD == li; };

    for (int k = 0; k < st.nops; ++k) {
        const EngOp opv = eng_fetch_op(st.ops, k);
        const EngOp* op = &opv;
        if ((ENG_STAMPS && st.dbg != nullptr) && c.cu == 0 && c.lane == 0 && li == 0) {  // diagnostic: when the loader reaches the op, and what it has announced by then
            st.dbg[k * 16 + 4] = __builtin_amdgcn_s_memrealtime();
            st.dbg[k * 16 + 5] = (uint64_t)seq | ((uint64_t)(li + pub * NLOAD) << 32);
            if (k > 0) st.dbg[(k - 1) * 16 + 12] = stalled;  // ring-full time while issuing the previous op's slots
            stalled = 0;
        }
        if (op->type == PARROT_ENG_GEMV) {
            if (op->norm_kind != 0) {
                // the norm's weights (K bf16, constants; LayerNorm: the bias behind them) ride the ring too: a plain load of
                // them sat in front of the hand-off polls (vmcnt is in order) with an HBM miss of 1 - 2 us; every consumer
                // wave reads its share
                // (a second norm of the same input - the MLP's of a parallel-residual block - follows in a slot of its own)
                // (weight and bias share a slot when both fit - K <= 4096 - else they take one each)
                const int npw = (op->K * 2 + 1023) >> 10;
                const int64_t last = (int64_t)op->K * 2 - 16;
                for (int which = 0; which < (op->norm2_w != nullptr ? 2 : 1); ++which) {
                    const unsigned char* nw = reinterpret_cast<const unsigned char*>(which ? op->norm2_w : op->norm_w);
                    const unsigned char* nb = reinterpret_cast<const unsigned char*>(which ? op->norm2_b : op->norm_b);
                    const bool split = nb != nullptr && 2 * npw > 16;
                    for (int half = 0; half < (split ? 2 : 1); ++half) {
                        if (!own()) {
                            ++seq;
                            continue;
                        }
                        const int np = (nb != nullptr && !split) ? 2 * npw : npw;
                        const uint32_t target = acquire(np, k);
                        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)((seq % NSLOT) * ENG_SLOT_BYTES));
                        const unsigned char* first = half ? nb : nw;
                        for (int j = 0; j < npw; ++j) eng_dma<false>(first + min((int64_t)j * 1024 + c.lane * 16, last), dst + (unsigned)(j * 1024));
                        if (nb != nullptr && !split)
                            for (int j = 0; j < npw; ++j)
                                eng_dma<false>(nb + min((int64_t)j * 1024 + c.lane * 16, last), dst + (unsigned)((npw + j) * 1024));
                        commit(target, np, CF::NC);
                    }
                }
            }
            int bs, nb, bstep;
            eng_block_range(op, c.cu, bs, nb, bstep);
            const int nq = op->nq;
            if (CF::WF == PARROT_ENG_W_E8) {
                // int8 (LLM.int8): a piece is 8 rows x 128 columns, a slot = one unit = up to 16 pieces (2048 columns; the last
                // unit of a row may be shorter) + the block's 8 row scales SCB (32 bytes, fetched by every lane pair) in the
                // metadata place of EVERY slot: the outlier part of a unit dequantises single weights
                const int pt = (op->K + 127) >> 7;  // pieces per block
                for (int bl = 0; bl < nb; ++bl) {
                    const int b = bs + bl * bstep;
                    for (int Q = 0; Q < nq; ++Q) {
                        if (!own()) {
                            ++seq;
                            continue;
                        }
                        const int npc = min(16, pt - 16 * Q);
                        const uint32_t target = acquire(npc + 1, k);
                        const unsigned char* src = reinterpret_cast<const unsigned char*>(op->W) + ((int64_t)b * pt + 16 * Q) * 1024 + c.lane * 16;
                        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)((seq % NSLOT) * ENG_SLOT_BYTES));
                        for (int j = 0; j < npc; ++j) eng_dma<true>(src + j * 1024, dst + (unsigned)(j * 1024));
                        eng_dma<false>(reinterpret_cast<const unsigned char*>(op->bias) + (int64_t)b * 32 + (c.lane & 1) * 16, dst + (unsigned)ENG_META_OFF);
                        commit(target, npc + 1, 1);
                    }
                }
            } else if (CF::WF == PARROT_ENG_W_E16) {
                // bf16: a slot is one unit = 16 pieces (8 rows x 1024 columns); the first slot of a block carries the block's
                // 8 bias values (every lane fetches the same 16 bytes) where the int4 layout has its metadata
                const int pt = (op->K + 63) >> 6;  // pieces per block (the last unit of a row may be short)
                for (int bl = 0; bl < nb; ++bl) {
                    const int b = bs + bl * bstep;
                    for (int Q = 0; Q < nq; ++Q) {
                        if (!own()) {
                            ++seq;
                            continue;
                        }
                        const bool wb = Q == 0 && op->bias != nullptr;
                        const int npc = min(16, pt - 16 * Q);
                        const int np = npc + (wb ? 1 : 0);
                        const uint32_t target = acquire(np, k);
                        const unsigned char* src = reinterpret_cast<const unsigned char*>(op->W) + ((int64_t)b * pt + 16 * Q) * 1024 + c.lane * 16;
                        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)((seq % NSLOT) * ENG_SLOT_BYTES));
                        if (npc == 16) {
                            for (int j = 0; j < 16; ++j) eng_dma<true>(src + j * 1024, dst + (unsigned)(j * 1024));
                        } else {
                            for (int j = 0; j < npc; ++j) eng_dma<true>(src + j * 1024, dst + (unsigned)(j * 1024));
                        }
                        if (wb) eng_dma<false>(reinterpret_cast<const unsigned char*>(op->bias) + (int64_t)b * 16, dst + (unsigned)ENG_META_OFF);
                        commit(target, np, 1);
                    }
                }
            } else {
                const int spb = (nq + 3) >> 2;
                const int64_t block_bytes = (int64_t)(4 * nq + spb) * 1024;
                for (int bl = 0; bl < nb; ++bl) {
                    const int b = bs + bl * bstep;
                    for (int sib = 0; sib < spb; ++sib) {
                        if (!own()) {
                            ++seq;
                            continue;
                        }
                        const int nqs = min(4, nq - 4 * sib);
                        const int np = 4 * nqs + 1;
                        const uint32_t target = acquire(np, k);
                        const unsigned char* src = reinterpret_cast<const unsigned char*>(op->W) + (int64_t)b * block_bytes +
                                                   (int64_t)sib * ENG_SLOT_BYTES + c.lane * 16;
                        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)((seq % NSLOT) * ENG_SLOT_BYTES));
                        for (int j = 0; j < np - 1; ++j) eng_dma<true>(src + j * 1024, dst + (unsigned)(j * 1024));
                        eng_dma<true>(src + (np - 1) * 1024, dst + (unsigned)ENG_META_OFF);
                        commit(target, np, nqs);
                    }
                }
            }
        } else if (op->epilogue != 2) {  // (the second half of a split attention op streams nothing)
            const EngKeys ky = eng_keys<HS>(st, c.cu, c.pos);
            const int64_t grp_bytes = (int64_t)st.S * HS * 2;
            const unsigned char* kg = reinterpret_cast<const unsigned char*>(op->k_cache) + (int64_t)ky.gr * grp_bytes;
            const unsigned char* vg = reinterpret_cast<const unsigned char*>(op->v_cache) + (int64_t)ky.gr * grp_bytes;
            for (int u0 = 0; u0 < ky.nunits; u0 += 8) {
                if (!own()) {
                    ++seq;
                    continue;
                }
                const int nu = min(8, ky.nunits - u0);
                const uint32_t target = acquire(2 * nu, k);
                const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)((seq % NSLOT) * ENG_SLOT_BYTES));
                for (int uu = 0; uu < nu; ++uu) {
                    // rows past the group's last one are clamped to it (loaded, never used)
                    const int64_t off = min((int64_t)(ky.kb + (u0 + uu) * KPP) * HS * 2 + c.lane * 16, grp_bytes - 16);
                    eng_dma<false>(kg + off, dst + (unsigned)(2 * uu * 1024));
                    eng_dma<false>(vg + off, dst + (unsigned)((2 * uu + 1) * 1024));
                }
                commit(target, 2 * nu, nu);
            }
        }
    }
    while (pub < mine) publish_oldest();
}

// ------------------------------------------------------------------------------------------ consumer side
struct EngCons {
    int cw;           // consumer wave 0 .. NC - 1
    uint32_t cb_gen;  // consumer barrier generation
    int seq;          // ring sequence number of the current op's first slot
    int bc;           // running block count of this CU (result buffer index)
    float best;       // lm_head: this wave's best logit so far (its epilogue lanes)
    int best_i;
    uint64_t waited;  // diagnostic: 100 MHz ticks this wave spent waiting for ring slots in the current op
    unsigned gate_spins;  // diagnostic: polls of the gather's first granule
};

// barrier over the consumer waves (the loader never joins)
template <class CF>
__device__ __forceinline__ void eng_cbar(const EngState& st, const EngCtx& c, EngCons& w) {
    lds_drain();
    w.cb_gen += CF::NC;
    if (c.lane == 0) lds_add(c.fx + EF_CB, 1u);
    eng_wait_lds_ge_tight<CF::SPIN>(st, c, EF_CB, w.cb_gen, 0x20000000u | w.cb_gen);
}
template <class CF>
__device__ __forceinline__ void eng_wait_full(const EngState& st, const EngCtx& c, EngCons& w, int seq) {
    if ((ENG_STAMPS && st.dbg != nullptr)) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        eng_wait_lds_ge(st, c, EF_FULL + (seq % CF::NSLOT) * 4, (uint32_t)(seq + 1), 0x30000000u | (uint32_t)seq);
        w.waited += __builtin_amdgcn_s_memrealtime() - t0;
        return;
    }
    eng_wait_lds_ge(st, c, EF_FULL + (seq % CF::NSLOT) * 4, (uint32_t)(seq + 1), 0x30000000u | (uint32_t)seq);
}
template <class CF>
__device__ __forceinline__ void eng_release(const EngCtx& c, int seq) {
    lds_drain();  // this wave's reads of the slot have returned
    if (c.lane == 0) lds_add(c.fx + EF_CONS + (seq % CF::NSLOT) * 4, 1u);
}
__device__ __forceinline__ void eng_stamp(const EngState& st, const EngCtx& c, const EngCons& w, int k, int i) {
    if ((ENG_STAMPS && st.dbg != nullptr) && c.cu == 0 && w.cw == 0 && c.lane == 0) st.dbg[k * 16 + i] = __builtin_amdgcn_s_memrealtime();
    // every CU's {input ready, own units done} times of every op (the 100 MHz clock is chip-wide): who is late
    if ((ENG_STAMPS && st.dbg_all != nullptr) && (i == 1 || i == 2) && w.cw == 0 && c.lane == 0)
        st.dbg_all[((int64_t)k * ENG_WGS + c.cu) * 2 + (i - 1)] = __builtin_amdgcn_s_memrealtime();
}

// Wait until the 64 granules at p (one per lane; lanes with !need are not checked) carry this launch's tag; returns the
// data words.  v holds the first attempt.
__device__ __forceinline__ uint32_t eng_gran_wait(const EngState& st, const EngCtx& c, const uint64_t* p, uint64_t v, bool need,
                                                 uint32_t code, unsigned* nspins = nullptr) {
    unsigned spins = 0;
    uint64_t t0 = 0;
    while (!ENG_STUB_HANDOFF && !__all(!need || (uint32_t)(v >> 32) == c.epoch)) {
        __builtin_amdgcn_s_sleep(ENG_POLL_SLEEP);
        if ((++spins & 15u) == 0) {
            if (eng_aborted(c)) break;
            if (ld_err(st) != 0) {
                eng_abort_local(c);
                break;
            }
            if (eng_timed_out(t0)) {
                eng_fail(st, c, code);
                break;
            }
        }
        v = ld_gran(p);
    }
    if (nspins != nullptr) *nspins = spins;
    return (uint32_t)v;
}

// ---- input vector of a GEMV op -> LDS activation buffer (normalised bf16, per-group sums)
template <class CF>
__device__ __forceinline__ void eng_gather(const EngState& st, const EngCtx& c0, EngCons& w, const EngOp* op, int k) {
    EngCtx c = c0;
    asm volatile("" : "+v"(c.lane));  // an opaque copy per op: addresses derived from the lane are not hoisted out of the op loop
    const int K = op->K;
    const int npairs = K >> 1;
    const int ngr = (K + 127) >> 7;
    const int ngr_pad = op->nq * 8;
    uint32_t xv[CF::MAXG];
    if (op->in_embedding) {
        glb_cu32_t e32 = (glb_cu32_t)c.emb;
#pragma unroll
        for (int i = 0; i < CF::MAXG; ++i) {
            const int g = w.cw + CF::NC * i;
            xv[i] = 0;
            if (g < ngr) {
                const int pr = 64 * g + c.lane;
                const uint32_t v = e32[min(pr, npairs - 1)];
                xv[i] = pr < npairs ? v : 0u;
            }
        }
    } else {
        const uint64_t* in = op->in;
        // ONE wave polls ONE granule (the vector's last: a single 8-byte request per poll and CU - 3840 waves polling 512 B
        // each cost every hand-off its latency) until the producers are about done; the others wait on an LDS word; then
        // every wave sweeps its share until every tag matches
        if (w.cw == 0) {
            // two polls in flight (the check of one waits for that one only: loads return in order): the arrival is seen
            // half a round trip after it happened instead of a whole one
            const uint64_t* gp = in + (npairs - 1);
            unsigned gs = 0;
            uint64_t gt0 = 0;
            uint64_t va = ld_gran(gp), vb = 0;
            for (; !ENG_STUB_HANDOFF;) {
                if (ENG_GATE_DEEP) vb = ld_gran(gp);
                if ((uint32_t)(va >> 32) == c.epoch) break;
                va = ld_gran(gp);
                if (ENG_GATE_DEEP && (uint32_t)(vb >> 32) == c.epoch) break;
                if (!ENG_GATE_DEEP) __builtin_amdgcn_s_sleep(ENG_POLL_SLEEP);
                if ((++gs & 15u) == 0) {
                    if (eng_aborted(c)) break;
                    if (ld_err(st) != 0) {
                        eng_abort_local(c);
                        break;
                    }
                    if (eng_timed_out(gt0)) {
                        eng_fail(st, c, 0x40000000u | (uint32_t)k);
                        break;
                    }
                }
            }
            w.gate_spins = 2 * gs;
            if (c.lane == 0) lds_st(c.fx + EF_GATE, (uint32_t)(k + 1));
        } else {
            eng_wait_lds_ge(st, c, EF_GATE, (uint32_t)(k + 1), 0x42000000u | (uint32_t)k);
        }
        eng_stamp(st, c, w, k, 6);
        unsigned spins = 0;
        uint64_t st0 = 0;
        for (;;) {
            uint64_t gv[CF::MAXG];
#pragma unroll
            for (int i = 0; i < CF::MAXG; ++i) {
                const int g = w.cw + CF::NC * i;
                gv[i] = (uint64_t)c.epoch << 32;
                if (g < ngr) gv[i] = ld_gran(in + min(64 * g + c.lane, npairs - 1));
            }
            bool ok = true;
#pragma unroll
            for (int i = 0; i < CF::MAXG; ++i) {
                const int pr = 64 * (w.cw + CF::NC * i) + c.lane;
                ok = ok && ((uint32_t)(gv[i] >> 32) == c.epoch || pr >= npairs);
                xv[i] = pr < npairs ? (uint32_t)gv[i] : 0u;
            }
            if (ENG_STUB_HANDOFF || __all(ok)) break;
            __builtin_amdgcn_s_sleep(ENG_POLL_SLEEP);
            if ((++spins & 15u) == 0) {
                if (eng_aborted(c)) break;
                if (ld_err(st) != 0) {
                    eng_abort_local(c);
                    break;
                }
                if (eng_timed_out(st0)) {
                    eng_fail(st, c, 0x41000000u | (uint32_t)k);
                    break;
                }
            }
        }
    }
    eng_stamp(st, c, w, k, 7);
    float r = 1.f;
    NormArgs na;
    na.kind = op->norm_kind;
    na.eps = op->norm_eps;
    na.rsqrt_mode = st.rsqrt_mode;
    na.d = K;
    float mean = 0.f;
    if (na.kind != 0) {
        // row statistics: every wave's partial sum -> LDS, summed in wave order by everyone (LayerNorm: mean first, then
        // the squared deviations, two passes as norm.hip)
        float* stat = reinterpret_cast<float*>(c.fx + EF_STAT);
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < CF::MAXG; ++i) s1 += norm_stat1(xv[i], na.kind);  // zero pairs add nothing
        s1 = wave_sum_to_lane63(s1);
        if (c.lane == 63) stat[w.cw] = s1;
        eng_cbar<CF>(st, c, w);
        eng_stamp(st, c, w, k, 8);
        float tot = 0.f;
        for (int i = 0; i < CF::NC; ++i) tot += stat[i];
        if (na.kind == 2) {
            mean = tot / (float)K;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < CF::MAXG; ++i)
                if (64 * (w.cw + CF::NC * i) + c.lane < npairs) s2 += norm_stat2(xv[i], mean);
            s2 = wave_sum_to_lane63(s2);
            if (c.lane == 63) stat[16 + w.cw] = s2;
            eng_cbar<CF>(st, c, w);
            eng_stamp(st, c, w, k, 9);
            tot = 0.f;
            for (int i = 0; i < CF::NC; ++i) tot += stat[16 + i];
        }
        r = norm_scale(na, tot);
    }
    const int nb_off = ((K * 2 + 1023) >> 10) << 10;  // LayerNorm bias behind the weights
    auto round = [&](bool second) {
        const unsigned char* nslot = c.ring;
        const unsigned char* bslot = c.ring;
        const bool has_nb = na.kind == 2 && (second ? op->norm2_b : op->norm_b) != nullptr;
        const bool split = has_nb && 2 * nb_off > 16 * 1024;  // weight and bias in a slot each (K > 4096)
        if (na.kind != 0) {  // the norm weights arrived through the ring (in front of the op's weights)
            eng_wait_full<CF>(st, c, w, w.seq);
            nslot = c.ring + (w.seq % CF::NSLOT) * ENG_SLOT_BYTES;
            bslot = nslot + nb_off;
            if (split) {
                eng_wait_full<CF>(st, c, w, w.seq + 1);
                bslot = c.ring + ((w.seq + 1) % CF::NSLOT) * ENG_SLOT_BYTES;
            }
            eng_stamp(st, c, w, k, 10);
        }
        const bool to1 = (op->buf != 0) != second;
        unsigned char* dstb = to1 ? c.buf1 : c.buf0;
        float* dxs = reinterpret_cast<float*>(c.fx + EF_XS) + (to1 ? 128 : 0);
        if (CF::WF == PARROT_ENG_W_E8) {
            // ---- LLM.int8 activation quantiser (oracle/int8.py::quantize_act_rows for one row; the arithmetic of w8.hip's
            // fused kernel): fp16 cast, entries with |a| >= threshold are outliers (kept in fp16, zero in the int8 copy,
            // excluded from the absmax), the others q = rint(a * (127 / absmax)).  LDS image: int8 [nq * 2048] then the
            // outlier list {column | fp16 << 16}, in a fixed order (wave, group round, lane, element).
            // The quantiser sits on every hand-off's critical path (measured, round 3: with its per-element work stubbed the
            // token takes 1739 us instead of 2124).  Outliers are rare - a handful of feature dimensions in a trained model, none
            // at all in most 128-element groups - so both passes take a SHORT path for a wave's group without one: fp16
            // magnitudes compare as 15-bit integers (a >= thr <=> its pattern >= that of the smallest fp16 >= thr; inf / NaN
            // patterns are above every threshold), the absmax is an integer max3, no ballots, no prefix counts.  A group with
            // an outlier (or an inf / NaN) anywhere in the wave takes the long path: the arithmetic below, unchanged.
            unsigned char* q8 = c.fx + ef_q8<CF>();
            const float thr = op->threshold;
            const uint32_t thr16 = thr > 0.f ? (uint32_t)__half_as_ushort(__float2half_ru(thr)) : 0x7c00u;  // (no threshold: only inf / NaN leave the short path)
            uint32_t hv[CF::MAXG];
            float mx = 0.f;
            uint32_t mxb = 0u;  // pattern of the largest magnitude seen on the short path
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < CF::MAXG; ++i) {
                const int g = w.cw + CF::NC * i;
                hv[i] = 0u;
                if (!ENG_STUB_Q8 && g < ngr) {
                    uint32_t o = xv[i];
                    const int pr = 64 * g + c.lane;
                    if (na.kind != 0) {
                        const uint32_t nwp = *reinterpret_cast<const uint32_t*>(nslot + min(pr, npairs - 1) * 4);
                        uint32_t nbp = 0u;
                        if (has_nb) nbp = *reinterpret_cast<const uint32_t*>(bslot + min(pr, npairs - 1) * 4);
                        o = norm_apply(o, nwp, nbp, na.kind, mean, r);
                    }
                    if (pr >= npairs) o = 0u;
                    const __half h0 = __float2half(bflo(o)), h1 = __float2half(bfhi(o));
                    hv[i] = (uint32_t)__half_as_ushort(h0) | ((uint32_t)__half_as_ushort(h1) << 16);
                    const uint32_t m0 = hv[i] & 0x7fffu, m1 = (hv[i] >> 16) & 0x7fffu;
                    if (!__any(m0 >= thr16 || m1 >= thr16)) {
                        mxb = max(mxb, max(m0, m1));
                        continue;
                    }
                    const float a0 = fabsf(__half2float(h0)), a1 = fabsf(__half2float(h1));
                    const bool o0 = thr > 0.f && a0 >= thr, o1 = thr > 0.f && a1 >= thr;
                    if (!o0) mx = fmaxf(mx, a0);
                    if (!o1) mx = fmaxf(mx, a1);
                    cnt += __popcll(__ballot(o0)) + __popcll(__ballot(o1));
                }
            }
            mx = fmaxf(mx, __half2float(__ushort_as_half((unsigned short)mxb)));
            // (non-negative values: the DPP moves' zero fill is neutral; the wave's maximum lands in lane 63)
            mx = fmaxf(mx, dpp0<0xB1>(mx));
            mx = fmaxf(mx, dpp0<0x4E>(mx));
            mx = fmaxf(mx, dpp0<0x141>(mx));
            mx = fmaxf(mx, dpp0<0x140>(mx));
            mx = fmaxf(mx, dpp0<0x142, 0xA>(mx));
            mx = fmaxf(mx, dpp0<0x143, 0xC>(mx));
            const int eq_r = second ? EQ_ROUND2 : 0;  // this round's set of the waves' maxima / counts
            if (c.lane == 63) {
                reinterpret_cast<float*>(q8 + eq_r + EQ_MAX)[w.cw] = mx;
                reinterpret_cast<int*>(q8 + eq_r + EQ_CNT)[w.cw] = cnt;
            }
            if (na.kind != 0) {
                eng_release<CF>(c, w.seq);
                if (split) eng_release<CF>(c, w.seq + 1);
                w.seq += split ? 2 : 1;
            }
            eng_cbar<CF>(st, c, w);
            int base = 0, total = 0;
            mx = 0.f;
            for (int t = 0; t < CF::NC; ++t) {
                const int ct = reinterpret_cast<const int*>(q8 + eq_r + EQ_CNT)[t];
                if (t < w.cw) base += ct;
                total += ct;
                mx = fmaxf(mx, reinterpret_cast<const float*>(q8 + eq_r + EQ_MAX)[t]);
            }
            const float inv = mx > 0.f ? __fdiv_rn(127.0f, mx) : 0.f;
            uint32_t* olist = reinterpret_cast<uint32_t*>(dstb + op->nq * 2048);
            const uint64_t lt = (1ull << c.lane) - 1ull;
#pragma unroll
            for (int i = 0; i < CF::MAXG; ++i) {
                const int g = w.cw + CF::NC * i;
                if (ENG_STUB_Q8 && g < op->nq * 16) *reinterpret_cast<uint16_t*>(dstb + (64 * g + c.lane) * 2) = 0;
                if (!ENG_STUB_Q8 && g < op->nq * 16) {  // (the padding up to whole units is written as zeros)
                    const float a0 = __half2float(__ushort_as_half((unsigned short)(hv[i] & 0xffffu)));
                    const float a1 = __half2float(__ushort_as_half((unsigned short)(hv[i] >> 16)));
                    if (!__any((hv[i] & 0x7fffu) >= thr16 || ((hv[i] >> 16) & 0x7fffu) >= thr16)) {  // the short path: no outlier in the wave's group
                        const int q0 = (int)rintf(__fmul_rn(a0, inv)), q1 = (int)rintf(__fmul_rn(a1, inv));
                        *reinterpret_cast<uint16_t*>(dstb + (64 * g + c.lane) * 2) = (uint16_t)((q0 & 0xff) | ((q1 & 0xff) << 8));
                        continue;
                    }
                    const bool o0 = thr > 0.f && fabsf(a0) >= thr, o1 = thr > 0.f && fabsf(a1) >= thr;
                    const int q0 = o0 ? 0 : (int)rintf(__fmul_rn(a0, inv)), q1 = o1 ? 0 : (int)rintf(__fmul_rn(a1, inv));
                    *reinterpret_cast<uint16_t*>(dstb + (64 * g + c.lane) * 2) = (uint16_t)((q0 & 0xff) | ((q1 & 0xff) << 8));
                    const uint64_t b0 = __ballot(o0), b1 = __ballot(o1);
                    if (b0 | b1) {
                        const int col = 2 * (64 * g + c.lane);
                        const int p0 = base + __popcll(b0 & lt) + __popcll(b1 & lt);
                        if (o0 && p0 < ENG_Q8_CAP) olist[p0] = (uint32_t)col | (hv[i] << 16);
                        if (o1 && p0 + (o0 ? 1 : 0) < ENG_Q8_CAP) olist[p0 + (o0 ? 1 : 0)] = (uint32_t)(col + 1) | (hv[i] & 0xffff0000u);
                        base += __popcll(b0) + __popcll(b1);
                    }
                }
            }
            if (w.cw == 0 && c.lane == 0) {
                reinterpret_cast<float*>(q8 + EQ_SA)[to1 ? 1 : 0] = mx;
                reinterpret_cast<int*>(q8 + EQ_NO)[to1 ? 1 : 0] = min(total, ENG_Q8_CAP);
                if (total > ENG_Q8_CAP) eng_fail(st, c, 0x70000000u | (uint32_t)k);  // more outliers than the list holds
            }
            eng_stamp(st, c, w, k, 11);
            return;
        }
#pragma unroll
        for (int i = 0; i < CF::MAXG; ++i) {
            const int g = w.cw + CF::NC * i;
            if (g < ngr_pad) {
                uint32_t o = 0;
                if (g < ngr) {
                    o = xv[i];
                    const int pr = 64 * g + c.lane;
                    if (na.kind != 0) {
                        const uint32_t nwp = *reinterpret_cast<const uint32_t*>(nslot + min(pr, npairs - 1) * 4);
                        uint32_t nbp = 0u;
                        if (has_nb) nbp = *reinterpret_cast<const uint32_t*>(bslot + min(pr, npairs - 1) * 4);
                        o = pr < npairs ? norm_apply(o, nwp, nbp, na.kind, mean, r) : 0u;
                    }
                }
                *reinterpret_cast<uint32_t*>(dstb + g * ENG_GROUP_STRIDE + c.lane * 4) = o;
                if (CF::WF == PARROT_ENG_W_E4) {  // the int4 arithmetic needs the groups' sums
                    const float t = wave_sum_to_lane63(bflo(o) + bfhi(o));
                    if (c.lane == 63) dxs[g] = t;
                }
            }
        }
        eng_stamp(st, c, w, k, 11);
        if (na.kind != 0) {
            eng_release<CF>(c, w.seq);
            if (split) eng_release<CF>(c, w.seq + 1);
            w.seq += split ? 2 : 1;
        }
    };
    round(false);
    // a second norm of the same input (parallel residual: the MLP's norm_2; the row statistics are the input's): the
    // up-projection that follows finds its input in the other buffer and gathers nothing.  (Straight-line, not a loop over
    // the two: the loop cost the models without a second norm 6 % of their token rate.)
    if (na.kind != 0 && op->norm2_w != nullptr) round(true);
    eng_cbar<CF>(st, c, w);
}

// ---- one Linear: the CU's blocks, units dealt round-robin over the consumer waves
template <class CF>
__device__ __forceinline__ void eng_gemv(const EngState& st, const EngCtx& c0, EngCons& w, const EngOp* op, int k) {
    constexpr int MAXQ = CF::MAXQ, NSLOT = CF::NSLOT;
    eng_stamp(st, c0, w, k, 0);
    if (!op->no_gather) eng_gather<CF>(st, c0, w, op, k);  // (else: the previous Linear's gather left this op's input in its buffer)
    EngCtx c = c0;
    asm volatile("" : "+v"(c.lane));
    if (w.cw == 0 && c.lane == 0) lds_st(c.fx + EF_CUR, (uint32_t)k);
    eng_stamp(st, c, w, k, 1);
    int bs, nb, bstep;
    eng_block_range(op, c.cu, bs, nb, bstep);
    constexpr bool e16 = CF::WF != PARROT_ENG_W_E4;  // one unit per slot (bf16 and int8 weights)
    constexpr bool e8 = CF::WF == PARROT_ENG_W_E8;
    const int nq = op->nq, spb = e16 ? nq : (nq + 3) >> 2;
    const unsigned char* buf = op->buf ? c.buf1 : c.buf0;
    const float* xs = reinterpret_cast<const float*>(c.fx + EF_XS) + op->buf * 128;
    float* red = reinterpret_cast<float*>(c.fx + EF_RED);
    float* redb = reinterpret_cast<float*>(c.fx + EF_REDB);
    float* resid = reinterpret_cast<float*>(c.fx + EF_RESID);
    const int r = c.lane & 7, p = c.lane >> 3;
    const int epi = op->epilogue;
    const bool has_bias = !e8 && op->bias != nullptr;  // (E8: that field holds the rows' scales)
    const int total = nb * nq;
    // a K-chunk op keeps its rows' running sums in result slot MAXQ - 1 of the rows' 8 block buffers: the chunk must leave that
    // slot free and the CU must own at most 8 blocks (the host builds the table that way; a table that does not is refused here)
    if (op->acc != 0 && (nq >= MAXQ || nb > 8) && w.cw == 0 && c.lane == 0) eng_fail(st, c, 0x71000000u | (uint32_t)k);
    // units are dealt round-robin over the consumer waves: wave cw takes units cw, cw + NC, ...; (bl, Q) = (local block,
    // unit of the row) are stepped without a division per unit
    int bl = w.cw / nq, Q = w.cw - bl * nq;
    const int step_b = CF::NC / nq, step_q = CF::NC - step_b * nq;
    for (int idx = w.cw; idx < total; idx += CF::NC) {
        const int b = bs + bl * bstep;
        const int rb = (w.bc + bl) % ENG_RED;
        const int seq = w.seq + bl * spb + (e16 ? Q : (Q >> 2));
        eng_wait_full<CF>(st, c, w, seq);
        const unsigned char* slot = c.ring + (seq % NSLOT) * ENG_SLOT_BYTES;
        float v;
        float scb_r = 0.f;  // E8: the weight row's scale
        if (e8) {
            // int8 unit = the slot's pieces: lane (r, p) holds columns 2048 Q + 128 i + 16 p .. + 15 of row r in piece i.
            // C32 is exact; the outlier columns (fp16 activations x dequantised weights, LLM.int8's mixed-precision part) that
            // fall into this unit are summed by the row's first lane in list order
            const int npc = min(16, ((op->K + 127) >> 7) - 16 * Q);
            const unsigned char* wq = slot + c.lane * 16;
            const unsigned char* xq = buf + 2048 * Q + p * 16;
            scb_r = *reinterpret_cast<const float*>(slot + ENG_META_OFF + r * 4);
            int acc = 0;
#pragma unroll 4
            for (int i = 0; i < npc; ++i) {
                const uint4 wv = *reinterpret_cast<const uint4*>(wq + i * 1024);
                const uint4 xv = *reinterpret_cast<const uint4*>(xq + i * 128);
                acc = __builtin_amdgcn_sdot4((int)wv.x, (int)xv.x, acc, false);
                acc = __builtin_amdgcn_sdot4((int)wv.y, (int)xv.y, acc, false);
                acc = __builtin_amdgcn_sdot4((int)wv.z, (int)xv.z, acc, false);
                acc = __builtin_amdgcn_sdot4((int)wv.w, (int)xv.w, acc, false);
            }
            acc += __builtin_amdgcn_update_dpp(0, acc, 0x128, 0xF, 0xF, true);  // row_ror:8, then the two lane swaps: 8 lanes of a row
            {
                const auto sw = __builtin_amdgcn_permlane16_swap((unsigned)acc, (unsigned)acc, false, false);
                acc = (int)sw[0] + (int)sw[1];
            }
            {
                const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)acc, (unsigned)acc, false, false);
                acc = (int)sw[0] + (int)sw[1];
            }
            const int no = *reinterpret_cast<const int*>(c.fx + ef_q8<CF>() + EQ_NO + op->buf * 4);
            float osum = 0.f;
            if (no > 0) {
                // the 8 lanes of a row share the list (entries p, p + 8, ...), then one sum over them: a fixed order
                const uint32_t* olist = reinterpret_cast<const uint32_t*>(buf + nq * 2048);
                for (int i = p; i < no; i += 8) {
                    const uint32_t e = olist[i];
                    const int col = (int)(e & 0xffffu) - 2048 * Q;
                    if (col >= 0 && col < 2048) {
                        const int wb = *reinterpret_cast<const signed char*>(slot + (col >> 7) * 1024 + ((((col & 127) >> 4) << 3) + r) * 16 + (col & 15));
                        osum += __half2float(__ushort_as_half((unsigned short)(e >> 16))) * eng_rhalf(__fdiv_rn(__fmul_rn((float)wb, scb_r), 127.0f));
                    }
                }
                osum = row8_allsum(osum);
            }
            if (c.lane < 8) {
                reinterpret_cast<int*>(red)[(rb * MAXQ + Q) * 8 + c.lane] = acc;
                reinterpret_cast<float*>(c.fx + ef_red2<CF>())[(rb * MAXQ + Q) * 8 + c.lane] = osum;
            }
            v = 0.f;
        } else if (e16) {
            // bf16 unit = the slot's 16 pieces: lane (r, p) holds columns 1024 Q + 64 i + 8 p .. + 7 of row r in piece i
            const unsigned char* wq = slot + c.lane * 16;
            const unsigned char* xq = buf + (8 * Q) * ENG_GROUP_STRIDE + p * 16;
            float p0 = 0.f, p1 = 0.f;
            const int npc = min(16, ((op->K + 63) >> 6) - 16 * Q);  // pieces of this unit (a row's last unit may be short)
#pragma unroll 1
            for (int i4 = 0; i4 + 4 <= npc; i4 += 4) {  // four pieces (32 registers of operands) per round
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int i = i4 + ii;
                    const uint4 wv = *reinterpret_cast<const uint4*>(wq + i * 1024);
                    const uint4 xv = *reinterpret_cast<const uint4*>(xq + (i >> 1) * ENG_GROUP_STRIDE + (ii & 1) * 128);
                    p0 = dot2_bf16(wv.x, xv.x, p0);
                    p1 = dot2_bf16(wv.y, xv.y, p1);
                    p0 = dot2_bf16(wv.z, xv.z, p0);
                    p1 = dot2_bf16(wv.w, xv.w, p1);
                }
            }
#pragma unroll 1
            for (int i = npc & ~3; i < npc; ++i) {  // (the pieces a short unit does not have were never loaded: not read)
                const uint4 wv = *reinterpret_cast<const uint4*>(wq + i * 1024);
                const uint4 xv = *reinterpret_cast<const uint4*>(xq + (i >> 1) * ENG_GROUP_STRIDE + (i & 1) * 128);
                p0 = dot2_bf16(wv.x, xv.x, p0);
                p1 = dot2_bf16(wv.y, xv.y, p1);
                p0 = dot2_bf16(wv.z, xv.z, p0);
                p1 = dot2_bf16(wv.w, xv.w, p1);
            }
            v = p0 + p1;
            if (Q == 0 && has_bias && c.lane < 8) redb[rb * 8 + c.lane] = bf2f(*reinterpret_cast<const bf16_t*>(slot + ENG_META_OFF + c.lane * 2));
        } else if (ENG_STUB_UNITS) {
            v = 0.f;  // (diagnostic build: no arithmetic, the slot is only awaited and released)
        } else {
            const int qq = Q & 3;
            const uint32_t mt = *reinterpret_cast<const uint32_t*>(slot + ENG_META_OFF + (qq * 64 + c.lane) * 4);
            const int G = 8 * Q + p;
            const unsigned char* xg = buf + G * ENG_GROUP_STRIDE;
            const unsigned char* wq = slot + qq * 4096 + c.lane * 16;
            // one accumulator pair over the quad's four pieces (= the lane's whole quantisation group)
            float p0 = 0.f, p1 = 0.f;
            const uint32_t mask = 0x000F000Fu;
            uint32_t magic = 0x43004300u;
            asm("" : "+v"(magic));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 wv = *reinterpret_cast<const uint4*>(wq + i * 1024);
                const uint32_t dw[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint4 xv = *reinterpret_cast<const uint4*>(xg + i * 64 + d * 16);
                    p0 = dot2_bf16(and_or(dw[d], mask, magic), xv.x, p0);
                    p1 = dot2_bf16(and_or(dw[d] >> 4, mask, magic), xv.y, p1);
                    p0 = dot2_bf16(and_or(dw[d] >> 8, mask, magic), xv.z, p0);
                    p1 = dot2_bf16(and_or(dw[d] >> 12, mask, magic), xv.w, p1);
                }
            }
            const float acc = p0 + p1;
            v = bflo(mt) * (acc - (128.0f + bfhi(mt)) * xs[G]);
        }
        if (!e8) {
            v = row8_allsum(v);
            if (c.lane < 8) red[(rb * MAXQ + Q) * 8 + c.lane] = v;
        }
        lds_drain();  // this wave's reads of the slot have returned, its partial sums are written
        uint32_t t = 0;
        if (c.lane == 0) {
            lds_add(c.fx + EF_CONS + (seq % NSLOT) * 4, 1u);
            t = lds_add(c.fx + EF_DONE + rb * 4, 1u);
        }
        t = __builtin_amdgcn_readfirstlane(t);
        if ((int)t == nq - 1) {
            // ---- this wave finished the block's last unit: fixed-order sum, epilogue, publish
            asm volatile("" ::: "memory");  // the partial sums are read behind the counter, not speculated above it
            lds_st(c.fx + EF_DONE + rb * 4, 0u);
            float a = 0.f;
            if (e8) {
                // mm_dequant's separately rounded products (w8.hip), then the outlier part, both rounded to fp16
                int c32 = 0;
                float osum = 0.f;
                for (int q = 0; q < nq; ++q) {
                    c32 += reinterpret_cast<const int*>(red)[(rb * MAXQ + q) * 8 + r];
                    osum += reinterpret_cast<const float*>(c.fx + ef_red2<CF>())[(rb * MAXQ + q) * 8 + r];
                }
                const float sa = *reinterpret_cast<const float*>(c.fx + ef_q8<CF>() + EQ_SA + op->buf * 4);
                a = eng_rhalf(__fmul_rn(__fmul_rn(__fmul_rn((float)c32, ENG_MM_DEQUANT), sa), scb_r));
                if (*reinterpret_cast<const int*>(c.fx + ef_q8<CF>() + EQ_NO + op->buf * 4) > 0) a = eng_rhalf(a + eng_rhalf(osum));
            } else {
                for (int q = 0; q < nq; ++q) a += red[(rb * MAXQ + q) * 8 + r];
            }
            if (op->acc != 0) {
                // this op is one K-chunk of a Linear whose input does not fit LDS (Falcon-40B's down-projection, K = 32768):
                // the chunks' sums meet in a per-row accumulator - the result slot of unit MAXQ - 1, which a chunk (nq <
                // MAXQ, checked on the host) never fills - in chunk order; the last chunk goes on to the epilogue
                const int lr = (bl * 8 + r) & 63;
                float* ca = red + (((lr >> 3) * MAXQ + (MAXQ - 1)) * 8 + (lr & 7));
                if (op->acc >= 2) a += *ca;
                if (op->acc <= 2) {
                    if (c.lane < 8) *ca = a;
                    goto unit_done;
                }
            }
            if (has_bias) a += redb[rb * 8 + r];
            float o = rbf(a);
            if (epi == PARROT_EPI_RESIDUAL) {
                const int lr = (bl * 8 + r) & 63;  // the CU's own rows: the same blocks in every n_embd-row op
                const float res = op->res_embedding ? bf2f(((glb_cu16_t)c.emb)[b * 8 + r]) : resid[op->res_in * 64 + lr];
                o = rbf(res + o);
                if (c.lane < 8) resid[op->res_out * 64 + lr] = o;
            } else if (epi == PARROT_EPI_GELU) {
                o = gelu_erf(o);
            } else if (epi == PARROT_EPI_SWIGLU) {
                const float gate = __shfl(o, (c.lane + 4) & 63, 64);  // lanes 0..3: fc_1 rows, 4..7: the same rows of fc_2
                o = rbf(rbf(silu(o)) * gate);
            }
            const uint32_t ob = f2bf(o);
            const uint32_t nb = (uint32_t)__shfl((int)ob, (c.lane + 1) & 63, 64);
            if (epi == PARROT_ENG_EPI_LOGITS) {
                if (c.lane < 8) {
                    ((__attribute__((address_space(1))) bf16_t*)op->out)[b * 8 + r] = (bf16_t)ob;
                    float lv = bf2f((bf16_t)ob);
                    if (lv != lv) lv = -INFINITY;
                    const int li = b * 8 + r;
                    if (li < st.V && (w.best_i == 0x7fffffff || lv > w.best || (lv == w.best && li < w.best_i))) {
                        w.best = lv;
                        w.best_i = li;
                    }
                }
            } else if (op->publish) {
                const int rows = epi == PARROT_EPI_SWIGLU ? 4 : 8;
                if (c.lane < rows && (c.lane & 1) == 0)
                    st_gran(reinterpret_cast<uint64_t*>(op->out) + ((b * rows + c.lane) >> 1), ob | (nb << 16), c.epoch);
            }
        }
    unit_done:
        bl += step_b;
        Q += step_q;
        if (Q >= nq) {
            Q -= nq;
            ++bl;
        }
    }
    w.seq += nb * spb;
    w.bc += nb;
    eng_stamp(st, c, w, k, 2);
    if ((ENG_STAMPS && st.dbg != nullptr) && c.cu == 0 && w.cw == 0 && c.lane == 0) st.dbg[k * 16 + 3] = w.waited | ((uint64_t)w.gate_spins << 40);
    w.waited = 0;
}

// ---- attention op: split + RoPE + KV append + softmax(q k^T / sqrt(hs)) v over the CU's key range, partial states to the
// group's leader CU, which merges them into the heads
// Two halves, run back to back for the sequential-residual block (phase 0) or as two ops with a part of the MLP's
// up-projection between them (parallel residual: phases 1 and 2 - the partial states travel while weights stream):
//   local:   the CU's key range -> its partial state, to the group's leader (or kept in LDS when the CU is the only split)
//   combine: the leader merges the splits into the heads and hands them over
// The scratch lives in LDS activation buffer op->buf; op->no_gather = 1 asks for a barrier first (the buffer's last readers
// may still be at work: the waves come here straight from another Linear's units).
template <class CF, int HS, int HQ>
__device__ __forceinline__ bool eng_attn_local(const EngState& st, const EngCtx& c0, EngCons& w, const EngOp* op, int k) {
    constexpr int LPR = HS / 8, KPP = 64 / LPR, PW = HS + 2;
    EngCtx c = c0;
    asm volatile("" : "+v"(c.lane));
    eng_stamp(st, c, w, k, 0);
    const EngKeys ky = eng_keys<HS>(st, c.cu, c.pos);
    if (op->no_gather) eng_cbar<CF>(st, c, w);
    if (!ky.part) return false;
    unsigned char* sc = op->buf ? c.buf1 : c.buf0;
    uint32_t* raw = reinterpret_cast<uint32_t*>(sc);                                      // [(HQ + 2) * HS / 2] bf16 pairs
    float* wpart = reinterpret_cast<float*>(sc + (HQ + 2) * HS * 2);                      // [NC][HQ][PW]
    float* stage = wpart + CF::NC * HQ * PW;                                              // [HQ][nsplit][PW]
    const float* rope = reinterpret_cast<const float*>(c.fx + EF_ROPE);
    const int dl = c.lane % LPR, j = c.lane / LPR;
    const int n_elem = st.n_elem, half_n = n_elem >> 1;

    // ---- the group's rows of the QKV vector
    {
        // rows of the K/V group in the QKV vector: q_per_kv query heads, k, v (model.py:204-213); this virtual group's are
        // query heads jv * HQ .. + HQ - 1, then k, v
        constexpr int NPAIR = (HQ + 2) * HS / 2, NLOAD = (NPAIR + 63) / 64, RP = HS / 2;
        const uint64_t* in = op->in + (int64_t)ky.gr * (st.q_per_kv + 2) * RP;
        auto src_pair = [&](int pr) {  // pair pr of the virtual group's rows -> its place in the K/V group's rows
            const int rr = pr / RP;
            return (rr < HQ ? ky.jv * HQ + rr : st.q_per_kv + (rr - HQ)) * RP + (pr - rr * RP);
        };
        if (!ENG_ATTN_GATE) {
        } else if (w.cw == 0) {  // one poller per CU (see eng_gather)
            (void)eng_gran_wait(st, c, in + src_pair(NPAIR - 1), 0, true, 0x52000000u | (uint32_t)k);
            if (c.lane == 0) lds_st(c.fx + EF_GATE, (uint32_t)(k + 1));
        } else if (w.cw < NLOAD) {
            eng_wait_lds_ge(st, c, EF_GATE, (uint32_t)(k + 1), 0x53000000u | (uint32_t)k);
        }
        for (int t = w.cw; t < NLOAD; t += CF::NC) {
            const int pr = 64 * t + c.lane;
            const uint64_t* p = in + src_pair(min(pr, NPAIR - 1));
            const uint32_t d = eng_gran_wait(st, c, p, ld_gran(p), pr < NPAIR, 0x50000000u | (uint32_t)k);
            if (pr < NPAIR) raw[pr] = d;
        }
    }
    eng_cbar<CF>(st, c, w);
    eng_stamp(st, c, w, k, 1);
    if (w.cw == 0 && c.lane == 0) lds_st(c.fx + EF_CUR, (uint32_t)k);
    // this lane's 8 dims of every query head (RoPE, rounded to bf16, scaled), of the new key (RoPE) and the new value.
    // The rotary width is a multiple of 16 (checked on the host), so a lane's 8 dims lie on one side of the rotation.
    const bf16_t* rawb = reinterpret_cast<const bf16_t*>(raw);
    const int d0 = dl * 8;
    const bool rot = d0 < n_elem, lo = d0 < half_n;
    const int dpart = rot ? (lo ? d0 + half_n : d0 - half_n) : d0;
    float cs[8], sn[8];
    {
        const int dr = rot ? d0 : 0;
        const float4 c0 = *reinterpret_cast<const float4*>(rope + dr), c1 = *reinterpret_cast<const float4*>(rope + dr + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(rope + 128 + dr), s1 = *reinterpret_cast<const float4*>(rope + 128 + dr + 4);
        cs[0] = c0.x; cs[1] = c0.y; cs[2] = c0.z; cs[3] = c0.w; cs[4] = c1.x; cs[5] = c1.y; cs[6] = c1.z; cs[7] = c1.w;
        sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
    }
    auto rope8 = [&](int row, uint32_t (&out)[4]) {  // the lane's 8 dims of a row, rotated, as 4 bf16 pairs
        const uint4 av = *reinterpret_cast<const uint4*>(rawb + row * HS + d0);
        const uint4 bv = *reinterpret_cast<const uint4*>(rawb + row * HS + dpart);
        const uint32_t a[4] = {av.x, av.y, av.z, av.w}, bq[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float o0 = bflo(bq[e]), o1 = bfhi(bq[e]);
            if (lo) {
                o0 = -o0;
                o1 = -o1;
            }
            // x * cos + rotate_half(x) * sin: both products and the sum rounded separately (no FMA), as the reference's promoted ops
            const float r0 = __fadd_rn(__fmul_rn(bflo(a[e]), cs[2 * e]), __fmul_rn(o0, sn[2 * e]));
            const float r1 = __fadd_rn(__fmul_rn(bfhi(a[e]), cs[2 * e + 1]), __fmul_rn(o1, sn[2 * e + 1]));
            const uint32_t pk = (uint32_t)f2bf(r0) | ((uint32_t)f2bf(r1) << 16);
            out[e] = rot ? pk : a[e];
        }
    };
    const float scale = 1.0f / sqrtf((float)HS);
    float qf[HQ][8];
    uint32_t knew[4], vnew[4];
    rope8(HQ, knew);
    {
        const uint4 vv = *reinterpret_cast<const uint4*>(rawb + (HQ + 1) * HS + d0);
        vnew[0] = vv.x; vnew[1] = vv.y; vnew[2] = vv.z; vnew[3] = vv.w;
    }
#pragma unroll
    for (int h = 0; h < HQ; ++h) {
        uint32_t qp[4];
        rope8(h, qp);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qf[h][2 * e] = bflo(qp[e]) * scale;
            qf[h][2 * e + 1] = bfhi(qp[e]) * scale;
        }
    }
    const int slot_new = c.pos % st.S;
    if (w.cw == 0 && c.lane < LPR && ky.jv == 0 && slot_new >= ky.kb && slot_new < ky.ke) {  // KV append by the owner of the new slot
        const int64_t off = ((int64_t)ky.gr * st.S + slot_new) * HS + dl * 8;
        typedef __attribute__((address_space(1))) u32x4_t* glb_u32x4_t;
        *(glb_u32x4_t)(reinterpret_cast<bf16_t*>(op->k_cache) + off) = u32x4_t{knew[0], knew[1], knew[2], knew[3]};
        *(glb_u32x4_t)(reinterpret_cast<bf16_t*>(op->v_cache) + off) = u32x4_t{vnew[0], vnew[1], vnew[2], vnew[3]};
    }

    float m[HQ], l[HQ], acc[HQ][8];
#pragma unroll
    for (int h = 0; h < HQ; ++h) {
        m[h] = -INFINITY;
        l[h] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[h][e] = 0.f;
    }
    for (int u = w.cw; u < ky.nunits; u += CF::NC) {
        const int seq = w.seq + (u >> 3);
        eng_wait_full<CF>(st, c, w, seq);
        const unsigned char* slot = c.ring + (seq % CF::NSLOT) * ENG_SLOT_BYTES + (u & 7) * 2048;
        uint4 kv = *reinterpret_cast<const uint4*>(slot + c.lane * 16);
        uint4 vv = *reinterpret_cast<const uint4*>(slot + 1024 + c.lane * 16);
        const int kidx = ky.kb + u * KPP + j;
        const bool ok = kidx < ky.ke;
        if (kidx == slot_new) {
            kv = make_uint4(knew[0], knew[1], knew[2], knew[3]);
            vv = make_uint4(vnew[0], vnew[1], vnew[2], vnew[3]);
        }
        const uint32_t kd[4] = {kv.x, kv.y, kv.z, kv.w}, vd[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int h = 0; h < HQ; ++h) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s = fmaf(qf[h][2 * e], bflo(kd[e]), s);
                s = fmaf(qf[h][2 * e + 1], bfhi(kd[e]), s);
            }
            s = eng_group_sum<LPR>(s);
            if (ok) {
                const float mn = fmaxf(m[h], s);
                const float corr = __expf(m[h] - mn), pp = __expf(s - mn);
                l[h] = l[h] * corr + pp;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[h][2 * e] = acc[h][2 * e] * corr + pp * bflo(vd[e]);
                    acc[h][2 * e + 1] = acc[h][2 * e + 1] * corr + pp * bfhi(vd[e]);
                }
                m[h] = mn;
            }
        }
        eng_release<CF>(c, seq);
    }
    w.seq += (ky.nunits + 7) >> 3;
    eng_stamp(st, c, w, k, 6);
    // ---- merge the key rows of the wave, then the waves of the CU
#pragma unroll
    for (int h = 0; h < HQ; ++h) {
        const float M = eng_slot_allreduce<LPR, true>(m[h]);
        const float f = (m[h] == -INFINITY) ? 0.f : __expf(m[h] - M);
        const float L = eng_slot_allreduce<LPR, false>(l[h] * f);
        float* wp = wpart + (w.cw * HQ + h) * PW;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = eng_slot_allreduce<LPR, false>(acc[h][e] * f);
            if (c.lane < LPR) wp[dl * 8 + e] = a;
        }
        if (c.lane == 0) {
            wp[HS] = M;
            wp[HS + 1] = L;
        }
    }
    eng_cbar<CF>(st, c, w);
    eng_stamp(st, c, w, k, 7);
    if (w.cw == 0) {
#pragma unroll
        for (int h = 0; h < HQ; ++h) {
            float M = -INFINITY;
            for (int t = 0; t < CF::NC; ++t) M = fmaxf(M, wpart[(t * HQ + h) * PW + HS]);
            float L = 0.f, o0 = 0.f, o1 = 0.f;
            for (int t = 0; t < CF::NC; ++t) {
                const float* wp = wpart + (t * HQ + h) * PW;
                const float f = (wp[HS] == -INFINITY) ? 0.f : __expf(wp[HS] - M);
                L += wp[HS + 1] * f;
                o0 += wp[c.lane % HS] * f;
                if (HS > 64) o1 += wp[(c.lane + 64) % HS] * f;
            }
            if (ky.ns == 1) {  // the only split: the state stays in LDS for the combine below
                float* sg = stage + h * PW;
                if (c.lane < HS) sg[c.lane] = o0;
                if (HS > 64) sg[64 + c.lane] = o1;
                if (c.lane == 0) sg[HS] = M;
                if (c.lane == 1) sg[HS + 1] = L;
                continue;
            }
            uint64_t* pg = op->part + ((int64_t)(ky.g * HQ + h) * ky.ns + ky.s) * PW;
            if (c.lane < HS) st_gran(pg + c.lane, __float_as_uint(o0), c.epoch);
            if (HS > 64) st_gran(pg + 64 + c.lane, __float_as_uint(o1), c.epoch);
            if (c.lane == 0) st_gran(pg + HS, __float_as_uint(M), c.epoch);
            if (c.lane == 1) st_gran(pg + HS + 1, __float_as_uint(L), c.epoch);
        }
    }
    return true;
}

template <class CF, int HS, int HQ>
__device__ __forceinline__ void eng_attn_combine(const EngState& st, const EngCtx& c0, EngCons& w, const EngOp* op, int k) {
    constexpr int PW = HS + 2;
    EngCtx c = c0;
    asm volatile("" : "+v"(c.lane));
    const EngKeys ky = eng_keys<HS>(st, c.cu, c.pos);
    unsigned char* sc = op->buf ? c.buf1 : c.buf0;
    float* stage = reinterpret_cast<float*>(sc + (HQ + 2) * HS * 2) + CF::NC * HQ * PW;
    // the group's leader CU merges the splits into the heads: every wave fetches a share of the partial states.  The leader is
    // split g % ns of group g: with split 0 every leader sat on XCD 0 (workgroup id % 8), whose CUs then ran late into
    // every later hand-off
    if (ky.s == ky.g % ky.ns) {
        const int cnt = HQ * ky.ns * PW;  // the group's heads lie back to back
        const uint64_t* pg = op->part + (int64_t)ky.g * HQ * ky.ns * PW;
        constexpr int NLD = (2 * 8 * PW + 63) / 64, NPW = (NLD + CF::NC - 1) / CF::NC;  // HQ <= 2, nsplit <= 8
        if (!ENG_ATTN_GATE) {
        } else if (w.cw == 0) {  // one poller per CU
            (void)eng_gran_wait(st, c, pg + (cnt - 1), 0, true, 0x54000000u | (uint32_t)k);
            if (c.lane == 0) lds_st(c.fx + EF_GATE2, (uint32_t)(k + 1));
        } else {
            eng_wait_lds_ge(st, c, EF_GATE2, (uint32_t)(k + 1), 0x55000000u | (uint32_t)k);
        }
        unsigned spins = 0;
        uint64_t ct0 = 0;
        for (; ky.ns > 1;) {  // flat sweep: every load in flight at once, repeated until every tag matches
            uint64_t gv[NPW];
#pragma unroll
            for (int t = 0; t < NPW; ++t) gv[t] = ld_gran(pg + min(64 * (w.cw + CF::NC * t) + c.lane, cnt - 1));
            bool ok = true;
#pragma unroll
            for (int t = 0; t < NPW; ++t) {
                const int i = 64 * (w.cw + CF::NC * t) + c.lane;
                ok = ok && ((uint32_t)(gv[t] >> 32) == c.epoch || i >= cnt);
                if (i < cnt) stage[i] = __uint_as_float((uint32_t)gv[t]);
            }
            if (ENG_STUB_HANDOFF || __all(ok)) break;
            __builtin_amdgcn_s_sleep(ENG_POLL_SLEEP);
            if ((++spins & 15u) == 0) {
                if (eng_aborted(c)) break;
                if (ld_err(st) != 0) {
                    eng_abort_local(c);
                    break;
                }
                if (eng_timed_out(ct0)) {
                    eng_fail(st, c, 0x51000000u | (uint32_t)k);
                    break;
                }
            }
        }
        eng_cbar<CF>(st, c, w);
        if (w.cw < HQ) {  // one wave per head
            const int h = w.cw;
            const float* sg = stage + h * ky.ns * PW;
            float M = -INFINITY;
            for (int sp = 0; sp < ky.ns; ++sp) M = fmaxf(M, sg[sp * PW + HS]);
            float L = 0.f, y0 = 0.f, y1 = 0.f;
            const int dd = (2 * c.lane) % HS;
            for (int sp = 0; sp < ky.ns; ++sp) {
                const float ms = sg[sp * PW + HS];
                const float f = (ms == -INFINITY) ? 0.f : __expf(ms - M);
                L += sg[sp * PW + HS + 1] * f;
                y0 += sg[sp * PW + dd] * f;
                y1 += sg[sp * PW + dd + 1] * f;
            }
            const uint32_t pk = (uint32_t)f2bf(y0 / L) | ((uint32_t)f2bf(y1 / L) << 16);
            if (c.lane < HS / 2)
                st_gran(reinterpret_cast<uint64_t*>(op->out) + (((int64_t)(ky.g * HQ + h) * HS) >> 1) + c.lane, pk, c.epoch);
        }
    }
    eng_cbar<CF>(st, c, w);  // the scratch is free again (the next op's input goes into this buffer)
}

template <class CF, int HS, int HQ>
__device__ __forceinline__ void eng_attn(const EngState& st, const EngCtx& c, EngCons& w, const EngOp* op, int k) {
    const int phase = op->epilogue;  // 0: the whole op; 1 / 2: its halves
    if (phase != 2) {
        const bool part = eng_attn_local<CF, HS, HQ>(st, c, w, op, k);
        if (phase == 0) {
            if (part)
                eng_attn_combine<CF, HS, HQ>(st, c, w, op, k);
            else
                eng_cbar<CF>(st, c, w);
        }
    } else {
        eng_stamp(st, c, w, k, 0);
        if (eng_keys<HS>(st, c.cu, c.pos).part) eng_attn_combine<CF, HS, HQ>(st, c, w, op, k);
    }
    eng_stamp(st, c, w, k, 2);
    if ((ENG_STAMPS && st.dbg != nullptr) && c.cu == 0 && w.cw == 0 && c.lane == 0) st.dbg[k * 16 + 3] = w.waited;
    w.waited = 0;
}

template <class CF, int HS, int HQ>
__device__ __forceinline__ void eng_consumer(const EngState& st, const EngCtx& c, int cw) {
    EngCons w;
    if (CF::SPIN == 1) __builtin_amdgcn_s_setprio(1);
    w.cw = cw;
    w.cb_gen = 0;
    w.seq = 0;
    w.bc = 0;
    w.best = -INFINITY;
    w.best_i = 0x7fffffff;
    w.waited = 0;
    w.gate_spins = 0;
    for (int k = 0; k < st.nops; ++k) {
        const EngOp opv = eng_fetch_op(st.ops, k);
        const EngOp* op = &opv;
        if (op->type == PARROT_ENG_GEMV)
            eng_gemv<CF>(st, c, w, op, k);
        else
            eng_attn<CF, HS, HQ>(st, c, w, op, k);
    }
    if (!st.greedy) {
        if (c.cu == 0 && cw == 0 && c.lane == 0) {
            st.epoch[0] = c.epoch + 1;
            if (st.host_words != nullptr) __hip_atomic_store((glb_u32_t*)(st.host_words + 1), c.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    // ---- greedy sampling: the CU's arg-max candidate -> CU 0 -> tokens[pos + 1], pos += 1
    float bv = w.best;
    int bi = w.best_i;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) {
            bv = ov;
            bi = oi;
        }
    }
    if (c.lane == 0) {
        reinterpret_cast<float*>(c.fx + EF_BESTV)[cw] = bv;
        reinterpret_cast<int*>(c.fx + EF_BESTI)[cw] = bi;
    }
    eng_cbar<CF>(st, c, w);
    if (cw != 0) return;
    bv = -INFINITY;
    bi = 0x7fffffff;
    for (int t = 0; t < CF::NC; ++t) {
        const float ov = reinterpret_cast<float*>(c.fx + EF_BESTV)[t];
        const int oi = reinterpret_cast<int*>(c.fx + EF_BESTI)[t];
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) {
            bv = ov;
            bi = oi;
        }
    }
    if (c.lane == 0) {
        st_gran(st.arg + 2 * c.cu, __float_as_uint(bv), c.epoch);
        st_gran(st.arg + 2 * c.cu + 1, (uint32_t)bi, c.epoch);
    }
    if (c.cu != 0) return;
    bv = -INFINITY;
    bi = 0x7fffffff;
    for (int t0 = 0; t0 < 2 * ENG_WGS; t0 += 64) {  // lane parity = {value, index}; candidates t0/2 + lane/2
        const uint64_t* p = st.arg + t0 + c.lane;
        const uint32_t d = eng_gran_wait(st, c, p, ld_gran(p), true, 0x60000000u);
        const uint32_t dn = (uint32_t)__shfl_xor((int)d, 1, 64);
        const float ov = __uint_as_float((c.lane & 1) ? dn : d);
        const int oi = (int)((c.lane & 1) ? d : dn);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) {
            bv = ov;
            bi = oi;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) {
            bv = ov;
            bi = oi;
        }
    }
    if (c.lane == 0) {
        st.tokens[c.pos + 1] = (bi == 0x7fffffff) ? 0 : bi;
        st.pos[0] = c.pos + 1;
        st.epoch[0] = c.epoch + 1;
        // progress word for a host watchdog: the epoch of the last launch that ran to its end
        if (st.host_words != nullptr) __hip_atomic_store((glb_u32_t*)(st.host_words + 1), c.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int HS, int HQ, int BIG, int WFMT>
__global__ void __launch_bounds__(ENG_THREADS)
eng_token_kernel(EngState st) {
    typedef EngCfg<BIG, WFMT> CF;
    extern __shared__ __attribute__((aligned(16))) unsigned char eng_smem[];
    EngCtx c;
    c.ring = eng_smem;
    c.buf0 = eng_smem + CF::NSLOT * ENG_SLOT_BYTES;
    c.buf1 = c.buf0 + st.lds_buf0_bytes;
    c.fx = c.buf1 + st.lds_buf1_bytes;
    c.epoch = st.epoch[0];
    c.pos = st.pos[0];
    c.cu = blockIdx.x;
    c.lane = threadIdx.x & 63;
    int64_t tok = st.tokens[c.pos];
    if (tok < 0 || tok >= st.V) tok = 0;
    c.emb = reinterpret_cast<const bf16_t*>(st.wte) + tok * st.d;
    // control words and the RoPE row of this position
    for (int i = threadIdx.x; i < (EF_RED - EF_STAT) / 4; i += ENG_THREADS) reinterpret_cast<uint32_t*>(c.fx + EF_STAT)[i] = 0u;
    if ((int)threadIdx.x < 2 * st.n_elem) {
        const int which = threadIdx.x / st.n_elem, d = threadIdx.x % st.n_elem;
        const __half* tab = reinterpret_cast<const __half*>(which ? st.rope_sin : st.rope_cos);
        reinterpret_cast<float*>(c.fx + EF_ROPE)[which * 128 + d] = __half2float(tab[(int64_t)c.pos * st.n_elem + d]);
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < CF::NLOAD)
        eng_loader<CF, HS>(st, c, wave);
    else
        eng_consumer<CF, HS, HQ>(st, c, wave - CF::NLOAD);
}

// ------------------------------------------------------------------------------------------ E4 repack
// one thread per 16-byte unit of the E4 image
__global__ void __launch_bounds__(256)
e4_repack_kernel(const uint8_t* __restrict__ q1, const bf16_t* __restrict__ s1, const bf16_t* __restrict__ z1,
                 const uint8_t* __restrict__ q2, const bf16_t* __restrict__ s2, const bf16_t* __restrict__ z2, int N, int K,
                 int nblocks, int nq, uint4* __restrict__ e4) {
    const int spb = (nq + 3) >> 2;
    const int block_units = (4 * nq + spb) * 64;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (int64_t)nblocks * block_units) return;
    const int B = (int)(tid / block_units), wb = (int)(tid % block_units);
    const int sib = min(wb / (17 * 64), spb - 1);
    const int rem = wb - sib * 17 * 64;
    const int piece = rem >> 6, ln = rem & 63;
    const int nqs = min(4, nq - 4 * sib);
    const int ng = (K + 127) >> 7;
    const bool dual = q2 != nullptr;
    uint32_t dw[4] = {0, 0, 0, 0};
    if (piece < 4 * nqs) {
        const int qq = piece >> 2, i = piece & 3;
        const int r = ln & 7, p = ln >> 3;
        const int t = 4 * (8 * (4 * sib + qq) + p) + i;  // 32-column slice of the row
        const uint8_t* q = (dual && r >= 4) ? q2 : q1;
        const int row = dual ? B * 4 + (r & 3) : B * 8 + r;
        if (t * 32 < K) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint32_t v = 0;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const uint32_t b = q[(int64_t)(t * 16 + d * 4 + ii) * N + row];
                    v |= (b & 0xFu) << (4 * ii);
                    v |= (b >> 4) << (16 + 4 * ii);
                }
                dw[d] = v;
            }
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int wd = ln * 4 + jj;
            const int qq = wd >> 6, l = wd & 63;
            const int r = l & 7, p = l >> 3;
            const int G = 8 * (4 * sib + qq) + p;
            const bf16_t* s = (dual && r >= 4) ? s2 : s1;
            const bf16_t* z = (dual && r >= 4) ? z2 : z1;
            const int row = dual ? B * 4 + (r & 3) : B * 8 + r;
            if (qq < nqs && G < ng) dw[jj] = (uint32_t)s[(int64_t)row * ng + G] | ((uint32_t)z[(int64_t)row * ng + G] << 16);
        }
    }
    e4[tid] = make_uint4(dw[0], dw[1], dw[2], dw[3]);
}

// ------------------------------------------------------------------------------------------ E16 repack
// bf16 weights: per 8 rows (a block) ceil(K / 64) pieces of 1 KiB; in piece j lane l holds the 8 columns 64 j + 8 (l / 8) .. + 7
// of row l % 8 (zero past K).  A unit = one ring slot = 16 pieces (1024 columns); a row's last unit may be shorter - the
// padding up to a whole unit is neither stored nor streamed.  One thread per 16-byte unit of the image.
__global__ void __launch_bounds__(256)
e16_repack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2, int N, int K, int nblocks, int pt, uint4* __restrict__ e16) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (int64_t)nblocks * pt * 64) return;
    const int ln = (int)(tid & 63);
    const int64_t bj = tid >> 6;
    const int B = (int)(bj / pt), j = (int)(bj % pt);
    const int r = ln & 7, p = ln >> 3;
    const bool dual = w2 != nullptr;
    const bf16_t* w = (dual && r >= 4) ? w2 : w1;
    const int row = dual ? B * 4 + (r & 3) : B * 8 + r;
    const int k0 = 64 * j + 8 * p;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (k0 < K) v = *reinterpret_cast<const uint4*>(w + (int64_t)row * K + k0);
    e16[tid] = v;
}

// ------------------------------------------------------------------------------------------ E8 repack
// int8 weights (LLM.int8's CB, row-major): per 8 rows (a block; SwiGLU pair: 4 + 4) ceil(K / 128) pieces of 1 KiB; in piece j
// lane l holds the 16 columns 128 j + 16 (l / 8) .. + 15 of row l % 8 (zero past K).  One thread per 16-byte unit.
__global__ void __launch_bounds__(256)
e8_repack_kernel(const int8_t* __restrict__ w1, const int8_t* __restrict__ w2, int N, int K, int nblocks, int pt, uint4* __restrict__ e8) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (int64_t)nblocks * pt * 64) return;
    const int ln = (int)(tid & 63);
    const int64_t bj = tid >> 6;
    const int B = (int)(bj / pt), j = (int)(bj % pt);
    const int r = ln & 7, p = ln >> 3;
    const bool dual = w2 != nullptr;
    const int8_t* w = (dual && r >= 4) ? w2 : w1;
    const int row = dual ? B * 4 + (r & 3) : B * 8 + r;
    const int k0 = 128 * j + 16 * p;
    uint32_t dw[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 16; ++e)
        if (k0 + e < K) dw[e >> 2] |= (uint32_t)(uint8_t)w[(int64_t)row * K + k0 + e] << (8 * (e & 3));
    e8[tid] = make_uint4(dw[0], dw[1], dw[2], dw[3]);
}

static int e4_shape(int N, int K, int dual, int* nblocks, int* nq) {
    PARROT_REQUIRE(N > 0 && K > 0, "e4: N and K must be positive (N=%d K=%d)", N, K);
    PARROT_UNSUPPORTED(K % 32 == 0, "e4: K=%d must be a multiple of 32", K);
    PARROT_UNSUPPORTED(N % (dual ? 4 : 8) == 0, "e4: N=%d must be a multiple of %d", N, dual ? 4 : 8);
    *nq = (K + 1023) / 1024;
    PARROT_UNSUPPORTED(*nq <= ENG_MAXQ_BIG, "e4: K=%d is beyond the %d columns the stream engine is built for", K, ENG_MAXQ_BIG * 1024);
    *nblocks = dual ? N / 4 : N / 8;
    return PARROT_OK;
}

static int64_t eng_attn_scratch_bytes(int hs, int hq, int nsplit) {
    return (int64_t)(hq + 2) * hs * 2 + (int64_t)(ENG_NC + nsplit) * hq * (hs + 2) * 4;
}
static bool eng_is_big(int kmax) { return kmax > ENG_MAXQ_STD * 1024; }

}  // namespace parrot

using namespace parrot;

extern "C" {

int64_t parrot_e4_bytes(int N, int K, int dual) {
    int nblocks, nq;
    const int rc = e4_shape(N, K, dual, &nblocks, &nq);
    if (rc != PARROT_OK) return rc;
    return (int64_t)nblocks * (4 * nq + (nq + 3) / 4) * 1024;
}

int parrot_e4_repack(const void* q1, const void* s1, const void* z1, const void* q2, const void* s2, const void* z2, int N,
                     int K, void* e4, void* stream) {
    PARROT_REQUIRE(q1 && s1 && z1 && e4, "e4_repack: null pointer");
    PARROT_REQUIRE((q2 != nullptr) == (s2 != nullptr) && (q2 != nullptr) == (z2 != nullptr), "e4_repack: the second matrix needs all three buffers");
    PARROT_REQUIRE(aligned16(e4), "e4_repack: the E4 buffer must be 16-byte aligned");
    int nblocks, nq;
    const int rc = e4_shape(N, K, q2 != nullptr, &nblocks, &nq);
    if (rc != PARROT_OK) return rc;
    const int64_t units = (int64_t)nblocks * (4 * nq + (nq + 3) / 4) * 64;
    const int64_t blocks = (units + 255) / 256;
    PARROT_UNSUPPORTED(blocks < (1ll << 31), "e4_repack: matrix too large");
    return launch(K_E4_REPACK, e4_repack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)q1,
                  (const bf16_t*)s1, (const bf16_t*)z1, (const uint8_t*)q2, (const bf16_t*)s2, (const bf16_t*)z2, N, K, nblocks, nq,
                  (uint4*)e4);
}

int64_t parrot_e16_bytes(int N, int K, int dual) {
    int nblocks, nq;
    const int rc = e4_shape(N, K, dual, &nblocks, &nq);  // the same block / unit grid as E4
    if (rc != PARROT_OK) return rc;
    return (int64_t)nblocks * ((K + 63) / 64) * 1024;
}

int parrot_e16_repack(const void* w1, const void* w2, int N, int K, void* e16, void* stream) {
    PARROT_REQUIRE(w1 && e16, "e16_repack: null pointer");
    PARROT_REQUIRE(aligned16(e16) && aligned16(w1) && (w2 == nullptr || aligned16(w2)), "e16_repack: buffers must be 16-byte aligned");
    int nblocks, nq;
    const int rc = e4_shape(N, K, w2 != nullptr, &nblocks, &nq);
    if (rc != PARROT_OK) return rc;
    const int pt = (K + 63) / 64;
    const int64_t blocks = ((int64_t)nblocks * pt * 64 + 255) / 256;
    PARROT_UNSUPPORTED(blocks < (1ll << 31), "e16_repack: matrix too large");
    return launch(K_E4_REPACK, e16_repack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w1,
                  (const bf16_t*)w2, N, K, nblocks, pt, (uint4*)e16);
}

int64_t parrot_e8_bytes(int N, int K, int dual) {
    PARROT_REQUIRE(N > 0 && K > 0, "e8: N and K must be positive (N=%d K=%d)", N, K);
    PARROT_UNSUPPORTED(N % (dual ? 4 : 8) == 0, "e8: N=%d must be a multiple of %d", N, dual ? 4 : 8);
    PARROT_UNSUPPORTED((K + 2047) / 2048 <= ENG_MAXQ_STD, "e8: K=%d is beyond the %d columns the stream engine is built for", K, ENG_MAXQ_STD * 2048);
    return (int64_t)(dual ? N / 4 : N / 8) * ((K + 127) / 128) * 1024;
}

int parrot_e8_repack(const void* w1, const void* w2, int N, int K, void* e8, void* stream) {
    PARROT_REQUIRE(w1 && e8, "e8_repack: null pointer");
    PARROT_REQUIRE(aligned16(e8), "e8_repack: the E8 buffer must be 16-byte aligned");
    const int64_t bytes = parrot_e8_bytes(N, K, w2 != nullptr);
    if (bytes < 0) return (int)bytes;
    const int nblocks = w2 != nullptr ? N / 4 : N / 8, pt = (K + 127) / 128;
    const int64_t blocks = ((int64_t)nblocks * pt * 64 + 255) / 256;
    PARROT_UNSUPPORTED(blocks < (1ll << 31), "e8_repack: matrix too large");
    return launch(K_E4_REPACK, e8_repack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const int8_t*)w1,
                  (const int8_t*)w2, N, K, nblocks, pt, (uint4*)e8);
}

// E8: the activation image is int8 (whole units of 2048 columns) + the outlier list
int64_t parrot_eng_lds_bytes_e8(int K, int hs, int q_per_kv, int nsplit) {
    PARROT_REQUIRE(K > 0, "eng_lds_bytes_e8: K must be positive");
    const int nq = (K + 2047) / 2048;
    PARROT_UNSUPPORTED(nq <= ENG_MAXQ_STD && nq * 16 <= (16 - ENG_NLOAD_MULTI) * ENG_MAXG_E8_BIG, "stream engine: K=%d is beyond what the int8 build takes", K);
    int64_t b = (int64_t)nq * 2048 + ENG_Q8_CAP * 4;
    if (hs > 0) {
        const int64_t a = eng_attn_scratch_bytes(hs, q_per_kv, nsplit);
        if (a > b) b = a;
    }
    return (b + 15) / 16 * 16;
}

int64_t parrot_eng_lds_bytes(int K, int hs, int q_per_kv, int nsplit) {
    PARROT_REQUIRE(K > 0, "eng_lds_bytes: K must be positive");
    const int nq = (K + 1023) / 1024;
    PARROT_UNSUPPORTED(nq <= ENG_MAXQ_BIG && nq * 8 <= ENG_NC * ENG_MAXG_BIG, "stream engine: K=%d is beyond what it is built for", K);
    int64_t b = (int64_t)nq * 8 * ENG_GROUP_STRIDE;
    if (hs > 0) {
        const int64_t a = eng_attn_scratch_bytes(hs, q_per_kv, nsplit);
        if (a > b) b = a;
    }
    return (b + 15) / 16 * 16;
}

// which build runs (wide = 6 ring slots, inputs up to 16384) and its dynamic LDS: the narrow one when the inputs allow it and
// it fits the CU
static int64_t eng_pick_build(int kmax, int wfmt, int buf0_bytes, int buf1_bytes, bool* big) {
    const bool e16 = (wfmt & 3) != PARROT_ENG_W_E4 || (wfmt & PARROT_ENG_W_TWO_LOADERS);  // two loaders: 6 ring slots
    const bool e8 = (wfmt & 3) == PARROT_ENG_W_E8;
    int64_t lds = 0;
    // (int8: units of 2048 columns; the narrow build gathers up to 14 * 7 groups of 128 = 12544 columns, the wide one 22528)
    for (int b = (e8 ? ((kmax + 2047) / 2048) * 16 > (16 - ENG_NLOAD_MULTI) * ENG_MAXG_E8 : eng_is_big(kmax)) ? 1 : 0; b < 2; ++b) {
        const int nslot = e16 ? (b ? EngCfg<1, PARROT_ENG_W_E16>::NSLOT : EngCfg<0, PARROT_ENG_W_E16>::NSLOT) : (b ? ENG_NSLOT_BIG : ENG_NSLOT_STD);
        lds = (int64_t)nslot * ENG_SLOT_BYTES + buf0_bytes + buf1_bytes + EF_RED + ENG_RED * (b ? ENG_MAXQ_BIG : ENG_MAXQ_STD) * 32 * (e8 ? 2 : 1) + (e8 ? ENG_Q8_STATE : 0);
        *big = b != 0;
        if (lds <= 160 * 1024) break;
    }
    return lds;
}

int64_t parrot_eng_lds_total(int kmax, int wfmt, int buf0_bytes, int buf1_bytes) {
    PARROT_REQUIRE(kmax > 0 && buf0_bytes > 0 && buf1_bytes > 0, "eng_lds_total: sizes must be positive");
    PARROT_REQUIRE((wfmt & 3) <= PARROT_ENG_W_E8 && (wfmt & ~7) == 0, "eng_lds_total: unknown weight format %d", wfmt);
    bool big;
    const int64_t lds = eng_pick_build(kmax, wfmt, buf0_bytes, buf1_bytes, &big);
    PARROT_UNSUPPORTED(lds <= 160 * 1024, "stream engine: needs %lld B of LDS", (long long)lds);
    return lds;
}

int parrot_eng_step(const parrot_eng_state_t* state_host, void* stream) {
    PARROT_REQUIRE(state_host != nullptr, "eng_step: null state");
    EngState st = *state_host;
    PARROT_REQUIRE(st.ops && st.nops >= 1 && st.tokens && st.pos && st.epoch && st.err && st.wte && st.rope_cos && st.rope_sin && st.arg,
                   "eng_step: null pointer in state");
    PARROT_UNSUPPORTED(st.hs == 64 || st.hs == 128, "stream engine: head size %d not built (64, 128)", st.hs);
    PARROT_REQUIRE(st.q_per_kv >= 1 && st.vper >= 1 && st.q_per_kv % st.vper == 0 && (st.q_per_kv / st.vper == 1 || st.q_per_kv / st.vper == 2),
                   "eng_step: q_per_kv=%d must be vper=%d virtual groups of 1 or 2 query heads", st.q_per_kv, st.vper);
    const int hq = st.q_per_kv / st.vper;
    PARROT_REQUIRE(st.n_groups >= 1 && st.nsplit >= 1 && st.nsplit <= 8 && st.n_groups * st.vper * st.nsplit <= ENG_WGS,
                   "eng_step: n_groups * vper * nsplit must fit the %d workgroups", ENG_WGS);
    PARROT_UNSUPPORTED(st.n_elem % 16 == 0 && st.n_elem >= 0 && st.n_elem <= st.hs && st.n_elem <= 128, "stream engine: rotary width %d must be a multiple of 16, at most the head size", st.n_elem);
    PARROT_REQUIRE(st.S >= 1 && st.V >= 1 && st.d >= 1, "eng_step: bad S / V / d");
    PARROT_REQUIRE(st.lds_buf0_bytes > 0 && st.lds_buf0_bytes % 16 == 0 && st.lds_buf1_bytes > 0 && st.lds_buf1_bytes % 16 == 0,
                   "eng_step: LDS buffer sizes must be positive multiples of 16");
    PARROT_REQUIRE(st.attn_buf == 0 || st.attn_buf == 1, "eng_step: attn_buf must be 0 or 1");
    PARROT_REQUIRE((st.attn_buf ? st.lds_buf1_bytes : st.lds_buf0_bytes) >= eng_attn_scratch_bytes(st.hs, hq, st.nsplit),
                   "eng_step: LDS buffer %d smaller than the attention scratch", st.attn_buf);
    PARROT_REQUIRE(st.kmax >= 1, "eng_step: kmax (the largest input of any op) must be set");
    if ((st.wfmt & 3) == PARROT_ENG_W_E8) {
        const int64_t need = (int64_t)((st.kmax + 2047) / 2048) * 2048 + ENG_Q8_CAP * 4;
        PARROT_UNSUPPORTED((st.kmax + 2047) / 2048 <= ENG_MAXQ_STD, "stream engine: K=%d is beyond what the int8 build takes", st.kmax);
        PARROT_REQUIRE(st.lds_buf0_bytes >= need || st.lds_buf1_bytes >= need, "eng_step: no LDS buffer holds an int8 input of kmax=%d elements", st.kmax);
    } else {
        const int nqm = (st.kmax + 1023) / 1024;
        PARROT_UNSUPPORTED(nqm <= ENG_MAXQ_BIG, "stream engine: K=%d is beyond what it is built for", st.kmax);
        PARROT_REQUIRE(st.lds_buf0_bytes >= (int64_t)nqm * 8 * ENG_GROUP_STRIDE || st.lds_buf1_bytes >= (int64_t)nqm * 8 * ENG_GROUP_STRIDE,
                       "eng_step: no LDS buffer holds an input of kmax=%d elements", st.kmax);
    }
    const int64_t lds_total = parrot_eng_lds_total(st.kmax, st.wfmt, st.lds_buf0_bytes, st.lds_buf1_bytes);
    if (lds_total < 0) return (int)lds_total;
    const size_t lds = (size_t)lds_total;
    bool big;
    (void)eng_pick_build(st.kmax, st.wfmt, st.lds_buf0_bytes, st.lds_buf1_bytes, &big);
    PARROT_REQUIRE((st.wfmt & 3) <= PARROT_ENG_W_E8 && (st.wfmt & ~7) == 0, "eng_step: unknown weight format %d", st.wfmt);
    const bool e16 = (st.wfmt & 3) == PARROT_ENG_W_E16, e8 = (st.wfmt & 3) == PARROT_ENG_W_E8;
    const bool two = (st.wfmt & 3) == PARROT_ENG_W_E4 && (st.wfmt & PARROT_ENG_W_TWO_LOADERS);
    hipStream_t s = (hipStream_t)stream;
#define PARROT_ENG_GO(HSV, HQV, BIGV, WFV)                                                                               \
    do {                                                                                                                 \
        static bool attr_set = false;                                                                                    \
        if (!attr_set) {                                                                                                 \
            hipError_t e = hipFuncSetAttribute((const void*)eng_token_kernel<HSV, HQV, BIGV, WFV>,                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                 \
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");                                               \
            attr_set = true;                                                                                             \
        }                                                                                                                \
        return launch(K_ENG_TOKEN, eng_token_kernel<HSV, HQV, BIGV, WFV>, dim3(ENG_WGS), dim3(ENG_THREADS), lds, s, st); \
    } while (0)
#define PARROT_ENG_GO2(HSV, HQV)                                      \
    do {                                                              \
        if (e8 && big) PARROT_ENG_GO(HSV, HQV, 1, PARROT_ENG_W_E8);   \
        if (e8) PARROT_ENG_GO(HSV, HQV, 0, PARROT_ENG_W_E8);          \
        if (e16 && big) PARROT_ENG_GO(HSV, HQV, 1, PARROT_ENG_W_E16); \
        if (e16) PARROT_ENG_GO(HSV, HQV, 0, PARROT_ENG_W_E16);        \
        if (two && big) PARROT_ENG_GO(HSV, HQV, 1, PARROT_ENG_W_E4 | PARROT_ENG_W_TWO_LOADERS); \
        if (two) PARROT_ENG_GO(HSV, HQV, 0, PARROT_ENG_W_E4 | PARROT_ENG_W_TWO_LOADERS);        \
        if (big) PARROT_ENG_GO(HSV, HQV, 1, PARROT_ENG_W_E4);         \
        PARROT_ENG_GO(HSV, HQV, 0, PARROT_ENG_W_E4);                  \
    } while (0)
    if (st.hs == 128) {
        if (hq == 1) PARROT_ENG_GO2(128, 1);
        PARROT_ENG_GO2(128, 2);
    }
    if (hq == 1) PARROT_ENG_GO2(64, 1);
    PARROT_ENG_GO2(64, 2);
#undef PARROT_ENG_GO2
#undef PARROT_ENG_GO
}

}  // extern "C"
