/*
 * parrot_hip.h — C ABI of libparrot_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary for the quantized-decode hot path of Lit-GPT
 * (griff4692/lit-parrot).  The reference has no FFI of its own for this path:
 * every op below is reached there through a PyTorch / Triton / bitsandbytes
 * call.  Each entry point cites the reference call site it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *  - plain pointers and ints only; every pointer is a DEVICE pointer unless
 *    its name ends in `_host`;
 *  - the caller owns every buffer; the library never allocates or frees
 *    tensor memory and keeps no mutable state besides the profiling sink (and, once per kernel, the LDS-size
 *    attribute of its code object); it reads no environment variable - the A/B switches and stamp hooks behind
 *    DESIGN.md's measurements exist only in the diagnostic build (-DPARROT_DIAG);
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work
 *    on it (no synchronisation), so a call sequence can be hipGraph-captured;
 *  - return value: 0 on success, negative PARROT_E* on failure; the message is
 *    available from parrot_last_error() (thread local);
 *  - "bf16" buffers hold IEEE bfloat16 bit patterns (uint16_t);
 *  - "rows" (M) are token rows of the activation matrix, row m of `x` starts at
 *    x + m*ldx elements.
 */
#ifndef PARROT_HIP_H
#define PARROT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PARROT_ABI_VERSION 1

#define PARROT_OK 0
#define PARROT_EINVAL (-1)       /* bad argument (shape, alignment, null pointer) */
#define PARROT_EHIP (-2)         /* a HIP runtime call failed */
#define PARROT_EUNSUPPORTED (-3) /* shape outside what the kernels are built for */

/* epilogues of the GEMV/GEMM entry points */
#define PARROT_EPI_NONE 0     /* out = bf16(acc + bias)                                   */
#define PARROT_EPI_RESIDUAL 1 /* out = bf16(residual + bf16(acc + bias))  model.py:171,178-179 */
#define PARROT_EPI_GELU 2     /* out = bf16(gelu_erf(bf16(acc + bias)))   model.py:284-287 */
#define PARROT_EPI_SWIGLU 3   /* out = bf16(bf16(silu(bf16(acc1))) * bf16(acc2)) model.py:297-301 (needs 2nd weight) */

/* Optional norm fused in front of a Linear (norm_1 / norm_2 / ln_f of lit_gpt/model.py:167,170,178,109): the
 * Linear's input rows are normalised on the fly.  Host struct, device pointers inside; NULL or kind 0 = none. */
typedef struct parrot_norm {
    int kind;           /* 0 none, 1 RMSNorm (lit_gpt/rmsnorm.py), 2 LayerNorm (torch.nn.LayerNorm) */
    const void* weight; /* bf16 [K] */
    const void* bias;   /* bf16 [K] or NULL (LayerNorm only) */
    float eps;
    int rsqrt_mode;     /* RMSNorm only, see parrot_rmsnorm */
} parrot_norm_t;

int parrot_version(void);
const char* parrot_last_error(void);

/* ---- profiling sink (bench.py only): while enabled every kernel launch is
 * bracketed by HIP events on its own stream (hipExtLaunchKernelGGL). ---------- */
int parrot_prof_begin(void);
/* synchronises, fills up to `cap` entries, returns the number of distinct kernels */
int parrot_prof_end(int cap, int* kernel_ids_host, double* total_ms_host, int64_t* launches_host);
const char* parrot_kernel_name(int kernel_id);

/* ---- int4 (GPTQ format) -------------------------------------------------------
 * Reference format (quantize/gptq.py:216-231): quant_weight uint8, logical
 * (N, K/2) with strides (1, N) i.e. memory [K/2][N]; byte j of row o holds
 * column 2j in the low nibble and 2j+1 in the high nibble (:240-241);
 * scales/zeros (N, ceil(K/group)) bf16; w = (q - zero) * scale (:249-251).
 * Kernel-native format "W4K": see DESIGN.md §3.                                   */
int64_t parrot_w4_packed_bytes(int N, int K, int group);
/* direction 0: reference -> W4K; 1: W4K -> reference (round trip is exact) */
int parrot_w4_repack(void* quant_weight_ref, void* scales, void* zeros, int N, int K, int group,
                     void* packed, int direction, void* stream);
/* y[m, :] = epilogue(x[m, :] @ dequant(W).T)  — replaces ColBlockQuantizedLinear.forward
 * (quantize/gptq.py:254-264) and qlinear_4bit_weight (:156-201) for M <= 8 rows.
 * packed2 is the second weight of PARROT_EPI_SWIGLU (fc_2), else NULL.           */
int parrot_w4_gemv(const void* packed, const void* packed2, const void* x, int ldx, int M,
                   const void* bias, const void* residual, int ldr, void* out, int ldo, int N,
                   int K, int group, int epilogue, const parrot_norm_t* norm, void* stream);
/* same contract for any M (prefill) on the matrix cores (v_mfma_f32_32x32x16_bf16; the int4 rows are expanded to
 * bf16 in LDS).  M <= 8 forwards to the GEMV.  For M > 8 a norm is taken for groups of 64 / 128 with K % group == 0 and
 * K <= 16384 (one launch writes the normalised rows and their per-group sums; PARROT_EUNSUPPORTED otherwise: apply
 * parrot_rmsnorm / parrot_layernorm first and pass norm == NULL), and
 * `workspace` holds parrot_gemm_workspace_floats(M, N, K, group, epilogue) floats: the normalised rows of a fused norm,
 * the per-group activation sums (int4) and, for launches with too few tiles to fill the chip (short prompts), the
 * split-K partial results that a second stage sums in a fixed order.  group = 0 asks for the bf16 GEMM's needs.   */
int64_t parrot_gemm_workspace_floats(int M, int N, int K, int group, int epilogue);
int parrot_w4_gemm(const void* packed, const void* packed2, const void* x, int ldx, int M,
                   const void* bias, const void* residual, int ldr, void* out, int ldo, int N,
                   int K, int group, int epilogue, const parrot_norm_t* norm, void* workspace,
                   void* stream);

/* ---- 4-bit codebook weights: bitsandbytes NF4 / FP4 (reference quantize/bnb.py:62-75 `Linear4bit`, selected by
 * lit_gpt/utils.py:36-68 "bnb.nf4", "bnb.nf4-dq", "bnb.fp4", "bnb.fp4-dq") ----------------------------------------
 * bitsandbytes format: flat row-major uint8, two weights per byte (FIRST weight in the HIGH nibble), one fp32 absmax per
 * block of 64 consecutive weights, w = code[q] * absmax with a 16-entry codebook.  Kernel-native format: W4K records
 *  with group = block and the group metadata word = the block's fp32 absmax; built with parrot_w4_repack
 * from the nibbles re-laid as (N, K/2) strides (1, N) low-nibble-first and the absmax words split into two 16-bit halves
 * (scales = low half, zeros = high half).  With compress_statistics ("-dq") the host de-nests absmax to fp32 first.
 * parrot_w4c_gemv replaces bnb.matmul_4bit for M <= 8 rows (same epilogues / fused norm as parrot_w4_gemv);
 * code16_bf16 = the codebook rounded to bf16, one value per 32-bit word (bf16 bits in the low half; 16 words, device).
 * parrot_w4c_dequant replaces bnb.functional.dequantize_4bit: out (N, K) bf16 = bf16(code16_f32[q] * absmax); the
 * prefill multiplies it with parrot_bf16_gemm (bitsandbytes' own MatMul4Bit is dequantise + F.linear).            */
int parrot_w4c_gemv(const void* packed, const void* packed2, const void* code16_bf16, const void* x, int ldx, int M,
                    const void* bias, const void* residual, int ldr, void* out, int ldo, int N, int K, int block,
                    int epilogue, const parrot_norm_t* norm, void* stream);
int parrot_w4c_dequant(const void* packed, const void* code16_f32, void* out, int ldo, int N, int K, int block,
                       void* stream);
/* any M (prefill) on the matrix cores with the GEMV's numerics (bf16 codebook, absmax applied per block): the codebook variant of
 * parrot_w4_gemm's kernel (packed weights by LDS-DMA, fragments looked up in registers in front of the MFMAs).  M <= 8 forwards
 * to the GEMV; M > 8: norm == NULL, K % 64 == 0, workspace = parrot_gemm_workspace_floats(M, N, K, block, epilogue) floats.    */
int parrot_w4c_gemm(const void* packed, const void* packed2, const void* code16_bf16, const void* x, int ldx, int M,
                    const void* bias, const void* residual, int ldr, void* out, int ldo, int N, int K, int block,
                    int epilogue, const parrot_norm_t* norm, void* workspace, void* stream);

/* ---- dense bf16 Linear (torch.nn.Linear on the bf16 path, lit_gpt/model.py:29,188,190,281-295)
 * W is (N, K) row-major bf16.                                                    */
int parrot_bf16_gemv(const void* W, const void* W2, const void* x, int ldx, int M, const void* bias,
                     const void* residual, int ldr, void* out, int ldo, int N, int K, int epilogue,
                     const parrot_norm_t* norm, void* stream);
int parrot_bf16_gemm(const void* W, const void* W2, const void* x, int ldx, int M, const void* bias,
                     const void* residual, int ldr, void* out, int ldo, int N, int K, int epilogue,
                     const parrot_norm_t* norm, void* workspace, void* stream);

/* ---- LLM.int8 (quantize/bnb.py:18-60; arithmetic = bitsandbytes MatMul8bitLt) ----
 * quantize the rows of a checkpoint weight as `double_quant(weight.contiguous().half())` does (quantize/bnb.py:54-60): the value
 * is first rounded to fp16 whatever its dtype - w_dtype 0: bf16, 1: fp16, 2: fp32 -, CB = rne(127*W16/absmax_row), SCB = absmax_row */
int parrot_w8_quantize_rows(const void* W, int w_dtype, int N, int K, void* CB_int8, void* SCB_f32, void* stream);
/* activation prep of one call (its M token rows): fp16-round x; entries with |x| >= threshold are outliers (zero in the int8
 * copy, excluded from their row's absmax); the OUTLIER COLUMNS are those with an outlier in any row of the call (LLM.int8's
 * feature dimensions; bitsandbytes MatMul8bitLt: unique(colidx)) and are cleared in the int8 copy of every row.
 * xq int8 [M][K]; xout fp32 [M][K]: the fp16 activations (readers use the listed columns only); sca fp32 [M]; nout int32 [M];
 * oidx int32 [M][K]: the first nout[m] entries of row m list the outlier columns (for M > 1 the same ascending list in every
 * row); colflag: K int32 of scratch, needed when M > 1 */
int parrot_w8_prep_act(const void* x, int ldx, int M, int K, float threshold, void* xq, void* xout,
                       void* sca, void* nout, void* oidx, void* colflag, const parrot_norm_t* norm, void* stream);
/* out = epilogue(cast_bf16(fp16(fp16(C32*SCA*SCB/127^2 + bias) + fp16(outlier_part)))).
 * For PARROT_EPI_SWIGLU the second weight (fc_2) follows the first in the same buffers:
 * CB holds 2N rows ([fc_1; fc_2]) and SCB 2N scales. */
int parrot_w8_gemv(const void* CB, const void* SCB, const void* xq, const void* xout, const void* sca,
                   const void* nout, const void* oidx, int M, const void* bias, const void* residual, int ldr, void* out,
                   int ldo, int N, int K, int epilogue, void* stream);
/* The two calls above for ONE token row in one launch (decode): x bf16 [K] -> (optional norm) -> fp16 cast, outlier
 * split at `threshold`, row-absmax int8 - computed per workgroup in LDS - then the int8 GEMV, the fp16 outlier part and
 * the epilogue.  Same arithmetic as parrot_w8_prep_act + parrot_w8_gemv; K <= 16384.                                 */
int parrot_w8_gemv_fused(const void* CB, const void* SCB, const void* x, float threshold, const void* bias,
                         const void* residual, void* out, int N, int K, int epilogue,
                         const parrot_norm_t* norm, void* stream);
/* The same contract as parrot_w8_gemv for any M (prompts): v_mfma_i32_32x32x32_i8 on 128 x 128 tiles staged by LDS-DMA (K % 128 == 0;
 * other shapes and M <= 8 forward to parrot_w8_gemv), exact int32 sums to `workspace` (parrot_w8_gemm_workspace_bytes bytes), then one
 * element-wise pass: mm_dequant, outlier columns, epilogue - the GEMV's arithmetic.                                           */
int64_t parrot_w8_gemm_workspace_bytes(int M, int N, int K, int epilogue);
int parrot_w8_gemm(const void* CB, const void* SCB, const void* xq, const void* xout, const void* sca, const void* nout,
                   const void* oidx, int M, const void* bias, const void* residual, int ldr, void* out, int ldo, int N, int K,
                   int epilogue, void* workspace, void* stream);

/* ---- norms (lit_gpt/rmsnorm.py:17-21; torch.nn.LayerNorm via lit_gpt/config.py:86-92) ---- */
/* rsqrt_mode 0: rsqrt evaluated in fp32 and rounded to bf16 once (what torch's GPU kernels do);
 * 1: sqrt rounded to bf16, then its reciprocal rounded to bf16 — torch's CPU scalar path, which is
 * what a CPU run of the reference does for the one-value-per-row tensor of rmsnorm.py:19-20
 * (kept for bit-parity tests against CPU-generated golden vectors).                      */
int parrot_rmsnorm(const void* x, int ldx, const void* weight, void* out, int ldo, int M, int d,
                   float eps, int rsqrt_mode, void* stream);
int parrot_layernorm(const void* x, int ldx, const void* weight, const void* bias, void* out, int ldo,
                     int M, int d, float eps, void* stream);

/* ---- attention (lit_gpt/model.py:194-275, apply_rope :330-336) --------------------
 * qkv: [M][n_groups*(q_per_kv+2)*hs] bf16, interleaved per group (model.py:208-214).
 * rope_cos/sin: fp16 [rows][n_elem] (model.py:304-327); rope_local = 0: the full table, row m uses
 * table row *pos + m; rope_local = 1: already indexed per row as Block.forward receives it
 * (model.py:88-89), row m uses table row m.
 * pos: device int32*, position of row 0 (row m is at *pos + m; rows are consecutive positions).
 * k_cache/v_cache: bf16 [n_groups][S][hs] (GQA-native).  Slot = position % S,
 * which is the reference's roll-left-and-append-last window (model.py:238-245)
 * up to the order of the slots.                                                   */
int parrot_qkv_rope_kvappend(const void* qkv, int ldqkv, int M, const void* rope_cos,
                             const void* rope_sin, int n_elem, int rope_local, const int32_t* pos, int n_groups,
                             int q_per_kv, int hs, int S, void* q_out, void* k_cache, void* v_cache,
                             void* stream);
/* y[m] = softmax(q[m] k^T / sqrt(hs)) v over slots 0..min(*pos+m, S-1); y: [M][n_head*hs] bf16.
 * workspace: fp32, at least parrot_attn_workspace_floats(...) elements.
 * softmax_mode (this entry point and parrot_attn_fused_decode): 0 = the probabilities stay fp32 until the division (default:
 * closest to the exact result); 1 = parity runs: as torch's CPU flash-attention kernel computes the bf16 reference
 * (lit_gpt/model.py:256-275): p = exp(s - max over the key block) rounded to bf16 before P.V, the denominator from the
 * unrounded p - one key block, so nsplit == 1 and S <= 512, else PARROT_EUNSUPPORTED.                                      */
int64_t parrot_attn_workspace_floats(int M, int n_head, int hs, int nsplit);
int parrot_attn_decode(const void* q, int M, const int32_t* pos, const void* k_cache,
                       const void* v_cache, int n_groups, int q_per_kv, int hs, int S, int nsplit,
                       void* workspace, void* y, int ldy, int softmax_mode, void* stream);

/* Prompt rows on the matrix cores: y[m] = softmax(q[m] k^T / sqrt(hs), keys 0 .. *pos + m) v for the M rows of one prefill call
 * (lit_gpt/model.py:256-275 with the causal mask of :126-128), flash-attention style (32 queries per wave, key blocks of 32,
 * v_mfma_f32_32x32x16_bf16 for Q.K^T and P.V, P rounded to bf16).  Requires *pos + M <= S (no ring wrap inside the call).
 * q: [M][n_head*hs] roped (parrot_qkv_rope_kvappend), caches as for parrot_attn_decode; vT_scratch: bf16,
 * parrot_attn_prefill_scratch_elems(n_groups, hs, S) elements (the call's V rows transposed to [group][dim][slot]).            */
int64_t parrot_attn_prefill_scratch_elems(int n_groups, int hs, int S);
int parrot_attn_prefill(const void* q, int M, const int32_t* pos, const void* k_cache, const void* v_cache,
                        void* vT_scratch, int n_groups, int q_per_kv, int hs, int S, void* y, int ldy, void* stream);

/* Decode step (one new token) of CausalSelfAttention in ONE launch: q/k/v split + RoPE + KV append + attention over
 * slots 0..min(*pos, S-1) + cross-split combine (lit_gpt/model.py:208-247).  qkv: one row; y: [n_head*hs] bf16.
 * workspace as for parrot_attn_decode (M = 1); tickets: n_head zero-initialised uint32 (one per group and chunk of query heads; re-armed by the kernel). */
int parrot_attn_fused_decode(const void* qkv, const void* rope_cos, const void* rope_sin, int n_elem,
                             const int32_t* pos, int n_groups, int q_per_kv, int hs, int S, int nsplit,
                             void* workspace, void* tickets, void* k_cache, void* v_cache, void* y,
                             int softmax_mode, void* stream);
/* ---- small ops of the step ----------------------------------------------------------
 * x[m] = wte[tokens[(pos ? *pos : 0) + m]]   (lit_gpt/model.py:99)                    */
int parrot_embedding(const void* wte, int d, const int64_t* tokens, const int32_t* pos, int M,
                     void* out, int ldo, void* stream);
/* greedy step of generate() (generate/base.py:136-153 with top_k=1):
 * tokens[*pos + 1] = argmax(logits) (lowest index on ties); then *pos += 1.          */
/* ---- GPTQ quantiser, the column loop of one 128-column block (quantize/gptq.py:397-431) -------------------------
 * W: fp32 [rows][ldw] working copy of the weights (updated in place inside the block), Hinv: fp32 upper Cholesky factor
 * of the inverse Hessian [cols][ldh], Q: fp32 [rows][ldq] receives the values on the grid, Err: fp32 [rows][128] the
 * scaled errors for the caller's trailing update W[:, behind] -= Err @ Hinv[block, behind].  groupsize 0: the grid
 * parameters scales/zeros[row*ngroups + 0] are given (per-channel); else a divisor of 128: recomputed at every group start
 * from the current columns (find_params_weight, :317-347) and stored (round_bf16: the scale is rounded to bf16 first, the
 * precision it is stored with in a bf16 checkpoint).  loss_rows: fp32 [rows], += the row's loss.                       */
int parrot_gptq_block(void* W, int ldw, int rows, int col0, int ncols, const void* Hinv, int ldh, void* Q, int ldq,
                      void* Err, void* scales, void* zeros, int ngroups, int groupsize, int maxq, int round_bf16,
                      void* loss_rows, void* stream);

/* Chat loop (chat/base.py:80-87): after a sampling step wrote tokens[*pos], latch the first stop sequence that the generated
 * tokens end with.  stop_flat: the sequences back to back (int64), stop_off: n_stop + 1 offsets, longest: the longest
 * sequence (the reference's look-back buffer: nothing matches before `longest` tokens were generated), first_gen: device
 * int32, index of the first generated token.  flag: int32[2], {-1, 0} before; {generated-token number of the hit, length
 * of the matched sequence} after.                                                                                      */
int parrot_stop_check(const int64_t* tokens, const int32_t* pos, const int32_t* first_gen, const int64_t* stop_flat,
                      const int32_t* stop_off, int n_stop, int longest, int32_t* flag, void* stream);
int parrot_argmax_advance(const void* logits, int V, int64_t* tokens, int32_t* pos, void* stream);
/* The sampling step of generate() for top_k != 1 (generate/base.py:136-153; chat/base.py:60-72) in one launch, with the arithmetic
 * of the torch device ops the reference runs there: logits / temperature (bf16), keep the values >= the top_k-th largest
 * (top_k <= 0: all), softmax (fp32 inside, bf16 out), then multinomial's draw = argmax(probs / q) with q the bf16
 * Exponential(1) noise that the CALLER draws with torch (`empty_like(probs).exponential_(1)`: exactly the call
 * torch.multinomial makes, so the same generator state gives the same token; it can be captured in a hipGraph), lowest index on
 * ties; tokens[*pos + 1] = idx, *pos += 1.  logits, noise_exp1: bf16 [V]; probs_out: bf16 [V] or NULL (the probabilities, for
 * tests).  torch.multinomial's validity checks (two host syncs per token) are not reproduced.                                 */
int parrot_topk_sample(const void* logits, int V, float temperature, int top_k, const void* noise_exp1, void* probs_out,
                       int64_t* tokens, int32_t* pos, void* stream);

/* ---- stream engine: ONE launch per decode token ------------------------------------------------------------
 * The whole token (generate/base.py:131-153 for one iteration: embedding, every Block of lit_gpt/model.py:158-180,
 * ln_f, lm_head, greedy sampling) runs as one launch of 256 workgroups, one per CU (csrc/engine.hip).  In every
 * workgroup one or two LOADER waves stream that CU's share of every op's weights - and of the K/V cache rows the CU
 * attends over - through a ring of LDS slots by LDS-DMA, in a fixed order that never waits for a data dependency, only
 * for a free slot; 15 or 14 CONSUMER waves compute from LDS.  Activation vectors pass between CUs as 8-byte {data, tag}
 * granules (tag = the step's epoch) written with write-through stores and polled with L1-bypassing loads: there is no
 * grid barrier.  Weight layouts, both per 8 output rows (a block) and 1024 input columns (a unit):
 *   E4  (GPTQ int4, group 128): four 1-KiB pieces in which lane l holds the 32-column slice of row l % 8 in quantisation
 *       group 8 * unit + l / 8, plus one metadata piece per four units;
 *   E16 (bf16): up to sixteen 1-KiB pieces, lane l of piece i holds columns 1024 unit + 64 i + 8 (l / 8) .. + 7 of row l % 8
 *       (a row's last unit has only the pieces K needs);
 *   E8  (LLM.int8, units of 2048 columns): up to sixteen pieces, lane l of piece j holds columns 128 j + 16 (l / 8) .. + 15
 *       of row l % 8; the activation quantiser (fp16 cast, outliers at `threshold`, row absmax, int8) runs in the gather.
 * A CU owns a contiguous range of blocks of every Linear.  Supported: every Linear int4 (group 128 or one per row) without
 * bias, every Linear LLM.int8 without bias, or every Linear bf16 (bias allowed); RMSNorm or LayerNorm; SwiGLU or GELU MLP; sequential or parallel residual; head size
 * 64 / 128, any q_per_kv (virtual groups of 1 or 2 query heads, `vper`); n_embd up to 8192; inputs of up to 16384
 * elements per op (wider Linears as K-chunk ops, `acc`).  Everything else keeps the multi-launch step.               */
#define PARROT_ENG_GEMV 0
#define PARROT_ENG_ATTN 1
#define PARROT_ENG_EPI_LOGITS 4 /* lm_head: plain bf16 logits + the CU's arg-max candidate */
#define PARROT_ENG_WGS 256
#define PARROT_ENG_W_E4 0
#define PARROT_ENG_W_E16 1
#define PARROT_ENG_W_E8 2
#define PARROT_ENG_W_TWO_LOADERS 4 /* flag on the STATE's wfmt (and for parrot_eng_lds_total) with E4: two loader waves, 14 consumers,
                                      6 ring slots - what bf16 and int8 weights always run with */

typedef struct parrot_eng_op {
    int32_t type;          /* PARROT_ENG_* */
    int32_t epilogue;      /* GEMV: PARROT_EPI_NONE / RESIDUAL / GELU / SWIGLU or PARROT_ENG_EPI_LOGITS;
                              ATTN: 0 the whole op, 1 / 2 its halves as two ops (1: the CU's keys -> partial state; 2: the
                              group's leader merges the partial states into the heads) with other ops between them */
    int32_t K;             /* input elements (GEMV) */
    int32_t nblocks;       /* 8-row blocks of the matrix: N / 8, or N / 4 for the SwiGLU pair */
    int32_t nq;            /* units per block = ceil(K / 1024) */
    int32_t buf;           /* GEMV: which of the two LDS activation buffers this op's input uses (consecutive GEMVs
                              alternate); ATTN: which one holds the scratch (= the state's attn_buf) */
    int32_t norm_kind;     /* norm fused in front: 0 none, 1 RMSNorm, 2 LayerNorm */
    float norm_eps;
    int32_t in_embedding;  /* the input vector is wte[tokens[pos]] (first block) */
    int32_t res_embedding; /* the residual is wte[tokens[pos]] (first block) */
    int32_t wfmt;          /* PARROT_ENG_W_E4 (int4, group 128) or PARROT_ENG_W_E16 (bf16); = the state's */
    int32_t res_in;        /* RESIDUAL: the residual is the CU's rows of 0: x (the block's input), 1: x + attention branch */
    int32_t res_out;       /* RESIDUAL: ... and the sum becomes the CU's rows of 0 / 1 */
    int32_t publish;       /* 1: the outputs go out as granules (`out`); 0: they stay in the CU (parallel residual: x + attn) */
    int32_t no_gather;     /* GEMV 1: the input is already in LDS buffer `buf`: an earlier Linear normalised it there
                              (norm2_w); ATTN 1: barrier first (the scratch buffer's last readers may still be at work) */
    int32_t blk_part;      /* GEMV: this op covers part blk_part of blk_parts of every CU's blocks of the Linear */
    int32_t blk_parts;     /*       (0 or 1: all of them); not with PARROT_EPI_RESIDUAL */
    int32_t acc;           /* GEMV over one K-chunk of a Linear whose input does not fit LDS: 0 the whole Linear; 1 first chunk
                              (the rows' sums are kept in the CU), 2 a middle chunk (added), 3 the last chunk (added, then
                              the epilogue).  W / in / K describe the chunk; needs nq < 11 (16 in the wide build) */
    float threshold;       /* E8: LLM.int8's outlier threshold (quantize/bnb.py:26-33: 6.0) */
    int32_t reserved;
    const void* W;         /* E4 / E16 / E8 weights */
    const void* norm_w;    /* K bf16 */
    const void* norm_b;    /* K bf16 or NULL (LayerNorm) */
    const void* bias;      /* E16: N bf16 or NULL; E8: the rows' scales SCB in block order, 8 fp32 per block (SwiGLU pair: 4 + 4) */
    const void* norm2_w;   /* NULL, or a second norm of the same input and kind (the MLP's norm_2 of a parallel-residual block:
                              model.py:166-171): its result goes to the other LDS buffer, for the next op (no_gather = 1) */
    const void* norm2_b;
    const uint64_t* in;    /* K / 2 input granules (GEMV), (q_per_kv + 2) * hs / 2 per group: the QKV vector (ATTN) */
    void* out;             /* output granules; LOGITS: V bf16, plain stores */
    uint64_t* part;        /* ATTN: n_head * nsplit * (hs + 2) fp32 granules {acc[hs], m, l} */
    void* k_cache;         /* ATTN: [n_groups][S][hs] bf16 */
    void* v_cache;
} parrot_eng_op_t;

typedef struct parrot_eng_state {
    const parrot_eng_op_t* ops; /* device array */
    int32_t nops;
    int32_t d;                  /* n_embd */
    int64_t* tokens;
    int32_t* pos;
    uint32_t* epoch;            /* 1 word, starts at 1; the launch tags its granules with it and leaves epoch + 1 */
    uint32_t* err;              /* 1 word, zero-initialised; non-zero after a bounded wait gave up: the code of the FIRST failure
                                   (0xT0000000 | op or ring sequence number; T: 1 loader / ring slot, 2 consumer barrier, 3 slot
                                   never landed, 4 / 5 / 6 hand-off of a Linear / attention / arg-max, 7 table check) */
    const void* wte;
    const void* rope_cos;       /* fp16 [block_size][n_elem] */
    const void* rope_sin;
    int32_t n_elem, n_groups, q_per_kv, hs, S, V, rsqrt_mode;
    int32_t nsplit;             /* CUs per virtual group in the attention op: min(8, 256 / (n_groups * vper)) */
    int32_t greedy;             /* 1: tokens[pos + 1] = argmax(logits), pos += 1 inside the launch */
    int32_t lds_buf0_bytes, lds_buf1_bytes; /* from parrot_eng_lds_bytes */
    int32_t kmax;               /* the largest K of any op: above 11264 the build with a 6-slot ring runs */
    int32_t wfmt;               /* PARROT_ENG_W_*: the one weight format of every Linear of the launch (| PARROT_ENG_W_TWO_LOADERS) */
    int32_t attn_buf;           /* which LDS buffer the attention ops use as scratch (must hold it) */
    int32_t vper;               /* virtual groups per K/V group: q_per_kv / vper = 1 or 2 query heads share one CU's pass over
                                   the group's keys (GQA / MQA: every virtual group streams the K/V rows for its own heads) */
    uint64_t* arg;              /* 2 * 256 granules: every CU's arg-max candidate {value, index} */
    uint64_t* dbg;              /* NULL, or nops * 16 words: 100 MHz stamps of workgroup 0 (diagnostic runs only) */
    uint64_t* dbg_all;          /* NULL, or nops * 256 * 2 words: {input ready, units done} stamps of every workgroup */
    uint32_t* host_words;       /* NULL, or 2 words of PINNED HOST memory (device-accessible): [0] the first error code, [1] the
                                   epoch of the last launch that ran to its end - a host watchdog reads them without a HIP call */
} parrot_eng_state_t;

/* bytes of the E4 image of an (N, K) int4 matrix with group 128 (dual = 1: the SwiGLU pair, N rows each) */
int64_t parrot_e4_bytes(int N, int K, int dual);
/* reference format (quant_weight / scales / zeros of quantize/gptq.py:216-231, group 128) -> E4; q2 / s2 / z2 = fc_2 of a
 * SwiGLU pair or NULL */
int parrot_e4_repack(const void* q1, const void* s1, const void* z1, const void* q2, const void* s2, const void* z2,
                     int N, int K, void* e4, void* stream);
/* the same for bf16 weights: E16 image of an (N, K) nn.Linear weight (row-major bf16; w2 = fc_2 of a SwiGLU pair or NULL) */
int64_t parrot_e16_bytes(int N, int K, int dual);
int parrot_e16_repack(const void* w1, const void* w2, int N, int K, void* e16, void* stream);
/* the same for LLM.int8 weights: E8 image of an (N, K) int8 matrix CB (row-major; w2 = fc_2 of a SwiGLU pair or NULL): per 8
 * rows ceil(K / 128) pieces of 1 KiB, lane l of piece j = columns 128 j + 16 (l / 8) .. + 15 of row l % 8 */
int64_t parrot_e8_bytes(int N, int K, int dual);
int parrot_e8_repack(const void* w1, const void* w2, int N, int K, void* e8, void* stream);
/* LDS bytes of one activation buffer of the int8 build: int8 image in whole units of 2048 columns + the outlier list */
int64_t parrot_eng_lds_bytes_e8(int K, int hs, int q_per_kv, int nsplit);
/* LDS bytes of one activation buffer for inputs of up to K elements (attention scratch for hs / q_per_kv included when
 * hs > 0); PARROT_EUNSUPPORTED when the step does not fit the CU */
int64_t parrot_eng_lds_bytes(int K, int hs, int q_per_kv, int nsplit);
/* dynamic LDS of the launch for these buffer sizes (ring + buffers + control area); PARROT_EUNSUPPORTED beyond 160 KiB */
int64_t parrot_eng_lds_total(int kmax, int wfmt, int buf0_bytes, int buf1_bytes);
/* one token, enqueued on `stream` (graph-capturable) */
int parrot_eng_step(const parrot_eng_state_t* state_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PARROT_HIP_H */
