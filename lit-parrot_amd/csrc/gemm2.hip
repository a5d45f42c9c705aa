// Prefill Linear on bf16 weights, second generation: 128 x 128 x 64 tiles staged global -> LDS by LDS-DMA.
//
// Reference: torch.nn.Linear on bf16 over the T prompt rows (lit_gpt/model.py:29,188,190,281-295).
// out[M, N] = epilogue(x[M, K] @ W[N, K]^T + b): both operands K-contiguous, so a K-tile of either is 128 rows x 128 B.
//
// Why a second kernel (gemm.hip stays for int4 and SwiGLU): the first one stages through registers + ds_write in 32-deep
// K-tiles; with 64 x 64 tiles each wave issues 2 MFMAs per barrier and the loop is bound by LDS traffic and barriers
// (measured 13 % of the bf16 matrix peak, no better with more workgroups per CU or split-K: tools/ab_gemm_tiles.sh).
// Here:  * global_load_lds_dwordx4: one wave instruction moves 8 rows x 128 B straight into LDS (no VGPRs, no ds_write pass);
//          a ring of three LDS buffers keeps the tiles of steps t+1 and t+2 in flight while step t is multiplied (counted
//          vmcnt + raw s_barrier), ONE barrier per 64-deep K-step;
//        * the LDS image is lane-linear per instruction (hardware), so the bank-conflict swizzle is applied on the SOURCE
//          side: 16-byte slot s of row r is stored at slot s ^ ((r >> 1) & 7) - rows are 128 B, two per 256-B bank row, and the
//          16 rows a ds_read_b128 lane group touches land on 16 different 16-byte slots;
//        * waves as WM x 2, each 64 x 64 = 2 x 2 tiles of v_mfma_f32_32x32x16_bf16, 16 MFMAs per wave and barrier: 4 waves on
//          a 128 x 128 tile, or 8 waves on a 256 x 128 tile with a ring of three (prompts of whole 256-row tiles).
// Short launches (few 128 x 128 tiles) split K; the second stage (gemm.hip's fixed-order sum + epilogue) is shared.
#include <stdlib.h>

#include <type_traits>

#include "parrot_common.h"
#include "w4_plan.h"

namespace parrot {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int G2M = 128, G2N = 128, G2K = 64;
#ifndef PARROT_G2_PRIO
#define PARROT_G2_PRIO 0
#endif
constexpr bool G2_PRIO = PARROT_G2_PRIO;  // raise the wave priority over the MFMA cluster (build-time A/B)
constexpr int G2_TILE16 = 128 * 8;  // 16-byte units of one operand tile (128 rows x 128 B)
#ifndef G2_B_AUX
#define G2_B_AUX 0  // cache policy bits of the weight tiles' LDS-DMA (2 = nt: A/B in profiles/r03a)
#endif

// Workgroup id -> (m tile, n tile, K split).  Workgroups are dealt to the 8 XCDs round-robin and each XCD has its own L2, so
// the tiles that share operands must (a) sit on one XCD and (b) run at the same time: XCD x owns a contiguous range of
// n tiles; its workgroups walk super-blocks of GM m-tiles x GN n-tiles (GM * GN = what is resident on an XCD at once), m
// fastest.  Without this every 128-row band of W is fetched from HBM once per m tile (measured: 26 % of the MFMA peak at
// M = 2048 with the plain x-fastest grid).
struct G2Map {
    int MT, NT, GM, GN, xcd_ok;
};
__device__ __forceinline__ void g2_tile_of(const G2Map& mp, int id, int& mt, int& nt, int& z) {
    const int per_z = mp.MT * mp.NT;
    z = id / per_z;
    int t = id - z * per_z;
    if (!mp.xcd_ok) {
        mt = t % mp.MT;
        nt = t / mp.MT;
        return;
    }
    const int x = t & 7, j = t >> 3, NTx = mp.NT >> 3;
    const int B = mp.GM * mp.GN;
    const int sb = j / B, r = j - sb * B;
    const int mblocks = mp.MT / mp.GM;
    const int mb = sb % mblocks, nb = sb / mblocks;
    mt = mb * mp.GM + r % mp.GM;
    nt = x * NTx + nb * mp.GN + r / mp.GM;
}

// G2_NBUF = LDS ring depth, WM = wave rows of the tile (waves are WM x 2, each 64 x 64):
//   WM = 2: 128 x 128 tile, 4 waves; ring of 2 (64 KB, two workgroups per CU) or 3 (96 KB, one per CU)
//   WM = 4: 256 x 128 tile, 8 waves, ring of 3 (144 KB, one workgroup per CU): for prompts of 256 rows and more.  The 128 x 128
//           shape loads 32 KB per 2 MFLOP-step and keeps ONE step per workgroup in flight (LDS holds no more twice per CU):
//           at 512 rows it runs at the rate the operands arrive from L2 / the Infinity Cache (9.5 TB/s of tile loads = 64
//           flop per byte = 600 TFLOP/s; profiles/r03b_stablelm*_kernel_stats.csv).  This shape loads 48 KB per 4 MFLOP
//           (87 flop per byte) and keeps TWO steps (96 KB) in flight.
template <bool SPLIT, int G2_NBUF, int WM>
__global__ void __launch_bounds__(WM * 128)
gemm2_kernel(const bf16_t* __restrict__ A, int lda, int M, const bf16_t* __restrict__ W, int N, int K,
             const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr, bf16_t* out, int ldo, int epi, int ksplit,
             float* __restrict__ part, G2Map mp) {
    constexpr int TM = WM * 64;                 // tile rows
    constexpr int NWAVES = WM * 2;
    constexpr int A16 = TM * 8;                 // 16-byte units of the A tile (TM rows x 128 B)
    constexpr int STAGE16 = A16 + G2_TILE16;    // [A | B]
    constexpr int A_LOADS = TM / 8 / NWAVES;    // LDS-DMA wave instructions (8 rows x 128 B each) per wave and K-step: 4
    constexpr int B_LOADS = 16 / NWAVES;        // 4 (WM = 2) or 2 (WM = 4)
    static_assert(A_LOADS * NWAVES * 8 == TM && B_LOADS * NWAVES == 16, "tile pieces must deal out evenly");
    // ONE LDS object (a second one beside an LDS-DMA target can cost a vmcnt(0) in front of every fragment read)
    __shared__ __attribute__((aligned(1024))) uint4 smem[G2_NBUF * STAGE16];  // [buffer][A rows | B rows][row * 8 + slot]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int mt_, nt_, z_;
    g2_tile_of(mp, blockIdx.x, mt_, nt_, z_);
    const int m0 = mt_ * TM, n0 = nt_ * G2N, zsplit = z_;
    const int lr = lane & 31, lh = lane >> 5;
    const int ktiles = K / G2K;
    // split z owns K-steps [z * ktiles / ksplit, (z + 1) * ktiles / ksplit): the ranges need not be equal
    const int kt_begin = SPLIT ? (int)((int64_t)zsplit * ktiles / ksplit) : 0;
    const int kt_end = SPLIT ? (int)((int64_t)(zsplit + 1) * ktiles / ksplit) : ktiles;

    // LDS-DMA assignment: wave w issues instructions w * A_LOADS .. of A and w * B_LOADS .. of B (8 rows each).
    // Lane l of an instruction: row r = r0 + l / 8, LDS slot l % 8 <- global slot (l % 8) ^ ((r >> 1) & 7).
    const int l_row = lane >> 3, l_slot = lane & 7;
    const bf16_t* a_src[A_LOADS];
    const bf16_t* b_src[B_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int r = (wave * A_LOADS + i) * 8 + l_row;
        a_src[i] = A + (int64_t)min(m0 + r, M - 1) * lda + (l_slot ^ ((r >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int r = (wave * B_LOADS + i) * 8 + l_row;
        b_src[i] = W + (int64_t)min(n0 + r, N - 1) * K + (l_slot ^ ((r >> 1) & 7)) * 8;
    }
    auto issue = [&](int kt, int buf) {
        uint4* st = smem + buf * STAGE16;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + (int64_t)kt * G2K),
                                             (__attribute__((address_space(3))) void*)&st[(wave * A_LOADS + i) * 64], 16, 0, 0);
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + (int64_t)kt * G2K),
                                             (__attribute__((address_space(3))) void*)&st[A16 + (wave * B_LOADS + i) * 64], 16, 0, G2_B_AUX);
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment rows of this lane and their swizzle keys (constant over the K loop)
    int a_row[2], b_row[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_row[i] = wm * 64 + i * 32 + lr;
        b_row[i] = wn * 64 + i * 32 + lr;
    }

    // Ring of G2_NBUF buffers.  An LDS-DMA load counts on vmcnt and retires in order (A_LOADS + B_LOADS per wave and tile):
    // before the barrier of step t every wave waits until at most the loads of tile t+1 are outstanding, i.e. ITS pieces of
    // tile t have landed; past the barrier everybody's have, and everybody is done reading the buffer of step t-1, which the
    // loads of tile t+2 issued right behind the barrier overwrite.  A raw s_barrier: __syncthreads() would drain vmcnt(0).
    issue(kt_begin, 0);
    if (G2_NBUF > 2 && kt_begin + 1 < kt_end) issue(kt_begin + 1, 1);
    int buf = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if (G2_NBUF > 2 && kt + 1 < kt_end) {
            if constexpr (A_LOADS + B_LOADS == 8)
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + G2_NBUF - 1 < kt_end) issue(kt + G2_NBUF - 1, buf >= 1 ? buf - 1 : G2_NBUF - 1);  // (buf + NBUF - 1) % NBUF
        const uint4* sa = smem + buf * STAGE16;
        const uint4* sb = sa + A16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int slot = ks * 2 + lh;
            bf16x8_t af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = __builtin_bit_cast(bf16x8_t, sa[a_row[i] * 8 + (slot ^ ((a_row[i] >> 1) & 7))]);
                bfr[i] = __builtin_bit_cast(bf16x8_t, sb[b_row[i] * 8 + (slot ^ ((b_row[i] >> 1) & 7))]);
            }
            if (G2_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            if (G2_PRIO) __builtin_amdgcn_s_setprio(0);
        }
        buf = buf + 1 == G2_NBUF ? 0 : buf + 1;
    }

    // C layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if (SPLIT && m0 + TM <= M && n0 + G2N <= N) {  // interior tile (wave-uniform): 64 stores off one base, no per-element bounds
        float* d0 = part + ((int64_t)zsplit * M + m0 + wm * 64 + 4 * lh) * N + n0 + wn * 64 + lr;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) d0[(int64_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * N + j * 32] = acc[i][j][r];
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M && col < N) {
                    if constexpr (SPLIT)
                        part[((int64_t)zsplit * M + row) * N + col] = acc[i][j][r];
                    else
                        out[(int64_t)row * ldo + col] = apply_epilogue(epi, acc[i][j][r], 0.f, bias, residual ? residual + (int64_t)row * ldr : nullptr, col);
                }
            }
        }
}

// tile rows: 256 (8 waves, one workgroup per CU) for prompts that fill whole 256-row tiles as well as 128-row ones
int gemm2_wm(int M) {
    const int env = tune_env("PARROT_GEMM2_WM", 0);  // PARROT_GEMM2_WM = 2 | 4 (A/B)
    if (env == 2 || env == 4) return env;
    return (M + 255) / 256 * 256 == (M + 127) / 128 * 128 ? 4 : 2;
}

// K splits when the tiles alone leave most of the chip idle: aim at ~512 workgroups of 128 x 128 (two fit a CU) or 256 of
// 256 x 128 (one per CU), whole 64-deep steps, at least 8 per split
int gemm2_ksplit(int M, int N, int K) {
    const int wm = gemm2_wm(M), tm = wm * 64;
    const int64_t tiles = (int64_t)((M + tm - 1) / tm) * ((N + G2N - 1) / G2N);
    const int ktiles = K / G2K;
    const int target = tune_env("PARROT_GEMM2_SPLIT_TARGET", wm == 4 ? 256 : 512);  // PARROT_GEMM2_SPLIT_TARGET: workgroups to aim at when splitting
    const int full = wm == 4 ? 128 : 256;  // more tiles than this: no split
    int ks = tiles > full ? 1 : (int)(target / (tiles > 0 ? tiles : 1));
    if (ks > 8) ks = 8;
    while (ks > 1 && ktiles / ks < 8) --ks;  // (uneven ranges are fine: split z owns steps [z kt / ks, (z + 1) kt / ks))
    return ks < 1 ? 1 : ks;
}

bool gemm2_enabled() {
    const int env = tune_env("PARROT_GEMM2", 1);  // PARROT_GEMM2=0: A/B against the first-generation kernel
    return env != 0;
}

static int g2_largest_divisor_le(int n, int cap) {
    for (int d = cap < n ? cap : n; d > 1; --d)
        if (n % d == 0) return d;
    return 1;
}

int gemm2_launch(const void* W, const void* x, int ldx, int M, const void* bias, const void* residual, int ldr, void* out, int ldo,
                 int N, int K, int epilogue, float* part, hipStream_t st, int* ksplit_out) {
    const int nbuf_env = tune_env("PARROT_GEMM2_NBUF", 0);  // PARROT_GEMM2_NBUF = 2 | 3 (A/B)
    const int ks = gemm2_ksplit(M, N, K);
    *ksplit_out = ks;
    PARROT_REQUIRE(ks == 1 || part != nullptr, "bf16_gemm: this shape splits K %d ways and needs the workspace of parrot_gemm_workspace_floats", ks);
    const int wm = gemm2_wm(M), tm = wm * 64;
    G2Map mp;
    mp.MT = (M + tm - 1) / tm;
    mp.NT = (N + G2N - 1) / G2N;
    const int64_t total = (int64_t)mp.MT * mp.NT * ks;
    PARROT_UNSUPPORTED(total < (1ll << 31), "bf16_gemm: too many tiles");
    // measured (tools/ab_gemm2.sh, StableLM-3B): 2 x 2-deep beats the 3-deep ring at one workgroup per CU at 512 and at 2048 rows
    const int nbuf = wm == 4 || nbuf_env == 3 ? 3 : 2;
    const int resident_per_xcd = nbuf == 3 ? 32 : 64;
    mp.xcd_ok = (mp.NT % 8 == 0);
    mp.GM = g2_largest_divisor_le(mp.MT, 8);
    mp.GN = mp.xcd_ok ? g2_largest_divisor_le(mp.NT / 8, resident_per_xcd / mp.GM > 0 ? resident_per_xcd / mp.GM : 1) : 1;
    const dim3 grid((unsigned)total);
#define PARROT_G2_GO(SPLITV, NBUFV, WMV)                                                                                      \
    return launch(K_BF16_GEMM, gemm2_kernel<SPLITV, NBUFV, WMV>, grid, dim3(WMV * 128), 0, st, (const bf16_t*)x, ldx, M, (const bf16_t*)W, N, K, \
                  (const bf16_t*)bias, (const bf16_t*)residual, ldr, (bf16_t*)out, ldo, epilogue, ks, part, mp)
    if (wm == 4) {
        if (ks > 1) PARROT_G2_GO(true, 3, 4);
        PARROT_G2_GO(false, 3, 4);
    }
    if (ks > 1) {
        if (nbuf == 3) PARROT_G2_GO(true, 3, 2);
        PARROT_G2_GO(true, 2, 2);
    }
    if (nbuf == 3) PARROT_G2_GO(false, 3, 2);
    PARROT_G2_GO(false, 2, 2);
#undef PARROT_G2_GO
}

// ====================================================================================================================
// int4 (W4K) weights on the same structure: parrot_w4_gemm for prompts (reference quantize/gptq.py:156-201, :254-264).
//   * A (activations) arrives by LDS-DMA exactly as above;
//   * B stays PACKED in LDS (LDS-DMA too: 4 KB per K-step instead of 16): a lane's B fragments of one K-step are the four
//     dwords of one W4K slice - one ds_read_b128 - each expanded in registers to eight bf16 values 128 + q (exact;
//     ((dword >> 4i) & 0x000F000F) | 0x43004300 is two of them) right in front of its MFMA.  No register staging, no
//     ds_write pass, and a ring of three stages fits twice on a CU (first version: registers -> expand -> ds_write_b128 x 4 into
//     a bf16 image with a one-deep prefetch; at 128 rows every K-step then waited a full HBM latency);
//   * at every quantisation-group boundary the group's partial product is folded into the result with the column's scale /
//     zero and the row's activation sum:  total += scale * (acc - (128 + zero) * sum_g(x)) - the numerics of the decode GEMV
//     (w4.hip) and of the first-generation kernel.  sum_g(x) comes from a pre-pass in a [group][row] layout; the sums and the
//     {scale, zero} words of a group reach LDS by 4-byte LDS-DMA one group ahead (no VGPR-destination load in the loop: the
//     compiler would answer one with vmcnt(0) and drain the ring).
// Takes group sizes that are multiples of 64 (whole K-steps per group); everything else stays on gemm.hip.
// 16 lanes per (row, group): lane c takes the 16-byte pieces c, c + 16, ... of the group (256 contiguous bytes per 16 lanes for
// groups of 128), then a 16-lane butterfly; the first version's one thread per (row, group) read 64 different lines per wave
// instruction and took 4.4 us on 128 x 4096 activations.
__global__ void __launch_bounds__(256)
gemm2_xsum_kernel(const bf16_t* __restrict__ x, int ldx, int M, int Mpad, int K, int G, int ngroups, float* __restrict__ xsT) {
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int c0 = threadIdx.x & 15;
    if (t >= (int64_t)Mpad * ngroups) return;  // (whole 16-lane groups leave together)
    const int g = (int)(t / Mpad), m = (int)(t % Mpad);
    float s = 0.f;
    if (m < M) {
        const int k0 = g * G, k1 = min(K, k0 + G);
        const uint4* p = reinterpret_cast<const uint4*>(x + (int64_t)m * ldx + k0);
        for (int c = c0; c < (k1 - k0) / 8; c += 16) {
            const uint4 v = p[c];
            s += (bflo(v.x) + bfhi(v.x)) + (bflo(v.y) + bfhi(v.y)) + (bflo(v.z) + bfhi(v.z)) + (bflo(v.w) + bfhi(v.w));
        }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 8);
    if (c0 == 0) xsT[t] = s;
}

// LDS-DMA issued from inline asm.  With the builtin, hipcc put an s_waitcnt vmcnt(0) in front of the first fragment read of every
// K-step of the int4 kernel (it cannot prove that the stage being read is not the stage in flight) and the ring drained every
// step; an asm load is absent from its bookkeeping, and the loop below counts vmcnt itself.  M0 = the wave's LDS byte address.
__device__ __forceinline__ void glds16_asm(const void* gsrc, const void* lds_dst_uniform) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst_uniform);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}
__device__ __forceinline__ void glds4_asm(const void* gsrc, const void* lds_dst_uniform) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst_uniform);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

// LDS of the int4 kernel (16-byte units): ring of 3 stages {A image 128 x 8 slots, packed B 128 rows x 2 slices} + two
// parities of group metadata {128 activation sums fp32, 128 {scale, zero} words}: 63.5 KB, two workgroups per CU.
// WN = wave columns: 2 -> 128 x 128 tile, 4 waves, ring of 3 stages, two workgroups per CU (63.5 KB each);
//                    4 -> 128 x 256 tile, 8 waves, ring of 6 stages, one workgroup per CU (147 KB): for launches with ONE m-tile
//                         (prompts up to 128 rows), which are bound by the weight stream - a third of every stage comes from HBM
//                         instead of a fifth, and five stages are in flight instead of two.
// CB = codebook weights (bitsandbytes NF4 / FP4, parrot_w4c_gemm): the nibbles index a 16-entry bf16 codebook kept in LDS as one
// private column per lane at LDS offset 0 (the GEMV's scheme: w4_plan.h w4c_slice_lookup), the group word is the block's fp32
// absmax and the fold is total += absmax * acc (groups of 64 = one K-step): the numerics of the decode kernel.
__device__ __forceinline__ void w4c_dword_lookup(uint32_t dw, uint32_t cb_addr, uint32_t (&o)[4]) {
    uint32_t a0 = cb_addr, a1 = cb_addr, lo[4], hi[4];
    const uint32_t lo4 = dw, hi4 = dw >> 4;
#define G2_CB_READ(R, A, SRC, SEL)                                                                                         \
    asm volatile("v_and_b32_sdwa %1, 15, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:" SEL "\n\t"  \
                 "ds_read_u16 %0, %1"                                                                                      \
                 : "=&v"(R), "+v"(A)                                                                                       \
                 : "v"(SRC)                                                                                                \
                 : "memory")
    G2_CB_READ(lo[0], a0, lo4, "BYTE_0");
    G2_CB_READ(hi[0], a1, lo4, "BYTE_2");
    G2_CB_READ(lo[1], a0, hi4, "BYTE_0");
    G2_CB_READ(hi[1], a1, hi4, "BYTE_2");
    G2_CB_READ(lo[2], a0, lo4, "BYTE_1");
    G2_CB_READ(hi[2], a1, lo4, "BYTE_3");
    G2_CB_READ(lo[3], a0, hi4, "BYTE_1");
    G2_CB_READ(hi[3], a1, hi4, "BYTE_3");
#undef G2_CB_READ
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])
                 :
                 : "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = lo[q] | (hi[q] << 16);
}

template <bool SPLIT, bool SWI, int WN, bool CB>
__global__ void __launch_bounds__(WN * 128) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm2_w4_kernel(const bf16_t* __restrict__ A, int lda, int M, const uint4* __restrict__ Wq, const uint4* __restrict__ Wq2, int N,
                int K, const float* __restrict__ xsT, int Mpad, const bf16_t* __restrict__ bias, const bf16_t* residual, int ldr,
                bf16_t* out, int ldo, int epi, W4Plan plan, int ksplit, float* __restrict__ part, float* __restrict__ part2, G2Map mp,
                const uint32_t* __restrict__ code) {
    constexpr int TN = WN * 64;                      // tile columns
    constexpr int NWAVES = 2 * WN;
    constexpr int W4_NBUF = WN == 2 ? 3 : 6;
    constexpr int W4_STAGE16 = G2_TILE16 + TN * 2;   // A image + packed B (two 16-byte slices per row)
    constexpr int W4_META16 = 32 + TN / 4;           // 128 activation sums + TN {scale, zero} words
    constexpr int A_PER_WAVE = 16 / NWAVES;          // 1-KB LDS-DMA pieces of the A tile per wave
    constexpr int TILE_LOADS = A_PER_WAVE + 1;       // LDS-DMA loads per wave and K-step
    constexpr int CB16 = CB ? 256 : 0;               // codebook columns (16 entries x 64 lanes x 4 B) in front of the ring
    __shared__ __attribute__((aligned(4096))) uint4 smem_all[CB16 + W4_NBUF * W4_STAGE16 + 2 * W4_META16];
    uint4* const smem = smem_all + CB16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave layout: WN columns of 64 x 64 waves (two row blocks x two column blocks each), or - the 128 x 128 tile -
    // four waves side by side, each ALL 128 rows x 32 columns: every wave expands only its own 32 columns of weights (half
    // the expansion VALU per wave and step) and reads all four row blocks of A instead (twice the A fragment reads)
#ifndef G2W_L22  // (-DG2W_L22: the 2 x 2 layout on the 128 x 128 tile too - A/B in profiles/r03a: 1 x 4 is 3 - 5 % faster at 512 rows, equal at 128)
    constexpr bool L14 = (WN == 2);
#else
    constexpr bool L14 = false;
#endif
    constexpr int IM = L14 ? 4 : 2, JN = L14 ? 1 : 2;
    const int wm = L14 ? 0 : wave / WN, wn = L14 ? wave : wave % WN;
    const int rowb = L14 ? 0 : wm * 64, colb = L14 ? wn * 32 : wn * 64;  // first tile row / column of the wave
    if constexpr (CB) {  // (visible after the barrier that opens the first pass)
        if (((uint32_t)(uintptr_t)smem_all & 0xFFFFu) != 0) __builtin_trap();  // the lookups write the nibble into byte 1 of an address based at 0
        for (int e = wave; e < 16; e += NWAVES) reinterpret_cast<uint32_t*>(smem_all)[e * 64 + lane] = code[e];
    }
    const uint32_t cb_addr = lane * 4;
    int mt_, nt_, z_;
    g2_tile_of(mp, blockIdx.x, mt_, nt_, z_);
    // SPLIT: the K range AND, for a SwiGLU pair, the matrix (fc_1 / fc_2) are grid dimensions - z = pass * ksplit + split: every
    // workgroup makes ONE pass and writes one slab of partial sums.  (The pair used to run as a two-pass loop in the split kernel
    // too: that variant needed 256 VGPRs + 14 - 16 spilled ones, 60 - 68 bytes of scratch per lane; the two-pass loop is now
    // only in the unsplit kernel, which keeps the gate in registers for the fused epilogue.)
    const int pass_begin = SPLIT ? z_ / ksplit : 0, pass_end = SPLIT ? pass_begin + 1 : (SWI ? 2 : 1);
    const int m0 = mt_ * G2M, n0 = nt_ * TN, zsplit = SPLIT ? z_ - pass_begin * ksplit : 0;
    const int lr = lane & 31, lh = lane >> 5;
    const int ktiles = K / G2K;
    const int Gt = plan.Gs / 2;  // K-steps per quantisation group
    // split z owns the whole quantisation groups [z * G / ksplit, (z + 1) * G / ksplit) (G = number of groups): ranges need not be equal
    const int G_all = (ktiles + Gt - 1) / Gt;
    const int kt_begin = SPLIT ? (int)((int64_t)zsplit * G_all / ksplit) * Gt : 0;
    const int kt_end = SPLIT ? min((int)((int64_t)(zsplit + 1) * G_all / ksplit) * Gt, ktiles) : ktiles;
    const int g_end = (kt_end + Gt - 1) / Gt;

    // ---- LDS-DMA sources.  Everything this kernel reads from global memory inside the K loop goes straight to LDS, so that all
    // outstanding loads are counted on vmcnt in issue order and waited for with exact counts (5 per wave and K-step).
    const int l_row = lane >> 3, l_slot = lane & 7;
    const bf16_t* a_src[A_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
        const int r = (wave * A_PER_WAVE + i) * 8 + l_row;
        a_src[i] = A + (int64_t)min(m0 + r, M - 1) * lda + (l_slot ^ ((r >> 1) & 7)) * 8;
    }
    // packed B: wave w brings rows 32 w .. 32 w + 31, lane l the 16-byte unit l of that KB: row 32 w + l / 2, physical half l % 2,
    // which holds slice 2 kt + (half ^ ((row >> 3) & 1)) - rows 8 apart swap halves, so that the 16 lanes of a ds_read_b128 group
    // (one row each, same half) land on 16 different 16-byte slots
    const int bl_row = wave * 32 + (lane >> 1);
    const int bl_half = (lane & 1) ^ ((bl_row >> 3) & 1);
#ifdef G2W_STUB_BSAME  // diagnostic: every workgroup streams the SAME 128 weight rows (L2 hits): is the loop waiting for B from HBM?
    const int64_t bl_rec = (int64_t)bl_row * plan.row16 + bl_half;
#else
    const int64_t bl_rec = (int64_t)min(n0 + bl_row, N - 1) * plan.row16 + bl_half;
#endif
    // metadata: waves 0, 1 bring the activation sums of rows 64 w + lane, waves 2, 3 the {scale, zero} words of columns 64 (w - 2) + lane
    const int64_t ml_rec = (int64_t)min(n0 + max(wave - 2, 0) * 64 + lane, N - 1) * plan.row16;
    const float* xs_src = xsT + m0 + (wave & 1) * 64 + lane;

    int a_row[IM], c_row[JN], b_unit[JN];
#pragma unroll
    for (int i = 0; i < IM; ++i) a_row[i] = rowb + i * 32 + lr;
#pragma unroll
    for (int j = 0; j < JN; ++j) {
        c_row[j] = colb + j * 32 + lr;
        b_unit[j] = c_row[j] * 2 + (lh ^ ((c_row[j] >> 3) & 1));
    }
    uint4* const meta = smem + W4_NBUF * W4_STAGE16;

    f32x16_t total[IM][JN];
    uint32_t gate[SWI && !SPLIT ? IM : 1][SWI && !SPLIT ? JN : 1][8];
    static_assert(!(SPLIT && SWI), "a split SwiGLU pair runs the plain split kernel with the pass in the grid");
    // The K loop is bound by instruction ISSUE, not by the matrix pipe (16 MFMAs per wave and step are 512 pipe cycles; the first
    // version of this loop issued 512 instructions per step - 203 scalar ones for slab walks, 64-bit addresses and M0 saves,
    // 64 v_mov to restart the accumulators - and one wave issues one instruction per 4 cycles: 1.0 us per step, measured alone
    // on a CU and no better with two workgroups, profiles/r03a).  So: per-lane 32-bit offsets against wave-uniform 64-bit
    // bases that advance by scalar adds (saddr form of the LDS-DMA), one asm statement per tile with M0 stepped in place, slab
    // and metadata positions kept as cursors, the K loop nested by quantisation group so that a group's first step starts its
    // accumulators with a zero-C MFMA (no v_mov), and (x >> s) & mask | magic as shift + v_and_or_b32.
    const uint32_t b_off = (uint32_t)(bl_rec * 16);
    uint32_t a_off[A_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) a_off[i] = (uint32_t)((const char*)a_src[i] - (const char*)A);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;  // LDS byte address of stage 0 (wave-uniform)
    const uint32_t mask4 = 0x000F000Fu;
    // The expansion is written in plain C with the magic constant hidden in a VGPR, so that hipcc selects v_and_or_b32 (one
    // literal operand) ITSELF.  The same instruction issued from inline asm (w4_plan.h and_or, fine in front of a dot2) is
    // invisible to the compiler's hazard recognizer: an MFMA that read its B operand one wait state behind the asm's write got
    // the old register - run-to-run different results in exactly the sub-tile whose MFMA follows the expansion (found with
    // tools/probes/w4_gemm_repeat2.py; the compiler puts the wait states in when it sees the VALU write).
    uint32_t magic = 0x43004300u;
    asm("" : "+v"(magic));
    for (int pass = pass_begin; pass < pass_end; ++pass) {
        const uint4* Wp = pass ? Wq2 : Wq;
        f32x16_t acc[IM][JN];
#pragma unroll
        for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int j = 0; j < JN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = total[i][j][r] = 0.f;
        // ---- issue cursors (wave-uniform): the next K-step to issue, its slab, the byte positions of its A columns / B slices
        int i_slab = 0;
        while (i_slab + 1 < plan.nslabs && 2 * kt_begin >= plan.slab[i_slab + 1].slice0) ++i_slab;
        int i_next = i_slab + 1 < plan.nslabs ? plan.slab[i_slab + 1].slice0 : 0x7fffffff;  // first slice of the next slab
        int kt_issue = kt_begin;
        const char* a_ptr = reinterpret_cast<const char*>(A) + (int64_t)kt_begin * (G2K * 2);
        const char* b_ptr = reinterpret_cast<const char*>(Wp + plan.slab[i_slab].w_off16 + (2 * kt_begin - plan.slab[i_slab].slice0));
        auto issue_tile = [&](int buf) {
            if (2 * kt_issue >= i_next) {  // (rare: a slab holds up to 32 K-steps)
                ++i_slab;
                i_next = i_slab + 1 < plan.nslabs ? plan.slab[i_slab + 1].slice0 : 0x7fffffff;
                b_ptr = reinterpret_cast<const char*>(Wp + plan.slab[i_slab].w_off16);
            }
            const uint32_t st = lds0 + (uint32_t)buf * (W4_STAGE16 * 16);
            const uint32_t da = __builtin_amdgcn_readfirstlane(st + (uint32_t)wave * (A_PER_WAVE * 1024));
            const uint32_t db = __builtin_amdgcn_readfirstlane(st + G2_TILE16 * 16 + (uint32_t)wave * 1024);
            unsigned keep;
#ifdef G2W_STUB_NOLOAD
            (void)keep; (void)da; (void)db;
#elif !defined(G2W_OLD_ISSUE)
            if constexpr (A_PER_WAVE == 4)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %3, %7\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %4, %7\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %5, %7\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %6, %7\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %8, %9\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "s"(da), "s"(db), "v"(a_off[0]), "v"(a_off[1]), "v"(a_off[A_PER_WAVE > 2 ? 2 : 0]),
                               "v"(a_off[A_PER_WAVE > 3 ? 3 : 0]), "s"(a_ptr), "v"(b_off), "s"(b_ptr)
                             : "memory", "scc");
            else
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %3, %5\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %6, %7\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "s"(da), "s"(db), "v"(a_off[0]), "v"(a_off[1]), "s"(a_ptr), "v"(b_off), "s"(b_ptr)
                             : "memory", "scc");
#else
            {
                uint4* stp = smem + buf * W4_STAGE16;
#pragma unroll
                for (int i = 0; i < A_PER_WAVE; ++i) glds16_asm(a_ptr + a_off[i], stp + (wave * A_PER_WAVE + i) * 64);
                glds16_asm(b_ptr + b_off, stp + G2_TILE16 + wave * 64);
                (void)keep; (void)da; (void)db;
            }
#endif
            a_ptr += G2K * 2;
            b_ptr += 32;
            ++kt_issue;
        };
        int m_slab = i_slab;
        auto issue_meta = [&](int g) {  // one 4-byte LDS-DMA per wave
            uint4* mb = meta + (g & 1) * W4_META16;
            if (wave < 2) {
                if constexpr (!CB) glds4_asm(xs_src + (int64_t)g * Mpad, reinterpret_cast<float*>(mb) + wave * 64);
            } else if (wave < 2 + WN) {
                while (m_slab + 1 < plan.nslabs && 2 * g * Gt >= plan.slab[m_slab + 1].slice0) ++m_slab;  // (groups are issued in order)
                const uint32_t* msrc = reinterpret_cast<const uint32_t*>(Wp + ml_rec + plan.slab[m_slab].meta_off16) + (g - plan.slab[m_slab].g0);
                glds4_asm(msrc, reinterpret_cast<uint32_t*>(mb + 32) + (wave - 2) * 64);
            }
        };
        // Ring of NBUF stages; per wave and K-step TILE_LOADS LDS-DMA loads, retired in issue order.  Before the barrier of step t
        // each wave waits until only the loads of steps t+1 .. t+NBUF-2 may be outstanding: its pieces of step t (and any metadata
        // issued before) have landed; past the barrier everybody's have, and the stage of step t-1 is free for step t+NBUF-1.  The
        // metadata of group g+1 is issued at the start of group g (before that step's tile, so that the newest loads are always
        // whole tiles) into the other parity, which the fold of group g-1 finished reading before this step's barrier.
        __syncthreads();  // previous pass done with the stages and the metadata
        const int g_begin = kt_begin / Gt;
        issue_meta(g_begin);
#pragma unroll
        for (int d = 0; d < W4_NBUF - 1; ++d)
            if (kt_begin + d < kt_end) issue_tile(d);
        int buf = 0, kt = kt_begin;
        auto step = [&](auto first_tag, int g) {
            constexpr bool FIRST = decltype(first_tag)::value;
            // tiles still allowed in flight: min(NBUF - 2, steps left after this one); the count is an immediate
            const int ahead = min(W4_NBUF - 2, kt_end - 1 - kt);
            if (W4_NBUF > 5 && ahead >= 4)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * TILE_LOADS) : "memory");
            else if (W4_NBUF > 4 && ahead == 3)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * TILE_LOADS) : "memory");
            else if (W4_NBUF > 3 && ahead == 2)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * TILE_LOADS) : "memory");
            else if (ahead == 1)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * TILE_LOADS) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (FIRST && g + 1 < g_end) issue_meta(g + 1);
            if (kt_issue < kt_end) issue_tile(buf >= 1 ? buf - 1 : W4_NBUF - 1);
            const uint4* sa = smem + buf * W4_STAGE16;
            const uint4* sb = sa + G2_TILE16;
            // this lane's 2 x 4 dwords of packed weights: k-block lh * 4 + ks of the step goes to MFMA ks (any assignment of the
            // step's eight 8-k blocks to (MFMA, lane half) is valid as long as A uses the same one)
            uint32_t bw[JN][4];
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                const uint4 bq = sb[b_unit[j]];
                bw[j][0] = bq.x, bw[j][1] = bq.y, bw[j][2] = bq.z, bw[j][3] = bq.w;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int slot = lh * 4 + ks;
                bf16x8_t af[IM], bfr[JN];
#pragma unroll
                for (int i = 0; i < IM; ++i) af[i] = __builtin_bit_cast(bf16x8_t, sa[a_row[i] * 8 + (slot ^ ((a_row[i] >> 1) & 7))]);
#pragma unroll
                for (int i = 0; i < JN; ++i) {
                    uint32_t o[4];
                    if constexpr (CB) {
                        w4c_dword_lookup(bw[i][ks], cb_addr, o);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = ((bw[i][ks] >> (4 * q)) & mask4) | magic;  // (plain C: see the note at `magic`)
                    }
                    bfr[i] = __builtin_bit_cast(bf16x8_t, make_uint4(o[0], o[1], o[2], o[3]));
                }
#pragma unroll
                for (int i = 0; i < IM; ++i)
#pragma unroll
                    for (int j = 0; j < JN; ++j) {
#ifdef G2W_STUB_NOMFMA
                        if (true) {
                            asm volatile("" ::"v"(af[i]), "v"(bfr[j]));
                        } else
#endif
#ifdef G2W_NO_ZEROC
                        if (false) {
#else
                        if (FIRST && ks == 0) {  // the group's accumulators start here: C = 0 is an operand, not 64 v_mov
#endif
                            const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], zero, 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                        }
                    }
            }
            buf = buf + 1 == W4_NBUF ? 0 : buf + 1;
            ++kt;
        };
        for (int g = g_begin; g < g_end; ++g) {
            const int kg_end = min((g + 1) * Gt, kt_end);
            step(std::true_type{}, g);
            while (kt < kg_end) step(std::false_type{}, g);
#ifndef G2W_STUB_NOFOLD
            {  // group end: fold
                const uint4* mb = meta + (g & 1) * W4_META16;
                const float* xs_l = reinterpret_cast<const float*>(mb);
                const uint32_t* mt_l = reinterpret_cast<const uint32_t*>(mb + 32);
                float4 xs4[IM][4];
                if constexpr (!CB) {
#pragma unroll
                    for (int i = 0; i < IM; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) xs4[i][q] = *reinterpret_cast<const float4*>(xs_l + rowb + i * 32 + 8 * q + 4 * lh);
                }
#pragma unroll
                for (int j = 0; j < JN; ++j) {
                    const uint32_t mt = mt_l[c_row[j]];
                    const float sc = CB ? __uint_as_float(mt) : bflo(mt), zz = CB ? 0.f : 128.0f + bfhi(mt);
#pragma unroll
                    for (int i = 0; i < IM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if constexpr (CB) {
                                total[i][j][r] += sc * acc[i][j][r];
                            } else {
                                const float4 x4 = xs4[i][r >> 2];
                                const float xs = (r & 3) == 0 ? x4.x : ((r & 3) == 1 ? x4.y : ((r & 3) == 2 ? x4.z : x4.w));
                                total[i][j][r] += sc * (acc[i][j][r] - zz * xs);
                            }
#ifdef G2W_NO_ZEROC
                            acc[i][j][r] = 0.f;
#endif
                        }
                }
            }
#else
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < JN; ++j) total[i][j] += acc[i][j];
#endif
        }
        if constexpr (SPLIT) {
            float* dst = (pass ? part2 : part) + (int64_t)zsplit * M * N;
            if (m0 + G2M <= M && n0 + TN <= N) {  // interior tile (wave-uniform): 64 stores off one base, no per-element bounds
                float* d0 = dst + (int64_t)(m0 + rowb + 4 * lh) * N + n0;
#pragma unroll
                for (int i = 0; i < IM; ++i)
#pragma unroll
                    for (int j = 0; j < JN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) d0[(int64_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * N + c_row[j]] = total[i][j][r];
            } else {
#pragma unroll
                for (int i = 0; i < IM; ++i)
#pragma unroll
                    for (int j = 0; j < JN; ++j) {
                        const int col = n0 + c_row[j];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = m0 + rowb + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            if (row < M && col < N) dst[(int64_t)row * N + col] = total[i][j][r];
                        }
                    }
            }
        } else if (SWI && pass == 0) {
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < JN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const bf16_t g0 = f2bf(silu(rbf(total[i][j][r]))), g1 = f2bf(silu(rbf(total[i][j][r + 1])));
                        gate[SWI && !SPLIT ? i : 0][SWI && !SPLIT ? j : 0][r >> 1] = (uint32_t)g0 | ((uint32_t)g1 << 16);
                    }
        }
    }
    if constexpr (SPLIT) return;
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int col = n0 + c_row[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + rowb + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M && col < N) {
                    bf16_t o;
                    if (SWI) {
                        const uint32_t gp = gate[SWI && !SPLIT ? i : 0][SWI && !SPLIT ? j : 0][r >> 1];
                        o = f2bf(((r & 1) ? bfhi(gp) : bflo(gp)) * rbf(total[i][j][r]));
                    } else {
                        o = apply_epilogue(epi, total[i][j][r], 0.f, bias, residual ? residual + (int64_t)row * ldr : nullptr, col);
                    }
                    out[(int64_t)row * ldo + col] = o;
                }
            }
        }
}

// the int4 kernel takes whole K-steps per group and slab boundaries on K-step boundaries
bool gemm2_w4_takes(const W4Plan& plan, int K) {
    if (!gemm2_enabled() || K % G2K != 0 || plan.Gs % 2 != 0) return false;
    for (int c = 0; c < plan.nslabs; ++c)
        if (plan.slab[c].slice0 % 2 != 0) return false;
    return true;
}

// tile shape of the int4 kernel (wave columns)
int gemm2_w4_wn(int M) {
    const int env = tune_env("PARROT_GEMM2_W4_WN", 0);  // PARROT_GEMM2_W4_WN = 2 | 4 (A/B)
    (void)M;
    // measured (Llama-2-7B int4, 128- and 32-token prompts, uneven split-K): WN = 2 3.97 / 3.80 ms, WN = 4 4.77 / 4.33 ms - the wide
    // shape does not pay; it stays selectable for A/B
    return env == 4 ? 4 : 2;
}

int gemm2_w4_ksplit(int M, int N, int K, const W4Plan& plan) {
    const int wn = gemm2_w4_wn(M);
    const int tn = wn * 64;
    const int64_t tiles = (int64_t)((M + G2M - 1) / G2M) * ((N + tn - 1) / tn);
    const int ktiles = K / G2K, Gt = plan.Gs / 2;
    const int env_target = tune_env("PARROT_GEMM2_W4_SPLIT_TARGET", 0), env_max = tune_env("PARROT_GEMM2_W4_KSMAX", 8);  // (A/B)
    const int target = env_target > 0 ? env_target : (wn == 4 ? 256 : 512);  // workgroups that fill the chip
    // launches with fewer tiles than this split K; 256 tiles still split two ways: two workgroups fit a CU (Falcon-40B 128-token prefill 25.8 -> 23.2 ms)
    const int thr = tune_env("PARROT_GEMM2_W4_SPLIT_BELOW", 257);
    int ks = tiles >= (wn == 4 ? 128 : thr) ? 1 : (int)(target / (tiles > 0 ? tiles : 1));
    if (ks > env_max) ks = env_max;
    const int G_all = (ktiles + Gt - 1) / Gt;
    while (ks > 1 && (G_all / ks < 1 || ktiles / ks < 4)) --ks;  // whole groups per split, ranges need not be equal
    return ks < 1 ? 1 : ks;
}

int64_t gemm2_w4_xs_floats(int M, const W4Plan& plan) { return (int64_t)((M + G2M - 1) / G2M) * G2M * plan.ngroups; }

int gemm2_w4_launch(const void* Wq, const void* Wq2, const void* x, int ldx, int M, const void* bias, const void* residual, int ldr,
                    void* out, int ldo, int N, int K, int epilogue, float* workspace, const W4Plan& plan, hipStream_t st, int* ksplit_out,
                    float** part_out, float** part2_out, const void* code, bool have_xs) {
    const int Mpad = (M + G2M - 1) / G2M * G2M;
    const int64_t nxs = code ? 0 : (int64_t)Mpad * plan.ngroups;  // codebook weights need no activation sums
    int rc = PARROT_OK;
    if (!code && !have_xs)  // (have_xs: the caller's fused norm wrote the sums at the head of the workspace)
        rc = launch(K_GEMM_XSUM, gemm2_xsum_kernel, dim3((unsigned)((nxs * 16 + 255) / 256)), dim3(256), 0, st, (const bf16_t*)x, ldx, M, Mpad, K,
                    plan.Gs * 32, plan.ngroups, workspace);
    if (rc != PARROT_OK) return rc;
    const int ks = gemm2_w4_ksplit(M, N, K, plan);
    *ksplit_out = ks;
    float* part = workspace + nxs;
    float* part2 = part + (int64_t)ks * M * N;
    *part_out = part;
    *part2_out = epilogue == PARROT_EPI_SWIGLU ? part2 : nullptr;
    const int wn = gemm2_w4_wn(M);
    G2Map mp;
    mp.MT = (M + G2M - 1) / G2M;
    mp.NT = (N + wn * 64 - 1) / (wn * 64);
    const bool swi = epilogue == PARROT_EPI_SWIGLU;
    const int64_t total = (int64_t)mp.MT * mp.NT * ks * (ks > 1 && swi ? 2 : 1);  // split SwiGLU pair: fc_1 and fc_2 as a grid dimension
    PARROT_UNSUPPORTED(total < (1ll << 31), "w4_gemm: too many tiles");
    const int resident_per_xcd = wn == 4 ? 32 : 64;
    mp.xcd_ok = (mp.NT % 8 == 0);
    mp.GM = g2_largest_divisor_le(mp.MT, 8);
    mp.GN = mp.xcd_ok ? g2_largest_divisor_le(mp.NT / 8, resident_per_xcd / mp.GM > 0 ? resident_per_xcd / mp.GM : 1) : 1;
    const dim3 grid((unsigned)total);
#define PARROT_G2W_GO_CB(SPLITV, SWIV, WNV, CBV)                                                                                    \
    return launch(CBV ? K_W4C_GEMM : K_W4_GEMM, gemm2_w4_kernel<SPLITV, SWIV, WNV, CBV>, grid, dim3(WNV * 128), 0, st, (const bf16_t*)x, ldx, \
                  M, (const uint4*)Wq, (const uint4*)Wq2, N, K, (const float*)workspace, Mpad, (const bf16_t*)bias, (const bf16_t*)residual, \
                  ldr, (bf16_t*)out, ldo, epilogue, plan, ks, part, part2, mp, (const uint32_t*)code)
#define PARROT_G2W_GO(SPLITV, SWIV, WNV) PARROT_G2W_GO_CB(SPLITV, SWIV, WNV, false)
    if (code) {  // codebook weights: the 128 x 128 shape only
        if (ks > 1) PARROT_G2W_GO_CB(true, false, 2, true);
        if (swi) PARROT_G2W_GO_CB(false, true, 2, true);
        PARROT_G2W_GO_CB(false, false, 2, true);
    }
    if (wn == 4) {
        if (ks > 1) PARROT_G2W_GO(true, false, 4);
        if (swi) PARROT_G2W_GO(false, true, 4);
        PARROT_G2W_GO(false, false, 4);
    }
    if (ks > 1) PARROT_G2W_GO(true, false, 2);
    if (swi) PARROT_G2W_GO(false, true, 2);
    PARROT_G2W_GO(false, false, 2);
#undef PARROT_G2W_GO
#undef PARROT_G2W_GO_CB
}

}  // namespace parrot
