// W4K layout plan (shared by w4.hip and persist.hip): how the K axis of an int4 matrix is cut into per-wave slabs.
#pragma once
#include "parrot_common.h"

namespace parrot {

constexpr int kMaxSlabs = 16;

struct W4Slab {
    int slice0;      // first slice (32-k unit) of the slab
    int nslices;     // <= 64
    int g0;          // first group overlapping the slab
    int ngroups;     // groups overlapping the slab
    int w_off16;     // offset of the slab's weight slices inside a row record, in 16-B units
    int meta_off16;  // offset of the slab's meta block inside a row record, in 16-B units
};

struct W4Plan {
    int nslabs;
    int row16;  // row record size in 16-B units
    int Gs;     // slices per group
    int nslices;
    int ngroups;
    W4Slab slab[kMaxSlabs];
};

inline int w4_make_plan(int N, int K, int group, W4Plan* p) {
    PARROT_REQUIRE(N > 0 && K > 0, "w4: N and K must be positive (N=%d K=%d)", N, K);
    PARROT_UNSUPPORTED(K % 32 == 0, "w4: K=%d must be a multiple of 32", K);
    if (group <= 0 || group > K) group = K;  // tile_cols = -1 -> per-channel (quantize/gptq.py:210)
    PARROT_UNSUPPORTED(group % 32 == 0, "w4: group size %d must be a multiple of 32", group);
    const int nslices = K / 32;
    const int Gs = group / 32;
    // slab boundaries fall on group starts when a group fits a slab; otherwise (per-channel scales, huge groups) on EVEN slices,
    // i.e. on the 64-deep K-steps of the prompt GEMM (gemm2_w4_takes: a slab that starts on an odd slice sent Falcon-7B's
    // per-channel gptq.int4 prompts to the first-generation kernel - 45.8 ms for 128 tokens)
    const int unit = Gs <= 64 ? Gs : (nslices % 2 == 0 ? 2 : 1);
    const int units = (nslices + unit - 1) / unit;
    int nslabs = (nslices + 63) / 64;
    while (nslabs <= kMaxSlabs && ((units + nslabs - 1) / nslabs) * unit > 64) ++nslabs;
    PARROT_UNSUPPORTED(nslabs <= kMaxSlabs, "w4: K=%d group=%d needs more than %d slabs", K, group, kMaxSlabs);
    p->nslabs = nslabs;
    p->Gs = Gs;
    p->nslices = nslices;
    p->ngroups = (nslices + Gs - 1) / Gs;
    int off = 0;
    for (int c = 0; c < nslabs; ++c) {
        const int u0 = (int)((int64_t)c * units / nslabs), u1 = (int)((int64_t)(c + 1) * units / nslabs);
        W4Slab& s = p->slab[c];
        s.slice0 = u0 * unit;
        const int slice1 = (u1 * unit < nslices) ? u1 * unit : nslices;
        s.nslices = slice1 - s.slice0;
        s.g0 = s.slice0 / Gs;
        s.ngroups = (slice1 - 1) / Gs - s.g0 + 1;
        s.w_off16 = off;
        off += s.nslices;
        s.meta_off16 = off;
        off += (s.ngroups * 4 + 15) / 16;
    }
    for (int c = nslabs; c < kMaxSlabs; ++c) p->slab[c] = W4Slab{0, 0, 0, 0, 0, 0};
    p->row16 = off;
    return PARROT_OK;
}

// fp32 sum over one 16-byte slice (32 weights) of x[k] * (128 + q[k]); xr = the lane's 16 packed bf16 activation pairs
// (a & mask) | magic in ONE instruction.  The compiler emits v_and_b32 + v_or_b32 with two literals here (gfx9 VOP3
// takes no literal and one scalar operand): with the magic held in a VGPR and the mask in an SGPR the fused form is legal,
// and it is 16 instructions less per 32 weights in a loop that is VALU-bound.
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask_s, uint32_t magic_v) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask_s), "v"(magic_v));
    return r;
}
__device__ __forceinline__ float w4_slice_dot(const uint4 w, const uint32_t (&xr)[16]) {
    const uint32_t dw[4] = {w.x, w.y, w.z, w.w};
    float p0 = 0.f, p1 = 0.f;
    const uint32_t mask = 0x000F000Fu;
    uint32_t magic = 0x43004300u;
    asm("" : "+v"(magic));  // keep it in a VGPR (not re-materialised as a literal per use)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t pk = and_or(dw[d] >> (4 * i), mask, magic);
            if (i & 1)
                p1 = dot2_bf16(pk, xr[4 * d + i], p1);
            else
                p0 = dot2_bf16(pk, xr[4 * d + i], p0);
        }
    }
    return p0 + p1;
}

// Codebook weights (NF4 / FP4): the 32 nibbles of a slice -> 16 bf16 pairs {code[q[2j]], code[q[2j+1]]} through the lane's
// private codebook column in LDS (cb_col = table + lane * 4, entry e at byte e * 256).  Nibble order inside a dword as in
// W4K: bits 4i <-> k = 8d + 2i, bits 16 + 4i <-> k = 8d + 2i + 1.
// One instruction per address: v_and_b32_sdwa takes byte b of the (possibly >> 4) weight dword, masks the nibble and writes
// it into byte 1 of an address register whose byte 0 permanently holds lane * 4 (dst_unused:UNUSED_PRESERVE), i.e.
// address = nibble * 256 + lane * 4.  The two codebook values of a pair are read with ds_read_u16 (zero-extended) and joined
// by one v_lshl_or_b32.  (ds_read_u16_d16_hi does NOT merge into the other half on this part: with SRAM ECC the D16 loads
// clear the unused half - measured with tools/probes/sdwa_probe.hip - which is also why the compiler never emits them.)
// The compiler does not track LDS reads issued from inline asm: the slice's reads are waited for in one place below.
#define W4C_READ(R, A, SRC, SEL)                                                                                         \
    asm volatile("v_and_b32_sdwa %1, 15, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:" SEL "\n\t" \
                 "ds_read_u16 %0, %1"                                                                                     \
                 : "=&v"(R), "+v"(A)                                                                                      \
                 : "v"(SRC)                                                                                               \
                 : "memory")
__device__ __forceinline__ void w4c_slice_lookup(const uint4 w, uint32_t cb_addr, uint32_t (&pairs)[16]) {
    const uint32_t dw[4] = {w.x, w.y, w.z, w.w};
    uint32_t a0 = cb_addr, a1 = cb_addr;  // byte 0 = lane * 4 (table at LDS offset 0), byte 1 is rewritten per lookup
    uint32_t lo[16], hi[16];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t lo4 = dw[d], hi4 = dw[d] >> 4;
        W4C_READ(lo[4 * d + 0], a0, lo4, "BYTE_0");
        W4C_READ(hi[4 * d + 0], a1, lo4, "BYTE_2");
        W4C_READ(lo[4 * d + 1], a0, hi4, "BYTE_0");
        W4C_READ(hi[4 * d + 1], a1, hi4, "BYTE_2");
        W4C_READ(lo[4 * d + 2], a0, lo4, "BYTE_1");
        W4C_READ(hi[4 * d + 2], a1, lo4, "BYTE_3");
        W4C_READ(lo[4 * d + 3], a0, hi4, "BYTE_1");
        W4C_READ(hi[4 * d + 3], a1, hi4, "BYTE_3");
    }
    // (an asm statement takes at most 30 operands: the wait carries the low halves, an empty statement behind it the high ones)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]),
                   "+v"(lo[8]), "+v"(lo[9]), "+v"(lo[10]), "+v"(lo[11]), "+v"(lo[12]), "+v"(lo[13]), "+v"(lo[14]), "+v"(lo[15])
                 :
                 : "memory");
    asm volatile(""
                 : "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7]),
                   "+v"(hi[8]), "+v"(hi[9]), "+v"(hi[10]), "+v"(hi[11]), "+v"(hi[12]), "+v"(hi[13]), "+v"(hi[14]), "+v"(hi[15])
                 :
                 : "memory");
#pragma unroll
    for (int j = 0; j < 16; ++j) pairs[j] = lo[j] | (hi[j] << 16);
}
#undef W4C_READ
__device__ __forceinline__ float w4c_pairs_dot(const uint32_t (&pairs)[16], const uint32_t (&xr)[16]) {
    float p0 = 0.f, p1 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        p0 = dot2_bf16(pairs[j], xr[j], p0);
        p1 = dot2_bf16(pairs[j + 1], xr[j + 1], p1);
    }
    return p0 + p1;
}

}  // namespace parrot
