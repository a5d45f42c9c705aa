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
    const int unit = Gs <= 64 ? Gs : 1;  // slab boundaries fall on group starts when a group fits a slab
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
// address = nibble * 256 + lane * 4.  ds_read_u16_d16 / _d16_hi deposit the two codebook values of a pair in the halves of
// one register.  The compiler does not track LDS reads issued from inline asm: the caller waits with w4c_wait().
#define W4C_PAIR(P, A, SRC_LO, SEL_LO, SRC_HI, SEL_HI)                                                              \
    asm volatile("v_and_b32_sdwa %1, 15, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:" SEL_LO "\n\t" \
                 "ds_read_u16_d16 %0, %1\n\t"                                                                       \
                 "v_and_b32_sdwa %1, 15, %3 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:" SEL_HI "\n\t" \
                 "ds_read_u16_d16_hi %0, %1"                                                                         \
                 : "=&v"(P), "+v"(A)                                                                                 \
                 : "v"(SRC_LO), "v"(SRC_HI)                                                                          \
                 : "memory")
__device__ __forceinline__ void w4c_slice_lookup(const uint4 w, uint32_t cb_addr, uint32_t (&pairs)[16]) {
    const uint32_t dw[4] = {w.x, w.y, w.z, w.w};
    uint32_t a0 = cb_addr, a1 = cb_addr;  // byte 0 = lane * 4 (+ table base), byte 1 is rewritten per lookup
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t lo4 = dw[d], hi4 = dw[d] >> 4;
        W4C_PAIR(pairs[4 * d + 0], a0, lo4, "BYTE_0", lo4, "BYTE_2");
        W4C_PAIR(pairs[4 * d + 1], a1, hi4, "BYTE_0", hi4, "BYTE_2");
        W4C_PAIR(pairs[4 * d + 2], a0, lo4, "BYTE_1", lo4, "BYTE_3");
        W4C_PAIR(pairs[4 * d + 3], a1, hi4, "BYTE_1", hi4, "BYTE_3");
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(pairs[0]), "+v"(pairs[1]), "+v"(pairs[2]), "+v"(pairs[3]), "+v"(pairs[4]), "+v"(pairs[5]), "+v"(pairs[6]),
                   "+v"(pairs[7]), "+v"(pairs[8]), "+v"(pairs[9]), "+v"(pairs[10]), "+v"(pairs[11]), "+v"(pairs[12]),
                   "+v"(pairs[13]), "+v"(pairs[14]), "+v"(pairs[15])
                 :
                 : "memory");
}
#undef W4C_PAIR
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
