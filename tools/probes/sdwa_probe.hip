// Probe: does v_and_b32_sdwa with an inline constant / a VGPR mask and UNUSED_PRESERVE do what w4c_slice_lookup assumes,
// and do ds_read_u16_d16 / _d16_hi fill the two halves of one register?   hipcc --offload-arch=gfx950 -O2 sdwa_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe(const uint32_t* in, uint32_t* out) {
    __shared__ __attribute__((aligned(4096))) uint32_t tbl[16 * 64];
    const int lane = threadIdx.x;
    for (int e = 0; e < 16; ++e) tbl[e * 64 + lane] = 0xAB00u + e * 16 + (lane & 15);  // low half identifies (e, lane)
    __syncthreads();
    const uint32_t dw = in[lane];
    uint32_t base = (uint32_t)(uintptr_t)tbl + lane * 4;
    uint32_t a = base, b = base, m = 15;
    asm volatile("v_and_b32_sdwa %0, 15, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2" : "+v"(a) : "v"(dw));
    asm volatile("v_and_b32_sdwa %0, %2, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2" : "+v"(b) : "v"(dw), "v"(m));
    uint32_t p = 0xFFFFFFFFu, a0 = base, a1 = base;
    asm volatile("v_and_b32_sdwa %1, %4, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0\n\t"
                 "ds_read_u16_d16 %0, %1\n\t"
                 "v_and_b32_sdwa %1, %4, %3 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2\n\t"
                 "ds_read_u16_d16_hi %0, %1\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "+v"(p), "+v"(a0)
                 : "v"(dw), "v"(dw), "v"(m)
                 : "memory");
    (void)a1;
    out[lane * 4 + 0] = a;
    out[lane * 4 + 1] = b;
    out[lane * 4 + 2] = p;
    out[lane * 4 + 3] = base;
}

int main() {
    uint32_t h_in[64], h_out[256], *d_in, *d_out;
    for (int i = 0; i < 64; ++i) h_in[i] = 0x9C3A5E71u * (i + 1) + 0x1234567u;
    hipMalloc(&d_in, sizeof h_in);
    hipMalloc(&d_out, sizeof h_out);
    hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_in, d_out);
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    int bad_a = 0, bad_b = 0, bad_p = 0;
    for (int l = 0; l < 64; ++l) {
        const uint32_t dw = h_in[l], base = h_out[l * 4 + 3];
        const uint32_t want_addr = base | (((dw >> 16) & 15u) << 8);
        const uint32_t lo = 0xAB00u + (dw & 15u) * 16 + (l & 15), hi = 0xAB00u + ((dw >> 16) & 15u) * 16 + (l & 15);
        bad_a += h_out[l * 4 + 0] != want_addr;
        bad_b += h_out[l * 4 + 1] != want_addr;
        bad_p += h_out[l * 4 + 2] != (lo | (hi << 16));
        if (l < 4) printf("lane %d dw=%08x base=%08x a=%08x b=%08x want=%08x p=%08x wantp=%08x\n", l, dw, base, h_out[l * 4], h_out[l * 4 + 1], want_addr, h_out[l * 4 + 2], lo | (hi << 16));
    }
    printf("inline-constant sdwa wrong: %d/64, vgpr-mask sdwa wrong: %d/64, d16 pair wrong: %d/64\n", bad_a, bad_b, bad_p);
    return 0;
}
