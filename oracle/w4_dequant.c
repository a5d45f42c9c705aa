/* Plain-C restatement of the int4 unpack + dequant of the reference's ColBlockQuantizedLinear
 * (quantize/gptq.py:243-252) and of a dequant-GEMV on top of it.  TEST INFRASTRUCTURE ONLY.
 *
 * quant_weight memory is [K/2][N] bytes (logical (N, K/2) with strides (1, N), gptq.py:216-222); byte j of
 * output row o holds column 2j in its low nibble and 2j+1 in its high nibble (gptq.py:240-241, :246-247).
 * scales / zeros are [N][ceil(K/tile_cols)] floats.  All arithmetic in float.
 */
#include <stdint.h>
#include <stddef.h>

void oracle_w4_dequant(const uint8_t* qw, const float* scales, const float* zeros, int N, int K, int tile_cols,
                       float* out /* [N][K] */) {
    const int ng = (K + tile_cols - 1) / tile_cols;
    for (int o = 0; o < N; ++o)
        for (int k = 0; k < K; ++k) {
            const uint8_t b = qw[(size_t)(k / 2) * N + o];
            const int q = (k & 1) ? (b >> 4) : (b & 15);
            const int g = k / tile_cols;
            out[(size_t)o * K + k] = ((float)q - zeros[(size_t)o * ng + g]) * scales[(size_t)o * ng + g];
        }
}

/* y[o] = sum_k x[k] * dequant(o, k), accumulated in double to serve as a summation-order-free check */
void oracle_w4_gemv(const uint8_t* qw, const float* scales, const float* zeros, const float* x, int N, int K,
                    int tile_cols, double* y) {
    const int ng = (K + tile_cols - 1) / tile_cols;
    for (int o = 0; o < N; ++o) {
        double acc = 0.0;
        for (int k = 0; k < K; ++k) {
            const uint8_t b = qw[(size_t)(k / 2) * N + o];
            const int q = (k & 1) ? (b >> 4) : (b & 15);
            const int g = k / tile_cols;
            acc += (double)x[k] * (double)(((float)q - zeros[(size_t)o * ng + g]) * scales[(size_t)o * ng + g]);
        }
        y[o] = acc;
    }
}
