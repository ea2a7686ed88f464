// conv11.hip -- conv1_1 (3 -> 64 channels, K = 27) fused with read_image_data's arithmetic, bf16 (gfx950).
//
// conv1_1 is HBM-bound (output 128 B/pixel against 27 MACs x 64): the kernel reads the decoded uint8 crop (or the
// preprocessed float tensor) directly, forms the 27-wide im2col row (pixel - mean, zero outside the image) for 256
// output pixels in LDS, multiplies by the [64][32] weight tile with two v_mfma_f32_32x32x16_bf16 per 32x32 tile,
// adds bias, applies ReLU, stages the bf16 result through LDS and writes every output pixel's 64 channels as one
// contiguous 128-byte row (16 B per lane).  No im2col matrix is ever written to HBM.
// Replaces: read_image_data's tail (lrcn.jl:766-772) + convx/relux for the first layer (lrcn.jl:724-725).
#include "common.h"
#include "kernels.h"

namespace {

template <bool U8>
__global__ __launch_bounds__(256) void conv11_kernel(const void *src, int N, int S, float m0, float m1, float m2,
                                                     const bf16_t *w /* [64][32] */, const float *bias, bf16_t *out) {
    // LDS: A tile 256 rows x 64 B (swizzled 16-B chunks), W tile 64 rows x 64 B, C tile 256 rows x 128 B
    __shared__ __attribute__((aligned(16))) unsigned char smem[256 * 64 + 64 * 64 + 256 * 128];
    unsigned char *As = smem, *Ws = smem + 256 * 64, *Cs = smem + 256 * 64 + 64 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t M = (int64_t)N * S * S;
    const int64_t mrow0 = (int64_t)blockIdx.x * 256;

    // ---- weights -> LDS (64 rows x 4 chunks of 16 B = 256 chunks, one per thread) ----
    {
        const int r = tid >> 2, c = tid & 3;
        *reinterpret_cast<uint4 *>(Ws + r * 64 + ((c ^ ((r >> 2) & 3)) << 4)) = *reinterpret_cast<const uint4 *>(w + r * 32 + c * 8);
    }
    // ---- im2col row of pixel m = mrow0 + tid -> LDS ----
    {
        const int64_t m = mrow0 + tid;
        bf16_t row[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) row[k] = (bf16_t)0.0f;
        if (m < M) {
            const PixDecode p = decode_pixel((int)m, S, S);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                const int y = p.y + kh - 1, x = p.x + kw - 1;  // internal (y, x) = reference (dim 2, dim 1)
                if ((unsigned)y < (unsigned)S && (unsigned)x < (unsigned)S) {
                    if (U8) {  // crop img[n][row = x][col = y][c]   (lrcn.jl:766-771)
                        const uint8_t *px = reinterpret_cast<const uint8_t *>(src) + (((int64_t)p.n * S + x) * S + y) * 3;
                        row[tap * 3 + 0] = (bf16_t)((float)px[0] - m0);
                        row[tap * 3 + 1] = (bf16_t)((float)px[1] - m1);
                        row[tap * 3 + 2] = (bf16_t)((float)px[2] - m2);
                    } else {  // preprocessed (S,S,3,N) column-major float tensor
                        const float *f = reinterpret_cast<const float *>(src);
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            row[tap * 3 + c] = (bf16_t)f[(int64_t)x + (int64_t)S * (y + (int64_t)S * (c + 3ll * p.n))];
                    }
                }
            }
        }
        const uint4 *rv = reinterpret_cast<const uint4 *>(row);
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<uint4 *>(As + tid * 64 + ((c ^ ((tid >> 2) & 3)) << 4)) = rv[c];
    }
    __syncthreads();

    // ---- MFMA: wave w owns rows [64w, 64w+64) x 64 columns; K = 32 = two 32x32x16 steps ----
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.0f;
    const int r31 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        uint4 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 64 + i * 32 + r31;
            af[i] = *reinterpret_cast<const uint4 *>(As + row * 64 + (((j * 2 + hh) ^ ((row >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int row = n * 32 + r31;
            bf[n] = *reinterpret_cast<const uint4 *>(Ws + row * 64 + (((j * 2 + hh) ^ ((row >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                    __builtin_bit_cast(bf16x8, bf[n]), acc[i][n], 0, 0, 0);
    }
    // ---- bias + ReLU -> bf16 -> LDS C tile [256][64] (row = 128 B) ----
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = n * 32 + r31;
            const float b = bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                *reinterpret_cast<bf16_t *>(Cs + row * 128 + col * 2) = (bf16_t)fmaxf(acc[i][n][r] + b, 0.0f);
            }
        }
    __syncthreads();
    // ---- coalesced write-out: 8 lanes x 16 B per pixel ----
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int row = pass * 32 + (tid >> 3), c = tid & 7;
        const int64_t m = mrow0 + row;
        if (m < M) {
            const PixDecode p = decode_pixel((int)m, S, S);
            bf16_t *dst = out + (((int64_t)p.n * S + p.y) * S + p.x) * 64 + c * 8;
            *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(Cs + row * 128 + c * 16);
        }
    }
}

}  // namespace

void k_conv11_fused(hipStream_t st, int src_is_u8, const void *src, int N, int S, float m0, float m1, float m2, const void *w,
                    const float *bias, void *out) {
    const int64_t M = (int64_t)N * S * S;
    const unsigned grid = (unsigned)((M + 255) / 256);
    if (src_is_u8)
        hipLaunchKernelGGL(conv11_kernel<true>, dim3(grid), dim3(256), 0, st, src, N, S, m0, m1, m2, (const bf16_t *)w, bias,
                           (bf16_t *)out);
    else
        hipLaunchKernelGGL(conv11_kernel<false>, dim3(grid), dim3(256), 0, st, src, N, S, m0, m1, m2, (const bf16_t *)w, bias,
                           (bf16_t *)out);
}
