// fp8.hip -- OCP e4m3 plumbing of the VGG convolution stack (BASELINE config 5: fp8 conv stack for caption generation).
//
// Quantisation scheme (DESIGN.md "fp8 convolution stack"):
//   weights      w8[co][tap][ci] = e4m3(w / sw[co]),  sw[co] = amax_co / 448          (per output channel)
//   activations  x8 = e4m3(x / sa),                    sa = margin * amax / 448        (per tensor, from lrcn_vgg_calibrate)
//   layer        y8 = e4m3(relu(acc * escale[co] + ebias[co])),  escale = sa_in sw / sa_out,  ebias = b / sa_out
// (the 2x2 max-pool, where fused, commutes with the positive scale and the monotone rounding).
// The reference has no reduced-precision path (lrcn.jl:724-728 runs conv4 in Float32); parity of this path is stated
// against the fp32 CPU oracle with the tolerance written in tests/test_gpu_fp8.py.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float clamp448(float v) { return fminf(fmaxf(v, -448.f), 448.f); }
__device__ __forceinline__ unsigned char to_e4m3(float v) {
    v = clamp448(v);
    return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false) & 0xFF);
}
__device__ __forceinline__ float from_e4m3(unsigned char b) { return __builtin_amdgcn_cvt_f32_fp8((int)b, 0); }

__device__ __forceinline__ float block_max(float v, float *sh) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, sh[i]);
    __syncthreads();
    return m;
}

// w(a,b,ci,co) at a + 3*(b + 3*(ci + Cin*co)) (lrcn.jl:724 conv4 weights)  ->  out[co][tap = b*3 + a][ci] e4m3, sw[co]
__global__ __launch_bounds__(256) void quant_conv_w_fp8_kernel(const float *w, int Cin, int Cout, unsigned char *out, float *sw) {
    __shared__ float sh[4];
    const int co = blockIdx.x;
    const float *wc = w + (int64_t)9 * Cin * co;
    float m = 0.0f;
    for (int i = threadIdx.x; i < 9 * Cin; i += 256) m = fmaxf(m, fabsf(wc[i]));
    m = block_max(m, sh);
    const float s = m > 0.0f ? m / 448.0f : 1.0f;
    if (threadIdx.x == 0) sw[co] = s;
    const float inv = 1.0f / s;
    for (int i = threadIdx.x; i < 9 * Cin; i += 256) {
        const int ci = i % Cin, tap = i / Cin;
        const int b = tap / 3, a = tap - 3 * b;
        out[(int64_t)co * 9 * Cin + i] = to_e4m3(wc[a + 3 * (b + 3 * ci)] * inv);
    }
}

template <typename T> __global__ __launch_bounds__(256) void amax_kernel(const T *x, int64_t n, float *out) {
    __shared__ float sh[4];
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(to_f32(x[i])));
    m = block_max(m, sh);
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned *>(out), __float_as_uint(m));  // m >= 0: uint order = float order
}

// 8 elements per thread: one 16-byte bf16 load -> one 8-byte e4m3 store (n % 8 == 0, 16-byte aligned buffers)
__global__ __launch_bounds__(256) void cast_bf16_fp8_kernel(const bf16_t *x, int64_t n8, float inv_scale, uint2 *out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + 8 * i);
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = clamp448((float)v[k] * inv_scale);
        uint2 o;
        o.x = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
        o.x = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], o.x, true);
        o.y = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
        o.y = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], o.y, true);
        out[i] = o;
    }
}
__global__ __launch_bounds__(256) void cast_fp8_bf16_kernel(const uint2 *x, int64_t n8, float scale, bf16_t *out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const uint2 v = x[i];
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[k] = (bf16_t)(from_e4m3((unsigned char)(v.x >> (8 * k))) * scale);
            o[4 + k] = (bf16_t)(from_e4m3((unsigned char)(v.y >> (8 * k))) * scale);
        }
        *reinterpret_cast<bf16x8 *>(out + 8 * i) = o;
    }
}

__global__ void fp8_epilogue_params_kernel(const float *b, const float *sw, int Cout, float sa_in, float sa_out, float *escale,
                                           float *ebias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cout) return;
    escale[c] = sa_in * sw[c] / sa_out;
    ebias[c] = b[c] / sa_out;
}

// parity-probe layouts (lrcn_conv3x3_fp8): reference (W,H,C,N) f32 <-> NHWC e4m3
__global__ void ref_to_nhwc_fp8_kernel(const float *x, int W, int H, int C, int N, float inv_scale, unsigned char *out) {
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int xx = (int)((i / C) % W), yy = (int)((i / ((int64_t)C * W)) % H);
        const int n = (int)(i / ((int64_t)C * W * H));
        out[i] = to_e4m3(x[(int64_t)xx + (int64_t)W * (yy + (int64_t)H * (c + (int64_t)C * n))] * inv_scale);
    }
}
__global__ void nhwc_fp8_to_ref_kernel(const unsigned char *in, int W, int H, int C, int N, float scale, float *out) {
    const int64_t total = (int64_t)N * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W), yy = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C);
        const int n = (int)(i / ((int64_t)W * H * C));
        out[i] = from_e4m3(in[(((int64_t)n * H + yy) * W + xx) * C + c]) * scale;
    }
}

inline unsigned grid_for(int64_t n) {
    const int64_t b = (n + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

void k_quant_conv_w_fp8(hipStream_t st, const float *w, int Cin, int Cout, void *out, float *sw) {
    hipLaunchKernelGGL(quant_conv_w_fp8_kernel, dim3(Cout), dim3(256), 0, st, w, Cin, Cout, (unsigned char *)out, sw);
}
void k_amax(hipStream_t st, int in_f32, const void *x, int64_t n, float *out) {
    if (in_f32)
        hipLaunchKernelGGL(amax_kernel<float>, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(256), 0, st, (const float *)x, n, out);
    else
        hipLaunchKernelGGL(amax_kernel<bf16_t>, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(256), 0, st, (const bf16_t *)x, n, out);
}
void k_cast_bf16_fp8(hipStream_t st, const void *x, int64_t n, float inv_scale, void *out) {
    hipLaunchKernelGGL(cast_bf16_fp8_kernel, dim3(grid_for(n / 8)), dim3(256), 0, st, (const bf16_t *)x, n / 8, inv_scale, (uint2 *)out);
}
void k_cast_fp8_bf16(hipStream_t st, const void *x, int64_t n, float scale, void *out) {
    hipLaunchKernelGGL(cast_fp8_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, st, (const uint2 *)x, n / 8, scale, (bf16_t *)out);
}
void k_fp8_epilogue_params(hipStream_t st, const float *b, const float *sw, int Cout, float sa_in, float sa_out, float *escale, float *ebias) {
    hipLaunchKernelGGL(fp8_epilogue_params_kernel, dim3((Cout + 255) / 256), dim3(256), 0, st, b, sw, Cout, sa_in, sa_out, escale, ebias);
}
void k_ref_to_nhwc_fp8(hipStream_t st, const float *x, int W, int H, int C, int N, float inv_scale, void *out) {
    hipLaunchKernelGGL(ref_to_nhwc_fp8_kernel, dim3(grid_for((int64_t)N * H * W * C)), dim3(256), 0, st, x, W, H, C, N, inv_scale,
                       (unsigned char *)out);
}
void k_nhwc_fp8_to_ref(hipStream_t st, const void *in, int W, int H, int C, int N, float scale, float *out) {
    hipLaunchKernelGGL(nhwc_fp8_to_ref_kernel, dim3(grid_for((int64_t)N * H * W * C)), dim3(256), 0, st, (const unsigned char *)in, W, H, C, N,
                       scale, out);
}
