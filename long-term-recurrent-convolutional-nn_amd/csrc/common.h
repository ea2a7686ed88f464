// common.h -- shared device/host helpers for liblrcn_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LRCN_WAVE 64

template <typename T> struct TypeInfo;
template <> struct TypeInfo<float> { static constexpr int chunk = 4; static constexpr int bk = 32; };
template <> struct TypeInfo<bf16_t> { static constexpr int chunk = 8; static constexpr int bk = 64; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: remember it per (kernel instantiation, device).
// `done` is one function-local static per kernel instantiation; bit d = set on device ordinal d (ordinals >= 64 just re-set it).
#include <atomic>
typedef std::atomic<uint64_t> LdsAttrMask;
static inline hipError_t set_max_lds(const void *kern, int lds, LdsAttrMask &done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = dev < 64 ? (1ull << dev) : 0ull;
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Window-major pixel order used by every convolution's GEMM row index m (see DESIGN.md "conv rows"):
//   m = ((n*(H/2) + wy)*(W/2) + wx)*4 + dy*2 + dx   <->  pixel (n, y = 2wy+dy, x = 2wx+dx)
// so the four pixels of one 2x2 max-pool window are four consecutive rows (= four consecutive accumulator
// registers of one lane in the 32x32 MFMA C layout) and the pooled output is simply row m/4.
struct PixDecode {
    int n, y, x;
};
// The same with the two divisions as multiply-high by host-made reciprocals (fastdiv_inv below): exact while (m >> 2) * d < 2^32 for
// both divisors d = W / 2 and H / 2; iw == 0 selects the dividing form.  (gemm_8p.hip decodes 4..8 rows per lane in a tile's prologue
// and 16 in its store loop: with 32-bit divisions those were 2.6k + 8k of the tile's 18.5k cycles outside the K loop.)
__device__ __forceinline__ PixDecode decode_pixel(int m, int H, int W);
__device__ __forceinline__ PixDecode decode_pixel_fast(int m, int H, int W, unsigned iw, unsigned ih) {
    if (iw == 0) return decode_pixel(m, H, W);
    const unsigned sub = (unsigned)m & 3u, w = (unsigned)m >> 2, W2 = (unsigned)W >> 1, H2 = (unsigned)H >> 1;
    const unsigned q1 = __umulhi(w, iw), wx = w - q1 * W2;
    const unsigned q2 = __umulhi(q1, ih), wy = q1 - q2 * H2;
    PixDecode p;
    p.n = (int)q2;
    p.y = (int)(2 * wy + (sub >> 1));
    p.x = (int)(2 * wx + (sub & 1));
    return p;
}
__device__ __forceinline__ PixDecode decode_pixel(int m, int H, int W) {
    const int sub = m & 3;
    int w = m >> 2;
    const int W2 = W >> 1, H2 = H >> 1;
    const int wx = w % W2;
    w /= W2;
    const int wy = w % H2;
    PixDecode p;
    p.n = w / H2;
    p.y = 2 * wy + (sub >> 1);
    p.x = 2 * wx + (sub & 1);
    return p;
}

// floor(2^32 / d) + 1: umulhi(n, inv) == n / d for every n with n * d < 2^32 (0 when d < 2: no reciprocal, divide)
inline unsigned fastdiv_inv(unsigned d) { return d < 2 ? 0u : (unsigned)((1ull << 32) / d + 1); }
