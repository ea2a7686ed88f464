// kernels.hip -- the HBM-bound kernels around the MFMA contractions: embedding gather/scatter, fused LSTM cell
// forward/backward, fused log-softmax + NLL + dlogits, dropout (counter hash), layout transposes, multi-tensor Adam,
// VGG weight repacks / im2col for conv1_1 / preprocessing, top-K for beam search.  One wave = 64 lanes everywhere.
#include "kernels.h"

#include "common.h"
#include "gemm.h"

#define DISPATCH_T(dtype, ...)                    \
    do {                                          \
        if ((dtype) == GEMM_T_BF16) {             \
            using T = bf16_t;                     \
            __VA_ARGS__;                          \
        } else {                                  \
            using T = float;                      \
            __VA_ARGS__;                          \
        }                                         \
    } while (0)

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float hash_uniform(uint64_t seed, uint64_t stream, uint64_t idx) {
    const uint64_t h = mix64(mix64(seed ^ (stream * 0xD1342543DE82EF95ull)) ^ idx);
    return (float)(h >> 40) * (1.0f / 16777216.0f);
}
// Dropout multiplier of element (s, b, j) of a (T+1) x [B x ncols] tensor  (Knet dropout: x .* (rand .> p) ./ (1-p)).
__device__ __forceinline__ float drop_mult(const DropSpec &d, int s, int b, int j, int B, int ncols) {
    if (d.mask) return d.mask[((int64_t)s * ncols + j) * B + b];
    if (d.p <= 0.0f) return 1.0f;
    const uint64_t idx = ((uint64_t)s * B + b) * (uint64_t)ncols + j;
    return hash_uniform(d.seed, (uint64_t)d.which, idx) > d.p ? 1.0f / (1.0f - d.p) : 0.0f;
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void build_tokens_kernel(const int32_t *tokens, int T, int B, int V, int32_t *tok_in, int32_t *tok_tgt, double *zero_acc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && zero_acc) *zero_acc = 0.0;  // the log-likelihood accumulator of softmax_xent (saves a memset launch)
    const int S = T + 1;
    if (i >= S * B) return;
    const int s = i / B, b = i - s * B;
    int in = (s == 0) ? 1 : tokens[(s - 1) * B + b];
    int tg = (s < T) ? tokens[s * B + b] : 0;
    // out-of-range ids would fault the gather (the reference raises BoundsError, lrcn.jl:556/569): clamp to unk so that nothing
    // faults, and raise the sticky flag zero_acc[1] -- the next synchronising call (lrcn_last_loss / loss_host / lrcn_sync)
    // reports LRCN_EINVAL
    if ((unsigned)in >= (unsigned)V || (unsigned)tg >= (unsigned)V) {
        if (zero_acc) zero_acc[1] = 1.0;
        if ((unsigned)in >= (unsigned)V) in = 2;
        if ((unsigned)tg >= (unsigned)V) tg = 2;
    }
    tok_in[i] = in;
    tok_tgt[i] = tg;
}

template <typename T>
__global__ void embed_gather_kernel(const T *wembT, int64_t ld_w, const int32_t *tok_in, int S, int B, int E, DropSpec d,
                                    T *xemb, int64_t ld_x) {
    const int m = blockIdx.x;
    const int s = m / B, b = m - s * B;
    const T *src = wembT + (int64_t)tok_in[m] * ld_w;
    T *dst = xemb + (int64_t)m * ld_x;
    for (int e = threadIdx.x; e < E; e += blockDim.x) dst[e] = from_f32<T>(to_f32(src[e]) * drop_mult(d, s, b, e, B, E));
}

__global__ void embed_scatter_kernel(const float *dxemb, int64_t ld_dx, const int32_t *tok_in, int S, int B, int E, int V,
                                     DropSpec d, float *dwembed) {
    const int m = blockIdx.x;
    const int s = m / B, b = m - s * B;
    const int tok = tok_in[m];
    const float *src = dxemb + (int64_t)m * ld_dx;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const float v = src[e] * drop_mult(d, s, b, e, B, E);
        if (v != 0.0f) atomicAdd(dwembed + (int64_t)e * V + tok, v);
    }
}

// ---- embedding gradient, E-contiguous form (dual of the gather, lrcn.jl:556/569 under AutoGrad) ----
// The Wembed gradient of the ABI is V x E column-major (memory [E][V]): a row of dXemb scattered straight into it touches E different
// cache lines per token (64 lanes -> 64 lines per wave instruction).  Instead: (1) rows are summed per token into a ROW-MAJOR f32
// staging array stage[V][ld] -- lanes run along e, 256-byte coalesced atomics (or ordered sums, below) -- and (2) one dense transpose
// writes every element of the column-major gradient (no memset) and puts the zeros back into the staging rows it found non-zero.
__global__ __launch_bounds__(256) void embed_scatter_rm_kernel(const float *dxemb, int64_t ld_dx, const int32_t *tok_in, int S, int B, int E,
                                                               DropSpec d, float *stage, int64_t ld_s) {
    const int m = blockIdx.x;
    const int s = m / B, b = m - s * B;
    float *dst = stage + (int64_t)tok_in[m] * ld_s;
    const float *src = dxemb + (int64_t)m * ld_dx;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const float v = src[e] * drop_mult(d, s, b, e, B, E);
        if (v != 0.0f) atomicAdd(dst + e, v);
    }
}
// Sparse exchange of the embedding gradient (data parallelism): a rank's contribution to d Wembed is its (T+1) B rows of d(x_lstm) (dropout
// multiplier applied) with their token ids -- 1.5 MB at 32 rows against the 42.6 MB dense V x E gradient.  This kernel writes those rows
// E-contiguous into the caller's buffer; the ranks all-gather rows + ids and every rank sums ALL of them in one fixed order
// (sort_token_rows_kernel + embed_segsum_kernel below: bit-identical results on every rank, as an all-reduce would give).
__global__ __launch_bounds__(256) void embed_rows_export_kernel(const float *dxemb, int64_t ld_dx, int S, int B, int E, DropSpec d, float *out) {
    const int m = blockIdx.x;
    const int s = m / B, b = m - s * B;
    const float *src = dxemb + (int64_t)m * ld_dx;
    float *dst = out + (int64_t)m * E;
    for (int e = threadIdx.x; e < E; e += blockDim.x) dst[e] = src[e] * drop_mult(d, s, b, e, B, E);
}
// LRCN_OPT_DETERMINISTIC: keys (token, row) sorted by ONE workgroup (bitonic network in LDS, n <= 8192 padded to a power of two), so
// that the rows of a token are consecutive and in row order ...
__global__ __launch_bounds__(1024) void sort_token_rows_kernel(const int32_t *tok_in, int M, int P2, unsigned long long *keys_out) {
    extern __shared__ unsigned long long sk[];
    for (int i = threadIdx.x; i < P2; i += blockDim.x)
        sk[i] = i < M ? (((unsigned long long)(unsigned)tok_in[i] << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P2; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = sk[i], b = sk[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { sk[i] = b; sk[l] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < M; i += blockDim.x) keys_out[i] = sk[i];
}
// ... and one wave per (segment head, 256-column slice) adds the segment's rows in that order: plain stores, a fixed summation order.
__global__ __launch_bounds__(64) void embed_segsum_kernel(const float *dxemb, int64_t ld_dx, const unsigned long long *keys, int M, int B, int E,
                                                          DropSpec d, float *stage, int64_t ld_s) {
    const int i = blockIdx.x;
    const unsigned tok = (unsigned)(keys[i] >> 32);
    if (i > 0 && (unsigned)(keys[i - 1] >> 32) == tok) return;  // not the first row of its token
    const int e0 = blockIdx.y * 256 + threadIdx.x * 4;
    if (e0 >= E) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = i; j < M && (unsigned)(keys[j] >> 32) == tok; ++j) {
        const int m = (int)(unsigned)keys[j];
        const int s = m / B, b = m - s * B;
        const float *src = dxemb + (int64_t)m * ld_dx;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (e0 + k < E) acc[k] += src[e0 + k] * drop_mult(d, s, b, e0 + k, B, E);
    }
    float *dst = stage + (int64_t)tok * ld_s;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (e0 + k < E) dst[e0 + k] = acc[k];
}
// stage[V][ld_s] (row-major, f32) -> dwembed (V x E column-major: [E][V]); every element of dwembed is written; non-zero staging
// values are replaced by zeros, so the staging array is all-zero again when the kernel ends.  64 x 64 tiles through LDS.
__global__ __launch_bounds__(256) void embed_stage_to_grad_kernel(float *stage, int64_t ld_s, int V, int E, float *dwembed) {
    __shared__ float tile[64][65];
    const int tv = (V + 63) / 64;
    const int v0 = (blockIdx.x % tv) * 64, e0 = (blockIdx.x / tv) * 64;
    const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = v0 + rr + 16 * i, e = e0 + 4 * q;
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (v < V) {
            float *src = stage + (int64_t)v * ld_s + e;
            if (e + 3 < E && (ld_s % 4) == 0) {
                const float4 f = *reinterpret_cast<const float4 *>(src);
                x[0] = f.x; x[1] = f.y; x[2] = f.z; x[3] = f.w;
                if (f.x != 0.f || f.y != 0.f || f.z != 0.f || f.w != 0.f) *reinterpret_cast<float4 *>(src) = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (e + k < E) {
                        x[k] = src[k];
                        if (x[k] != 0.f) src[k] = 0.f;
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[rr + 16 * i][4 * q + k] = x[k];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = e0 + rr + 16 * i, v = v0 + 4 * q;
        if (e >= E) continue;
        float *dst = dwembed + (int64_t)e * V + v;
        if (v + 3 < V && (V % 4) == 0) {
            *reinterpret_cast<float4 *>(dst) = make_float4(tile[4 * q][rr + 16 * i], tile[4 * q + 1][rr + 16 * i], tile[4 * q + 2][rr + 16 * i],
                                                           tile[4 * q + 3][rr + 16 * i]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (v + k < V) dst[k] = tile[4 * q + k][rr + 16 * i];
        }
    }
}
// LRCN_OPT_DETERMINISTIC: the loss is the sum of the per-row log p(target) taken by one workgroup in a fixed order
__global__ __launch_bounds__(256) void sum_rows_f64_kernel(const double *rows, int M, double *out) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < M; i += 256) a += rows[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out += sh[0];
}

template <typename T>
__global__ void lstm_fwd_kernel(const float *G, int64_t ld_g, const float *c_prev, int B, int H, T *acts, int64_t ld_a,
                                float *c_new, T *h_new, int64_t ld_h, float *h_new_f32) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (j >= H) return;
    const float *g = G + (int64_t)b * ld_g;
    const float f = sigm(g[j]), i = sigm(g[H + j]), o = sigm(g[2 * H + j]), ch = tanhf(g[3 * H + j]);
    const float cp = c_prev ? c_prev[(int64_t)b * H + j] : 0.0f;
    const float c = cp * f + i * ch;
    const float h = o * tanhf(c);
    T *a = acts + (int64_t)b * ld_a;
    a[j] = from_f32<T>(f);
    a[H + j] = from_f32<T>(i);
    a[2 * H + j] = from_f32<T>(o);
    a[3 * H + j] = from_f32<T>(ch);
    c_new[(int64_t)b * H + j] = c;
    h_new[(int64_t)b * ld_h + j] = from_f32<T>(h);
    if (h_new_f32) h_new_f32[(int64_t)b * H + j] = h;
}

template <typename T>
__global__ void lstm_bwd_kernel(const T *acts, int64_t ld_a, const float *c_prev, const float *c_new, const float *dh_a,
                                int64_t ld_dha, float *dh_b, int dh_b_read, float *dc, int dc_zero, int B, int H, T *dz, int64_t ld_dz, int nslab) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (j >= H) return;
    const T *a = acts + (int64_t)b * ld_a;
    const float f = to_f32(a[j]), i = to_f32(a[H + j]), o = to_f32(a[2 * H + j]), g = to_f32(a[3 * H + j]);
    const float tc = tanhf(c_new[(int64_t)b * H + j]);
    float dh = dh_a[(int64_t)b * ld_dha + j];
    if (nslab > 0) {  // recurrent dh from step s+1 as the K slices' partial sums [nslab][B][H] of a split-K GEMM without a reduce launch, summed
        if (dh_b_read)  // here in slice order (fixed: deterministic); the slabs are overwritten whole by step s-1's GEMM, nothing to zero
            for (int q = 0; q < nslab; ++q) dh += dh_b[((int64_t)q * B + b) * H + j];
    } else if (dh_b) {  // recurrent dh from step s+1; left zeroed for the split-K GEMM that accumulates step s-1's into it
        if (dh_b_read) dh += dh_b[(int64_t)b * H + j];
        dh_b[(int64_t)b * H + j] = 0.0f;
    }
    const float dov = dh * tc;
    const float dcv = (dc_zero ? 0.0f : dc[(int64_t)b * H + j]) + dh * o * (1.0f - tc * tc);
    const float cp = c_prev ? c_prev[(int64_t)b * H + j] : 0.0f;
    T *z = dz + (int64_t)b * ld_dz;
    z[j] = from_f32<T>(dcv * cp * f * (1.0f - f));
    z[H + j] = from_f32<T>(dcv * g * i * (1.0f - i));
    z[2 * H + j] = from_f32<T>(dov * o * (1.0f - o));
    z[3 * H + j] = from_f32<T>(dcv * i * (1.0f - g * g));
    dc[(int64_t)b * H + j] = dcv * f;
}

template <typename T>
__global__ void concat_x2_kernel(T *x2, int64_t ld_x2, const float *xcnn, int64_t ld_xc, int S, int B, int nl, int nr, DropSpec d) {
    const int m = blockIdx.x;
    const int s = m / B, b = m - s * B;
    T *row = x2 + (int64_t)m * ld_x2;
    for (int j = threadIdx.x; j < nl + nr; j += blockDim.x) {
        const float v = (j < nl) ? to_f32(row[j]) : xcnn[(int64_t)b * ld_xc + (j - nl)];
        row[j] = from_f32<T>(v * drop_mult(d, s, b, j, B, nl + nr));
    }
}

template <typename T>
__global__ void dx2_mask_reduce_kernel(T *dx2, int64_t ld, int S, int B, int nl, int nr, DropSpec d, float *dxcnn, int64_t ld_dxc) {
    const int b = blockIdx.x, j = blockIdx.y * blockDim.x + threadIdx.x;
    if (j >= nl + nr) return;
    float acc = 0.0f;
    for (int s = 0; s < S; ++s) {
        T *p = dx2 + (int64_t)(s * B + b) * ld + j;
        const float v = to_f32(*p) * drop_mult(d, s, b, j, B, nl + nr);
        *p = from_f32<T>(v);
        acc += v;
    }
    if (j >= nl) dxcnn[(int64_t)b * ld_dxc + (j - nl)] = acc;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float block_max(float v, float *sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, sh[i]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, float *sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = 0.0f;
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float *logits, int64_t ld_l, const int32_t *tgt, int M,
                                                           int V, float scale, double *logp_sum, T *dlog, int64_t ld_d, double *logp_rows) {
    __shared__ float sh[8];
    const int m = blockIdx.x;
    const float *row = logits + (int64_t)m * ld_l;
    float mx = -INFINITY;
    for (int v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, row[v]);
    mx = block_max(mx, sh);
    float se = 0.0f;
    for (int v = threadIdx.x; v < V; v += blockDim.x) se += expf(row[v] - mx);
    se = block_sum(se, sh);
    const float lse = mx + logf(se);
    const int t = tgt[m];
    if (threadIdx.x == 0) {
        if (logp_rows) logp_rows[m] = (double)(row[t] - lse);
        else atomicAdd(logp_sum, (double)(row[t] - lse));
    }
    if (dlog) {
        T *drow = dlog + (int64_t)m * ld_d;
        for (int v = threadIdx.x; v < V; v += blockDim.x) {
            const float p = expf(row[v] - lse);
            drow[v] = from_f32<T>((p - (v == t ? 1.0f : 0.0f)) * scale);
        }
    }
}

template <typename T> __device__ __forceinline__ void store4(T *p, const float *v) {
    struct alignas(4 * sizeof(T)) V4 { T e[4]; };
    V4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o.e[k] = from_f32<T>(v[k]);
    *reinterpret_cast<V4 *>(p) = o;
}

// The same for V <= 1024 Q with the row held in registers: one expf per element (e = expf(x - max), p = e / sum) instead of
// two, 16-byte loads; the log-likelihood term is unchanged (x[t] - (max + logf(sum))).
template <typename T, int Q>
__global__ __launch_bounds__(256) void softmax_xent_reg_kernel(const float *logits, int64_t ld_l, const int32_t *tgt, int M, int V,
                                                               float scale, double *logp_sum, T *dlog, int64_t ld_d, double *logp_rows) {
    __shared__ float sh[8];
    const int m = blockIdx.x;
    const float *row = logits + (int64_t)m * ld_l;  // ld_l % 4 == 0, 16-byte aligned rows
    float x[Q][4];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int v0 = 4 * (threadIdx.x + 256 * q);
        float4 f = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (v0 < V) f = *reinterpret_cast<const float4 *>(row + v0);
        x[q][0] = f.x;
        x[q][1] = v0 + 1 < V ? f.y : -INFINITY;
        x[q][2] = v0 + 2 < V ? f.z : -INFINITY;
        x[q][3] = v0 + 3 < V ? f.w : -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = fmaxf(mx, x[q][j]);
    mx = block_max(mx, sh);
    float se = 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[q][j] = expf(x[q][j] - mx);  // 0 for the padding
            se += x[q][j];
        }
    se = block_sum(se, sh);
    const int t = tgt[m];
    if (threadIdx.x == 0) {
        if (logp_rows) logp_rows[m] = (double)(row[t] - (mx + logf(se)));
        else atomicAdd(logp_sum, (double)(row[t] - (mx + logf(se))));
    }
    if (dlog) {
        T *drow = dlog + (int64_t)m * ld_d;
        const float inv = 1.0f / se;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int v0 = 4 * (threadIdx.x + 256 * q);
            if (v0 >= V) continue;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (x[q][j] * inv - (v0 + j == t ? 1.0f : 0.0f)) * scale;
            if (v0 + 3 < V && (ld_d % 4) == 0) {
                store4(drow + v0, o);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (v0 + j < V) drow[v0 + j] = from_f32<T>(o[j]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(const float *logits, int64_t ld_l, int M, int V, float *prob,
                                                           int64_t ld_p) {
    __shared__ float sh[8];
    const int m = blockIdx.x;
    const float *row = logits + (int64_t)m * ld_l;
    float mx = -INFINITY;
    for (int v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, row[v]);
    mx = block_max(mx, sh);
    float se = 0.0f;
    for (int v = threadIdx.x; v < V; v += blockDim.x) se += expf(row[v] - mx);
    se = block_sum(se, sh);
    const float lse = mx + logf(se);
    for (int v = threadIdx.x; v < V; v += blockDim.x) prob[(int64_t)m * ld_p + v] = expf(row[v] - lse);
}

template <typename Tin, typename Tout>
__global__ void transpose_kernel(const Tin *in, int64_t ld_in, int R, int C, Tout *out, int64_t ld_out, int shift) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? to_f32(in[(int64_t)r * ld_in + c]) : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[(int64_t)c * ld_out + r + shift] = from_f32<Tout>(tile[tx][i]);
    }
    if (shift > 0 && blockIdx.y == 0) {
        for (int i = ty; i < 32; i += 8) {
            const int c = c0 + i;
            if (c < C)
                for (int r = tx; r < shift; r += 32) out[(int64_t)c * ld_out + r] = from_f32<Tout>(0.0f);
        }
    }
    if (blockIdx.y == gridDim.y - 1) {  // zero the K-padding [R + shift, ld_out) of every output row
        for (int i = ty; i < 32; i += 8) {
            const int c = c0 + i;
            if (c < C)
                for (int r = R + shift + tx; r < ld_out; r += 32) out[(int64_t)c * ld_out + r] = from_f32<Tout>(0.0f);
        }
    }
}

// All shadow weights of one model in ONE launch (was 8 cast_rows + 9 transpose launches per step): for every 32 x 32 tile of
// a parameter's memory image [R][C] (f32) write the direct copy split at column `cs` (dA[r][c], dB[r][c - cs]) and / or the
// transposed copy (tA[c][r], tB[c - cs][r]) in T.  Padding columns of the destinations are never touched (zero since allocation).
template <typename T, bool ADAM = false> __global__ __launch_bounds__(256) void prepare_weights_kernel(const PrepPlan plan) {
    // 64 x 64 tiles, 16 bytes in / 8 bytes out per thread access (bf16); generic element-wise path for f32 shadows and edges
    __shared__ float tile[64][65];
    // plan.total tiles walked by gridDim.x workgroups: one tile per workgroup, or -- a capped grid (k_adam_shadows' grid_cap) -- a few
    // workgroups that walk them all, so that an update running BESIDE the backward pass streams at a fraction of the chip's bandwidth
    // instead of taking every CU from the latency-bound kernels of the recurrence
    for (int bid = blockIdx.x; bid < plan.total; bid += gridDim.x) {
    if (bid != (int)blockIdx.x) __syncthreads();  // the previous tile's transposed stores have read `tile`
    int d = 0;
#pragma unroll
    for (int k = 1; k < PREP_MAX; ++k)
        if (k < plan.n && bid >= plan.d[k].tile0) d = k;
    const PrepDesc &P = plan.d[d];
    const int t = bid - P.tile0;
    const int tc = (P.C + 63) / 64;
    const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
    const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;  // 16 column quads x 16 row groups
    const bool vec = (P.C % 4) == 0 && (P.cs % 4) == 0;
    auto al = [](const void *p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) % (4 * sizeof(T))) == 0 && (ld % 4) == 0; };
    const bool alA = al(P.dA, P.ldA), alB = al(P.dB, P.ldB), altA = al(P.tA, P.ldtA), altB = al(P.tB, P.ldtB);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + rr + 16 * i, c = c0 + 4 * q;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < P.R) {
            if (vec && c + 3 < P.C) {
                const float4 x = *reinterpret_cast<const float4 *>(P.src + (int64_t)r * P.C + c);
                v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c + k < P.C) v[k] = P.src[(int64_t)r * P.C + c + k];
            }
            if constexpr (ADAM) {  // update! on the loaded values (the arithmetic of adam_kernel), written back before the shadows are made
                const int64_t o = (int64_t)r * P.C + c;
                float gg[4] = {0.f, 0.f, 0.f, 0.f}, mm[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f};
                const bool v4 = vec && c + 3 < P.C;
                if (v4) {
                    const float4 a = *reinterpret_cast<const float4 *>(P.g + o), b = *reinterpret_cast<const float4 *>(P.m + o),
                                 d4 = *reinterpret_cast<const float4 *>(P.v + o);
                    gg[0] = a.x; gg[1] = a.y; gg[2] = a.z; gg[3] = a.w;
                    mm[0] = b.x; mm[1] = b.y; mm[2] = b.z; mm[3] = b.w;
                    vv[0] = d4.x; vv[1] = d4.y; vv[2] = d4.z; vv[3] = d4.w;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c + k < P.C) { gg[k] = P.g[o + k]; mm[k] = P.m[o + k]; vv[k] = P.v[o + k]; }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    mm[k] = plan.b1 * mm[k] + (1.0f - plan.b1) * gg[k];
                    vv[k] = plan.b2 * vv[k] + (1.0f - plan.b2) * gg[k] * gg[k];
                    v[k] -= plan.lr * (mm[k] / plan.c1) / (sqrtf(vv[k] / plan.c2) + plan.eps);
                }
                float *w = const_cast<float *>(P.src);
                if (v4) {
                    *reinterpret_cast<float4 *>(P.m + o) = make_float4(mm[0], mm[1], mm[2], mm[3]);
                    *reinterpret_cast<float4 *>(P.v + o) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                    *reinterpret_cast<float4 *>(w + o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c + k < P.C) { P.m[o + k] = mm[k]; P.v[o + k] = vv[k]; w[o + k] = v[k]; }
                }
            }
            if (P.dG) {  // gate-interleaved rows of the B side (element-wise: the destination is written once per step, 8 MB)
                const int rg = (r % P.giH) * 4 + r / P.giH;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c + k >= P.cs && c + k < P.C) reinterpret_cast<T *>(P.dG)[(int64_t)rg * P.ldG + (c + k - P.cs)] = from_f32<T>(v[k]);
            }
            const bool sideA = c + 3 < P.cs, sideB = c >= P.cs;
            const int rd = P.permH > 0 ? (r % P.permH) * 4 + r / P.permH : r;  // destination row of the direct copies
            if (vec && c + 3 < P.C && sideA && alA) {
                if (P.dA) store4(reinterpret_cast<T *>(P.dA) + (int64_t)rd * P.ldA + c, v);
            } else if (vec && c + 3 < P.C && sideB && alB) {
                if (P.dB) store4(reinterpret_cast<T *>(P.dB) + (int64_t)rd * P.ldB + (c - P.cs), v);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cc = c + k;
                    if (cc >= P.C) continue;
                    if (cc < P.cs) {
                        if (P.dA) reinterpret_cast<T *>(P.dA)[(int64_t)rd * P.ldA + cc] = from_f32<T>(v[k]);
                    } else if (P.dB) {
                        reinterpret_cast<T *>(P.dB)[(int64_t)rd * P.ldB + (cc - P.cs)] = from_f32<T>(v[k]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[rr + 16 * i][4 * q + k] = v[k];
    }
    if (!P.tA && !P.tB) continue;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + rr + 16 * i;  // source column = destination row
        if (c >= P.C) continue;
        T *dst = nullptr;
        if (c < P.cs) {
            if (P.tA) dst = reinterpret_cast<T *>(P.tA) + (int64_t)c * P.ldtA;
        } else if (P.tB) {
            dst = reinterpret_cast<T *>(P.tB) + (int64_t)(c - P.cs) * P.ldtB;
        }
        if (!dst) continue;
        const int r = r0 + 4 * q;
        if (r + 3 < P.R && (c < P.cs ? altA : altB)) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = tile[4 * q + k][rr + 16 * i];
            store4(dst + r, v);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (r + k < P.R) dst[r + k] = from_f32<T>(tile[4 * q + k][rr + 16 * i]);
        }
    }
    }
}

template <typename T>
__global__ void cast_rows_kernel(const float *in, int64_t ld_in, int R, int C, T *out, int64_t ld_out) {
    const int r = blockIdx.y;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ld_out; c += gridDim.x * blockDim.x)
        out[(int64_t)r * ld_out + c] = from_f32<T>(c < C ? in[(int64_t)r * ld_in + c] : 0.0f);
}
template <typename T>
__global__ void bias_act_cast_kernel(const float *in, int64_t ld_in, const float *bias, int relu, int R, int C, T *out, int64_t ld_out) {
    const int r = blockIdx.y;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ld_out; c += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (c < C) {
            v = in[(int64_t)r * ld_in + c] + (bias ? bias[c] : 0.0f);
            if (relu) v = fmaxf(v, 0.0f);
        }
        out[(int64_t)r * ld_out + c] = from_f32<T>(v);
    }
}
template <typename T>
__global__ void uncast_rows_kernel(const T *in, int64_t ld_in, int R, int C, float *out, int64_t ld_out) {
    const int r = blockIdx.y;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x)
        out[(int64_t)r * ld_out + c] = to_f32(in[(int64_t)r * ld_in + c]);
}

template <typename T> __global__ __launch_bounds__(256) void colsum_kernel(const T *z, int64_t ld, int M, int N, int rows_per_slab, float *out) {
    // block = 64 columns (16 lanes x 4 elements, one 8/16-byte load each) x 16 row groups over one slab of rows (blockIdx.y).
    // One slab: plain store (no zeroing, deterministic).  Several slabs: f32 atomics into an output the host zeroed.
    __shared__ float sh[16][65];
    const int q = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + 4 * q;
    const int m0 = blockIdx.y * rows_per_slab, m1 = min(M, m0 + rows_per_slab);
    struct alignas(4 * sizeof(T)) V4 { T e[4]; };
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {  // ld % 4 == 0 and N <= ld: the whole quad is inside the row
#pragma unroll 4
        for (int m = m0 + rg; m < m1; m += 16) {
            const V4 v = *reinterpret_cast<const V4 *>(z + (int64_t)m * ld + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += to_f32(v.e[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[rg][4 * q + k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cc = blockIdx.x * 64 + threadIdx.x;
        float t = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += sh[r][threadIdx.x];
        if (cc < N) {
            if (gridDim.y == 1) out[cc] = t;
            else atomicAdd(out + cc, t);
        }
    }
}

template <typename T> __global__ __launch_bounds__(256) void transpose_multi_kernel(const TrPlan plan) {
    __shared__ float tile[64][65];
    int d = 0;
#pragma unroll
    for (int k = 1; k < TR_MAX; ++k)
        if (k < plan.n && (int)blockIdx.x >= plan.d[k].tile0) d = k;
    const TrDesc &P = plan.d[d];
    const T *src = reinterpret_cast<const T *>(P.src);
    T *dst = reinterpret_cast<T *>(P.dst);
    const int t = blockIdx.x - P.tile0;
    const int tc = (P.C + 63) / 64, tr = P.R > 0 ? (P.R + 63) / 64 : 1;
    const int ty = t / tc;
    const int r0 = ty * 64, c0 = (t % tc) * 64;
    const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;
    struct alignas(4 * sizeof(T)) V4 { T e[4]; };
    const bool vin = (P.ld_src % 4) == 0 && (reinterpret_cast<uintptr_t>(src) % sizeof(V4)) == 0;
    const bool vout = (P.ld_dst % 4) == 0 && (P.shift % 4) == 0 && (reinterpret_cast<uintptr_t>(dst) % sizeof(V4)) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + rr + 16 * i, c = c0 + 4 * q;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < P.R && c < P.C) {
            if (vin && c + 3 < P.ld_src) {
                const V4 x = *reinterpret_cast<const V4 *>(src + (int64_t)r * P.ld_src + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = to_f32(x.e[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c + k < P.C) v[k] = to_f32(src[(int64_t)r * P.ld_src + c + k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[rr + 16 * i][4 * q + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + rr + 16 * i;  // source column = destination row
        if (c >= P.C) continue;
        T *row = dst + (int64_t)c * P.ld_dst;
        const int r = r0 + 4 * q;
        if (vout && r + 3 < P.R) {
            V4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o.e[k] = from_f32<T>(tile[4 * q + k][rr + 16 * i]);
            *reinterpret_cast<V4 *>(row + P.shift + r) = o;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (r + k < P.R) row[P.shift + r + k] = from_f32<T>(tile[4 * q + k][rr + 16 * i]);
        }
        if (ty == 0)
            for (int x = q; x < P.shift; x += 16) row[x] = from_f32<T>(0.0f);
        if (ty == tr - 1)
            for (int64_t x = P.R + P.shift + q; x < P.ld_dst; x += 16) row[x] = from_f32<T>(0.0f);
    }
}

__global__ void adam_kernel(AdamTensors t, float lr, float b1, float b2, float eps, float c1, float c2) {
    const int k = blockIdx.y;
    const int64_t n = t.n[k];
    float *w = t.w[k], *m = t.m[k], *v = t.v[k];
    const float *g = t.g[k];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        w[i] -= lr * (mi / c1) / (sqrtf(vi / c2) + eps);
    }
}

__global__ void init_uniform_kernel(float *w, int64_t n, float scale, uint64_t seed, int tensor) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = mix64(mix64(seed ^ ((uint64_t)(tensor + 1) * 0xD1342543DE82EF95ull)) ^ (uint64_t)i);
        const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
        w[i] = (float)(2.0 * (double)scale * u - (double)scale);
    }
}
__global__ void fill_kernel(float *w, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) w[i] = v;
}

// ---- VGG ----
template <typename T> __global__ void repack_conv_w_kernel(const float *w, int Cin, int Cout, int Cin_pad, T *out) {
    // w(a,b,ci,co) at a + 3*(b + 3*(ci + Cin*co));  out[co][tap=b*3+a][ci]
    const int64_t total = (int64_t)Cout * 9 * Cin_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin_pad);
        const int tap = (int)((i / Cin_pad) % 9);
        const int co = (int)(i / ((int64_t)Cin_pad * 9));
        const int b = tap / 3, a = tap - 3 * b;
        out[i] = from_f32<T>(ci < Cin ? w[a + 3 * (b + 3 * ((int64_t)ci + (int64_t)Cin * co))] : 0.0f);
    }
}
template <typename T> __global__ void repack_conv11_w_kernel(const float *w, int Cout, T *out, int64_t ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout * ld) return;
    const int k = i % ld, co = i / ld;
    float v = 0.0f;
    if (k < 27) {
        const int tap = k / 3, c = k - 3 * tap;
        const int b = tap / 3, a = tap - 3 * b;
        v = w[a + 3 * (b + 3 * (c + 3 * co))];
    }
    out[i] = from_f32<T>(v);
}
// out[i] = (bf16)(img[i] - mean[i % 3])  : read_image_data's arithmetic (lrcn.jl:770) in the crop's own layout [n][row][col][3]
// avg != NULL: the full averageImage (S,S,3) column-major instead of the three channel means; the reference subtracts it BEFORE
// its last H <-> W permutedims (lrcn.jl:770-771), so pixel (row r, col q, c) meets avg(q, r, c) = avg[q + S r + S^2 c]
__device__ __forceinline__ float avg_at(const float *avg, int64_t i, int S) {
    const int c = (int)(i % 3);
    const int64_t px = i / 3;
    const int q = (int)(px % S), r = (int)((px / S) % S);
    return avg[q + (int64_t)S * r + (int64_t)S * S * c];
}
// PAD = 2: the output is the crop inside a frame of PAD zero pixels on every side, out[n][S + 2 PAD][S + 2 PAD][3] (the frame is zero
// since allocation and never written): conv64.hip's raw-window DMA then reads conv1_1's zero padding as DATA -- no per-lane in-image tests,
// and a dword of the window never straddles the image edge (its element-shifted second copy needs that).
template <int PAD>
__global__ void img_u8_to_bf16_kernel(const uint8_t *img, int64_t n, float m0, float m1, float m2, const float *avg, int S, bf16_t *out) {
    // 12 bytes (4 pixels) per thread: channel phase is the same for every thread; S % 4 == 0, so the four pixels share an image row
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i0 = t * 12;
    if (i0 >= n) return;
    const int SP = S + 2 * PAD;
    auto opix = [&](int64_t pix) {  // flat pixel index (n, x, y) -> pixel index in the framed output
        const int64_t row = pix / S;           // n * S + x
        const int y = (int)(pix - row * S);
        const int64_t nn = row / S;
        const int x = (int)(row - nn * S);
        return ((nn * SP + x + PAD) * SP + y + PAD);
    };
    if (avg) {
        for (int64_t i = i0; i < n && i < i0 + 12; ++i) out[opix(i / 3) * 3 + i % 3] = (bf16_t)((float)img[i] - avg_at(avg, i, S));
        return;
    }
    if (i0 + 12 <= n && (S & 3) == 0) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(img + i0);
        const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
        const float mean[3] = {m0, m1, m2};
        bf16_t o[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const uint32_t w = k < 4 ? w0 : (k < 8 ? w1 : w2);
            o[k] = (bf16_t)((float)((w >> (8 * (k & 3))) & 0xFFu) - mean[k % 3]);
        }
        // 24 bytes at a 4-byte-aligned address (the framed pixel index of a thread's first pixel is even)
        uint32_t *q = reinterpret_cast<uint32_t *>(out + opix(i0 / 3) * 3);
        const uint32_t *ov = reinterpret_cast<const uint32_t *>(o);
#pragma unroll
        for (int k = 0; k < 6; ++k) q[k] = ov[k];
    } else {
        for (int64_t i = i0; i < n && i < i0 + 12; ++i)
            out[opix(i / 3) * 3 + i % 3] = (bf16_t)((float)img[i] - (i % 3 == 0 ? m0 : (i % 3 == 1 ? m1 : m2)));
    }
}
// conv1_1 weights for the fused conv1_1+conv1_2 kernels (conv64.hip FUSE, conv64f.hip): out[co][k'], k' = 8 lq + j:
//   lq < 3: kw = lq, kh = j / 3, c = j % 3 (the first 8 bytes of the 9-byte run of image row kw);  lq = 3: j < 3 -> kw = j, kh = 2, c = 2;
//   k' = 27, 28, 29: the f32 bias of channel co cut into three bf16 pieces, hi + mid + lo = b exactly (24 bits of mantissa in 3 x 8) --
//   conv64f.hip's im2col fragment holds 1.0 there, so the bias enters the f32 accumulation as data; conv64.hip's fragment holds 0 there.
__global__ void repack_conv11_w_fused_kernel(const float *w, const float *b, bf16_t *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 32) return;
    const int k = i & 31, co = i >> 5, lq = k >> 3, j = k & 7;
    float v = 0.0f;
    int kw = -1, kh = 0, c = 0;
    if (lq < 3) {
        kw = lq; kh = j / 3; c = j % 3;
    } else if (j < 3) {
        kw = j; kh = 2; c = 2;
    }
    if (kw >= 0) v = w[kw + 3 * (kh + 3 * (c + 3 * co))];  // reference layout (3,3,3,64) column-major: a = kw (dim 1), b = kh (dim 2)
    if (lq == 3 && j >= 3 && j < 6 && b) {
        float rest = b[co];
        for (int piece = 3; piece <= j; ++piece) {
            v = (float)(bf16_t)rest;
            rest -= v;
        }
    }
    out[i] = (bf16_t)v;
}
template <typename T> __global__ void repack_fc6_w_kernel(const float *w, T *out) {
    // out[o][(y*7+x)*512 + c] = w(o, x + 7y + 49c) = w[o + 4096*(x + 7y + 49c)]; tiled through LDS for coalescing both ways
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, o0 = blockIdx.y * 32;  // k = internal index
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i;  // internal k -> ref k
        const int c = k % 512, yx = k / 512, y = yx / 7, x = yx - 7 * y;
        const int kref = x + 7 * y + 49 * c;
        tile[i][tx] = w[(int64_t)(o0 + tx) + 4096ll * kref];
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) out[(int64_t)(o0 + i) * 25088 + k0 + tx] = from_f32<T>(tile[tx][i]);
}

template <typename T, bool U8>
__global__ void im2col11_kernel(const void *src, int N, int S, float m0, float m1, float m2, T *out, int64_t ld) {
    // one thread per (m, tap); writes 3 channels.  internal (y, x) = (dim 2, dim 1) of the reference tensor;
    // for the uint8 path reference dim 1 = image row, dim 2 = image col (lrcn.jl:766-771).
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t M = (int64_t)N * S * S;
    if (idx >= M * 9) return;  // columns [27, ld) are never read: the GEMM loader masks k >= K = 27
    const int tap = (int)(idx % 9);
    const int m = (int)(idx / 9);
    T *row = out + (int64_t)m * ld;
    const PixDecode p = decode_pixel(m, S, S);
    const int kh = tap / 3, kw = tap - 3 * kh;
    const int y = p.y + kh - 1, x = p.x + kw - 1;
    float v[3] = {0.0f, 0.0f, 0.0f};
    if ((unsigned)y < (unsigned)S && (unsigned)x < (unsigned)S) {
        if (U8) {
            const uint8_t *px = reinterpret_cast<const uint8_t *>(src) + (((int64_t)p.n * S + x) * S + y) * 3;  // row=x, col=y
            v[0] = (float)px[0] - m0;
            v[1] = (float)px[1] - m1;
            v[2] = (float)px[2] - m2;
        } else {
            const float *f = reinterpret_cast<const float *>(src);
            for (int c = 0; c < 3; ++c) v[c] = f[(int64_t)x + (int64_t)S * (y + (int64_t)S * (c + 3ll * p.n))];
        }
    }
    for (int c = 0; c < 3; ++c) row[tap * 3 + c] = from_f32<T>(v[c]);
}

__global__ void preprocess_u8_kernel(const uint8_t *img, int N, int S, float m0, float m1, float m2, const float *avg, float *out) {
    const int64_t total = (int64_t)N * 3 * S * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ii = (int)(i % S), j = (int)((i / S) % S), c = (int)((i / ((int64_t)S * S)) % 3);
        const int n = (int)(i / (3ll * S * S));
        const float mean = avg ? avg[j + (int64_t)S * ii + (int64_t)S * S * c] : (c == 0 ? m0 : (c == 1 ? m1 : m2));
        out[i] = (float)img[(((int64_t)n * S + ii) * S + j) * 3 + c] - mean;
    }
}

// read_image_data's geometry (lrcn.jl:755-765) for a batch of decoded images of different sizes: resize so that the shorter side is
// 224 and the other div(side * 224, shorter) (:756), centre crop at div offsets (:758-760), grey -> three channels (:762-764).
// Resampling: bilinear between pixel CENTRES (output (R, Q) of the nh x nw resized image samples the source at
// ((2R+1) h / (2 nh) - 1/2, (2Q+1) w / (2 nw) - 1/2), clamped to the image), computed in exact integer arithmetic with
// round-half-up, so that a host restatement reproduces every byte (Images.imresize's own kernel is unpinned).
struct ImgMeta {
    int64_t off;
    int h, w, ch, pad;
};
__global__ void resize_crop_u8_kernel(const uint8_t *src, const ImgMeta *meta, int N, int S, uint8_t *out) {
    const int64_t total = (int64_t)N * S * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % S), r = (int)((i / S) % S), n = (int)(i / ((int64_t)S * S));
        const ImgMeta m = meta[n];
        const int64_t h = m.h, w = m.w, sm = h < w ? h : w;
        const int64_t nh = h * S / sm, nw = w * S / sm;          // :756  div(size * 224, minimum(size))
        const int64_t R = r + (nh - S) / 2, Q = q + (nw - S) / 2;  // :758-760
        int64_t ny = (2 * R + 1) * h - nh, nx = (2 * Q + 1) * w - nw;  // 2 nh * sy, 2 nw * sx
        if (ny < 0) ny = 0;
        if (nx < 0) nx = 0;
        const int64_t y0 = ny / (2 * nh), fy = ny - y0 * 2 * nh, x0 = nx / (2 * nw), fx = nx - x0 * 2 * nw;
        const int64_t y1 = y0 + 1 < h ? y0 + 1 : h - 1, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
        const uint8_t *im = src + m.off;
        const int ch = m.ch;
        uint8_t v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int cs = ch >= 3 ? c : 0;
            const int64_t p00 = im[(y0 * w + x0) * ch + cs], p01 = im[(y0 * w + x1) * ch + cs], p10 = im[(y1 * w + x0) * ch + cs],
                          p11 = im[(y1 * w + x1) * ch + cs];
            const int64_t top = (2 * nw - fx) * p00 + fx * p01, bot = (2 * nw - fx) * p10 + fx * p11;
            v[c] = (uint8_t)(((2 * nh - fy) * top + fy * bot + 2 * nh * nw) / (4 * nh * nw));
        }
        uint8_t *o = out + i * 3;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    }
}

// feats(n, :) /= sum(feats(n, :))  (generate's input/sum(input), lrcn.jl:595-597; what the reference's `featsn` files hold).
// feats: N x F column-major f32; one workgroup per row.
__global__ void normalize_rows_kernel(float *feats, int N, int F) {
    __shared__ float sh[8];
    const int n = blockIdx.x;
    float s = 0.0f;
    for (int j = threadIdx.x; j < F; j += blockDim.x) s += feats[n + (int64_t)N * j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    float tot = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += sh[w];
    for (int j = threadIdx.x; j < F; j += blockDim.x) feats[n + (int64_t)N * j] /= tot;
}

template <typename T> __global__ void ref_to_nhwc_kernel(const float *x, int W, int H, int C, int N, T *out, int C_ld) {
    const int64_t total = (int64_t)N * H * W * C_ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C_ld);
        const int xx = (int)((i / C_ld) % W), yy = (int)((i / ((int64_t)C_ld * W)) % H);
        const int n = (int)(i / ((int64_t)C_ld * W * H));
        out[i] = from_f32<T>(c < C ? x[(int64_t)xx + (int64_t)W * (yy + (int64_t)H * (c + (int64_t)C * n))] : 0.0f);
    }
}
template <typename T> __global__ void nhwc_to_ref_kernel(const T *in, int W, int H, int C, int N, int C_ld, float *out) {
    const int64_t total = (int64_t)N * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W), yy = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C);
        const int n = (int)(i / ((int64_t)W * H * C));
        out[i] = to_f32(in[(((int64_t)n * H + yy) * W + xx) * C_ld + c]);
    }
}

// Beam bookkeeping of ONE decode step for N images at once (lrcn.jl:657-677 per image), one workgroup per image:
//   candidates (i, j) = hypothesis i of the image x its j-th best next word, probability topv * p[i] (linear float32 space);
//   step 1 expands hypothesis 1 only (:662-664); stable descending order (ties: lower candidate index); keep K; stop when
//   the best ends in eos or current > nword (:670) -> the image is frozen and its result recorded.
// seq_in/seq_out: [N*K][L] token histories (ping-pong), p: [N*K] in/out, parent[N*K]: state row to copy, last[N*K]: token fed next.
__global__ __launch_bounds__(256) void beam_update_kernel(const int32_t *topi, const float *topv, const int32_t *seq_in, int32_t *seq_out,
                                                          float *p, int32_t *parent, int32_t *last, int32_t *done, int32_t *ndone,
                                                          int32_t *res_tok, int32_t *res_len, float *res_p, int K, int L, int current,
                                                          int nword, int eos) {
    __shared__ float cp[1024];
    __shared__ float newp[32];
    __shared__ int sel[32];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int r0 = n * K;
    if (done[n]) {  // frozen: identity parent, histories carried over
        for (int k = tid; k < K; k += blockDim.x) {
            parent[r0 + k] = r0 + k;
            last[r0 + k] = eos;
        }
        for (int e = tid; e < K * L; e += blockDim.x) seq_out[(int64_t)r0 * L + e] = seq_in[(int64_t)r0 * L + e];
        return;
    }
    const int nexp = current == 1 ? 1 : K, C = nexp * K;
    for (int c = tid; c < C; c += blockDim.x) cp[c] = topv[(int64_t)(r0 + c / K) * K + c % K] * p[r0 + c / K];
    __syncthreads();
    for (int c = tid; c < C; c += blockDim.x) {
        const float v = cp[c];
        int rank = 0;
        for (int q = 0; q < C; ++q) rank += (cp[q] > v) || (cp[q] == v && q < c);
        if (rank < K) {
            sel[rank] = c;
            newp[rank] = v;
        }
    }
    __syncthreads();
    // K <= C always (C >= K): every rank 0..K-1 is filled.  New histories = parent's history + the chosen word.
    for (int e = tid; e < K * L; e += blockDim.x) {
        const int k = e / L, pos = e - k * L;
        const int c = sel[k], i = c / K;
        int32_t t = seq_in[(int64_t)(r0 + i) * L + pos];
        if (pos == current) t = topi[(int64_t)(r0 + i) * K + c % K];
        seq_out[(int64_t)(r0 + k) * L + pos] = t;
    }
    __syncthreads();
    for (int k = tid; k < K; k += blockDim.x) {
        const int c = sel[k];
        p[r0 + k] = newp[k];
        parent[r0 + k] = r0 + c / K;
        last[r0 + k] = topi[(int64_t)(r0 + c / K) * K + c % K];
    }
    if (tid == 0) {
        const int c = sel[0];
        const int best_tok = topi[(int64_t)(r0 + c / K) * K + c % K];
        if (best_tok == eos || current > nword) {
            done[n] = 1;
            atomicAdd(ndone, 1);
            res_len[n] = current + 1;
            res_p[n] = newp[0];
        }
    }
    __syncthreads();
    if (done[n])
        for (int pos = tid; pos <= current; pos += blockDim.x) res_tok[(int64_t)n * L + pos] = seq_out[(int64_t)r0 * L + pos];
}
// out[r][0..C) = in[r / K][0..C)  (replicate each image row K times)
template <typename T> __global__ void repeat_rows_kernel(const T *in, int64_t ld, int R, int K, int C, T *out) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[(int64_t)r * ld + c] = in[(int64_t)(r / K) * ld + c];
}

// Top-K of each row, descending, ties to the lower index (Julia's stable sortperm(rev=true), lrcn.jl:655).
// One 256-thread block per row; K rounds of block-wide argmax with the winners masked out. K <= 32.
__global__ __launch_bounds__(256) void topk_rows_kernel(const float *prob, int64_t ld, int R, int V, int K, int32_t *idx,
                                                        float *val) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int chosen[32];
    const int r = blockIdx.x;
    const float *row = prob + (int64_t)r * ld;
    for (int k = 0; k < K; ++k) {
        float bv = -INFINITY;
        int bi = 0x7FFFFFFF;
        for (int v = threadIdx.x; v < V; v += blockDim.x) {
            bool taken = false;
            for (int q = 0; q < k; ++q) taken |= (chosen[q] == v);
            const float x = row[v];
            if (!taken && (x > bv || (x == bv && v < bi))) {
                bv = x;
                bi = v;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if ((threadIdx.x & 63) == 0) {
            sv[threadIdx.x >> 6] = bv;
            si[threadIdx.x >> 6] = bi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w)
                if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) {
                    bv = sv[w];
                    bi = si[w];
                }
            chosen[k] = bi;
            idx[r * K + k] = bi;
            val[r * K + k] = bv;
        }
        __syncthreads();
    }
}

// softmax_rows + topk_rows in one pass for V <= 16384: the row of logits stays in registers; the K winners are the K largest
// LOGITS (softmax is monotone; ties to the lower index), their probabilities expf(x - (max + logf(sum))) are computed for
// those K only, and winners whose float probabilities coincide are put in ascending index order, which is the order
// topk_rows_kernel (Julia's stable sortperm(rev=true), lrcn.jl:655) gives on the probabilities.  The normaliser uses the
// hardware exponential (one instruction per element instead of ~20): 1e-6 relative on the reported probabilities.
// Q = float4 loads per thread: V <= 1024 Q.  Every thread keeps the best of its own not-yet-retired elements; a round is
// one block-wide argmax over those 256 candidates, after which only the winner's owner rescans its registers.
template <int Q>
__global__ __launch_bounds__(256) void softmax_topk_rows_kernel(const float *logits, int64_t ld, int R, int V, int K, int32_t *idx,
                                                                float *val) {
    __shared__ float sh[8];
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float wv[64];
    __shared__ int wi[64];
    const int r = blockIdx.x;
    const float *row = logits + (int64_t)r * ld;  // ld % 4 == 0, 16-byte aligned rows: columns [V, ld) may be read, never used
    float x[Q][4];
    // thread t owns columns 4 (t + 256 q) .. + 3
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int v0 = 4 * (threadIdx.x + 256 * q);
        float4 f = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (v0 < V) f = *reinterpret_cast<const float4 *>(row + v0);
        x[q][0] = f.x;
        x[q][1] = v0 + 1 < V ? f.y : -INFINITY;
        x[q][2] = v0 + 2 < V ? f.z : -INFINITY;
        x[q][3] = v0 + 3 < V ? f.w : -INFINITY;
    }
    auto local_best = [&](float &lv, int &li) {
        lv = -INFINITY;
        li = 0x7FFFFFFF;
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x[q][j] > lv) {  // increasing index: strict > keeps the lowest index among equals
                    lv = x[q][j];
                    li = 4 * (threadIdx.x + 256 * q) + j;
                }
    };
    float lv;
    int li;
    local_best(lv, li);
    const float mx = block_max(lv, sh);
    float se = 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) se += __expf(x[q][j] - mx);  // exp(-inf) = 0 for the padding
    se = block_sum(se, sh);
    const float lse = mx + logf(se);
    // Rounds 0 .. K-1 take the K largest logits.  The reference ranks the float32 PROBABILITIES with a stable sort (lrcn.jl:652-656),
    // and distinct logits can round to one float probability: if candidates beyond the K-th still share the K-th winner's
    // probability they belong to the same tie group, whose lowest INDICES win.  So the rounds go on (at most to 64 entries)
    // until the next candidate's probability differs; all of this is block-uniform.
    float pK = -1.0f;
    int n = 0;
    for (int k = 0; k < 64; ++k) {
        float bv = lv;
        int bi = li;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            sv[threadIdx.x >> 6] = bv;
            si[threadIdx.x >> 6] = bi;
        }
        __syncthreads();
        bv = sv[0];
        bi = si[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) {
                bv = sv[w];
                bi = si[w];
            }
        const float pv = expf(bv - lse);
        if (k >= K && (pv != pK || bi == 0x7FFFFFFF)) break;  // uniform: every thread holds the same (bv, bi)
        if (threadIdx.x == 0) {
            wi[k] = bi;
            wv[k] = pv;
        }
        n = k + 1;
        if (k == K - 1) pK = pv;
        if (((bi >> 2) & 255) == (int)threadIdx.x) {  // the owner retires the winner and finds its next candidate
            const int slot = bi >> 10, j0 = bi & 3;
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (q == slot && j == j0) x[q][j] = -INFINITY;
            local_best(lv, li);
        }
    }
    if (threadIdx.x == 0) {
        for (int k = 1; k < n; ++k) {  // equal probabilities (distinct logits, same float): ascending index
            const float v = wv[k];
            const int ix = wi[k];
            int q = k;
            while (q > 0 && wv[q - 1] == v && wi[q - 1] > ix) {
                wv[q] = wv[q - 1];
                wi[q] = wi[q - 1];
                --q;
            }
            wv[q] = v;
            wi[q] = ix;
        }
        for (int k = 0; k < K; ++k) {
            idx[r * K + k] = wi[k];
            val[r * K + k] = wv[k];
        }
    }
}

// The second half of GEMM_OUT_SMAX_TOPK (gemm.h SmaxEpi; round 6): one wave per row combines the row's nrec = V / 128 records {max, sum exp,
// SMAX_KC best logits + columns} into what softmax_topk_rows_kernel returns for the full row of logits: the K largest float32
// PROBABILITIES p = exp(x - lse), lse = max + log(sum), in descending order, equal probabilities by ascending column (lrcn.jl:652-656 --
// a stable descending sort of p).  As there, the rounds go on past K while the next candidate still shares the K-th probability
// (distinct logits that round to one float), then the tie group is put in index order.  A record keeps SMAX_KC = K + 1 candidates
// of its 128 columns, so a tie group that crosses the K boundary is exact as long as no more than SMAX_KC of it fall into one record.
// Lane l owns records l, l + 64, ...; every record's list is sorted, so a lane's best candidate is the best list HEAD, and retiring a
// candidate shifts that list (static indices only).
template <int NR>
__global__ __launch_bounds__(256) void softmax_topk_merge_kernel(const float *part, int nrec, int R, int K, int32_t *idx, float *val) {
    __shared__ float wv[4][64];
    __shared__ int wi[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + w;
    if (row >= R) return;   // wave-uniform; no block-wide barrier below
    const float *rp = part + (int64_t)row * nrec * SMAX_REC;
    float m[NR], sx[NR], v[NR][SMAX_KC];
    int ix[NR][SMAX_KC];
    float gm = -INFINITY;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int rec = lane + 64 * q;
        m[q] = -INFINITY;
        sx[q] = 0.0f;
#pragma unroll
        for (int j = 0; j < SMAX_KC; ++j) {
            v[q][j] = -INFINITY;
            ix[q][j] = 0x7FFFFFFF;
        }
        if (rec < nrec) {
            const float4 a = *reinterpret_cast<const float4 *>(rp + (int64_t)rec * SMAX_REC), b = *reinterpret_cast<const float4 *>(rp + (int64_t)rec * SMAX_REC + 4),
                         c = *reinterpret_cast<const float4 *>(rp + (int64_t)rec * SMAX_REC + 8), d = *reinterpret_cast<const float4 *>(rp + (int64_t)rec * SMAX_REC + 12);
            m[q] = a.x; sx[q] = a.y;
            v[q][0] = a.z; v[q][1] = a.w; v[q][2] = b.x; v[q][3] = b.y; v[q][4] = b.z; v[q][5] = b.w;
            ix[q][0] = __float_as_int(c.x); ix[q][1] = __float_as_int(c.y); ix[q][2] = __float_as_int(c.z); ix[q][3] = __float_as_int(c.w);
            ix[q][4] = __float_as_int(d.x); ix[q][5] = __float_as_int(d.y);
        }
        gm = fmaxf(gm, m[q]);
    }
    gm = wave_max(gm);
    float se = 0.0f;
#pragma unroll
    for (int q = 0; q < NR; ++q)
        if (m[q] != -INFINITY) se += sx[q] * __expf(m[q] - gm);
    se = wave_sum(se);
    const float lse = gm + logf(se);
    auto local_best = [&](float &lv, int &li) {
        lv = -INFINITY;
        li = 0x7FFFFFFF;
#pragma unroll
        for (int q = 0; q < NR; ++q)
            if (v[q][0] > lv || (v[q][0] == lv && ix[q][0] < li)) {
                lv = v[q][0];
                li = ix[q][0];
            }
    };
    float lv;
    int li;
    local_best(lv, li);
    float pK = -1.0f;
    int n = 0;
    for (int k = 0; k < 64; ++k) {
        float bv = lv;
        int bi = li;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        const float pv = expf(bv - lse);
        if (k >= K && (pv != pK || bi == 0x7FFFFFFF)) break;  // wave-uniform
        if (lane == 0) {
            wi[w][k] = bi;
            wv[w][k] = pv;
        }
        n = k + 1;
        if (k == K - 1) pK = pv;
        if (li == bi && bi != 0x7FFFFFFF) {  // the owner retires the winner: its list moves up by one
#pragma unroll
            for (int q = 0; q < NR; ++q)
                if (ix[q][0] == bi) {
#pragma unroll
                    for (int j = 0; j + 1 < SMAX_KC; ++j) {
                        v[q][j] = v[q][j + 1];
                        ix[q][j] = ix[q][j + 1];
                    }
                    v[q][SMAX_KC - 1] = -INFINITY;
                    ix[q][SMAX_KC - 1] = 0x7FFFFFFF;
                }
            local_best(lv, li);
        }
    }
    if (lane == 0) {
        for (int k = 1; k < n; ++k) {  // equal probabilities (distinct logits, same float): ascending index
            const float vv = wv[w][k];
            const int ii = wi[w][k];
            int q = k;
            while (q > 0 && wv[w][q - 1] == vv && wi[w][q - 1] > ii) {
                wv[w][q] = wv[w][q - 1];
                wi[w][q] = wi[w][q - 1];
                --q;
            }
            wv[w][q] = vv;
            wi[w][q] = ii;
        }
        for (int k = 0; k < K; ++k) {
            idx[row * K + k] = wi[w][k];
            val[row * K + k] = wv[w][k];
        }
    }
}

// Beam reordering of the four recurrent state tensors in one launch (lrcn.jl:673-676): out[i][r] = in[i][parent[r]], plus
// the K-contiguous T copies of h1 / h2 that the next step's recurrent GEMMs read.
struct GatherState {
    const float *in[4];
    float *out[4];
    void *hT[4];     // T copy of state i (or NULL)
    int64_t ldT[4];
    int C[4];
};
template <typename T> __global__ void gather_state_kernel(const GatherState g, const int32_t *parent) {
    const int r = blockIdx.x, i = blockIdx.y;
    const int C = g.C[i];
    const float *s = g.in[i] + (int64_t)parent[r] * C;
    float *o = g.out[i] + (int64_t)r * C;
    T *t = g.hT[i] ? reinterpret_cast<T *>(g.hT[i]) + (int64_t)r * g.ldT[i] : nullptr;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float v = s[c];
        o[c] = v;
        if (t) t[c] = from_f32<T>(v);
    }
}

// The batched beam decode's per-step gather (round 6), bf16 only: row r of the next step's [x | h1] operand = the embedding of hypothesis
// r's last token (lrcn.jl:650) next to h1 of its PARENT hypothesis (lrcn.jl:673-676), and the h2 block of [x2 | h2] likewise -- what
// embed_gather + gather_state did in two launches, without the four f32 state tensors' round trip (82 MB in, 82 MB out per step at 5120
// hypotheses: the cell state now stays where the epilogue wrote it and is READ through `parent`, LstmEpi::c_prev_idx).  16-byte vectors:
// every row starts 128-byte aligned (leading dimensions are multiples of 64 elements).  parent == NULL: the first step (h blocks zero).
__global__ __launch_bounds__(256) void decode_prep_kernel(const bf16_t *wembT, int64_t ld_w, const int32_t *last, const int32_t *parent, int E,
                                                          const bf16_t *h1, int64_t ld_h1, int H1, const bf16_t *h2, int64_t ld_h2, int H2,
                                                          bf16_t *xh1, int64_t ld_xh1, int64_t off_h1, bf16_t *xh2, int64_t ld_xh2, int64_t off_h2) {
    const int r = blockIdx.x;
    auto copy = [&](const bf16_t *src, bf16_t *dst, int n) {   // exactly n elements: whole 16-byte vectors, then a scalar tail (LRCN-1f keeps
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);  // x_cnn right behind the embedding columns)
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        for (int i = threadIdx.x; i < n / 8; i += 256) d4[i] = s4[i];
        for (int i = (n & ~7) + threadIdx.x; i < n; i += 256) dst[i] = src[i];
    };
    copy(wembT + (int64_t)last[r] * ld_w, xh1 + (int64_t)r * ld_xh1, E);
    if (parent) {
        const int pr = parent[r];
        copy(h1 + (int64_t)pr * ld_h1, xh1 + (int64_t)r * ld_xh1 + off_h1, H1);
        if (h2) copy(h2 + (int64_t)pr * ld_h2, xh2 + (int64_t)r * ld_xh2 + off_h2, H2);
    }
}

// The prep launch of the decode step with input-projection TABLES (lrcn_api.hip decode_tables; round 6): the gate GEMMs contract the hidden
// state alone, so only the parents' h move -- h1[parent] into the rows of A1, h2[parent] into the h block of A2 = [h1 Wproj | h2].
__global__ __launch_bounds__(256) void decode_prep_h_kernel(const int32_t *parent, const bf16_t *h1, int64_t ld_h1, int H1, const bf16_t *h2,
                                                            int64_t ld_h2, int H2, bf16_t *a1, int64_t ld_a1, bf16_t *a2, int64_t ld_a2, int64_t off_h2) {
    const int r = blockIdx.x, pr = parent[r];
    auto copy = [&](const bf16_t *src, bf16_t *dst, int n) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        for (int i = threadIdx.x; i < n / 8; i += 256) d4[i] = s4[i];
        for (int i = (n & ~7) + threadIdx.x; i < n; i += 256) dst[i] = src[i];
    };
    copy(h1 + (int64_t)pr * ld_h1, a1 + (int64_t)r * ld_a1, H1);
    copy(h2 + (int64_t)pr * ld_h2, a2 + (int64_t)r * ld_a2 + off_h2, H2);
}
__global__ void row_div_kernel(int32_t *out, int R, int K) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) out[r] = r / K;
}

__global__ void gather_rows_f32_kernel(const float *in, int64_t ld, const int32_t *src_row, int R, int C, float *out) {
    const int r = blockIdx.x;
    const float *s = in + (int64_t)src_row[r] * ld;
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[(int64_t)r * ld + c] = s[c];
}

__global__ void mul_f32_kernel(const float *a, const float *b, int64_t n, float *out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}

inline unsigned grid1d(int64_t n, int block = 256, int64_t cap = 8192) {
    int64_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

// ---------------------------------------------------------------- launchers
void k_build_tokens(hipStream_t st, const int32_t *tokens, int T, int B, int V, int32_t *tok_in, int32_t *tok_tgt, double *zero_acc) {
    const int n = (T + 1) * B;
    hipLaunchKernelGGL(build_tokens_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, tokens, T, B, V, tok_in, tok_tgt, zero_acc);
}
void k_embed_gather(hipStream_t st, int dtype, const void *wembT, int64_t ld_w, const int32_t *tok_in, int S, int B, int E,
                    DropSpec d, void *xemb, int64_t ld_x) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(embed_gather_kernel<T>, dim3(S * B), dim3(256), 0, st, (const T *)wembT, ld_w,
                                         tok_in, S, B, E, d, (T *)xemb, ld_x));
}
void k_embed_rows_export(hipStream_t st, const float *dxemb, int64_t ld_dx, int S, int B, int E, DropSpec d, float *out) {
    hipLaunchKernelGGL(embed_rows_export_kernel, dim3(S * B), dim3(256), 0, st, dxemb, ld_dx, S, B, E, d, out);
}
void k_embed_scatter(hipStream_t st, const float *dxemb, int64_t ld_dx, const int32_t *tok_in, int S, int B, int E, int V,
                     DropSpec d, float *dwembed) {
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(S * B), dim3(256), 0, st, dxemb, ld_dx, tok_in, S, B, E, V, d, dwembed);
}
bool k_embed_scatter_rm(hipStream_t st, const float *dxemb, int64_t ld_dx, const int32_t *tok_in, int S, int B, int E, int V, DropSpec d,
                        float *stage, int64_t ld_s, float *dwembed, unsigned long long *sort_keys) {
    const int M = S * B;
    if (sort_keys) {  // ordered sums
        int P2 = 1;
        while (P2 < M) P2 <<= 1;
        if (P2 > 8192) return false;
        hipLaunchKernelGGL(sort_token_rows_kernel, dim3(1), dim3(1024), sizeof(unsigned long long) * (size_t)P2, st, tok_in, M, P2, sort_keys);
        hipLaunchKernelGGL(embed_segsum_kernel, dim3(M, cdiv(E, 256)), dim3(64), 0, st, dxemb, ld_dx, sort_keys, M, B, E, d, stage, ld_s);
    } else {
        hipLaunchKernelGGL(embed_scatter_rm_kernel, dim3(M), dim3(256), 0, st, dxemb, ld_dx, tok_in, S, B, E, d, stage, ld_s);
    }
    hipLaunchKernelGGL(embed_stage_to_grad_kernel, dim3(cdiv(V, 64) * cdiv(E, 64)), dim3(256), 0, st, stage, ld_s, V, E, dwembed);
    return true;
}
void k_lstm_fwd(hipStream_t st, int dtype, const float *G, int64_t ld_g, const float *c_prev, int B, int H, void *acts,
                int64_t ld_a, float *c_new, void *h_new, int64_t ld_h, float *h_new_f32) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(lstm_fwd_kernel<T>, dim3(cdiv(H, 256), B), dim3(256), 0, st, G, ld_g, c_prev, B, H,
                                         (T *)acts, ld_a, c_new, (T *)h_new, ld_h, h_new_f32));
}
void k_lstm_bwd(hipStream_t st, int dtype, const void *acts, int64_t ld_a, const float *c_prev, const float *c_new,
                const float *dh_a, int64_t ld_dha, float *dh_b, int dh_b_read, float *dc, int dc_zero, int B, int H, void *dz, int64_t ld_dz, int nslab) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(lstm_bwd_kernel<T>, dim3(cdiv(H, 256), B), dim3(256), 0, st, (const T *)acts, ld_a,
                                         c_prev, c_new, dh_a, ld_dha, dh_b, dh_b_read, dc, dc_zero, B, H, (T *)dz, ld_dz, nslab));
}
void k_concat_x2(hipStream_t st, int dtype, void *x2, int64_t ld_x2, const float *xcnn, int64_t ld_xc, int S, int B, int nl, int nr,
                 DropSpec d) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(concat_x2_kernel<T>, dim3(S * B), dim3(256), 0, st, (T *)x2, ld_x2, xcnn, ld_xc, S,
                                         B, nl, nr, d));
}
void k_dx2_mask_reduce(hipStream_t st, int dtype, void *dx2, int64_t ld, int S, int B, int nl, int nr, DropSpec d, float *dxcnn,
                       int64_t ld_dxc) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(dx2_mask_reduce_kernel<T>, dim3(B, cdiv(nl + nr, 256)), dim3(256), 0, st, (T *)dx2, ld, S, B, nl,
                                         nr, d, dxcnn, ld_dxc));
}
void k_softmax_xent(hipStream_t st, int dtype, const float *logits, int64_t ld_l, const int32_t *tgt, int M, int V,
                    float scale, double *logp_sum, void *dlog, int64_t ld_d, double *logp_rows) {
    const bool reg = V <= 16384 && (ld_l % 4) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(dlog) & 15) == 0;
    if (reg) {
        const int q = (V + 1023) / 1024;
#define SX_LAUNCH(QQ)                                                                                                              \
    DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_xent_reg_kernel<T, QQ>), dim3(M), dim3(256), 0, st, logits, ld_l, tgt, M, V, scale, \
                                         logp_sum, (T *)dlog, ld_d, logp_rows))
        if (q <= 2) SX_LAUNCH(2);
        else if (q <= 4) SX_LAUNCH(4);
        else if (q <= 8) SX_LAUNCH(8);
        else if (q <= 12) SX_LAUNCH(12);
        else SX_LAUNCH(16);
#undef SX_LAUNCH
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL(softmax_xent_kernel<T>, dim3(M), dim3(256), 0, st, logits, ld_l, tgt, M, V, scale,
                                             logp_sum, (T *)dlog, ld_d, logp_rows));
    }
    if (logp_rows) hipLaunchKernelGGL(sum_rows_f64_kernel, dim3(1), dim3(256), 0, st, logp_rows, M, logp_sum);
}
void k_softmax_rows(hipStream_t st, const float *logits, int64_t ld_l, int M, int V, float *prob, int64_t ld_p) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(M), dim3(256), 0, st, logits, ld_l, M, V, prob, ld_p);
}
void k_transpose(hipStream_t st, int dtype, int in_f32, const void *in, int64_t ld_in, int R, int C, void *out,
                 int64_t ld_out, int shift) {
    const dim3 grid(cdiv(C, 32), cdiv(R, 32));
    DISPATCH_T(dtype, {
        if (in_f32)
            hipLaunchKernelGGL((transpose_kernel<float, T>), grid, dim3(256), 0, st, (const float *)in, ld_in, R, C, (T *)out,
                               ld_out, shift);
        else
            hipLaunchKernelGGL((transpose_kernel<T, T>), grid, dim3(256), 0, st, (const T *)in, ld_in, R, C, (T *)out, ld_out,
                               shift);
    });
}
void k_transpose_f32(hipStream_t st, const float *in, int64_t ld_in, int R, int C, float *out, int64_t ld_out) {
    hipLaunchKernelGGL((transpose_kernel<float, float>), dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, st, in, ld_in, R, C,
                       out, ld_out, 0);
}
void k_prepare_weights(hipStream_t st, int dtype, PrepPlan &plan) {
    int tiles = 0;
    for (int k = 0; k < plan.n; ++k) {
        plan.d[k].tile0 = tiles;
        tiles += cdiv(plan.d[k].R, 64) * cdiv(plan.d[k].C, 64);
    }
    if (tiles == 0) return;
    plan.total = tiles;
    DISPATCH_T(dtype, hipLaunchKernelGGL(prepare_weights_kernel<T>, dim3(tiles), dim3(256), 0, st, plan));
}
void k_adam_shadows(hipStream_t st, int dtype, PrepPlan &plan, int step, float lr, float b1, float b2, float eps) {
    int tiles = 0;
    for (int k = 0; k < plan.n; ++k) {
        plan.d[k].tile0 = tiles;
        tiles += cdiv(plan.d[k].R, 64) * cdiv(plan.d[k].C, 64);
    }
    if (tiles == 0) return;
    plan.lr = lr; plan.b1 = b1; plan.b2 = b2; plan.eps = eps;
    plan.c1 = (float)(1.0 - pow((double)b1, (double)step));
    plan.c2 = (float)(1.0 - pow((double)b2, (double)step));
    plan.total = tiles;
    const int grid = (plan.grid_cap > 0 && plan.grid_cap < tiles) ? plan.grid_cap : tiles;
    DISPATCH_T(dtype, hipLaunchKernelGGL((prepare_weights_kernel<T, true>), dim3(grid), dim3(256), 0, st, plan));
}
void k_cast_rows(hipStream_t st, int dtype, const float *in, int64_t ld_in, int R, int C, void *out, int64_t ld_out) {
    const dim3 grid(cdiv(ld_out, 256) > 64 ? 64 : cdiv(ld_out, 256), R);
    DISPATCH_T(dtype, hipLaunchKernelGGL(cast_rows_kernel<T>, grid, dim3(256), 0, st, in, ld_in, R, C, (T *)out, ld_out));
}
void k_bias_act_cast(hipStream_t st, int dtype, const float *in, int64_t ld_in, const float *bias, int relu, int R, int C, void *out,
                     int64_t ld_out) {
    const dim3 grid(cdiv(ld_out, 256) > 64 ? 64 : cdiv(ld_out, 256), R);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bias_act_cast_kernel<T>, grid, dim3(256), 0, st, in, ld_in, bias, relu, R, C, (T *)out, ld_out));
}
void k_uncast_rows(hipStream_t st, int dtype, const void *in, int64_t ld_in, int R, int C, float *out, int64_t ld_out) {
    const dim3 grid(cdiv(C, 256) > 64 ? 64 : cdiv(C, 256), R);
    DISPATCH_T(dtype, hipLaunchKernelGGL(uncast_rows_kernel<T>, grid, dim3(256), 0, st, (const T *)in, ld_in, R, C, out,
                                         ld_out));
}
void k_colsum(hipStream_t st, int dtype, const void *z, int64_t ld, int M, int N, float *out, bool deterministic) {
    // enough slabs of rows to give the chip ~2 blocks per CU; a single slab needs no zeroing and no atomics
    const int cb = cdiv(N, 64);
    int slabs = (M <= 512 || deterministic) ? 1 : cdiv(512, cb);  // few rows: one pass, no memset launch (the step is launch-bound there)
    if (slabs > cdiv(M, 64)) slabs = cdiv(M, 64);
    if (slabs < 1) slabs = 1;
    const int rows = cdiv(cdiv(M, slabs), 16) * 16;
    slabs = cdiv(M, rows);
    if (slabs > 1) (void)hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, st);
    DISPATCH_T(dtype, hipLaunchKernelGGL(colsum_kernel<T>, dim3(cb, slabs), dim3(256), 0, st, (const T *)z, ld, M, N, rows, out));
}
void k_transpose_multi(hipStream_t st, int dtype, TrPlan &plan) {
    int tiles = 0;
    for (int k = 0; k < plan.n; ++k) {
        plan.d[k].tile0 = tiles;
        tiles += (plan.d[k].R > 0 ? cdiv(plan.d[k].R, 64) : 1) * cdiv(plan.d[k].C, 64);
    }
    if (tiles == 0) return;
    DISPATCH_T(dtype, hipLaunchKernelGGL(transpose_multi_kernel<T>, dim3(tiles), dim3(256), 0, st, plan));
}
void k_adam(hipStream_t st, const AdamTensors &t, int step, float lr, float b1, float b2, float eps) {
    const float c1 = (float)(1.0 - pow((double)b1, (double)step)), c2 = (float)(1.0 - pow((double)b2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(1024, 9), dim3(256), 0, st, t, lr, b1, b2, eps, c1, c2);
}
void k_init_uniform(hipStream_t st, float *w, int64_t n, float scale, uint64_t seed, int tensor) {
    hipLaunchKernelGGL(init_uniform_kernel, dim3(grid1d(n)), dim3(256), 0, st, w, n, scale, seed, tensor);
}
void k_fill(hipStream_t st, float *w, int64_t n, float v) {
    hipLaunchKernelGGL(fill_kernel, dim3(grid1d(n)), dim3(256), 0, st, w, n, v);
}
void k_repack_conv_w(hipStream_t st, int dtype, const float *w, int Cin, int Cout, int Cin_pad, void *out) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(repack_conv_w_kernel<T>, dim3(grid1d((int64_t)Cout * 9 * Cin_pad)), dim3(256), 0, st,
                                         w, Cin, Cout, Cin_pad, (T *)out));
}
void k_repack_conv11_w(hipStream_t st, int dtype, const float *w, int Cout, void *out, int64_t ld) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(repack_conv11_w_kernel<T>, dim3(cdiv(Cout * ld, 256)), dim3(256), 0, st, w, Cout,
                                         (T *)out, ld));
}
void k_img_u8_to_bf16(hipStream_t st, const uint8_t *img, int64_t n, float m0, float m1, float m2, const float *avg, int S, void *out) {
    const int64_t threads = (n + 11) / 12;
    hipLaunchKernelGGL(img_u8_to_bf16_kernel<2>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, img, n, m0, m1, m2, avg, S,
                       (bf16_t *)out);
}
void k_resize_crop_u8(hipStream_t st, const uint8_t *src, const void *meta, int N, int S, uint8_t *out) {
    hipLaunchKernelGGL(resize_crop_u8_kernel, dim3(grid1d((int64_t)N * S * S)), dim3(256), 0, st, src, (const ImgMeta *)meta, N, S, out);
}
namespace {
__global__ void beam_init_kernel(int32_t *seq, int32_t *last, float *p, int R, int Lh, int bos) {
    const int64_t total = (int64_t)R * Lh;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i / Lh), j = (int)(i - (int64_t)q * Lh);
        seq[i] = j == 0 ? bos : 0;
        if (j == 0) {
            last[q] = bos;
            p[q] = 1.0f;
        }
    }
}
}  // namespace
void k_beam_init(hipStream_t st, int32_t *seq, int32_t *last, float *p, int R, int Lh, int bos) {
    hipLaunchKernelGGL(beam_init_kernel, dim3(grid1d((int64_t)R * Lh)), dim3(256), 0, st, seq, last, p, R, Lh, bos);
}
void k_normalize_rows(hipStream_t st, float *feats, int N, int F) {
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(N), dim3(256), 0, st, feats, N, F);
}
void k_repack_conv11_w_fused(hipStream_t st, const float *w, const float *b, void *out) {
    hipLaunchKernelGGL(repack_conv11_w_fused_kernel, dim3(8), dim3(256), 0, st, w, b, (bf16_t *)out);
}
void k_repack_fc6_w(hipStream_t st, int dtype, const float *w, void *out) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(repack_fc6_w_kernel<T>, dim3(25088 / 32, 4096 / 32), dim3(256), 0, st, w, (T *)out));
}
void k_im2col11_u8(hipStream_t st, int dtype, const uint8_t *img, int N, int S, float m0, float m1, float m2, void *out,
                   int64_t ld) {
    const int64_t n = (int64_t)N * S * S * 9;
    DISPATCH_T(dtype, hipLaunchKernelGGL((im2col11_kernel<T, true>), dim3(cdiv(n, 256)), dim3(256), 0, st, (const void *)img,
                                         N, S, m0, m1, m2, (T *)out, ld));
}
void k_im2col11_f32(hipStream_t st, int dtype, const float *x, int N, int S, void *out, int64_t ld) {
    const int64_t n = (int64_t)N * S * S * 9;
    DISPATCH_T(dtype, hipLaunchKernelGGL((im2col11_kernel<T, false>), dim3(cdiv(n, 256)), dim3(256), 0, st, (const void *)x, N,
                                         S, 0.f, 0.f, 0.f, (T *)out, ld));
}
void k_preprocess_u8(hipStream_t st, const uint8_t *img, int N, int S, float m0, float m1, float m2, const float *avg, float *out) {
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid1d((int64_t)N * 3 * S * S)), dim3(256), 0, st, img, N, S, m0, m1, m2, avg,
                       out);
}
void k_ref_to_nhwc(hipStream_t st, int dtype, const float *x, int W, int H, int C, int N, void *out, int C_ld) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(ref_to_nhwc_kernel<T>, dim3(grid1d((int64_t)N * H * W * C_ld)), dim3(256), 0, st, x,
                                         W, H, C, N, (T *)out, C_ld));
}
void k_nhwc_to_ref(hipStream_t st, int dtype, const void *in, int W, int H, int C, int N, int C_ld, float *out) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(nhwc_to_ref_kernel<T>, dim3(grid1d((int64_t)N * C * H * W)), dim3(256), 0, st,
                                         (const T *)in, W, H, C, N, C_ld, out));
}
void k_topk_rows(hipStream_t st, const float *prob, int64_t ld, int R, int V, int K, int32_t *idx, float *val) {
    hipLaunchKernelGGL(topk_rows_kernel, dim3(R), dim3(256), 0, st, prob, ld, R, V, K, idx, val);
}
void k_beam_update(hipStream_t st, const int32_t *topi, const float *topv, const int32_t *seq_in, int32_t *seq_out, float *p,
                   int32_t *parent, int32_t *last, int32_t *done, int32_t *ndone, int32_t *res_tok, int32_t *res_len, float *res_p, int N,
                   int K, int L, int current, int nword, int eos) {
    hipLaunchKernelGGL(beam_update_kernel, dim3(N), dim3(256), 0, st, topi, topv, seq_in, seq_out, p, parent, last, done, ndone, res_tok,
                       res_len, res_p, K, L, current, nword, eos);
}
void k_repeat_rows(hipStream_t st, int dtype, const void *in, int64_t ld, int N, int K, int C, void *out) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(repeat_rows_kernel<T>, dim3(N * K), dim3(256), 0, st, (const T *)in, ld, N * K, K, C, (T *)out));
}
bool k_softmax_topk_rows(hipStream_t st, const float *logits, int64_t ld, int R, int V, int K, int32_t *idx, float *val) {
    if (V > 16384 || K > 32 || (ld % 4) || (reinterpret_cast<uintptr_t>(logits) & 15)) return false;
    const int q = (V + 1023) / 1024;
    if (q <= 4)
        hipLaunchKernelGGL(softmax_topk_rows_kernel<4>, dim3(R), dim3(256), 0, st, logits, ld, R, V, K, idx, val);
    else if (q <= 8)
        hipLaunchKernelGGL(softmax_topk_rows_kernel<8>, dim3(R), dim3(256), 0, st, logits, ld, R, V, K, idx, val);
    else if (q <= 12)
        hipLaunchKernelGGL(softmax_topk_rows_kernel<12>, dim3(R), dim3(256), 0, st, logits, ld, R, V, K, idx, val);
    else
        hipLaunchKernelGGL(softmax_topk_rows_kernel<16>, dim3(R), dim3(256), 0, st, logits, ld, R, V, K, idx, val);
    return true;
}
void k_gather_state(hipStream_t st, int dtype, const float *const in[4], float *const out[4], void *const hT[4], const int64_t ldT[4],
                    const int C[4], const int32_t *parent, int R) {
    GatherState g;
    for (int i = 0; i < 4; ++i) {
        g.in[i] = in[i]; g.out[i] = out[i]; g.hT[i] = hT[i]; g.ldT[i] = ldT[i]; g.C[i] = C[i];
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(gather_state_kernel<T>, dim3(R, 4), dim3(256), 0, st, g, parent));
}
bool k_softmax_topk_merge(hipStream_t st, const float *part, int nrec, int R, int K, int32_t *idx, float *val) {
    if (K < 1 || K >= SMAX_KC || nrec < 1 || nrec > 256 || (reinterpret_cast<uintptr_t>(part) & 15)) return false;
    const dim3 grid((R + 3) / 4);
    if (nrec <= 64) hipLaunchKernelGGL(softmax_topk_merge_kernel<1>, grid, dim3(256), 0, st, part, nrec, R, K, idx, val);
    else if (nrec <= 128) hipLaunchKernelGGL(softmax_topk_merge_kernel<2>, grid, dim3(256), 0, st, part, nrec, R, K, idx, val);
    else hipLaunchKernelGGL(softmax_topk_merge_kernel<4>, grid, dim3(256), 0, st, part, nrec, R, K, idx, val);
    return true;
}
void k_decode_prep(hipStream_t st, const void *wembT, int64_t ld_w, const int32_t *last, const int32_t *parent, int R, int E, const void *h1,
                   int64_t ld_h1, int H1, const void *h2, int64_t ld_h2, int H2, void *xh1, int64_t ld_xh1, int64_t off_h1, void *xh2, int64_t ld_xh2,
                   int64_t off_h2) {
    hipLaunchKernelGGL(decode_prep_kernel, dim3(R), dim3(256), 0, st, (const bf16_t *)wembT, ld_w, last, parent, E, (const bf16_t *)h1, ld_h1, H1,
                       (const bf16_t *)h2, ld_h2, H2, (bf16_t *)xh1, ld_xh1, off_h1, (bf16_t *)xh2, ld_xh2, off_h2);
}
void k_decode_prep_h(hipStream_t st, const int32_t *parent, int R, const void *h1, int64_t ld_h1, int H1, const void *h2, int64_t ld_h2, int H2,
                     void *a1, int64_t ld_a1, void *a2, int64_t ld_a2, int64_t off_h2) {
    hipLaunchKernelGGL(decode_prep_h_kernel, dim3(R), dim3(256), 0, st, parent, (const bf16_t *)h1, ld_h1, H1, (const bf16_t *)h2, ld_h2, H2,
                       (bf16_t *)a1, ld_a1, (bf16_t *)a2, ld_a2, off_h2);
}
void k_row_div(hipStream_t st, int32_t *out, int R, int K) {
    hipLaunchKernelGGL(row_div_kernel, dim3((R + 255) / 256), dim3(256), 0, st, out, R, K);
}
void k_gather_rows_f32(hipStream_t st, const float *in, int64_t ld, const int32_t *src_row, int R, int C, float *out) {
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3(R), dim3(256), 0, st, in, ld, src_row, R, C, out);
}
void k_mul_f32(hipStream_t st, const float *a, const float *b, int64_t n, float *out) {
    hipLaunchKernelGGL(mul_f32_kernel, dim3(grid1d(n)), dim3(256), 0, st, a, b, n, out);
}
