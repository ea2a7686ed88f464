// gemm.hip -- NT GEMM / implicit-GEMM 3x3 convolution on the gfx950 matrix cores.
//
//   C[M][N] (+)= A[M][K] * B[N][K]^T (+ bias[N]) (ReLU) (2x2 max-pool over 4 consecutive rows)
//
// Both operands are K-contiguous, so both are staged with coalesced 16-byte loads into 128-byte LDS rows
// (one K-step = 128 bytes of K: 64 bf16 or 32 f32) and both fragment reads are ds_read_b128 from rows padded to
// 144 bytes (conflict-free for any 16 consecutive rows, cdna guide G4).  Within a K-step lane half hh of a wave
// owns bytes [64hh, 64hh+64) of its row for BOTH operands; the contraction index is a sum, so any assignment of k
// to (instruction, lane half) is valid as long as A and B use the same one.
//   bf16: v_mfma_f32_32x32x16_bf16, 4 per 32x32 tile per K-step;  f32: v_mfma_f32_32x32x2_f32 (exact fp32), 16.
// A-operand modes: PLAIN rows, or CONV3 = the im2col row of a 3x3/pad-1 convolution over NHWC activations formed
// on the fly (never in HBM): K-step kt -> tap (kh,kw) = kt / (Cin/BK), channel slice c0; the row is the Cin-slice of
// input pixel (y+kh-1, x+kw-1) or zeros outside the image.  Rows are in window-major pixel order (common.h) so the
// 2x2 max-pool is a max over 4 accumulator registers of one lane.
//
// Replaces, in the reference: cublasSgemm at lrcn.jl:529/545/550/558 (+ their AutoGrad duals), conv4+bias+relu+pool
// at lrcn.jl:724-726, fcx at lrcn.jl:728.
#include <cstdio>
#include "gemm.h"

#include <cstdlib>

#include "common.h"

namespace {

constexpr int ROWB = 144;  // LDS row stride: 128 B of K + 16 B pad

template <typename T> __device__ __forceinline__ uint4 mask_tail(uint4 v, int nvalid);
template <> __device__ __forceinline__ uint4 mask_tail<float>(uint4 v, int nvalid) {
    if (nvalid < 4) v.w = 0;
    if (nvalid < 3) v.z = 0;
    if (nvalid < 2) v.y = 0;
    if (nvalid < 1) v.x = 0;
    return v;
}
template <> __device__ __forceinline__ uint4 mask_tail<bf16_t>(uint4 v, int nvalid) {
    unsigned *d = reinterpret_cast<unsigned *>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (nvalid <= 2 * i)
            d[i] = 0;
        else if (nvalid == 2 * i + 1)
            d[i] &= 0xFFFFu;
    }
    return v;
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const uint4 &a, const uint4 &b, f32x16 &acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc,
                                                      0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const uint4 &a, const uint4 &b, f32x16 &acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

template <typename T, int BM, int BN, int AMODE>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs g) {
    constexpr int CE = TypeInfo<T>::chunk;  // elements per 16-byte chunk
    constexpr int BK = TypeInfo<T>::bk;     // elements per 128-byte K-step
    constexpr int AP = BM / 32, BP = BN / 32;
    constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
    constexpr int BUF = (BM + BN) * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (g.N + BN - 1) / BN;
    const int nt = blockIdx.x % tiles_n, mt = blockIdx.x / tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;
    const int M = g.M, N = g.N, K = g.K;

    const int kpt = (AMODE == GEMM_A_CONV3) ? g.Cin / BK : 1;  // K-steps per tap
    const int KT = (AMODE == GEMM_A_CONV3) ? 9 * kpt : (K + BK - 1) / BK;

    // ---- loader state: thread -> (row lrow + 32 i, 16-byte chunk lchunk) ----
    const int lrow = tid >> 3, lchunk = tid & 7;
    const T *a_base[AP];
    int a_y[AP], a_x[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + lrow + 32 * i;
        a_base[i] = nullptr;
        a_y[i] = a_x[i] = 0;
        if (m < M) {
            if (AMODE == GEMM_A_CONV3) {
                const PixDecode p = decode_pixel(m, g.H, g.W);
                a_base[i] = reinterpret_cast<const T *>(g.A) + (int64_t)p.n * g.H * g.W * g.Cin;
                a_y[i] = p.y;
                a_x[i] = p.x;
            } else {
                a_base[i] = reinterpret_cast<const T *>(g.A) + (int64_t)m * g.lda;
            }
        }
    }
    const T *b_base[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int n = n0 + lrow + 32 * i;
        b_base[i] = (n < N) ? reinterpret_cast<const T *>(g.B) + (int64_t)n * g.ldb : nullptr;
    }

    uint4 ra[AP], rb[BP];
    auto load_tile = [&](int kt) {
        const int kk = kt * BK + lchunk * CE;  // element offset along K (B operand, and A in PLAIN mode)
        const uint4 zero = make_uint4(0, 0, 0, 0);
        if (AMODE == GEMM_A_CONV3) {
            const int tap = kt / kpt, c0 = (kt - tap * kpt) * BK;
            const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int y = a_y[i] + kh - 1, x = a_x[i] + kw - 1;
                const bool ok = a_base[i] != nullptr && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
                ra[i] = ok ? *reinterpret_cast<const uint4 *>(a_base[i] + ((int64_t)y * g.W + x) * g.Cin + c0 +
                                                              lchunk * CE)
                           : zero;
            }
        } else {
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                uint4 v = zero;
                if (a_base[i] != nullptr && kk < K) {
                    v = *reinterpret_cast<const uint4 *>(a_base[i] + kk);
                    if (kk + CE > K) v = mask_tail<T>(v, K - kk);
                }
                ra[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            uint4 v = zero;
            if (b_base[i] != nullptr && kk < K) {
                v = *reinterpret_cast<const uint4 *>(b_base[i] + kk);
                if (kk + CE > K) v = mask_tail<T>(v, K - kk);
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char *As = smem + buf * BUF, *Bs = As + BM * ROWB;
#pragma unroll
        for (int i = 0; i < AP; ++i) *reinterpret_cast<uint4 *>(As + (lrow + 32 * i) * ROWB + lchunk * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < BP; ++i) *reinterpret_cast<uint4 *>(Bs + (lrow + 32 * i) * ROWB + lchunk * 16) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    const int frag_off = (lane & 31) * ROWB + (lane >> 5) * 64;
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) load_tile(kt + 1);
        const unsigned char *As = smem + cur * BUF + (wr * WTM) * ROWB + frag_off;
        const unsigned char *Bs = smem + cur * BUF + BM * ROWB + (wc * WTN) * ROWB + frag_off;
        uint4 af[TM][4], bf[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) af[i][j] = *reinterpret_cast<const uint4 *>(As + i * 32 * ROWB + j * 16);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[i][j] = *reinterpret_cast<const uint4 *>(Bs + i * 32 * ROWB + j * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int n = 0; n < TN; ++n) Mma<T>::run(af[i][j], bf[n][j], acc[i][n]);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
    const int hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = n0 + wc * WTN + n * 32 + (lane & 31);
            if (col >= N) continue;
            const float bias = g.bias ? g.bias[col] : 0.0f;
            const int rbase = m0 + wr * WTM + i * 32 + 4 * hh;
            if (g.out_mode == GEMM_OUT_POOL) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = rbase + 8 * q;  // first of 4 consecutive rows = one pool window
                    if (row >= M) continue;
                    float v = fmaxf(fmaxf(acc[i][n][4 * q], acc[i][n][4 * q + 1]),
                                    fmaxf(acc[i][n][4 * q + 2], acc[i][n][4 * q + 3])) + bias;
                    if (g.relu) v = fmaxf(v, 0.0f);
                    const int64_t off = (int64_t)(row >> 2) * g.ldc + col;
                    if (g.c_f32)
                        reinterpret_cast<float *>(g.C)[off] = v;
                    else
                        reinterpret_cast<T *>(g.C)[off] = from_f32<T>(v);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row >= M) continue;
                    int64_t off;
                    if (g.out_mode == GEMM_OUT_CONV) {
                        const PixDecode p = decode_pixel(row, g.H, g.W);
                        off = (((int64_t)p.n * g.H + p.y) * g.W + p.x) * g.ldc + col;
                    } else {
                        off = (int64_t)row * g.ldc + col;
                    }
                    float v = acc[i][n][r] + bias;
                    if (g.c_f32) {
                        float *c = reinterpret_cast<float *>(g.C) + off;
                        if (g.beta) v += *c;
                        if (g.relu) v = fmaxf(v, 0.0f);
                        *c = v;
                    } else {
                        T *c = reinterpret_cast<T *>(g.C) + off;
                        if (g.beta) v += to_f32(*c);
                        if (g.relu) v = fmaxf(v, 0.0f);
                        *c = from_f32<T>(v);
                    }
                }
            }
        }
    }
}

template <typename T, int BM, int BN, int AMODE> hipError_t launch_one(hipStream_t s, const GemmArgs &g) {
    constexpr int lds = 2 * (BM + BN) * ROWB;
    static LdsAttrMask attr_done{0};
    auto kern = gemm_nt_kernel<T, BM, BN, AMODE>;
    if (hipError_t e = set_max_lds(reinterpret_cast<const void *>(kern), lds, attr_done); e != hipSuccess) return e;
    const int64_t blocks = (int64_t)cdiv(g.M, BM) * cdiv(g.N, BN);
    if (blocks <= 0 || blocks > 0x7FFFFFFF) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, g);
    return hipGetLastError();
}

template <typename T> hipError_t dispatch(hipStream_t s, const GemmArgs &g) {
    const int64_t big = (int64_t)cdiv(g.M, 128) * cdiv(g.N, 128);
    const bool small = big < 160 || g.M <= 64 || g.N <= 64;
    if (g.a_mode == GEMM_A_CONV3)
        return small ? launch_one<T, 64, 64, GEMM_A_CONV3>(s, g) : launch_one<T, 128, 128, GEMM_A_CONV3>(s, g);
    return small ? launch_one<T, 64, 64, GEMM_A_PLAIN>(s, g) : launch_one<T, 128, 128, GEMM_A_PLAIN>(s, g);
}

}  // namespace

static thread_local char g_route[48] = "";
static thread_local int g_route_cfg = -1;
void gemm_debug_note_route(const char *route, int cfg) {
    if (route) {
        if (cfg >= 0) snprintf(g_route, sizeof(g_route), "%s:%d", route, cfg);
        else snprintf(g_route, sizeof(g_route), "%s", route);
    } else {
        g_route_cfg = cfg;  // tile config / slice count of the launch in flight (composed into the string by launch_gemm)
    }
}
const char *gemm_debug_last_route() { return g_route; }

static hipError_t launch_gemm_routed(hipStream_t stream, const GemmArgs &g, const char **route);
hipError_t launch_gemm(hipStream_t stream, const GemmArgs &g) {
    static const bool trace = getenv("LRCN_GEMM_TRACE") != nullptr;  // development: print the kernel family every GEMM takes
    const char *route = "?";
    g_route_cfg = -1;
    const hipError_t e = launch_gemm_routed(stream, g, &route);
    gemm_debug_note_route(route, g_route_cfg);
    if (trace) fprintf(stderr, "[gemm] M=%d N=%d K=%d lda=%ld ldb=%ld amode=%d out=%d beta=%d -> %s\n", g.M, g.N, g.K, (long)g.lda, (long)g.ldb, g.a_mode, g.out_mode, (int)g.beta, route);
    return e;
}
static hipError_t launch_gemm_routed(hipStream_t stream, const GemmArgs &g, const char **route) {
    if (g.dtype == GEMM_T_F8) {  // e4m3: the phase-interleaved convolution kernel or nothing
        int64_t blocks = 0;
        if (g.M <= 0 || !g.A || !g.B || !g.C || gemm_8p_config(g, &blocks) < 0) return hipErrorInvalidValue;
        *route = "8p-f8";
        return launch_gemm_8p(stream, g);
    }
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || !g.A || !g.B || !g.C) return hipErrorInvalidValue;
    // LRCN_GLDS=0 disables the direct-to-LDS path, LRCN_GLDS=force uses it whenever eligible (tests); default: when
    // the grid fills the chip.
    const char *knob = getenv("LRCN_GLDS");
    const char *knob8 = getenv("LRCN_8P");  // LRCN_8P=0 disables the phase-interleaved path, =force lowers its grid threshold
    const char *knobs = getenv("LRCN_SKINNY");  // LRCN_SKINNY=0 disables the skinny-M weight-streaming kernel
    const bool skinny_ok = !(knob && knob[0] == '0') && !(knobs && knobs[0] == '0') && gemm_skinny_eligible(g);
    if (!(knob && knob[0] == '0') && !(knob8 && knob8[0] == '0')) {
        int64_t blocks = 0;
        // (beside the capped convolution grids a launch that fills 3/4 of the FREE CUs in one round counts as filling the chip; LRCN_FREE_8P_MIN:
        // that fraction in percent, development knob)
        static const int free_pct = getenv("LRCN_FREE_8P_MIN") ? atoi(getenv("LRCN_FREE_8P_MIN")) : 75;
        if (gemm_8p_config(g, &blocks) >= 0 &&
            ((knob8 && knob8[0] == 'f') || blocks >= 128 || (g.free_cus > 0 && blocks * 100 >= (int64_t)g.free_cus * free_pct && blocks <= g.free_cus))) {
            *route = "8p";
            return launch_gemm_8p(stream, g);
        }
        // Beside the capped convolution grids only ~bg_cus CUs are free: a launch of many small workgroups runs in several rounds
        // on them, one of few large tiles (less operand traffic per FLOP) in one.  The recurrent GEMM of the B = 256 step
        // (256 x 4000 x 1024): 126 workgroups of 128 x 64 take 13 us alone but 50 us beside the VGG forward (4 rounds on 32 CUs);
        // 32 workgroups of 256 x 128 take 25 us either way.
        static const int bg_minN = getenv("LRCN_BG_MINN") ? atoi(getenv("LRCN_BG_MINN")) : 512;  // development knob
        if (g.bg_cus > 0 && g.a_mode == GEMM_A_PLAIN && g.M >= 256 && g.M <= 512 && g.N >= bg_minN) {
            GemmArgs h = g;
            h.cfg_pref = 2;
            if (gemm_8p_config(h, &blocks) >= 0 && blocks <= 2 * g.bg_cus) {
                // few tiles and a long K (the backward dh GEMM of the recurrence: 8 tiles x 63 K-tiles): K slices over the free CUs
                static const int bg_sk = getenv("LRCN_BG_SPLITK") ? atoi(getenv("LRCN_BG_SPLITK")) : 0;  // development knob (0 = off)
                if (bg_sk > 1 && blocks * bg_sk <= g.bg_cus && g.K >= 64 * 8 * bg_sk && g.c_f32 && !g.beta && !g.relu && g.out_mode == GEMM_OUT_PLAIN &&
                    g.ws && (size_t)bg_sk * g.M * g.N * sizeof(float) <= g.ws_bytes && (g.ldc % 4) == 0 && (g.N % 4) == 0 && !((uintptr_t)g.C & 15)) {
                    *route = "8p-bg-splitk";
                    h.splitk_forced = 1;
                    return launch_gemm_8p(stream, h, bg_sk);
                }
                *route = "8p-bg";
                return launch_gemm_8p(stream, h);
            }
        }
        if (skinny_ok && g.M <= 128) { *route = "skinny"; return launch_gemm_skinny(stream, g); }
        const char *ksk = getenv("LRCN_8P_SPLITK");  // kernel-development knob: 0 disables the split-K form
        const int sk = (ksk && ksk[0] == '0') ? 0 : gemm_8p_splitk(g, &blocks);
        const char *kth = getenv("LRCN_8P_SPLITK_MIN");
        if (sk > 1 && blocks >= (kth ? atoi(kth) : 96)) { *route = "8p-splitk"; return launch_gemm_8p(stream, g, sk); }
    }
    if (!(knob && knob[0] == '0') && gemm_glds_eligible(g)) {
        const int64_t blocks = gemm_glds_blocks(g);
        if (blocks > 0 && ((knob && knob[0] == 'f') || blocks >= 96)) { *route = "glds"; return launch_gemm_glds(stream, g); }
    }
    if (skinny_ok) { *route = "skinny-last"; return launch_gemm_skinny(stream, g); }
    if (!(knob && knob[0] == '0') && gemm_glds_eligible(g) && gemm_glds_blocks(g) >= 16) {  // few tiles, but still far ahead of gemm_nt
        *route = "glds-small";
        return launch_gemm_glds(stream, g);
    }  // 128 < M <= 256 with too few tiles for the paths above
    const int ce = g.dtype == GEMM_T_BF16 ? 8 : 4;
    if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || (g.ldb % ce)) return hipErrorInvalidValue;
    if (g.a_mode == GEMM_A_CONV3) {
        const int bk = g.dtype == GEMM_T_BF16 ? 64 : 32;
        if (g.Cin % bk || g.K != 9 * g.Cin || (g.H & 1) || (g.W & 1) || g.M % (g.H * g.W)) return hipErrorInvalidValue;
    } else if (g.lda % ce) {
        return hipErrorInvalidValue;
    }
    if (g.out_mode != GEMM_OUT_PLAIN && ((g.H & 1) || (g.W & 1) || g.H <= 0 || g.W <= 0)) return hipErrorInvalidValue;
    if (g.out_mode == GEMM_OUT_POOL && (g.beta || (g.M & 3))) return hipErrorInvalidValue;
    *route = "gemm_nt";
    return g.dtype == GEMM_T_BF16 ? dispatch<bf16_t>(stream, g) : dispatch<float>(stream, g);
}
