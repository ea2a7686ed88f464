// gemm_skinny.hip -- C[M][N] (+)= A[M][K] * B[N][K]^T for FEW rows (M <= 256) and many columns: the recurrent LSTM
// GEMMs (M = batch rows of one timestep), fc6/fc7 and the image embedding at small per-GPU batch.        (gfx950, bf16)
//
// These problems are weight streams: every B element is used by M/16 MFMAs only, so the job is to pull B through the chip
// once, at full width, with enough loads in flight -- not to tile for reuse (cdna guide 5, "GEMV / M <= 16" row, extended
// to M <= 256 with MFMA):
//   * a workgroup = 4 waves = 64 output columns, one 16-column n-tile per wave, ALL M rows; grid = N/64 x K-slices;
//   * each wave streams ITS OWN 16 weight rows (two 1-KiB LDS-DMA pieces per 64-deep K-tile, whole 128-byte lines),
//     PF K-tiles ahead; A (shared by the four waves, L2-resident) rides the same (PF + 1)-slot LDS ring; XOR-swizzled
//     128-byte rows; one barrier per K-tile;
//   * all staging is LDS-DMA and all fragment reads are inline asm with hand-counted s_waitcnt vmcnt / lgkmcnt (an ordinary
//     load in the loop makes hipcc drain the DMA queue with vmcnt(0): cdna guide, "three .s-level traps" (b); async
//     register-destination loads carried across the loop back-edge were tried and are unsafe: hipcc may copy the
//     destination registers before the data has landed);
//   * split-K (long K, few column tiles: fc6) writes f32 slabs that splitk_reduce combines (shared with gemm_8p.hip).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF> __device__ __forceinline__ uint4 lds_read16(unsigned addr) {
    uint4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}

// MT = m-tiles of 16 rows (M <= 16 MT).  PF = K-tiles in flight.
template <int MT, int PF> __global__ __launch_bounds__(256) void gemm_skinny_kernel(const GemmArgs g) {
    constexpr int ROWS = MT * 16;
    constexpr int A_TILE = ROWS * 128;             // bytes per K-tile of A in LDS
    constexpr int SLOT = A_TILE + 64 * 128;        // + the workgroup's 64 weight rows
    constexpr int NBUF = PF + 1;
    constexpr int APW = (MT * 2 + 3) / 4;          // DMA pieces (8 rows x 128 B) per wave per K-tile (MT = 2: 1, 4: 2, 8: 4, 16: 8)
    constexpr int IPT = APW + 2;                   // vector-memory operations per wave per K-tile
    constexpr int MC = MT < 4 ? MT : 4;            // m-tiles per fragment-read chunk
    static_assert(MT % MC == 0 && (NBUF * SLOT) <= 160 * 1024 && APW * 4 * 8 == ROWS, "geometry");
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int M = g.M, N = g.N;
    const int n0 = blockIdx.x * 64 + wave * 16;
    const int KT_all = g.K / 64;
    const int kbeg = (int)((int64_t)KT_all * blockIdx.y / gridDim.y);
    const int KT = (int)((int64_t)KT_all * (blockIdx.y + 1) / gridDim.y) - kbeg;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

    const bf16_t *Ab = reinterpret_cast<const bf16_t *>(g.A);
    const bf16_t *Zp = reinterpret_cast<const bf16_t *>(g.zero_page) + (lane & 7) * 8;
    // B staging: piece p (0, 1) of this wave = its weight rows n0 + 8 p .. + 7 (rows >= N: zero page)
    const bf16_t *Bb = reinterpret_cast<const bf16_t *>(g.B);
    int b_off[2];
    bool b_ok[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = p * 8 + (lane >> 3), n = n0 + row;
        b_ok[p] = n < N;
        b_off[p] = b_ok[p] ? n * (int)g.ldb + kbeg * 64 + (((lane & 7) ^ ((row >> 1) & 7)) << 3) : 0;
    }
    // A staging: piece p of wave w covers rows (w * APW + p) * 8 .. + 7; lane -> (row, swizzled source chunk)
    int a_off[APW];
    bool a_ok[APW];
#pragma unroll
    for (int p = 0; p < APW; ++p) {
        const int row = (wave * APW + p) * 8 + (lane >> 3);
        a_ok[p] = row < M;
        a_off[p] = a_ok[p] ? row * (int)g.lda + kbeg * 64 + (((lane & 7) ^ ((row >> 1) & 7)) << 3) : 0;
    }
    auto issue_tile = [&](int kt, int slot) {
        const bool live = kt < KT;
#pragma unroll
        for (int p = 0; p < APW; ++p) {
            const int piece = wave * APW + p;
            const bf16_t *src = (live & a_ok[p]) ? Ab + (a_off[p] + kt * 64) : Zp;
            const int dst = slot * SLOT + piece * 8 * 128;  // 4 APW pieces = MT * 16 rows exactly
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(smem + dst), 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bf16_t *src = (live & b_ok[p]) ? Bb + (b_off[p] + kt * 64) : Zp;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(smem + slot * SLOT + A_TILE + (wave * 16 + p * 8) * 128), 16, 0, 0);
        }
    };

    // fragment address: row l15 of a 16-row tile, chunk (4 s + lq) ^ swizzle(row)   [tile bases are multiples of 16 rows]
    unsigned fa[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) fa[s] = lds0 + l15 * 128 + (((4 * s + lq) ^ ((l15 >> 1) & 7)) << 4);

    f32x4v acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};

    static_for<0, PF>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        issue_tile(t, t % NBUF);
    });
    const int KTp = (KT + PF - 1) / PF * PF;
    for (int t0 = 0; t0 < KTp; t0 += PF) {
        static_for<0, PF>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            const int t = t0 + u;
            wait_vmcnt<(PF - 1) * IPT>();  // this wave's share of tile t (A pieces and its own B fragments) has landed
            __builtin_amdgcn_s_barrier();  // ... and everybody else's A pieces; slot (t - 1) % NBUF is free again
            issue_tile(t + PF, (t + PF) % NBUF);
            const unsigned base = (unsigned)((t % NBUF) * SLOT);
            const unsigned bbase = base + A_TILE + wave * 16 * 128;
            const uint4 b0 = lds_read16<0>(fa[0] + bbase), b1 = lds_read16<0>(fa[1] + bbase);
            static_for<0, MT / MC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                uint4 af[MC][2];
                static_for<0, MC>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    af[i][0] = lds_read16<(c * MC + i) * 16 * 128>(fa[0] + base);
                    af[i][1] = lds_read16<(c * MC + i) * 16 * 128>(fa[1] + base);
                });
                wait_lgkm<0>();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MC; ++i) {
                    acc[c * MC + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i][0]),
                                                                             __builtin_bit_cast(bf16x8, b0), acc[c * MC + i], 0, 0, 0);
                    acc[c * MC + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i][1]),
                                                                             __builtin_bit_cast(bf16x8, b1), acc[c * MC + i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }
    wait_vmcnt<0>();  // the dead tiles issued by the last iterations

    // ---- epilogue: D rows = m (4 lq + r), column = n0 + l15 ----
    const int col = n0 + l15;
    if (col >= N) return;
    if (gridDim.y > 1) {  // split-K: partial tile -> this slice's f32 slab [M][N]
        float *slab = reinterpret_cast<float *>(g.ws) + (size_t)blockIdx.y * M * N;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = i * 16 + 4 * lq + r;
                if (m < M) slab[(size_t)m * N + col] = acc[i][r];
            }
        return;
    }
    const float bias = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = i * 16 + 4 * lq + r;
            if (m >= M) continue;
            float v = acc[i][r] + bias;
            const int64_t off = (int64_t)m * g.ldc + col;
            if (g.c_f32) {
                float *c = reinterpret_cast<float *>(g.C) + off;
                if (g.beta) v += *c;
                if (g.relu) v = fmaxf(v, 0.0f);
                *c = v;
            } else {
                bf16_t *c = reinterpret_cast<bf16_t *>(g.C) + off;
                if (g.beta) v += (float)*c;
                if (g.relu) v = fmaxf(v, 0.0f);
                *c = (bf16_t)v;
            }
        }
}

template <int MT, int PF> hipError_t launch_one(hipStream_t s, const GemmArgs &g, int splitk) {
    constexpr int lds = (PF + 1) * (MT * 16 + 64) * 128;
    static LdsAttrMask attr_done{0};
    auto kern = gemm_skinny_kernel<MT, PF>;
    if (hipError_t e = set_max_lds(reinterpret_cast<const void *>(kern), lds, attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(g.N, 64), (unsigned)splitk), dim3(256), lds, s, g);
    return hipGetLastError();
}

}  // namespace

// PLAIN bf16 problems with M <= 256 rows; K a multiple of 64; the usual alignment rules of the direct-to-LDS path.
bool gemm_skinny_eligible(const GemmArgs &g) {
    if (g.dtype != GEMM_T_BF16 || !g.zero_page || g.a_mode != GEMM_A_PLAIN || g.out_mode != GEMM_OUT_PLAIN) return false;
    if (g.M < 1 || g.M > 256 || g.N < 16 || g.K < 64 || (g.K % 64)) return false;
    if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || (g.lda % 8) || (g.ldb % 8)) return false;
    if ((int64_t)g.M * g.lda >= (1ll << 31)) return false;
    return true;
}

// K-slices: enough workgroups to cover the chip when there are few column tiles and K is long (>= 16 K-tiles per slice)
int gemm_skinny_splitk(const GemmArgs &g) {
    if (!g.ws || (g.N % 4) || (g.ldc % 4) || ((uintptr_t)g.C & 15)) return 1;
    const int kt = g.K / 64, tiles = cdiv(g.N, 64);
    int s = 512 / tiles;
    if (s > kt / 16) s = kt / 16;
    if (s > 32) s = 32;
    while (s >= 2 && (size_t)s * g.M * g.N * sizeof(float) > g.ws_bytes) --s;
    return s < 2 ? 1 : s;
}

hipError_t launch_gemm_skinny(hipStream_t stream, const GemmArgs &g) {
    if (!gemm_skinny_eligible(g)) return hipErrorInvalidValue;
    const int sk = gemm_skinny_splitk(g);
    hipError_t e;
    if (g.M <= 32)
        e = launch_one<2, 4>(stream, g, sk);
    else if (g.M <= 64)
        e = launch_one<4, 4>(stream, g, sk);
    else if (g.M <= 128)
        e = launch_one<8, 3>(stream, g, sk);
    else
        e = launch_one<16, 2>(stream, g, sk);
    if (e != hipSuccess || sk == 1) return e;
    return launch_splitk_reduce(stream, g, sk);
}
