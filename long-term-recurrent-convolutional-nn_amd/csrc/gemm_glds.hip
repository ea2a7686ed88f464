// gemm_glds.hip -- the bf16 NT contraction / implicit-GEMM 3x3 convolution with direct-to-LDS staging (gfx950).
//
// Same math and epilogues as gemm.hip, restructured for the CDNA4 memory path (cdna guide section 5):
//   * both operand tiles are staged HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write);
//     the per-lane SOURCE address makes the A tile a row gather, so the im2col row of a 3x3/pad-1 convolution is
//     formed by the DMA itself: lane -> (row, 16-byte chunk), source = centre pixel + tap offset + channel slice,
//     or a 256-byte zero page when the tap falls outside the image / the row is beyond M;
//   * LDS rows are 128 bytes (64 bf16 of K) with NO padding (the DMA writes lane-linear 1 KiB pieces); bank conflicts
//     are removed by an XOR swizzle applied on the source chunk index and again on the ds_read_b128 address:
//     logical chunk j of row r lives at chunk position j ^ ((r >> 1) & 7)   (conflict-free for every 16-lane group);
//   * 256-row tiles, 8 waves (2 per SIMD), 2 LDS buffers: the DMA of K-step t+1 is in flight while the MFMAs of K-step
//     t run; one barrier per K-step;
//   * workgroups are renumbered so that the N-tiles sharing one im2col panel run on the same XCD (same L2).
// MFMA: v_mfma_f32_32x32x16_bf16; rows in window-major pixel order so 2x2 max-pool = max of 4 accumulator registers.
#include <cstdlib>

#include "common.h"
#include "gemm.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ uint4 lds_read16(unsigned addr) {
    uint4 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}

// T = bf16_t: v_mfma_f32_32x32x16_bf16 on 64-element K-steps.  T = float (the exact-fp32 VGG path, BASELINE configs[1]):
// the same staging / ring / swizzle on 32-element K-steps (still 128-byte rows), v_mfma_f32_32x32x2_f32 -- a lane's 16-byte
// fragment feeds four MFMAs (component c of A with component c of B; lane half hh = k index of the instruction), f32 output.
template <typename T, int WM, int WN, int TM, int TN, int AMODE, int NBUF>
__global__ __launch_bounds__(WM *WN * 64) void gemm_glds_kernel(const GemmArgs g) {
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int KE = 128 / (int)sizeof(T);  // elements per K-step (one 128-byte LDS row)
    constexpr int CE = 16 / (int)sizeof(T);   // elements per 16-byte chunk
    constexpr int NW = WM * WN;  // waves per workgroup (4 or 8); each wave stages 32 A rows, so BM = 32 NW
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(BM == NW * 32 && (NW == 8 || NW == 4), "each wave stages 32 rows of A");
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
    constexpr int BQ = BN / (8 * NW);  // B staging instructions per wave per K-step
    static_assert(BQ >= 1 && BQ * 8 * NW == BN, "B tile must split evenly over the waves");
    constexpr int IPT = 4 + BQ;        // DMA instructions per wave per K-step
    static_assert((NBUF - 2) * IPT <= 48, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int M = g.M, N = g.N;
    const int tiles_n = (N + BN - 1) / BN;
    // XCD-aware renumbering (bijective form, cdna guide T1): blocks b, b+8, ... share an XCD; give each XCD a contiguous
    // run of logical tiles so the N-tiles of one M-panel (consecutive logical ids) hit the same L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int nt = bid % tiles_n, mt = bid / tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;

    const T *Ab = reinterpret_cast<const T *>(g.A);
    const T *Bb = reinterpret_cast<const T *>(g.B);
    const T *Zp = reinterpret_cast<const T *>(g.zero_page) + (lane & 7) * CE;

    const int kpt = (AMODE == GEMM_A_CONV3) ? g.Cin / KE : g.K / KE;
    const int KT_all = (AMODE == GEMM_A_CONV3) ? 9 * kpt : kpt;
    // split-K: blockIdx.y owns K-steps [kbeg, kbeg + KT); partial sums are combined by f32 atomics in the epilogue
    const int kbeg = (int)((int64_t)KT_all * blockIdx.y / gridDim.y);
    const int KT = (int)((int64_t)KT_all * (blockIdx.y + 1) / gridDim.y) - kbeg;

    // ---- per-lane staging state ----
    // A: wave w, instruction q stages rows w*32 + q*8 + (lane>>3), chunk position lane&7
    int a_off[4];
    unsigned a_mask[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wave * 32 + q * 8 + (lane >> 3);
        const int m = m0 + row;
        const int src_chunk = (lane & 7) ^ ((row >> 1) & 7);
        a_off[q] = 0;
        a_mask[q] = 0;
        if (m < M) {
            if (AMODE == GEMM_A_CONV3) {
                const PixDecode p = decode_pixel(m, g.H, g.W);
                a_off[q] = ((p.n * g.H + p.y) * g.W + p.x) * g.Cin + src_chunk * CE;
                unsigned mk = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int y = p.y + t / 3 - 1, x = p.x + t % 3 - 1;
                    if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W) mk |= 1u << t;
                }
                a_mask[q] = mk;
            } else {
                a_off[q] = m * (int)g.lda + src_chunk * CE;
                a_mask[q] = 1;
            }
        }
    }
    // B: instruction j = wave + 8*i stages rows j*8 + (lane>>3)
    int b_off[BQ];
    bool b_ok[BQ];
#pragma unroll
    for (int i = 0; i < BQ; ++i) {
        const int row = (wave + NW * i) * 8 + (lane >> 3);
        const int n = n0 + row;
        const int src_chunk = (lane & 7) ^ ((row >> 1) & 7);
        b_ok[i] = n < N;
        b_off[i] = b_ok[i] ? n * (int)g.ldb + src_chunk * CE : 0;
    }

    // One DMA piece (1 KiB = 8 rows x 128 B) of the K-step `kt` tile into ring slot `buf`: p < 4 -> A piece p of this
    // wave's 32 rows, p >= 4 -> B piece.  live = false stages zeros (used for the tail, into a dead slot, so that the
    // K-step body is branch-free and the vmcnt bookkeeping uniform).
    struct KStep {
        int tap, koff, kb;
    };
    auto kstep_of = [&](int kt) {
        // K-step order for the convolution: channel slice OUTER, tap INNER (kt = slice*9 + tap): the nine taps of one
        // 64-channel slice re-read the same (patch + halo) pixels in nine consecutive K-steps -> L2 hits.
        KStep k;
        if (AMODE == GEMM_A_CONV3) {
            const int slice = kt / 9;
            k.tap = kt - 9 * slice;
            const int kh = k.tap / 3, kw = k.tap - 3 * kh;
            k.koff = ((kh - 1) * g.W + (kw - 1)) * g.Cin + slice * KE;
            k.kb = k.tap * g.Cin + slice * KE;
        } else {
            k.tap = 0;
            k.koff = kt * KE;
            k.kb = k.koff;
        }
        return k;
    };
    auto stage_piece = [&](int buf, const KStep &k, int p, bool live) {
        unsigned char *As = smem + buf * BUF;
        unsigned char *Bs = As + A_BYTES;
        if (p < 4) {
            const bool ok = live && ((a_mask[p] >> k.tap) & 1u);
            const T *src = ok ? Ab + (a_off[p] + k.koff) : Zp;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(As + (wave * 32 + p * 8) * 128), 16, 0, 0);
        } else {
            const int i = p - 4;
            const T *src = (live && b_ok[i]) ? Bb + (b_off[i] + k.kb) : Zp;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(Bs + (wave + NW * i) * 8 * 128), 16, 0, 0);
        }
    };

    // fragment read offsets: lane (r = lane&31, hh = lane>>5) reads logical chunks 4hh..4hh+3 of its row
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    int fo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fo[j] = (lane & 31) * 128 + ((((lane >> 5) * 4 + j) ^ (((lane & 31) >> 1) & 7)) << 4);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // NBUF-deep ring: the DMA of K-steps kt+1 .. kt+NBUF-1 is in flight while K-step kt is multiplied.  A K-step's
    // data is ordered for every wave's ds_read by: each wave's counted vmcnt for its own pieces, then the barrier.
    // (raw s_barrier: __syncthreads() would drain vmcnt to 0 -- cdna guide "Pipelining across barriers")
    // All DMA pieces of the next tile are issued in one burst right after the barrier, then the compiler schedules the
    // ds_reads against the MFMAs.  (Tried: cutting the K-step into sched_barrier-pinned chunks of [1 DMA piece, 2 A-fragment
    // reads, 2*TN MFMAs] -- 14 % SLOWER, 693 vs 803 TF over the VGG stack; the burst form is kept.)
#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t) {
        const KStep k = kstep_of(kbeg + (t < KT ? t : 0));
#pragma unroll
        for (int p = 0; p < IPT; ++p) stage_piece(t, k, p, t < KT);
    }
    for (int kt = 0; kt < KT; ++kt) {
        wait_vmcnt<(NBUF - 2) * IPT>();
        __builtin_amdgcn_s_barrier();
        const int nxt = kt + NBUF - 1;
        const bool live = nxt < KT && !(g.dbg & 1);
        const KStep kn = kstep_of(kbeg + (nxt < KT ? nxt : 0));
#pragma unroll
        for (int p = 0; p < IPT; ++p) stage_piece(nxt % NBUF, kn, p, live);
        if (g.dbg & 2) continue;
        const int cur = kt % NBUF;
        // Fragment reads are inline asm: hipcc orders every ordinary LDS load behind ALL outstanding LDS-DMA with
        // s_waitcnt vmcnt(0) (it cannot tell which ring slot a global_load_lds writes), which drained the ring at every
        // K-step.  Ordering is by hand: the counted vmcnt + barrier above, counted lgkmcnt below.
        const unsigned As = lds0 + cur * BUF + (wm * TM * 32) * 128;
        const unsigned Bs = lds0 + cur * BUF + A_BYTES + (wn * TN * 32) * 128;
        uint4 bfr[TN][4], af[2][4];
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[n][j] = lds_read16(Bs + n * 32 * 128 + fo[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) af[0][j] = lds_read16(As + fo[j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i + 1 < TM) {
#pragma unroll
                for (int j = 0; j < 4; ++j) af[(i + 1) & 1][j] = lds_read16(As + (i + 1) * 32 * 128 + fo[j]);
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    if constexpr (F32) {
                        const uint4 &a4 = af[i & 1][j], &b4 = bfr[n][j];
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a4.x), __uint_as_float(b4.x), acc[i][n], 0, 0, 0);
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a4.y), __uint_as_float(b4.y), acc[i][n], 0, 0, 0);
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a4.z), __uint_as_float(b4.z), acc[i][n], 0, 0, 0);
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a4.w), __uint_as_float(b4.w), acc[i][n], 0, 0, 0);
                    } else {
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i & 1][j]),
                                                                            __builtin_bit_cast(bf16x8, bfr[n][j]), acc[i][n], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    wait_vmcnt<0>();  // the tail's dummy pieces

    // ---- epilogue A (bf16 output, no accumulate): bias/ReLU/pool in registers -> bf16 C tile staged in LDS (the ring is
    // free now) -> whole rows written with 16 B per lane (one 2*BN-byte row segment per BN/8 lanes).  The direct
    // per-lane 2-byte stores of epilogue B cost the un-pooled convolution layers 15-45 % (profiles/r01 conv_bench).
    const int hh = lane >> 5;
    const bool staged = !F32 && !g.c_f32 && !g.beta && gridDim.y == 1 && (g.ldc % 8) == 0 && ((uintptr_t)g.C & 15) == 0 &&
                        (n0 + BN <= N || (N % 8) == 0);
    const bool c_f32 = F32 || g.c_f32;  // T = float: the output is float whether or not the caller says so
    if (staged) {
        constexpr int CSTR = BN * 2;  // bytes per staged row
        __builtin_amdgcn_s_barrier();  // every wave is done reading the last K-step (and no DMA is in flight)
        const bool pool = g.out_mode == GEMM_OUT_POOL;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const int lcol = (wn * TN + n) * 32 + (lane & 31);
                const int col = n0 + lcol;
                const float bias = (g.bias && col < N) ? g.bias[col] : 0.0f;
                const int lrow0 = (wm * TM + i) * 32 + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (pool) {
                        float v = fmaxf(fmaxf(acc[i][n][4 * q], acc[i][n][4 * q + 1]),
                                        fmaxf(acc[i][n][4 * q + 2], acc[i][n][4 * q + 3])) + bias;
                        if (g.relu) v = fmaxf(v, 0.0f);
                        *reinterpret_cast<bf16_t *>(smem + ((lrow0 + 8 * q) >> 2) * CSTR + lcol * 2) = (bf16_t)v;
                    } else {
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) {
                            float v = acc[i][n][4 * q + s2] + bias;
                            if (g.relu) v = fmaxf(v, 0.0f);
                            *reinterpret_cast<bf16_t *>(smem + (lrow0 + 8 * q + s2) * CSTR + lcol * 2) = (bf16_t)v;
                        }
                    }
                }
            }
        __syncthreads();
        const int rows_out = pool ? BM / 4 : BM;
        constexpr int CPR = BN / 8;  // 16-byte chunks per staged row
        bf16_t *Cb = reinterpret_cast<bf16_t *>(g.C);
        for (int idx = tid; idx < rows_out * CPR; idx += NW * 64) {
            const int lrow = idx / CPR, ch = idx - lrow * CPR;
            const int col = n0 + ch * 8;
            if (col >= N) continue;
            int64_t off;
            if (pool) {
                const int prow = (m0 >> 2) + lrow;
                if (prow >= (M >> 2)) continue;
                off = (int64_t)prow * g.ldc + col;
            } else {
                const int row = m0 + lrow;
                if (row >= M) continue;
                if (g.out_mode == GEMM_OUT_CONV) {
                    const PixDecode p = decode_pixel(row, g.H, g.W);
                    off = (((int64_t)p.n * g.H + p.y) * g.W + p.x) * g.ldc + col;
                } else {
                    off = (int64_t)row * g.ldc + col;
                }
            }
            *reinterpret_cast<uint4 *>(Cb + off) = *reinterpret_cast<const uint4 *>(smem + lrow * CSTR + ch * 16);
        }
        return;
    }

    // ---- epilogue B (f32 / accumulating / split-K outputs): direct stores from the 32x32 MFMA C layout
    //      (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = n0 + (wn * TN + n) * 32 + (lane & 31);
            if (col >= N) continue;
            const float bias = (g.bias && blockIdx.y == 0) ? g.bias[col] : 0.0f;
            const int rbase = m0 + (wm * TM + i) * 32 + 4 * hh;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = rbase + 8 * q;  // 4 consecutive rows row..row+3 = one 2x2 pool window
                if (row >= M) continue;
                if (g.out_mode == GEMM_OUT_POOL) {
                    float v = fmaxf(fmaxf(acc[i][n][4 * q], acc[i][n][4 * q + 1]),
                                    fmaxf(acc[i][n][4 * q + 2], acc[i][n][4 * q + 3])) + bias;
                    if (g.relu) v = fmaxf(v, 0.0f);
                    const int64_t off = (int64_t)(row >> 2) * g.ldc + col;
                    if (c_f32)
                        reinterpret_cast<float *>(g.C)[off] = v;
                    else
                        reinterpret_cast<bf16_t *>(g.C)[off] = (bf16_t)v;
                    continue;
                }
                int64_t off0, dstep_x, dstep_y;
                if (g.out_mode == GEMM_OUT_CONV) {
                    const PixDecode p = decode_pixel(row, g.H, g.W);  // sub-pixel 0 of the window
                    off0 = (((int64_t)p.n * g.H + p.y) * g.W + p.x) * g.ldc + col;
                    dstep_x = g.ldc;
                    dstep_y = (int64_t)g.W * g.ldc;
                } else {
                    off0 = (int64_t)row * g.ldc + col;
                    dstep_x = g.ldc;
                    dstep_y = 2 * g.ldc;
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (row + s >= M) continue;
                    const int64_t off = off0 + (s & 1) * dstep_x + (s >> 1) * dstep_y;
                    float v = acc[i][n][4 * q + s] + bias;
                    if (gridDim.y > 1) {  // split-K (launcher guarantees c_f32, PLAIN, no relu, C pre-zeroed unless beta)
                        atomicAdd(reinterpret_cast<float *>(g.C) + off, v);
                    } else if (c_f32) {
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
        }
    }
}

template <typename T, int WM, int WN, int TM, int TN, int AMODE, int NBUF> hipError_t launch_one(hipStream_t s, const GemmArgs &g, int splitk) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int lds = NBUF * (BM + BN) * 128;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static LdsAttrMask attr_done{0};
    auto kern = gemm_glds_kernel<T, WM, WN, TM, TN, AMODE, NBUF>;
    if (hipError_t e = set_max_lds(reinterpret_cast<const void *>(kern), lds, attr_done); e != hipSuccess) return e;
    const int64_t blocks = (int64_t)cdiv(g.M, BM) * cdiv(g.N, BN);
    if (blocks <= 0 || blocks > 0x7FFFFFFF) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)splitk), dim3(WM * WN * 64), lds, s, g);
    return hipGetLastError();
}

// Tile choice: the largest tile that still gives the chip >= 2 workgroups' worth of parallelism per 2 CUs (>= 200 blocks),
// else the candidate with the most blocks.  Returns the block count of the chosen config (0 = none fits).
struct Cfg {
    int bm, bn;
};
const Cfg kCfgs[4] = {{256, 256}, {256, 128}, {256, 64}, {128, 64}};
int choose_cfg(const GemmArgs &g, int64_t *blocks_out) {
    int best = -1;
    int64_t best_blocks = 0;
    for (int i = 0; i < 4; ++i) {
        if (g.M < kCfgs[i].bm) continue;
        if (kCfgs[i].bn > 64 && g.N <= kCfgs[i].bn / 2) continue;  // do not waste more than half a tile of columns
        const int64_t b = (int64_t)cdiv(g.M, kCfgs[i].bm) * cdiv(g.N, kCfgs[i].bn);
        if (b >= 200) {
            *blocks_out = b;
            return i;
        }
        if (b > best_blocks) {
            best_blocks = b;
            best = i;
        }
    }
    *blocks_out = best_blocks;
    return best;
}

// Split-K for skinny problems (few output tiles, long K): enough slices to give every CU a workgroup, each slice at
// least 4 K-steps.  Needs an f32 PLAIN output without ReLU; C is zeroed first unless the call accumulates (beta).
int choose_splitk(const GemmArgs &g, int64_t blocks) {
    if (!g.c_f32 || g.out_mode != GEMM_OUT_PLAIN || g.relu || g.a_mode != GEMM_A_PLAIN || g.deterministic) return 1;
    if (!g.beta && g.ldc != g.N) return 1;
    if (g.dtype != GEMM_T_BF16) return 1;
    const int kt = g.K / 64;
    int s = (int)(256 / (blocks > 0 ? blocks : 1));
    if (s > kt / 4) s = kt / 4;
    if (s > 16) s = 16;
    return s < 2 ? 1 : s;
}

template <typename T, int AMODE> hipError_t dispatch(hipStream_t s, const GemmArgs &g) {
    int64_t blocks = 0;
    const int cfg = choose_cfg(g, &blocks);
    const int sk = (AMODE == GEMM_A_PLAIN) ? choose_splitk(g, blocks) : 1;
    if (sk > 1 && !g.beta && !g.c_is_zero) {
        hipError_t e = hipMemsetAsync(g.C, 0, sizeof(float) * (size_t)g.M * g.N, s);
        if (e != hipSuccess) return e;
    }
    switch (cfg) {
        case 0: return launch_one<T, 2, 4, 4, 2, AMODE, 2>(s, g, sk);  // 256 x 256, 2 x 64 KiB
        case 1: return launch_one<T, 4, 2, 2, 2, AMODE, 3>(s, g, sk);  // 256 x 128, 3 x 48 KiB
        case 2: return launch_one<T, 4, 2, 2, 1, AMODE, 2>(s, g, sk);  // 256 x 64,  2 x 40 KiB (two workgroups per CU)
        case 3: return launch_one<T, 2, 2, 2, 1, AMODE, 3>(s, g, sk);  // 128 x 64 (4 waves), 3 x 24 KiB (two per CU)
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool gemm_glds_eligible(const GemmArgs &g) {
    if (g.dtype == GEMM_T_F32) {  // exact-fp32 convolutions only (the f32 LSTM GEMMs stay on gemm_nt: the 1e-5 parity path)
        if (g.a_mode != GEMM_A_CONV3 || !g.zero_page || g.M < 128 || g.beta) return false;
        if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || (g.ldb % 4) || (int64_t)g.N * g.ldb >= (1ll << 31)) return false;
        if (g.Cin % 32 || g.K != 9 * g.Cin || (g.H & 1) || (g.W & 1) || g.H <= 0 || g.W <= 0 || g.M % (g.H * g.W)) return false;
        if ((int64_t)g.M * g.Cin >= (1ll << 31) || g.out_mode == GEMM_OUT_PLAIN) return false;
        if (g.out_mode == GEMM_OUT_POOL && (g.M & 3)) return false;
        return true;
    }
    if (g.dtype != GEMM_T_BF16 || !g.zero_page || g.M < 128) return false;
    if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || (g.ldb % 8)) return false;
    if ((int64_t)g.N * g.ldb >= (1ll << 31)) return false;
    if (g.a_mode == GEMM_A_CONV3) {
        if (g.Cin % 64 || g.K != 9 * g.Cin || (g.H & 1) || (g.W & 1) || g.M % (g.H * g.W)) return false;
        if ((int64_t)g.M * g.Cin >= (1ll << 31)) return false;
    } else {
        if (g.K % 64 || g.lda % 8) return false;
        if ((int64_t)g.M * g.lda >= (1ll << 31)) return false;
    }
    if (g.out_mode != GEMM_OUT_PLAIN && ((g.H & 1) || (g.W & 1) || g.H <= 0 || g.W <= 0)) return false;
    if (g.out_mode == GEMM_OUT_POOL && (g.beta || (g.M & 3))) return false;
    return true;
}

hipError_t launch_gemm_glds(hipStream_t stream, const GemmArgs &g0) {
    if (!gemm_glds_eligible(g0)) return hipErrorInvalidValue;
    GemmArgs g = g0;
    const char *dbg = getenv("LRCN_DBG");
    g.dbg = dbg ? atoi(dbg) : 0;
    if (g.dtype == GEMM_T_F32) return dispatch<float, GEMM_A_CONV3>(stream, g);
    return g.a_mode == GEMM_A_CONV3 ? dispatch<bf16_t, GEMM_A_CONV3>(stream, g) : dispatch<bf16_t, GEMM_A_PLAIN>(stream, g);
}

int64_t gemm_glds_blocks(const GemmArgs &g) {
    int64_t b = 0;
    if (choose_cfg(g, &b) < 0) return 0;
    return b * choose_splitk(g, b);
}
