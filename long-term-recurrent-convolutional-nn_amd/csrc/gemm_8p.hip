// gemm_8p.hip -- phase-interleaved bf16 NT contraction / implicit-GEMM 3x3 convolution for gfx950 (the fast path).
//
// Same contract as gemm_glds.hip (GemmArgs, NT operands, im2col formed by the LDS-DMA source addresses, window-major conv
// rows, XOR-swizzled 128-byte LDS rows), re-scheduled the way a CDNA4 CU wants to be fed (cdna guide 5, "8-phase"):
//   * 256-row tiles, 8 waves (two per SIMD), v_mfma_f32_16x16x32_bf16 (holds a higher clock than 32x32x16 on random data);
//   * a K-tile (64 deep) is processed in 2 super-phases, two quadrants of the wave's output tile each.  A super-phase is
//       [ds_read the new fragments | issue LDS-DMA prefetch of the next K-tile | counted vmcnt]   s_barrier
//       [32 (or 16) MFMAs under s_setprio 1]                                                      s_barrier
//     (the first version ran 4 phases of 16 MFMAs: the per-barrier-pair overhead is roughly constant, halving the pairs
//     is +6-8 % on every convolution layer);
//   * waves 4..7 run ONE barrier behind waves 0..3, so on every SIMD one wave is in its MFMA cluster while its partner
//     reads LDS / issues DMA: the matrix pipe is never idle waiting for fragment reads;
//   * LDS = 2 K-tile buffers x {A half 0, A half 1, B half 0, B half 1}.  A half-tile slot is refilled a whole K-tile
//     after its last read and retired one super-phase after its DMA was issued; counted s_waitcnt vmcnt, never 0 in the loop;
//   * D = B_frag x A_frag ("swapped": lane = pixel, 4 registers = 4 consecutive channels) for plain/conv outputs, so
//     the epilogue moves 8/16 bytes per lane; D = A_frag x B_frag for the fused 2x2 max-pool (4 registers = the 4 pixels
//     of one pool window, SURVEY conv rows / DESIGN.md).
// Hazards (two wave groups one barrier apart): a half-tile is read one super-phase AFTER the one whose first barrier
// follows the wait that retires it (every wave retires its own pieces, the barrier publishes them); a slot is re-staged
// >= 3 barriers after the leading group's last ds_read of it (>= 2 after the trailing group's).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct KStep {
    int tap, koff, kb;
};

// compile-time loop: the ds_read offsets below must be literal immediates of an inline-asm statement
template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// Fragment reads are inline asm on purpose: hipcc orders every ordinary LDS load behind ALL outstanding LDS-DMA with
// s_waitcnt vmcnt(0) (it cannot tell which ring slot a global_load_lds writes), which would drain the prefetch ring at
// every phase.  The ordering that matters is established by hand: counted vmcnt + s_barrier (see the hazard note above).
template <int OFF> __device__ __forceinline__ uint4 lds_read16(unsigned addr) {
    uint4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}

typedef int i32x8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x8v pack8(const uint4 &lo, const uint4 &hi) {
    return i32x8v{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
}
// four f32 -> four OCP e4m3 bytes (round to nearest even, saturating at +-448)
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return w;
}

// F8: A and B are OCP e4m3 bytes, a K-tile is 128 elements (the same 128-byte LDS rows), one
// v_mfma_f32_16x16x128_f8f6f4 replaces two v_mfma_f32_16x16x32_bf16 (same cycles, twice the K), the staged epilogue
// multiplies by g.scale[col] before the bias and writes e4m3.
// max(x, 0) as ONE integer max on the bits (non-negative floats order like ints, every negative float and -0 is a negative int;
// a NaN keeps its payload if positive): fmaxf costs two VALU instructions (it quiets NaNs first), and the staged epilogue applies
// it to 128 accumulators per lane.  lo = 0: ReLU; lo = INT_MIN: identity.
__device__ __forceinline__ float relu_bits(float x, int lo) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > lo ? b : lo);
}
__device__ __forceinline__ float sigm8p(float x) { return 1.0f / (1.0f + expf(-x)); }

// Per-tile operand addressing of one lane: survives from the epilogue of tile t, where the persistent walk may already set it up for
// tile t + 1 and issue that tile's first K-tile (below), into tile t + 1's K loop.
template <int APW, int BPW> struct TileAddr {
    int m0, n0;
    unsigned a_voff[2][APW];  // byte offset of this lane's 16-byte chunk of the piece (OOB: row >= M)
    unsigned a_mask[2][APW];  // conv: bit t CLEAR = tap t reads inside the image
    unsigned b_voff[2][BPW];
};

// One output tile.  `pre`: ta already describes `tile` and its first K-tile's DMA is in flight (issued by the previous call).
// `next_tile` >= 0: the tile this workgroup visits next; where the epilogue leaves ring buffer 0 free early (the bf16 pooled
// epilogue) it sets ta up for that tile and issues its first K-tile BEFORE the global stores of this one, so that the set-up and the
// DMA's latency -- a cold HBM fetch of new image rows -- run under the store issue instead of after it.  Returns whether it did.
template <int WM, int WN, int MT, int NT, int AMODE, bool SWAP, bool F8, int EPI = 0>
__device__ __forceinline__ bool gemm8p_tile(const GemmArgs &g, unsigned char *smem, const int tile, const int ntiles_xy, const int next_tile,
                                            TileAddr<WM * (2 * MT * 16) / 128, WN * (2 * NT * 16) / 128> &ta, const bool pre) {
    constexpr int ES = F8 ? 1 : 2;       // bytes per element
    constexpr int KE = 128 / ES;         // elements per K-tile
    static_assert(WM * WN == 8, "8 waves");
    constexpr int QM = MT * 16, QN = NT * 16;   // quadrant = QM x QN of a wave's (2 QM) x (2 QN) output tile
    constexpr int WTM = 2 * QM, WTN = 2 * QN;
    constexpr int BM = WM * WTM, BN = WN * WTN;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
    constexpr int APW = BM / 128, BPW = BN / 128;  // DMA instructions per wave per half-tile
    static_assert(APW >= 1 && BPW >= 1 && APW * 128 == BM && BPW * 128 == BN, "tile must be a multiple of 128");
    constexpr int CPR = BN / 8;                     // 16-byte chunks per staged C row
    static_assert(CPR >= 16, "epilogue swizzle needs >= 16 chunks per row");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int grp = wave >> 2;  // waves 4..7 run one barrier behind waves 0..3
    // kernel-development stamps (GemmArgs::stamps, LRCN_STAMPS=1 through lrcn_bench_conv): wave 0 / lane 0 writes the shader clock at
    // the segment boundaries of this tile into a buffer nothing else reads; slots 7 / 6 = the 100 MHz wall counter at tile start / end
    auto stamp = [&](int k) {
        if (g.stamps && tid == 0) g.stamps[(size_t)tile * 8 + k] = k >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
    };
    stamp(7);
    stamp(0);
    const int M = g.M, N = g.N;
    const int tiles_n = (N + BN - 1) / BN;

    // Operands are addressed through buffer descriptors (buffer_load_dwordx4 ... lds): the per-lane byte offset of a piece is
    // loop-invariant, the K-step offset rides in the scalar soffset, and a lane whose row / tap falls outside the matrix /
    // image uses offset 0xFFFFFFFF >= num_records, for which the hardware writes zeros to LDS -- no 64-bit address arithmetic
    // and no zero-page pointers in the load segment (measured on gfx950: out-of-range lanes store 0, soffset is not part of
    // the range check).  The 3x3 taps reach (W + 1) pixels BEHIND a pixel, so the convolution's descriptor starts that far
    // before the tensor and soffset carries the difference.
    constexpr unsigned OOB = 0xFFFFFFFFu, NREC = 0xFFFFFF00u;
    const unsigned conv_back = (AMODE == GEMM_A_CONV3) ? (unsigned)((g.W + 1) * g.Cin * ES) : 0u;
    const unsigned char *Abase = reinterpret_cast<const unsigned char *>(g.A) - conv_back;
    const unsigned char *Bbase = reinterpret_cast<const unsigned char *>(g.B);
    auto rsrc_of = [](const unsigned char *base, bool live) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(base), 0, live ? NREC : 0u, 0x00020000);
    };

    const int kpt = (AMODE == GEMM_A_CONV3) ? g.Cin / KE : g.K / KE;
    const int KT_all = (AMODE == GEMM_A_CONV3) ? 9 * kpt : kpt;
    const int kbeg = (int)((int64_t)KT_all * blockIdx.y / gridDim.y);
    const int KT = (int)((int64_t)KT_all * (blockIdx.y + 1) / gridDim.y) - kbeg;

    // ---- staging geometry: half-tile `hf` of A = rows {wr' * WTM + hf * QM + i}; piece = 8 rows x 128 B ----
    auto a_piece_row = [&](int hf, int j) {
        const int hr = (wave * APW + j) * 8;
        return (hr / QM) * WTM + hf * QM + (hr % QM);
    };
    auto b_piece_row = [&](int hf, int j) {
        const int hr = (wave * BPW + j) * 8;
        return (hr / QN) * WTN + hf * QN + (hr % QN);
    };
    auto setup = [&](int t, TileAddr<APW, BPW> &o) {
        int bid = t;
        {   // bijective XCD renumbering: the N-tiles of one im2col panel (consecutive logical ids) share an XCD's L2
            // (workgroup w sits on XCD w % 8 and visits tiles w, w + G, ...: tile % 8 == w % 8 as long as 8 divides G)
            const int nwg = ntiles_xy, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        }
        const int nt = bid % tiles_n, mt = bid / tiles_n;
        o.m0 = mt * BM;
        o.n0 = nt * BN;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int j = 0; j < APW; ++j) {
                const int row = a_piece_row(hf, j) + (lane >> 3);
                const int m = o.m0 + row;
                const int src_chunk = (lane & 7) ^ ((row >> 1) & 7);
                o.a_voff[hf][j] = OOB;
                o.a_mask[hf][j] = ~0u;
                if (m < M) {
                    if (AMODE == GEMM_A_CONV3) {
                        const PixDecode p = decode_pixel_fast(m, g.H, g.W, g.inv_w2, g.inv_h2);
                        o.a_voff[hf][j] = (unsigned)((uint64_t)((p.n * g.H + p.y) * g.W + p.x) * (uint64_t)(g.Cin * ES) + src_chunk * 16);  // < NREC: launch check
                        // bit t = kh * 3 + kw SET = tap t falls outside the image (stored inverted: see stage_a): a border row / column
                        // knocks out one row / column of taps
                        o.a_mask[hf][j] = (p.y == 0 ? 0x007u : 0u) | (p.y == g.H - 1 ? 0x1C0u : 0u) | (p.x == 0 ? 0x049u : 0u) |
                                          (p.x == g.W - 1 ? 0x124u : 0u) | ~0x1FFu;
                    } else {
                        o.a_voff[hf][j] = (unsigned)((uint64_t)m * (uint64_t)(g.lda * ES) + src_chunk * 16);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < BPW; ++j) {
                const int row = b_piece_row(hf, j) + (lane >> 3);
                const int n = o.n0 + row;
                const int src_chunk = (lane & 7) ^ ((row >> 1) & 7);
                o.b_voff[hf][j] = n < N ? (unsigned)((uint64_t)n * (uint64_t)(g.ldb * ES) + src_chunk * 16) : OOB;
            }
        }
    };

    auto kstep_of = [&](int kt) {  // conv: channel slice OUTER, tap INNER (nine consecutive K-tiles re-read one patch: L2 hits)
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
    // PLAIN operands keep flat 64-bit addresses (global_load_lds_dwordx4, zero page for rows past the edge): with no tap
    // masks there was little arithmetic to save and the descriptor form measured 5 % slower at 4096^3.
    const unsigned char *Aflat = reinterpret_cast<const unsigned char *>(g.A);
    const unsigned char *Zp = reinterpret_cast<const unsigned char *>(g.zero_page) + (lane & 7) * 16;
    auto stage_a = [&](int buf, int hf, const KStep &k, bool live) {
        if constexpr (AMODE == GEMM_A_CONV3) {
            const __amdgpu_buffer_rsrc_t rs = rsrc_of(Abase, live);  // a dead K-tile (past the end) loads zeros: keeps the vmcnt bookkeeping uniform
            const int soff = (int)(conv_back + (unsigned)(k.koff * ES));
#pragma unroll
            for (int j = 0; j < APW; ++j) {
                // bit `tap` of the INVERTED mask, sign-extended (v_bfe_i32): 0 for a tap inside the image, all ones (= OOB) outside
                const unsigned vo = ta.a_voff[hf][j] | (unsigned)__builtin_amdgcn_sbfe((int)ta.a_mask[hf][j], k.tap, 1);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + buf * BUF + a_piece_row(hf, j) * 128), 16, (int)vo, soff, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < APW; ++j) {
                const bool ok = live && ta.a_voff[hf][j] != OOB;
                const unsigned char *src = ok ? Aflat + ((int64_t)ta.a_voff[hf][j] + (int64_t)k.koff * ES) : Zp;
                __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(smem + buf * BUF + a_piece_row(hf, j) * 128), 16, 0, 0);
            }
        }
    };
    auto stage_b = [&](int buf, int hf, const KStep &k, bool live) {
        if constexpr (AMODE == GEMM_A_CONV3) {
            const __amdgpu_buffer_rsrc_t rs = rsrc_of(Bbase, live);
            const int soff = k.kb * ES;
#pragma unroll
            for (int j = 0; j < BPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + buf * BUF + A_BYTES + b_piece_row(hf, j) * 128), 16,
                                                         (int)ta.b_voff[hf][j], soff, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < BPW; ++j) {
                const bool ok = live && ta.b_voff[hf][j] != OOB;
                const unsigned char *src = ok ? Bbase + ((int64_t)ta.b_voff[hf][j] + (int64_t)k.kb * ES) : Zp;
                __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(smem + buf * BUF + A_BYTES + b_piece_row(hf, j) * 128), 16, 0, 0);
            }
        }
    };

    // fragment read offsets (16x16x32 operand: lane -> row lane&15, 16-byte K chunk 4s + (lane>>4), swizzled;
    // 16x16x128 fp8 operand: 32 consecutive K bytes per lane = chunks 2(lane>>4) + s)
    unsigned fa[2], fb[2];  // per-lane LDS byte addresses (buffer 0) of the wave's first A / B fragment, K half s
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int kch = F8 ? 2 * (lane >> 4) + s : s * 4 + (lane >> 4);
        const unsigned fo = (lane & 15) * 128 + ((kch ^ (((lane & 15) >> 1) & 7)) << 4);
        const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
        fa[s] = base + (wr * WTM) * 128 + fo;
        fb[s] = base + A_BYTES + (wc * WTN) * 128 + fo;
    }

    auto issue_first = [&]() {  // all of the tile's first K-tile -> ring buffer 0
        const KStep k0 = kstep_of(kbeg);
        stage_a(0, 0, k0, true);
        stage_b(0, 0, k0, true);
        stage_b(0, 1, k0, true);
        stage_a(0, 1, k0, true);
    };
    if (!pre) {
        setup(tile, ta);
        issue_first();
    }
    stamp(1);  // prologue DMA issued
    const int m0 = ta.m0, n0 = ta.n0;  // copies: the epilogue may re-point ta at the next tile

    // bf16 tiles that leave through the LDS-staged epilogue (every convolution, bf16 GEMM outputs): the bias is the accumulators'
    // INITIAL value (a pooled window's four pixels share it, so max commutes) -- its loads hide under the prologue's DMA wait and the
    // epilogue is convert + ReLU + LDS write only (with the adds there, and the bias vectors loaded one by one in front of them,
    // staging took 5.3k cycles of a tile's 18.5k outside the K loop; the pooled form 7.5k)
    const bool staged = !F8 && EPI == 0 && !g.c_f32 && !g.beta && gridDim.y == 1 && (g.ldc % 8) == 0 && ((uintptr_t)g.C & 15) == 0 &&
                        (N % 8) == 0 && (!SWAP || (N % 4) == 0);
    const bool bias_init = staged && g.bias != nullptr;
    f32x4v acc[2][MT][2][NT];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x4v bv = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (bias_init) {
                if constexpr (SWAP) {  // registers 0..3 = four consecutive channels
                    const int col = n0 + wc * WTN + b * QN + n * 16 + (lane >> 4) * 4;
                    if (col < N) bv = *reinterpret_cast<const f32x4v *>(g.bias + col);
                } else {  // registers 0..3 = the four pixels of one window, lane & 15 = channel
                    const int col = n0 + wc * WTN + b * QN + n * 16 + (lane & 15);
                    const float sb = col < N ? g.bias[col] : 0.0f;
                    bv = f32x4v{sb, sb, sb, sb};
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[a][i][b][n] = bv;
        }

    uint4 af[MT][2], bf0[NT][2], bf1[NT][2];
    auto read_a = [&](int buf, auto mhc) {
        constexpr int mh = decltype(mhc)::value;
        const unsigned a0 = fa[0] + buf * BUF, a1 = fa[1] + buf * BUF;
        static_for<0, MT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            af[i][0] = lds_read16<(mh * QM + i * 16) * 128>(a0);
            af[i][1] = lds_read16<(mh * QM + i * 16) * 128>(a1);
        });
    };
    auto read_b = [&](int buf, auto nhc, uint4 (&bf)[NT][2]) {
        constexpr int nh = decltype(nhc)::value;
        const unsigned b0 = fb[0] + buf * BUF, b1 = fb[1] + buf * BUF;
        static_for<0, NT>([&](auto ic) {
            constexpr int n = decltype(ic)::value;
            bf[n][0] = lds_read16<(nh * QN + n * 16) * 128>(b0);
            bf[n][1] = lds_read16<(nh * QN + n * 16) * 128>(b1);
        });
    };
    constexpr std::integral_constant<int, 0> I0{};
    constexpr std::integral_constant<int, 1> I1{};
#define MFMA_BODY(mh, nh, BF)                                                                                              \
    do {                                                                                                                   \
        if constexpr (F8) {                                                                                                \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int n = 0; n < NT; ++n) {                \
                const i32x8v av = pack8(af[i][0], af[i][1]);                                                               \
                const i32x8v bv = pack8(BF[n][0], BF[n][1]);                                                               \
                acc[mh][i][nh][n] = SWAP ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bv, av, acc[mh][i][nh][n], 0, 0, 0, 0, 0, 0) \
                                         : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[mh][i][nh][n], 0, 0, 0, 0, 0, 0); \
            }                                                                                                              \
        } else {                                                                                                           \
            _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int n = 0; \
                                                                                                                   n < NT; ++n) { \
                const bf16x8 av = __builtin_bit_cast(bf16x8, af[i][s]);                                                    \
                const bf16x8 bv = __builtin_bit_cast(bf16x8, BF[n][s]);                                                    \
                acc[mh][i][nh][n] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, av, acc[mh][i][nh][n], 0, 0, 0)     \
                                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[mh][i][nh][n], 0, 0, 0);    \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)
// Two quadrants per barrier pair; the first starts as soon as ITS fragments have arrived (LGKM1 later reads may still fly).
#define MFMA_PAIR(mh1, nh1, BF1, mh2, nh2, BF2, LGKM1)                                                                     \
    do {                                                                                                                   \
        __builtin_amdgcn_s_barrier();                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LGKM1) : "memory");                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                                     \
        MFMA_BODY(mh1, nh1, BF1);                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        wait_lgkm0();                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        MFMA_BODY(mh2, nh2, BF2);                                                                                          \
        __builtin_amdgcn_s_setprio(0);                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        __builtin_amdgcn_s_barrier();                                                                                      \
    } while (0)

    // ---- prologue: the first K-tile (issued above, or by the previous tile's epilogue) ----
    {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        stamp(2);  // ... landed (and the previous tile's stores retired: one in-order counter)
        if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger
    }
    KStep kn = kstep_of(kbeg + (KT > 1 ? 1 : 0));  // tile t+1
    // Two super-phases per K-tile, one barrier pair each:
    //   [read A0, B0, B1 | stage A0, B0, B1 of tile t+1 | retire A1 of tile t]   32 MFMAs: quadrants (0,0), (0,1)
    //   [read A1         | stage A1 of tile t+1         | retire A0, B0, B1 of t+1]   32 MFMAs: quadrants (1,1), (1,0)
    // (four barrier pairs of 16 MFMAs per K-tile ran the matrix pipe 59 % of a resident workgroup's cycles: the barrier /
    // wait overhead per pair is about constant, so two pairs of 32 are +6-8 % on every conv layer.)
    // the K loop is unrolled by two so that the ring buffer of a K-tile is a compile-time constant (LDS addresses of the
    // fragment reads and DMA destinations need no per-tile VALU arithmetic in the load segment)
    auto k_tile = [&](int t, auto curc) {
        constexpr int cur = decltype(curc)::value, nxt = cur ^ 1;
        const bool live1 = t + 1 < KT;
        read_a(cur, I0);
        read_b(cur, I0, bf0);
        read_b(cur, I1, bf1);
        stage_a(nxt, 0, kn, live1);
        stage_b(nxt, 0, kn, live1);
        stage_b(nxt, 1, kn, live1);
        wait_vmcnt<APW + 2 * BPW>();  // A1 of this tile (read in the second super-phase)
        MFMA_PAIR(0, 0, bf0, 0, 1, bf1, 2 * NT);
        read_a(cur, I1);
        stage_a(nxt, 1, kn, live1);
        wait_vmcnt<APW>();            // A0, B0, B1 of tile t+1
        MFMA_PAIR(1, 1, bf1, 1, 0, bf0, 0);
        kn = kstep_of(kbeg + (t + 2 < KT ? t + 2 : 0));
    };
    for (int t = 0; t < KT; t += 2) {
        k_tile(t, I0);
        if (t + 1 < KT) k_tile(t + 1, I1);
    }
#undef MFMA_PAIR
#undef MFMA_BODY
    if (grp == 0) __builtin_amdgcn_s_barrier();  // un-stagger
    wait_vmcnt<0>();                             // the tail's dummy pieces
    __builtin_amdgcn_s_barrier();                // nobody reads or DMA-writes the ring any more
    stamp(3);  // main loop done

    const int l15 = lane & 15, lq = lane >> 4;
    // ---------------------------------------------------------------- LSTM epilogues (gemm.h LstmEpi): the cell math in the accumulators.
    // SWAP layout: a lane holds row (l15) x four consecutive columns.  FWD: those are the four gates of ONE unit (interleaved
    // weight rows); BWD: four consecutive units.
    if constexpr (EPI == GEMM_OUT_LSTM_FWD || EPI == GEMM_OUT_LSTM_BWD) {
        // 1. the f32 accumulator tile goes through LDS (the ring is free; 16-byte chunk c of row r at chunk c ^ (r & 31): the 16 lanes
        //    of a group write 16 different rows), so that 2. every global access of the cell math is coalesced along a row
        //    (a first version did the math straight from the MFMA layout -- 16 rows per memory instruction -- and ran 3-10x slower).
        constexpr int CPRF = BN / 4;  // 16-byte chunks per staged f32 row
        static_assert(CPRF == 32, "LSTM epilogues are built for the 256 x 128 tile");
        const LstmEpi &e = g.lstm;
        const int H = e.H;
        typedef __bf16 bf16x4e __attribute__((ext_vector_type(4)));
        // FWD (round 6): a thread owns FOUR CONSECUTIVE UNITS of a row -- chunks 4q .. 4q+3 of the staged row -- in 4 rows (r64 + 64 j), so
        // that every global access is a 16-byte (f32) / 8-byte (bf16) vector whatever the gate layout of Gx / acts ([f | i | o | g] blocks:
        // four units of one gate are contiguous), and everything the cell math READS (c_prev, Gx or the bias row) is requested BEFORE the
        // accumulators are staged: the loads fly under the LDS round trip instead of sixteen dependent 4-byte round trips per thread
        // (round 5's form: one unit per thread and iteration, 5 scalar loads + 6 scalar stores each -- 45 us fused against 27 + 9.6 us
        // as two launches at 256 rows, and 175 us per decode-step layer at 5120 hypotheses).
        const int q8 = tid & 7, r64 = tid >> 3;
        const int u0 = (n0 >> 2) + 4 * q8;          // first of the thread's four hidden units (H % 4 == 0: a quad is inside or outside as a whole)
        const bool uok = u0 < H;
        [[maybe_unused]] f32x4v cpv[4], gxv[4][4];
        if constexpr (EPI == GEMM_OUT_LSTM_FWD) {
            const f32x4v z4 = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (e.gx_bcast) {   // one row [4H] for every row of the tile (the bias of a decode step)
#pragma unroll
                for (int gt = 0; gt < 4; ++gt) gxv[0][gt] = uok ? *reinterpret_cast<const f32x4v *>(e.Gx + (int64_t)gt * H + u0) : z4;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = m0 + r64 + 64 * j;
                const bool ok = uok && row < M;
                const int64_t grow = (ok && e.gx_idx) ? e.gx_idx[row] : row;   // Gx as a table (per token / per image), or one row per GEMM row
                if (ok && e.c_prev) {
                    const int64_t prow = e.c_prev_idx ? e.c_prev_idx[row] : row;
                    cpv[j] = *reinterpret_cast<const f32x4v *>(e.c_prev + prow * H + u0);
                } else {
                    cpv[j] = z4;
                }
#pragma unroll
                for (int gt = 0; gt < 4; ++gt) {   // (statically indexed copies: a run-time row index would put the array in scratch)
                    if (!e.gx_bcast) gxv[j][gt] = ok ? *reinterpret_cast<const f32x4v *>(e.Gx + grow * 4 * H + (int64_t)gt * H + u0) : z4;
                    else if (j > 0) gxv[j][gt] = gxv[0][gt];
                }
            }
        }
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int lrow = wr * WTM + mh * QM + i * 16 + l15;
                        const int ch = (wc * WTN + nh * QN + n * 16 + 4 * lq) >> 2;
                        *reinterpret_cast<f32x4v *>(smem + lrow * (BN * 4) + ((ch ^ (lrow & 31)) << 4)) = acc[mh][i][nh][n];
                    }
        __syncthreads();
        if constexpr (EPI == GEMM_OUT_LSTM_FWD) {  // staged columns 4u .. 4u+3 = the gates f, i, o, g of unit u (interleaved weight rows)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lrow = r64 + 64 * j, row = m0 + lrow;
                if (!uok || row >= M) continue;
                f32x4v fv, iv, ov, gv, cn, hv;
#pragma unroll
                for (int k = 0; k < 4; ++k) {   // unit u0 + k
                    const f32x4v a = *reinterpret_cast<const f32x4v *>(smem + lrow * (BN * 4) + (((4 * q8 + k) ^ (lrow & 31)) << 4));
                    const float f = sigm8p(a[0] + gxv[j][0][k]), in = sigm8p(a[1] + gxv[j][1][k]), o = sigm8p(a[2] + gxv[j][2][k]),
                                chg = tanhf(a[3] + gxv[j][3][k]);
                    const float c = cpv[j][k] * f + in * chg;
                    fv[k] = f; iv[k] = in; ov[k] = o; gv[k] = chg; cn[k] = c;
                    hv[k] = o * tanhf(c);
                }
                if (e.acts) {
                    bf16_t *ac = reinterpret_cast<bf16_t *>(e.acts) + (int64_t)row * e.ld_a + u0;
                    bf16x4e t;
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] = (bf16_t)fv[k];
                    *reinterpret_cast<bf16x4e *>(ac) = t;
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] = (bf16_t)iv[k];
                    *reinterpret_cast<bf16x4e *>(ac + H) = t;
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] = (bf16_t)ov[k];
                    *reinterpret_cast<bf16x4e *>(ac + 2 * H) = t;
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[k] = (bf16_t)gv[k];
                    *reinterpret_cast<bf16x4e *>(ac + 3 * H) = t;
                }
                *reinterpret_cast<f32x4v *>(e.c_out + (int64_t)row * H + u0) = cn;
                bf16x4e hb;
#pragma unroll
                for (int k = 0; k < 4; ++k) hb[k] = (bf16_t)hv[k];
                *reinterpret_cast<bf16x4e *>(reinterpret_cast<bf16_t *>(e.h_new) + (int64_t)row * e.ld_h + u0) = hb;
                if (e.h_f32) *reinterpret_cast<f32x4v *>(e.h_f32 + (int64_t)row * H + u0) = hv;
            }
            return false;
        }
        for (int idx = tid; idx < BM * CPRF; idx += 512) {
            const int lrow = idx / CPRF, ch = idx - lrow * CPRF;
            const int row = m0 + lrow, col0 = n0 + 4 * ch;
            if (row >= M || col0 >= N) continue;
            const f32x4v a = *reinterpret_cast<const f32x4v *>(smem + lrow * (BN * 4) + ((ch ^ (lrow & 31)) << 4));
            {  // BWD: columns = four consecutive hidden units (H % 4 == 0 is checked at launch)
                const int u = col0;
                const int64_t o = (int64_t)row * H + u;
                const f32x4v dhe = *reinterpret_cast<const f32x4v *>(e.dh_ext + o), cn = *reinterpret_cast<const f32x4v *>(e.c_new + o),
                             dci = *reinterpret_cast<const f32x4v *>(e.dc + o);
                f32x4v cp = f32x4v{0.f, 0.f, 0.f, 0.f};
                if (e.c_prev) cp = *reinterpret_cast<const f32x4v *>(e.c_prev + o);
                const bf16_t *ac = reinterpret_cast<const bf16_t *>(e.acts) + (int64_t)row * e.ld_a + u;
                const bf16x4e fv = *reinterpret_cast<const bf16x4e *>(ac), iv = *reinterpret_cast<const bf16x4e *>(ac + H),
                              ov = *reinterpret_cast<const bf16x4e *>(ac + 2 * H), gv = *reinterpret_cast<const bf16x4e *>(ac + 3 * H);
                bf16x4e zf, zi, zo, zg;
                f32x4v dco;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dh = a[r] + dhe[r];
                    const float f = (float)fv[r], in = (float)iv[r], og = (float)ov[r], gg = (float)gv[r];
                    const float tc = tanhf(cn[r]);
                    const float dov = dh * tc;
                    const float dcv = dci[r] + dh * og * (1.0f - tc * tc);
                    zf[r] = (bf16_t)(dcv * cp[r] * f * (1.0f - f));
                    zi[r] = (bf16_t)(dcv * gg * in * (1.0f - in));
                    zo[r] = (bf16_t)(dov * og * (1.0f - og));
                    zg[r] = (bf16_t)(dcv * in * (1.0f - gg * gg));
                    dco[r] = dcv * f;
                }
                bf16_t *z = reinterpret_cast<bf16_t *>(e.dz_out) + (int64_t)row * e.ld_a + u;
                *reinterpret_cast<bf16x4e *>(z) = zf;
                *reinterpret_cast<bf16x4e *>(z + H) = zi;
                *reinterpret_cast<bf16x4e *>(z + 2 * H) = zo;
                *reinterpret_cast<bf16x4e *>(z + 3 * H) = zg;
                *reinterpret_cast<f32x4v *>(e.dc + o) = dco;
            }
        }
        return false;
    }
    // ---------------------------------------------------------------- softmax / top-K partials (gemm.h SmaxEpi): the logits stay on chip
    if constexpr (EPI == GEMM_OUT_SMAX_TOPK) {
        static_assert(BM == 256 && BN == 256 && SWAP && !F8, "built for the 256 x 256 tile");
        // Two phases, one per 128-column half h of the tile: the waves that own those columns (wc >> 1 == h) stage accumulator + bias as f32
        // (256 rows x 512 bytes = the whole ring; 16-byte chunk c of row r at chunk c ^ (r & 31)), columns past N as -inf; then thread t scans
        // 64 of them for row t >> 1 (part t & 1: columns [0, 32) + [64, 96) resp. [32, 64) + [96, 128) of the half) and carries {max, sum exp,
        // SMAX_KC best} over both phases: 128 columns per record.  Pass A of a phase finds the 64 values' maximum, pass B (a second LDS read: cheaper than 64
        // live registers beside the other half's accumulators) adds exp(x - max) and offers every value that beats the list's last entry.
        // List order: value descending, equal values by ascending column (columns are scanned ascending and only a strictly larger value
        // moves in front of an entry).
        const SmaxEpi &e = g.smax;
        const int srow = tid >> 1, part = tid & 1;
        float m_run = -INFINITY, s_run = 0.0f;
        float tv[SMAX_KC];
        int ti[SMAX_KC];
#pragma unroll
        for (int k = 0; k < SMAX_KC; ++k) {
            tv[k] = -INFINITY;
            ti[k] = 0x7FFFFFFF;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if ((wc >> 1) == h) {
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int lcol = (wc & 1) * WTN + nh * QN + n * 16 + 4 * lq;   // column inside the half
                        const int col = n0 + h * 128 + lcol;
                        f32x4v bias = f32x4v{0.f, 0.f, 0.f, 0.f};
                        const bool cok = col < N;                                    // N % 4 == 0 (checked at launch): a quad is in or out as a whole
                        if (g.bias && cok) bias = *reinterpret_cast<const f32x4v *>(g.bias + col);
#pragma unroll
                        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                            for (int i = 0; i < MT; ++i) {
                                const int lrow = wr * WTM + mh * QM + i * 16 + l15;
                                f32x4v a = acc[mh][i][nh][n] + bias;
                                if (!cok) a = f32x4v{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                                *reinterpret_cast<f32x4v *>(smem + lrow * 512 + (((lcol >> 2) ^ (lrow & 31)) << 4)) = a;
                            }
                    }
            }
            __syncthreads();
            const unsigned char *rowp = smem + srow * 512;
            const int sw = srow & 31;
            // the thread's 16 chunks of this half: chunk id cid(c) = (c & 7) | (part << 3) | ((c >> 3) << 4), i.e. columns [0, 32) + [64, 96) of the
            // half for part 0 and [32, 64) + [96, 128) for part 1, ascending in c.  `part` sits in bit 3 of the chunk id because a 16-lane group
            // of the LDS read holds 8 rows x 2 parts and the rows' swizzle varies bits 0..2: bit 3 keeps the two parts on different banks
            // (with part in bit 4 -- contiguous 64-column parts -- every read was a two-way conflict).
            auto cid = [&](int c) { return (c & 7) | (part << 3) | ((c >> 3) << 4); };
            float cm = -INFINITY;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4v x = *reinterpret_cast<const f32x4v *>(rowp + ((cid(c) ^ sw) << 4));
                cm = fmaxf(cm, fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
            }
            if (cm > m_run) {   // (cm = -inf -- every column of this part past N -- never enters)
                s_run *= __expf(m_run - cm);   // exp(-inf) = 0 the first time
                m_run = cm;
            }
            if (cm != -INFINITY) {
                float s4[4] = {0.f, 0.f, 0.f, 0.f};   // four independent partial sums: no 128-long dependent chain of adds
#pragma unroll 4
                for (int c = 0; c < 16; ++c) {
                    const f32x4v x = *reinterpret_cast<const f32x4v *>(rowp + ((cid(c) ^ sw) << 4));
#pragma unroll
                    for (int k = 0; k < 4; ++k) s4[k] += __expf(x[k] - m_run);
                    if (fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])) > tv[SMAX_KC - 1]) {   // one test per chunk: most chunks offer nothing
                        const int cbase = n0 + h * 128 + 4 * cid(c);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float xv = x[k];
                            if (xv > tv[SMAX_KC - 1]) {
                                float cv = xv;
                                int ci = cbase + k;
                                bool ins = false;
#pragma unroll
                                for (int j = 0; j < SMAX_KC; ++j) {
                                    ins = ins || cv > tv[j];
                                    if (ins) {
                                        const float ov = tv[j];
                                        const int oi = ti[j];
                                        tv[j] = cv; ti[j] = ci;
                                        cv = ov; ci = oi;
                                    }
                                }
                            }
                        }
                    }
                }
                s_run += (s4[0] + s4[1]) + (s4[2] + s4[3]);
            }
            __syncthreads();   // the ring is overwritten by the next phase / the next tile's DMA
        }
        const int row = m0 + srow;
        if (row < M) {
            float *rec = e.part + ((int64_t)row * e.nrec + (2 * (n0 >> 8) + part)) * SMAX_REC;
            *reinterpret_cast<f32x4v *>(rec) = f32x4v{m_run, s_run, tv[0], tv[1]};
            *reinterpret_cast<f32x4v *>(rec + 4) = f32x4v{tv[2], tv[3], tv[4], tv[5]};
            *reinterpret_cast<f32x4v *>(rec + 8) = f32x4v{__int_as_float(ti[0]), __int_as_float(ti[1]), __int_as_float(ti[2]), __int_as_float(ti[3])};
            *reinterpret_cast<f32x4v *>(rec + 12) = f32x4v{__int_as_float(ti[4]), __int_as_float(ti[5]), 0.0f, 0.0f};
        }
        return false;
    }
    // ---------------------------------------------------------------- epilogue A: bf16 tile staged through LDS
    if constexpr (F8) {
        // e4m3 tile staged through LDS: BN bytes per row, 16-byte chunk c of row r at chunk c ^ swz(r) (two 128-byte rows
        // share a 256-byte bank line when BN = 128).  Launch-side checks guarantee N % 16 == 0, ldc % 16 == 0, no beta/f32.
        constexpr int CSTR = BN, CPR8 = BN / 16, RPL = 256 / BN > 1 ? 256 / BN : 1;
        auto swz = [](int r) { return (r / RPL) & (CPR8 - 1); };
        if (!SWAP) {
            float scp[2][NT], biasp[2][NT];  // all of the lane's scales / biases first: one L2 round trip instead of 2 x 2 NT dependent ones
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int col = n0 + wc * WTN + nh * QN + n * 16 + l15;
                    scp[nh][n] = (g.scale && col < N) ? g.scale[col] : 1.0f;
                    biasp[nh][n] = (g.bias && col < N) ? g.bias[col] : 0.0f;
                }
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const int lcol = wc * WTN + nh * QN + n * 16 + l15;
                            const float sc = scp[nh][n];
                            const float bias = biasp[nh][n];
                            const f32x4v a = acc[mh][i][nh][n];
                            float v = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])) * sc + bias;  // sc > 0: max commutes
                            if (g.relu) v = fmaxf(v, 0.0f);
                            const int prow = (wr * WTM + mh * QM + i * 16) / 4 + lq;
                            const int pos = ((lcol >> 4) ^ swz(prow)) * 16 + (lcol & 15);
                            smem[prow * CSTR + pos] = (unsigned char)(pack_fp8x4(v, v, 0.f, 0.f) & 0xFF);
                        }
        } else {
            f32x4v biasv[2][NT], scv[2][NT];
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int col = n0 + wc * WTN + nh * QN + n * 16 + lq * 4;
                    biasv[nh][n] = f32x4v{0.f, 0.f, 0.f, 0.f};
                    scv[nh][n] = f32x4v{1.f, 1.f, 1.f, 1.f};
                    if (g.bias && col < N) biasv[nh][n] = *reinterpret_cast<const f32x4v *>(g.bias + col);
                    if (g.scale && col < N) scv[nh][n] = *reinterpret_cast<const f32x4v *>(g.scale + col);
                }
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int lcol = wc * WTN + nh * QN + n * 16 + lq * 4;
                    const f32x4v bias = biasv[nh][n], sc = scv[nh][n];
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                        for (int i = 0; i < MT; ++i) {
                            f32x4v a = acc[mh][i][nh][n] * sc + bias;
                            if (g.relu) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.0f);
                            }
                            const int lrow = wr * WTM + mh * QM + i * 16 + l15;
                            const int pos = ((lcol >> 4) ^ swz(lrow)) * 16 + (lcol & 15);
                            *reinterpret_cast<unsigned *>(smem + lrow * CSTR + pos) = pack_fp8x4(a[0], a[1], a[2], a[3]);
                        }
                }
        }
        __syncthreads();
        const int rows_out = SWAP ? BM : BM / 4;
        unsigned char *C8 = reinterpret_cast<unsigned char *>(g.C);
        for (int idx = tid; idx < rows_out * CPR8; idx += 512) {
            const int lrow = idx / CPR8, ch = idx - lrow * CPR8;
            const int col = n0 + ch * 16;
            if (col >= N) continue;
            int64_t off;
            if (!SWAP) {
                const int prow = (m0 >> 2) + lrow;
                if (prow >= (M >> 2)) continue;
                off = (int64_t)prow * g.ldc + col;
            } else {
                const int row = m0 + lrow;
                if (row >= M) continue;
                if (g.out_mode == GEMM_OUT_CONV) {
                    const PixDecode p = decode_pixel_fast(row, g.H, g.W, g.inv_w2, g.inv_h2);
                    off = (((int64_t)p.n * g.H + p.y) * g.W + p.x) * g.ldc + col;
                } else {
                    off = (int64_t)row * g.ldc + col;
                }
            }
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(smem + lrow * CSTR + ((ch ^ swz(lrow)) << 4)),
                                        reinterpret_cast<u32x4 *>(C8 + off));
        }
        return false;
    }
    if (staged) {
        constexpr int CSTR = BN * 2;  // bytes per staged row; 16-byte chunk c of row r lives at chunk c ^ (r & 15)
        const int relu_lo = g.relu ? 0 : (int)0x80000000;  // relu_bits: branch-free ReLU (or identity)
        // The pooled tile (BM / 4 rows) is staged in ring buffer 1's region: buffer 0 is then free for the next tile's first K-tile
        // while this tile's rows are still being stored.  The full tile needs both buffers (it IS the ring's size).
        unsigned char *const cst = smem + (SWAP ? 0 : BUF);
        static_assert(SWAP || (BM / 4) * BN * 2 <= BUF, "pooled tile fits one ring buffer");
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        const s16x2 relu_lo2 = g.relu ? s16x2{0, 0} : s16x2{(short)0x8000, (short)0x8000};  // ReLU on the PACKED bf16 pair: one integer max
        if (!SWAP) {  // fused 2x2 max-pool: registers 0..3 = the four pixels of one window (bias already inside), lane&15 = channel
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const int lcol = wc * WTN + nh * QN + n * 16 + l15;
                            const f32x4v a = acc[mh][i][nh][n];
                            const float v = relu_bits(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), relu_lo);
                            const int prow = (wr * WTM + mh * QM + i * 16) / 4 + lq;
                            const int pos = ((lcol >> 3) ^ (prow & 15)) * 16 + (lcol & 7) * 2;
                            *reinterpret_cast<bf16_t *>(cst + prow * CSTR + pos) = (bf16_t)v;
                        }
        } else {  // lane&15 = row (pixel), registers 0..3 = four consecutive channels (bias already inside)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int lcol = wc * WTN + nh * QN + n * 16 + lq * 4;
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                        for (int i = 0; i < MT; ++i) {
                            const f32x4v a = acc[mh][i][nh][n];
                            typedef short s16x4 __attribute__((ext_vector_type(4)));
                            // ONE vector conversion = two v_cvt_pk_bf16_f32.  Written element by element hipcc scalarises it into four
                            // single conversions plus two v_perm_b32 (192 instead of 64 vector-ALU instructions per wave and tile); the
                            // empty asm keeps the packed pair opaque so that the integer max below cannot pull it apart again.
                            typedef __bf16 bf16x4c __attribute__((ext_vector_type(4)));
                            uint2 pk = __builtin_bit_cast(uint2, __builtin_convertvector(a, bf16x4c));
                            asm volatile("" : "+v"(pk.x), "+v"(pk.y));
                            const s16x2 r0 = __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk.x), relu_lo2),
                                        r1 = __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk.y), relu_lo2);
                            const int lrow = wr * WTM + mh * QM + i * 16 + l15;
                            const int pos = ((lcol >> 3) ^ (lrow & 15)) * 16 + (lcol & 7) * 2;
                            *reinterpret_cast<s16x4 *>(cst + lrow * CSTR + pos) = s16x4{r0[0], r0[1], r1[0], r1[1]};
                        }
                }
        }
        __syncthreads();
        stamp(4);  // accumulators staged
        constexpr int rows_out = SWAP ? BM : BM / 4;
        bf16_t *Cb = reinterpret_cast<bf16_t *>(g.C);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        constexpr int ITER = rows_out * CPR / 512;  // 16-byte chunks per thread
        static_assert(ITER * 512 == rows_out * CPR, "whole passes of the workgroup");
        // The thread's chunks leave LDS in batches of four, each batch's stores after its reads (read-wait-store per chunk was ITER
        // dependent LDS round trips in front of the store issue).  The POOLED tile sits outside ring buffer 0, so with a next tile to
        // prepare its addresses are made and its first K-tile's DMA issued before this tile's stores: per tile, cycles issue + wait +
        // stores (`tools/tile_stamps.py` under TILE_STAMPS_CAP=224) conv2_2 7528 -> 6536, conv3_3 6540 -> 5368.  For the FULL tile
        // (which is the ring's size) the same was tried -- all 16 chunks into registers, a barrier, set-up + DMA, then the stores --
        // and LOST 320-800 cycles per tile: the next tile's wait is the retirement of these stores (one in-order vmcnt; edge tiles
        // skip stores, so a counted wait cannot exempt them), not the DMA's latency.
        const bool early = !SWAP && next_tile >= 0;  // workgroup-uniform
        int tid_e = tid;  // opaque copy: otherwise hipcc hoists every iteration's index arithmetic out of the persistent TILE loop and
        asm volatile("" : "+v"(tid_e));  // keeps ~26 registers alive across the K loop (spilled at kernel start, reloaded here per tile)
        auto store_chunk = [&](int k, const u32x4 &v) {
            const int idx = tid_e + 512 * k;
            const int lrow = idx / CPR, ch = idx - lrow * CPR;
            const int col = n0 + ch * 8;
            if (col >= N) return;
            int64_t off;
            if (!SWAP) {
                const int prow = (m0 >> 2) + lrow;
                if (prow >= (M >> 2)) return;
                off = (int64_t)prow * g.ldc + col;
            } else {
                const int row = m0 + lrow;
                if (row >= M) return;
                if (g.out_mode == GEMM_OUT_CONV) {
                    const PixDecode p = decode_pixel_fast(row, g.H, g.W, g.inv_w2, g.inv_h2);
                    off = (((int64_t)p.n * g.H + p.y) * g.W + p.x) * g.ldc + col;
                } else {
                    off = (int64_t)row * g.ldc + col;
                }
            }
            // non-temporal: the tile is not re-read by this kernel, keep the im2col panels and weights in L2 (+2 % on the VGG stack)
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(Cb + off));
        };
        auto read_chunk = [&](int k) {
            const int idx = tid_e + 512 * k, lrow = idx / CPR, ch = idx - lrow * CPR;
            return *reinterpret_cast<const u32x4 *>(cst + lrow * CSTR + ((ch ^ (lrow & 15)) << 4));
        };
        if (early) {
            setup(next_tile, ta);
            issue_first();
        }
        constexpr int HB = ITER > 4 ? 4 : ITER;
        static_assert(ITER % HB == 0, "whole batches");
        for (int kb = 0; kb < ITER; kb += HB) {
            u32x4 chunk[HB];
#pragma unroll
            for (int k = 0; k < HB; ++k) chunk[k] = read_chunk(kb + k);
#pragma unroll
            for (int k = 0; k < HB; ++k) store_chunk(kb + k, chunk[k]);
        }
        stamp(5);  // stores issued
        stamp(6);  // ... and the 100 MHz wall counter again: (stamp 5 - stamp 0) / (stamp 6 - stamp 7) = shader cycles per 10 ns
        return early;
    }

    // ---------------------------------------------------------------- split-K: this slice's partial tile -> its f32 slab
    if (gridDim.y > 1) {
        if constexpr (SWAP) {
            float *slab = reinterpret_cast<float *>(g.ws) + (size_t)blockIdx.y * M * N;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const int row = m0 + wr * WTM + mh * QM + i * 16 + l15, col0 = n0 + wc * WTN + nh * QN + n * 16 + 4 * lq;
                            if (row < M && col0 < N) *reinterpret_cast<f32x4v *>(slab + (size_t)row * N + col0) = acc[mh][i][nh][n];
                        }
        }
        return false;
    }

    // ---------------------------------------------------------------- epilogue B: direct stores (f32 / accumulate / odd N)
    // A persistent walk (the LSTM GEMMs beside the capped convolution grids: f32 outputs, K = 1000 -> 16 K-tiles per tile, so the per-tile
    // prologue and epilogue are a third of a tile's time) prepares the NEXT tile here: the direct epilogue does not touch LDS, so the ring is
    // free, and the next tile's addresses and first K-tile's DMA run under the issue of this tile's 64 store instructions per lane instead
    // of after them.  LRCN_DBG=16 turns it off (A/B).
    bool early_b = false;
    if (next_tile >= 0 && gridDim.y == 1 && !(g.dbg & 16)) {  // workgroup-uniform
        setup(next_tile, ta);
        issue_first();
        early_b = true;
    }
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const f32x4v a = acc[mh][i][nh][n];
                    const int rb = m0 + wr * WTM + mh * QM + i * 16, cb = n0 + wc * WTN + nh * QN + n * 16;
                    if (!SWAP) {  // pool: rows rb + 4 lq .. +3 are one window, col cb + l15
                        const int col = cb + l15, row = rb + 4 * lq;
                        if (col >= N || row >= M) continue;
                        float v = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])) + (g.bias ? g.bias[col] : 0.0f);
                        if (g.relu) v = fmaxf(v, 0.0f);
                        const int64_t off = (int64_t)(row >> 2) * g.ldc + col;
                        if (g.c_f32)
                            reinterpret_cast<float *>(g.C)[off] = v;
                        else
                            reinterpret_cast<bf16_t *>(g.C)[off] = (bf16_t)v;
                        continue;
                    }
                    const int row = rb + l15, col0 = cb + 4 * lq;
                    if (row >= M || col0 >= N) continue;
                    int64_t off0;
                    if (g.out_mode == GEMM_OUT_CONV) {
                        const PixDecode p = decode_pixel_fast(row, g.H, g.W, g.inv_w2, g.inv_h2);
                        off0 = (((int64_t)p.n * g.H + p.y) * g.W + p.x) * g.ldc + col0;
                    } else {
                        off0 = (int64_t)row * g.ldc + col0;
                    }
                    const bool vec = g.c_f32 && col0 + 3 < N && ((off0 & 3) == 0) && (((uintptr_t)g.C & 15) == 0);
                    if (vec) {
                        f32x4v v = a;
                        if (g.bias) v += *reinterpret_cast<const f32x4v *>(g.bias + col0);  // bias + col0: 16-byte aligned when col0 % 4 == 0
                        float *c = reinterpret_cast<float *>(g.C) + off0;
                        if (g.beta) v += *reinterpret_cast<const f32x4v *>(c);
                        if (g.relu) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                        }
                        *reinterpret_cast<f32x4v *>(c) = v;
                        continue;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (col0 + r >= N) continue;
                        float v = a[r] + (g.bias ? g.bias[col0 + r] : 0.0f);
                        if (g.c_f32) {
                            float *c = reinterpret_cast<float *>(g.C) + off0 + r;
                            if (g.beta) v += *c;
                            if (g.relu) v = fmaxf(v, 0.0f);
                            *c = v;
                        } else {
                            bf16_t *c = reinterpret_cast<bf16_t *>(g.C) + off0 + r;
                            if (g.beta) v += (float)*c;
                            if (g.relu) v = fmaxf(v, 0.0f);
                            *c = (bf16_t)v;
                        }
                    }
                }
    return early_b;
}

// One workgroup per output tile, or -- g.wg_cap > 0 -- a capped, persistent grid that walks the tiles: the data-parallel
// step caps the convolution grids below the CU count at small per-GPU batches so that the LSTM stream's chain of small
// dependent launches always finds idle CUs (DESIGN.md "Stream structure").
template <int WM, int WN, int MT, int NT, int AMODE, bool SWAP, bool F8, int EPI = 0>
__global__ __launch_bounds__(512) void gemm8p_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    const int ntiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    TileAddr<BM / 128, BN / 128> ta;
    if (g.tile_ctr == nullptr || gridDim.y > 1) {
        bool pre = false;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int next = (!(g.dbg & 8) && tile + (int)gridDim.x < ntiles) ? tile + (int)gridDim.x : -1;  // LRCN_DBG=8: no overlap
            pre = gemm8p_tile<WM, WN, MT, NT, AMODE, SWAP, F8, EPI>(g, smem, tile, ntiles, next, ta, pre);
            // not issued early: the staged output tile / ring are reused by the next tile's first DMA
            if (!pre && tile + (int)gridDim.x < ntiles) __syncthreads();
        }
        return;
    }
    // Dynamic tile scheduling for the capped persistent grids of the two-stream training step.  Queue q = the tiles {8 i + q}
    // (the ids that the XCD renumbering in gemm8p_tile keeps on one XCD's L2); a workgroup pulls from queue blockIdx.x & 7 and,
    // once that is empty, from the others.  The pull for the NEXT tile is issued before the current tile's arithmetic, so its
    // latency is hidden; over-claiming costs at most one tile at the tail.  Every workgroup leaves when all queues are empty.
    // The claimed tile id travels from lane 0 to the other seven waves through a per-workgroup GLOBAL slot (two slots, used
    // alternately; tile_ctr[8 + 2 blockIdx.x + parity]): the 512 x 128 configuration uses all 160 KiB of LDS, there is no
    // byte to spare for a broadcast word.
    auto pull = [&]() -> int {  // one lane
        const int q0 = blockIdx.x & 7;
        for (int k = 0; k < 8; ++k) {
            const int q = (q0 + k) & 7;
            const int nq = (ntiles - q + 7) >> 3;
            if (nq <= 0) continue;
            const int i = atomicAdd(g.tile_ctr + q, 1);
            if (i < nq) return 8 * i + q;
        }
        return -1;
    };
    // Both sides are RELAXED device-scope atomics (they execute at the XCD's L2, which both waves of one CU share): an
    // acquire / release pair here would invalidate L1 and write back L2 once per tile -- measured 25 % slower on the whole stack.
    // Ordering: lane 0's exchange has returned before it reaches the end-of-tile barrier; the readers come after it.
    int *slot = g.tile_ctr + 8 + 2 * blockIdx.x;
    const int lane = threadIdx.x & 63;
    auto publish = [&](int *p) {  // lane 0 of wave 0
        const int old = __hip_atomic_exchange(p, pull(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(old) : "memory");
    };
    auto fetch = [&](int *p) -> int {  // one atomic per wave, broadcast to its lanes
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __builtin_amdgcn_readfirstlane(t);
    };
    int par = 0;
    if (threadIdx.x == 0) publish(slot);
    __syncthreads();
    int tile = fetch(slot);
    while (tile >= 0) {
        par ^= 1;
        if (threadIdx.x == 0) publish(slot + par);  // the NEXT tile: its latency hides under this tile's prologue
        (void)gemm8p_tile<WM, WN, MT, NT, AMODE, SWAP, F8, EPI>(g, smem, tile, ntiles, -1, ta, false);  // next tile unknown until fetched: no early issue
        __syncthreads();  // the staged output tile / ring are free again; lane 0's exchange precedes every wave's fetch
        tile = fetch(slot + par);
    }
}

template <int WM, int WN, int MT, int NT, int AMODE, bool SWAP, bool F8 = false, int EPI = 0> hipError_t launch_one(hipStream_t s, const GemmArgs &g, int splitk) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int ring = 2 * (BM + BN) * 128, ctile = EPI == GEMM_OUT_SMAX_TOPK ? BM * 512 : BM * BN * (EPI ? 4 : 2);  // the LSTM epilogues stage the f32 tile, SMAX a 128-column half
    constexpr int lds = ring > ctile ? ring : ctile;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static LdsAttrMask attr_done{0};
    auto kern = gemm8p_kernel<WM, WN, MT, NT, AMODE, SWAP, F8, EPI>;
    if (hipError_t e = set_max_lds(reinterpret_cast<const void *>(kern), lds, attr_done); e != hipSuccess) return e;
    const int64_t blocks = (int64_t)cdiv(g.M, BM) * cdiv(g.N, BN);
    if (blocks <= 0 || blocks > 0x7FFFFFFF) return hipErrorInvalidValue;
    int64_t grid = blocks;
    if (g.wg_cap >= 8 && grid > g.wg_cap && g.tile_ctr && splitk == 1) {
        grid = g.wg_cap & ~7;  // workgroups pull tiles from the per-XCD queues: no reason to launch fewer than the cap
    } else if (g.wg_cap >= 8 && grid > g.wg_cap) {
        // the fewest workgroups that still finish in ceil(tiles / cap) rounds (conv5 at 256 images: 392 tiles, cap 224 -> two
        // rounds either way, 200 workgroups instead of 224), as a multiple of 8 so that tile -> XCD stays stable
        const int64_t cap = g.wg_cap & ~7, rounds = (blocks + cap - 1) / cap;
        grid = (((blocks + rounds - 1) / rounds) + 7) & ~7ll;
        if (grid > cap) grid = cap;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)splitk), dim3(512), lds, s, g);
    return hipGetLastError();
}

template <int AMODE, bool SWAP, bool F8 = false> hipError_t dispatch(hipStream_t s, const GemmArgs &g, int cfg, int splitk) {
    switch (cfg) {
        case 0: return launch_one<2, 4, 4, 2, AMODE, SWAP, F8>(s, g, splitk);  // 256 x 256
        case 1: return launch_one<4, 2, 2, 2, AMODE, SWAP, F8>(s, g, splitk);  // 256 x 128
        case 2: return launch_one<4, 2, 4, 2, AMODE, SWAP, F8>(s, g, splitk);  // 512 x 128 (all 160 KiB of LDS)
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

// Tile menu of the phase-interleaved path: 256 x 256 when that gives the chip >= 200 workgroups or N > 128, else 256 x 128.
static bool gemm_8p_f8_ok(const GemmArgs &g) {  // e4m3 3x3 convolution with an e4m3 NHWC (optionally 2x2-pooled) output
    if (g.dtype != GEMM_T_F8 || !g.zero_page || g.a_mode != GEMM_A_CONV3 || g.out_mode == GEMM_OUT_PLAIN) return false;
    if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || ((uintptr_t)g.C & 15) || (g.ldb % 16) || (g.ldc % 16) || (g.N % 16)) return false;
    if (g.Cin % 128 || g.K != 9 * g.Cin || (g.H & 1) || (g.W & 1) || g.H <= 0 || g.W <= 0 || g.M % (g.H * g.W)) return false;
    if ((int64_t)g.M * g.Cin >= (1ll << 31) || (int64_t)g.N * g.ldb >= (1ll << 31)) return false;
    if (g.c_f32 || g.beta || (g.out_mode == GEMM_OUT_POOL && (g.M & 3))) return false;
    return true;
}

int gemm_8p_config(const GemmArgs &g, int64_t *blocks) {
    if (g.dtype == GEMM_T_F8) {
        if (!gemm_8p_f8_ok(g)) return -1;
    } else if (g.dtype != GEMM_T_BF16 || !gemm_glds_eligible(g)) {
        return -1;
    }
    {   // 32-bit byte offsets inside the kernel (buffer descriptors / lane offsets): operands must stay below 4 GiB - margin
        const int64_t es = g.dtype == GEMM_T_F8 ? 1 : 2;
        const int64_t a_bytes = (g.a_mode == GEMM_A_CONV3 ? (int64_t)g.M * g.Cin + (int64_t)(g.W + 1) * g.Cin : (int64_t)g.M * g.lda) * es;
        if (a_bytes >= 0xFFFF0000ll || (int64_t)g.N * g.ldb * es >= 0xFFFF0000ll) return -1;
    }
    if ((g.M < 256 && g.dtype != GEMM_T_F8) || g.N < 128) return -1;  // e4m3 has no other kernel: M tails are masked rows
    const int ke = g.dtype == GEMM_T_F8 ? 128 : 64;
    const int kt = (g.a_mode == GEMM_A_CONV3) ? 9 * (g.Cin / ke) : g.K / ke;
    if (kt < 2) return -1;
    const int64_t b0 = (int64_t)cdiv(g.M, 256) * cdiv(g.N, 256), b1 = (int64_t)cdiv(g.M, 256) * cdiv(g.N, 128);
    int cfg = (g.N > 128 && (b0 >= 200 || g.N % 256 == 0 || g.N > 384)) ? 0 : 1;
    if (cfg == 0 && ((b0 < 200 && b1 >= 200) || (b0 < 128 && b1 >= 128))) cfg = 1;  // the narrower tile when only it fills the chip
    if (g.cfg_pref == 2 && g.N >= 128) cfg = 1;
    // beside the capped convolution grids (g.free_cus CUs free): the wider tile when only IT fits them in one round -- the logits GEMM of a
    // 32-row rank (384 x 10640): 84 tiles of 256 x 256 instead of 168 of 256 x 128 on 96 CUs
    if (g.free_cus > 0 && g.N > 128 && b0 <= g.free_cus && b1 > g.free_cus) cfg = 0;
    {
        static const char *fc = getenv("LRCN_8P_CFG");  // kernel-development knob: force the 256 x 256 (0) or 256 x 128 (1) tile
        if (fc && (fc[0] == '0' || fc[0] == '1') && g.N > 128) cfg = fc[0] - '0';
    }
    *blocks = cfg == 0 ? b0 : b1;
    const char *t = getenv("LRCN_8P_TALL");  // kernel-development knob: 0 disables the 512 x 128 tile
    if (cfg == 1 && g.N <= 128 && cdiv(g.M, 512) >= 400 && !(t && t[0] == '0')) {
        cfg = 2;
        *blocks = cdiv(g.M, 512);
    }
    return cfg;
}

// Split-K plan for skinny problems (few 256-row tiles, long K: fc6/fc7, the image-embedding and dH GEMMs): 256 x 128
// tiles, K cut into `splits` slices of >= 8 K-tiles so that tiles * splits ~ one workgroup per CU.  Each slice writes its
// partial tile to an f32 slab with 16-byte stores; splitk_reduce_kernel then sums the slabs and applies bias / accumulate
// / ReLU / output type.  (f32 atomics instead of slabs were 8x slower here: 16 scattered rows per wave instruction.)
// Returns the number of slices (0 = not applicable).
int gemm_8p_splitk(const GemmArgs &g, int64_t *blocks) {
    if (g.dtype != GEMM_T_BF16 || !gemm_glds_eligible(g) || g.M < 256 || g.N < 128 || (g.N % 4)) return 0;
    // plain contractions, and (small image batches: conv5 at 32 images is 50 square tiles) 3x3 convolutions with a bf16 NHWC output,
    // optionally 2x2-pooled: the slabs keep the kernel's window-major row order, splitk_reduce_conv_kernel undoes it
    const bool conv = g.a_mode == GEMM_A_CONV3 && g.out_mode != GEMM_OUT_PLAIN && !g.c_f32 && !g.beta && (g.M & 3) == 0;
    if (!conv && (g.a_mode != GEMM_A_PLAIN || g.out_mode != GEMM_OUT_PLAIN)) return 0;
    if (!g.ws || (g.ldc % 4) || ((uintptr_t)g.C & 15)) return 0;
    const int kt = conv ? 9 * (g.Cin / 64) : g.K / 64;
    const int64_t b1 = (int64_t)cdiv(g.M, 256) * cdiv(g.N, 128);
    // one workgroup per CU of the chip -- or of the CUs the capped convolution grids of the other stream leave free (g.free_cus): at 32 rows
    // per rank the dH / dX GEMMs (16 tiles, K = 10640 / 4000) were cut into 16 / 7 slices = 256 / 112 workgroups, three / two rounds on 96 CUs
    int s = (int)((g.free_cus > 0 ? g.free_cus : 256) / b1);
    // slices of >= 8 K-tiles; >= 12 for convolutions (measured per layer, ms without -> with: conv5 at 32 / 16 / 8 images 0.058 -> 0.045,
    // 0.045 -> 0.032, 0.044 -> 0.030; conv4_2 at 8 images 0.057 -> 0.045; but conv3_1 at 4 images, two slices of 9: 0.023 -> 0.030)
    const int min_slice = conv ? 12 : 8;
    if (s > kt / min_slice) s = kt / min_slice;
    if (s > 16) s = 16;
    while (s >= 2 && (size_t)s * g.M * g.N * sizeof(float) > g.ws_bytes) --s;
    if (s < 2) return 0;
    *blocks = b1 * s;
    return s;
}

namespace {
template <typename T> __global__ void splitk_reduce_kernel(const float *ws, int S, int M, int N, const float *bias, int beta, int relu,
                                                            void *C, int64_t ldc, int c_f32) {
    const int n4 = N >> 2;
    const int64_t total = (int64_t)M * n4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / n4), c4 = (int)(idx - (int64_t)m * n4) * 4;
        f32x4v v = *reinterpret_cast<const f32x4v *>(ws + (size_t)m * N + c4);
        for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4v *>(ws + ((size_t)s * M + m) * N + c4);
        if (bias) v += *reinterpret_cast<const f32x4v *>(bias + c4);
        if (c_f32) {
            float *c = reinterpret_cast<float *>(C) + (int64_t)m * ldc + c4;
            if (beta) v += *reinterpret_cast<const f32x4v *>(c);
            if (relu)
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
            *reinterpret_cast<f32x4v *>(c) = v;
        } else {
            T *c = reinterpret_cast<T *>(C) + (int64_t)m * ldc + c4;
            for (int r = 0; r < 4; ++r) {
                float x = v[r] + (beta ? to_f32(c[r]) : 0.0f);
                if (relu) x = fmaxf(x, 0.0f);
                c[r] = from_f32<T>(x);
            }
        }
    }
}
// The convolution form: slab rows are in the kernel's window-major pixel order (common.h decode_pixel: rows 4 w .. 4 w + 3 = one 2x2
// window).  One thread per (window, 4 channels): bias, ReLU, then either the four pixels to their NHWC places or their maximum to
// the pooled tensor's row w.  bf16 output.
__global__ void splitk_reduce_conv_kernel(const float *ws, int S, int M, int N, const float *bias, int relu, bf16_t *C, int64_t ldc, int H,
                                          int W, int pool) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const int n4 = N >> 2;
    const int64_t total = (int64_t)(M >> 2) * n4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(idx / n4), c4 = (int)(idx - (int64_t)w * n4) * 4;
        const f32x4v b = bias ? *reinterpret_cast<const f32x4v *>(bias + c4) : f32x4v{0.f, 0.f, 0.f, 0.f};
        f32x4v best = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const int m = 4 * w + sub;
            f32x4v v = *reinterpret_cast<const f32x4v *>(ws + (size_t)m * N + c4);
            for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4v *>(ws + ((size_t)s * M + m) * N + c4);
            v += b;
            if (relu)
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
            if (pool) {
                if (sub == 0) best = v;
                else
                    for (int r = 0; r < 4; ++r) best[r] = fmaxf(best[r], v[r]);
            } else {
                const PixDecode p = decode_pixel(m, H, W);
                bf16_t *c = C + (((int64_t)p.n * H + p.y) * W + p.x) * ldc + c4;
                bf16x4 o;
                for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
                *reinterpret_cast<bf16x4 *>(c) = o;
            }
        }
        if (pool) {
            bf16x4 o;
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)best[r];
            *reinterpret_cast<bf16x4 *>(C + (int64_t)w * ldc + c4) = o;
        }
    }
}
}  // namespace

// C = act((beta ? C : 0) + sum of the `splits` f32 slabs [M][N] in g.ws + bias), written in C's type (N, ldc multiples of 4)
hipError_t launch_splitk_reduce(hipStream_t stream, const GemmArgs &g, int splits) {
    if (g.a_mode == GEMM_A_CONV3) {
        const int64_t total = (int64_t)(g.M >> 2) * (g.N >> 2);
        const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        hipLaunchKernelGGL(splitk_reduce_conv_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float *>(g.ws), splits, g.M, g.N,
                           g.bias, g.relu, reinterpret_cast<bf16_t *>(g.C), g.ldc, g.H, g.W, g.out_mode == GEMM_OUT_POOL ? 1 : 0);
        return hipGetLastError();
    }
    const int64_t total = (int64_t)g.M * (g.N >> 2);
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float *>(g.ws), splits, g.M,
                       g.N, g.bias, g.beta, g.relu, g.C, g.ldc, g.c_f32);
    return hipGetLastError();
}

hipError_t launch_gemm_8p(hipStream_t stream, const GemmArgs &g0, int splitk) {
    GemmArgs g = g0;
    const int epi = g0.out_mode;
    if (epi == GEMM_OUT_LSTM_FWD || epi == GEMM_OUT_LSTM_BWD) {
        g.out_mode = GEMM_OUT_PLAIN;  // tile menu and operand checks are those of a plain contraction; the kernel template carries the mode
        g.cfg_pref = 2;
    }
    if (epi == GEMM_OUT_SMAX_TOPK) {  // 256 x 256 tiles, the logits reduced to per-row records in the epilogue: C is never written
        static_assert(SMAX_KC == 6 && SMAX_REC == 16, "record layout of the epilogue's four 16-byte stores");
        g.out_mode = GEMM_OUT_PLAIN;
        g.c_f32 = 1;
        g.C = g.smax.part;   // (operand checks want a non-null C)
        g.ldc = g.N;
        if (g.dtype != GEMM_T_BF16 || g.a_mode != GEMM_A_PLAIN || splitk > 1 || (g.N & 3) || g.M < 256 || g.N < 256 || !g.smax.part ||
            g.smax.nrec != 2 * cdiv(g.N, 256) || ((uintptr_t)g.smax.part & 15) || (g.bias && ((uintptr_t)g.bias & 15)) || !gemm_glds_eligible(g))
            return hipErrorInvalidValue;
        static const char *dbg5 = getenv("LRCN_DBG");
        g.dbg = dbg5 ? atoi(dbg5) : 0;
        g.inv_w2 = g.inv_h2 = 0;
        gemm_debug_note_route(nullptr, 0);
        return launch_one<2, 4, 4, 2, GEMM_A_PLAIN, true, false, GEMM_OUT_SMAX_TOPK>(stream, g, 1);
    }
    static const char *dbg = getenv("LRCN_DBG");  // kernel-development ablation flags (gemm.h)
    g.dbg = dbg ? atoi(dbg) : 0;
    g.inv_w2 = g.inv_h2 = 0;
    if ((g.a_mode == GEMM_A_CONV3 || g.out_mode == GEMM_OUT_CONV || g.out_mode == GEMM_OUT_POOL) && g.W >= 4 && g.H >= 4) {
        const uint64_t wmax = (uint64_t)((g.M + 511) >> 2) + 128, dmax = (uint64_t)(g.W > g.H ? g.W : g.H) >> 1;  // rows past M in the last tile included
        static const char *nofd = getenv("LRCN_FASTDIV");  // LRCN_FASTDIV=0: the dividing decode (tests compare the two)
        if (wmax * dmax < (1ull << 32) && !(nofd && nofd[0] == '0')) {
            g.inv_w2 = fastdiv_inv((unsigned)g.W >> 1);
            g.inv_h2 = fastdiv_inv((unsigned)g.H >> 1);
        }
    }
    int64_t blocks = 0;
    int cfg = gemm_8p_config(g, &blocks);
    if (cfg < 0) return hipErrorInvalidValue;
    gemm_debug_note_route(nullptr, splitk > 1 ? splitk : cfg);
    if (epi == GEMM_OUT_LSTM_FWD || epi == GEMM_OUT_LSTM_BWD) {  // 256 x 128 tiles, cell math in the epilogue
        if (g.dtype != GEMM_T_BF16 || g.a_mode != GEMM_A_PLAIN || splitk > 1 || cfg != 1 || (g.N & 3) ||
            (!g.lstm.acts && epi == GEMM_OUT_LSTM_BWD) || g.lstm.H < 1 || (g.lstm.H & 3) || (g.lstm.ld_a & 3))
            return hipErrorInvalidValue;
        return epi == GEMM_OUT_LSTM_FWD ? launch_one<4, 2, 2, 2, GEMM_A_PLAIN, true, false, GEMM_OUT_LSTM_FWD>(stream, g, 1)
                                               : launch_one<4, 2, 2, 2, GEMM_A_PLAIN, true, false, GEMM_OUT_LSTM_BWD>(stream, g, 1);
    }
    if (splitk > 1) {
        // the caller's slice count must be the planner's, unless it is the "beside the convolutions" route's own choice (g.splitk_forced:
        // the planner counts the whole chip's CUs, that route the free ones; its eligibility was checked by the router)
        if (!g.splitk_forced && gemm_8p_splitk(g, &blocks) != splitk) return hipErrorInvalidValue;
        hipError_t e = g.a_mode == GEMM_A_CONV3 ? dispatch<GEMM_A_CONV3, true>(stream, g, 1, splitk) : dispatch<GEMM_A_PLAIN, true>(stream, g, 1, splitk);
        if (e != hipSuccess) return e;
        if (g.splitk_no_reduce) return hipSuccess;   // the caller's next kernel sums the slabs (lstm_bwd_kernel's dh_b slabs)
        return launch_splitk_reduce(stream, g, splitk);
    } else {
        splitk = 1;
    }
    const bool swap = g.out_mode != GEMM_OUT_POOL;
    if (g.dtype == GEMM_T_F8)
        return swap ? dispatch<GEMM_A_CONV3, true, true>(stream, g, cfg, 1) : dispatch<GEMM_A_CONV3, false, true>(stream, g, cfg, 1);
    if (g.a_mode == GEMM_A_CONV3)
        return swap ? dispatch<GEMM_A_CONV3, true>(stream, g, cfg, splitk) : dispatch<GEMM_A_CONV3, false>(stream, g, cfg, splitk);
    return swap ? dispatch<GEMM_A_PLAIN, true>(stream, g, cfg, splitk) : dispatch<GEMM_A_PLAIN, false>(stream, g, cfg, splitk);
}
