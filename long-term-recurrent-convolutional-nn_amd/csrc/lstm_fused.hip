// lstm_fused.hip -- one launch per recurrent timestep for small and medium batches (bf16, gfx950; row blocks of 64):
//   forward : G[s] = Gx[s] + h[s-1] Wh'  fused with the cell update of step s           (lrcn.jl:529-536)
//   backward: dh_rec = dZ[s] Wh          fused with the cell backward of step s-1       (AutoGrad dual)
// At 32..64 rows per GPU (what 4..8-way data parallelism leaves of a 256 batch) the LSTM step is a chain of ~130 tiny
// launches per layer pass (GEMM + split-K combine + memset + cell kernel per timestep) whose cost is launch latency, not
// work.  Here a workgroup owns 16 hidden units for all rows: the four gate pre-activations of a unit (forward) or the
// four K-slices of the dh contraction (backward) are computed by the four WAVES of the workgroup, each an independent
// [M x 16] MFMA GEMM with its own LDS-DMA ring (no barrier in the K loop -- a wave only reads what it staged itself), and
// meet in LDS for the elementwise cell math.  63 workgroups for H = 1000 per block of 64 rows (grid.y).  Used up to
// B = 128 (LRCN_LSTM_FUSED_MAXB): at B = 256 it is faster alone (2.39 -> 2.21 ms per LSTM step) but its 252 LDS-heavy
// workgroups take more from the concurrently running convolutions than the separate launches do.
#include <type_traits>

#include "common.h"
#include "gemm.h"
#include "kernels.h"

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
template <int OFF> __device__ __forceinline__ uint4 lds_read16(unsigned addr) {
    uint4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
__device__ __forceinline__ float sigm_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// acc[i] (rows 16 i + 4 lq + r, column l15) = A[M][K-tiles kt0..kt1) * Brows[16][same K]'   for ONE wave.
//   A: [M][lda] bf16 (rows >= M: zero page); Brow: this lane's B row pointer for staging (NULL -> zero page); ring of NBUF slots
//   of (MT*16 + 16) rows x 128 B at LDS byte offset `ring` (wave-private).
//   BP = 8-row pieces of B staged per K-tile: 2 (16 B rows) or 1 (8 B rows; columns 8..15 of the result repeat columns 0..7).
template <int MT, int PF, int BP = 2>
__device__ __forceinline__ void wave_gemm(f32x4v (&acc)[MT], const bf16_t *A, int64_t lda, int M, const bf16_t *Bbase, int64_t ldb,
                                          int brow0, int brows_valid, int kt0, int kt1, unsigned char *smem, unsigned ring,
                                          const void *zero_page, int lane) {
    constexpr int NBUF = PF + 1, ROWS = MT * 16, SLOT = (ROWS + 8 * BP) * 128, APW = MT * 2, IPT = APW + BP;
    const int l15 = lane & 15, lq = lane >> 4;
    const bf16_t *Zp = reinterpret_cast<const bf16_t *>(zero_page) + (lane & 7) * 8;
    int a_off[APW];
    bool a_ok[APW];
#pragma unroll
    for (int p = 0; p < APW; ++p) {
        const int row = p * 8 + (lane >> 3);
        a_ok[p] = row < M;
        a_off[p] = a_ok[p] ? row * (int)lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3) : 0;
    }
    int b_off[BP];
    bool b_ok[BP];
#pragma unroll
    for (int p = 0; p < BP; ++p) {
        const int row = p * 8 + (lane >> 3);
        b_ok[p] = row < brows_valid;
        b_off[p] = b_ok[p] ? (brow0 + row) * (int)ldb + (((lane & 7) ^ ((row >> 1) & 7)) << 3) : 0;
    }
    const int KT = kt1 - kt0;
    auto issue = [&](int t) {
        const bool live = t < KT;
        const int slot = t % NBUF, ko = (kt0 + (live ? t : 0)) * 64;
#pragma unroll
        for (int p = 0; p < APW; ++p) {
            const bf16_t *src = (live & a_ok[p]) ? A + (a_off[p] + ko) : Zp;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(smem + ring + slot * SLOT + p * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            const bf16_t *src = (live & b_ok[p]) ? Bbase + (b_off[p] + ko) : Zp;
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(smem + ring + slot * SLOT + ROWS * 128 + p * 1024), 16, 0, 0);
        }
    };
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + ring;
    unsigned fa[2], fb[2];
    const int lb = BP == 1 ? (l15 & 7) : l15;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        fa[s] = lds0 + l15 * 128 + (((4 * s + lq) ^ ((l15 >> 1) & 7)) << 4);
        fb[s] = lds0 + lb * 128 + (((4 * s + lq) ^ ((lb >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int t = 0; t < PF; ++t) issue(t);
    for (int t = 0; t < KT; ++t) {
        wait_vmcnt<(PF - 1) * IPT>();  // tile t (staged by this wave alone) has landed; its slot's previous reads are long done
        issue(t + PF);
        const unsigned base = (unsigned)((t % NBUF) * SLOT);
        uint4 af[MT][2], b0, b1;
        // one asm statement: outputs exist only after the wait (no consumer can be scheduled above it)
        if constexpr (MT == 1) {
            asm volatile(
                "ds_read_b128 %0, %4 offset:%6\n\tds_read_b128 %1, %5 offset:%6\n\tds_read_b128 %2, %7\n\tds_read_b128 %3, %8\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(b0), "=&v"(b1), "=&v"(af[0][0]), "=&v"(af[0][1])
                : "v"(fb[0] + base), "v"(fb[1] + base), "n"(ROWS * 128), "v"(fa[0] + base), "v"(fa[1] + base)
                : "memory");
        } else if constexpr (MT == 2) {
            static_assert(BP == 2, "8-row B pieces: MT = 1 only");
            asm volatile(
                "ds_read_b128 %0, %6 offset:%8\n\tds_read_b128 %1, %7 offset:%8\n\t"
                "ds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\tds_read_b128 %4, %6 offset:2048\n\tds_read_b128 %5, %7 offset:2048\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(b0), "=&v"(b1), "=&v"(af[0][0]), "=&v"(af[0][1]), "=&v"(af[1][0]), "=&v"(af[1][1])
                : "v"(fa[0] + base), "v"(fa[1] + base), "n"(ROWS * 128)
                : "memory");
        } else {
            static_assert(MT == 4 && BP == 2, "MT is 1, 2 or 4");
            asm volatile(
                "ds_read_b128 %0, %10 offset:%12\n\tds_read_b128 %1, %11 offset:%12\n\t"
                "ds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\tds_read_b128 %4, %10 offset:2048\n\tds_read_b128 %5, %11 offset:2048\n\t"
                "ds_read_b128 %6, %10 offset:4096\n\tds_read_b128 %7, %11 offset:4096\n\tds_read_b128 %8, %10 offset:6144\n\t"
                "ds_read_b128 %9, %11 offset:6144\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(b0), "=&v"(b1), "=&v"(af[0][0]), "=&v"(af[0][1]), "=&v"(af[1][0]), "=&v"(af[1][1]), "=&v"(af[2][0]), "=&v"(af[2][1]),
                  "=&v"(af[3][0]), "=&v"(af[3][1])
                : "v"(fa[0] + base), "v"(fa[1] + base), "n"(ROWS * 128)
                : "memory");
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i][0]), __builtin_bit_cast(bf16x8, b0), acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i][1]), __builtin_bit_cast(bf16x8, b1), acc[i], 0, 0, 0);
        }
    }
    wait_vmcnt<0>();  // the dead tiles of the tail
}

template <int MT, int BP = 2> struct FusedGeom {
    static constexpr int PF = MT == 1 ? 8 : MT == 2 ? 4 : 2;
    static constexpr int WAVE_LDS = (PF + 1) * (MT * 16 + 8 * BP) * 128;
    static constexpr int XCH = 4 * MT * 16 * 17 * 4;  // exchange area: [4][M][17] f32 (placed over the rings after the K loops)
    static constexpr int LDS = 4 * WAVE_LDS > XCH ? 4 * WAVE_LDS : XCH;
};

struct RecFwdArgs {
    const bf16_t *h_prev;  // [B][ldh]
    const bf16_t *Wh;      // [4H][ldh]   rows g*H + u
    const float *Gx;       // [B][4H]     input-side pre-activations (+ bias) of this step
    const float *c_prev;   // [B][H]
    bf16_t *acts;          // [B][ld_a]   activated gates [f | i | o | g]
    float *c_new;          // [B][H]
    bf16_t *h_new;         // [B][ldh]
    const void *zero_page;
    int64_t ldh, ld_a;
    int B, H;
};

template <int MT> __global__ __launch_bounds__(256) void lstm_rec_fwd_kernel(const RecFwdArgs a) {
    typedef FusedGeom<MT> G;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = gate
    const int u0 = blockIdx.x * 16, H = a.H;
    const int r0 = blockIdx.y * (MT * 16);                       // row block (batches > 64 rows: several blocks of 64)
    const int M = a.B - r0 < MT * 16 ? a.B - r0 : MT * 16;
    const int valid = H - u0 < 16 ? H - u0 : 16;
    f32x4v acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
    wave_gemm<MT, G::PF>(acc, a.h_prev + (int64_t)r0 * a.ldh, a.ldh, M, a.Wh, a.ldh, wave * H + u0, valid, 0, (int)(a.ldh / 64), smem,
                         wave * G::WAVE_LDS, a.zero_page, lane);
    __syncthreads();  // every wave is done with its ring: the exchange area may overwrite it
    float *xch = reinterpret_cast<float *>(smem);
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[(wave * MT * 16 + i * 16 + 4 * lq + r) * 17 + l15] = acc[i][r];
    __syncthreads();
    for (int e = tid; e < M * 16; e += 256) {
        const int ml = e >> 4, u = e & 15, j = u0 + u, m = r0 + ml;
        if (j >= H) continue;
        const float *gx = a.Gx + (int64_t)m * 4 * H;
        const float f = sigm_f(xch[(0 * MT * 16 + ml) * 17 + u] + gx[j]);
        const float i = sigm_f(xch[(1 * MT * 16 + ml) * 17 + u] + gx[H + j]);
        const float o = sigm_f(xch[(2 * MT * 16 + ml) * 17 + u] + gx[2 * H + j]);
        const float ch = tanhf(xch[(3 * MT * 16 + ml) * 17 + u] + gx[3 * H + j]);
        const float c = a.c_prev[(int64_t)m * H + j] * f + i * ch;
        const float h = o * tanhf(c);
        bf16_t *ac = a.acts + (int64_t)m * a.ld_a;
        ac[j] = (bf16_t)f;
        ac[H + j] = (bf16_t)i;
        ac[2 * H + j] = (bf16_t)o;
        ac[3 * H + j] = (bf16_t)ch;
        a.c_new[(int64_t)m * H + j] = c;
        a.h_new[(int64_t)m * a.ldh + j] = (bf16_t)h;
    }
}

struct RecBwdArgs {
    const bf16_t *dz_s;    // [B][ld4]   dZ of step s
    const bf16_t *WhT;     // [H][ld4]   row u = column u of Wh'
    // cell backward of step s-1:
    const bf16_t *acts;    // [B][ld4]
    const float *c_prev;   // [B][H] or NULL (step s-1 = 0)
    const float *c_new;    // [B][H]
    const float *dh_ext;   // [B][H]
    float *dc;             // [B][H] in/out
    bf16_t *dz_out;        // [B][ld4]   dZ of step s-1
    const void *zero_page;
    int64_t ld4;
    int B, H;
};

// U = hidden units per workgroup: 16, or 8 with MT = 1 (16 rows x 8 units: 250 workgroups at 32 rows, H = 1000, each pulling
// 4 waves x 16 K-tiles x 3 KiB = 192 KiB through its CU's LDS-DMA path instead of 63 pulling 384 KiB -- see the forward kernel)
// U = 12 with MT = 2 (round 5): 84 workgroups instead of 63 beside the VGG forward's 160 (96 CUs free); rows 12..15 of a wave's B pieces
// come from the zero page.
template <int MT, int U = 16> __global__ __launch_bounds__(256) void lstm_rec_bwd_kernel(const RecBwdArgs a) {
    constexpr int BP = (U + 7) / 8;
    typedef FusedGeom<MT, BP> G;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = K slice
    const int u0 = blockIdx.x * U, H = a.H;
    const int r0 = blockIdx.y * (MT * 16);
    const int M = a.B - r0 < MT * 16 ? a.B - r0 : MT * 16;
    const int valid = H - u0 < U ? H - u0 : U;
    const int KT = (int)(a.ld4 / 64), per = (KT + 3) / 4;
    const int kt0 = wave * per < KT ? wave * per : KT, kt1 = (wave + 1) * per < KT ? (wave + 1) * per : KT;
    f32x4v acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
    wave_gemm<MT, G::PF, BP>(acc, a.dz_s + (int64_t)r0 * a.ld4, a.ld4, M, a.WhT, a.ld4, u0, valid, kt0, kt1, smem, wave * G::WAVE_LDS,
                             a.zero_page, lane);
    __syncthreads();
    float *xch = reinterpret_cast<float *>(smem);
    const int l15 = lane & 15, lq = lane >> 4;
    if (l15 < U) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) xch[(wave * MT * 16 + i * 16 + 4 * lq + r) * 17 + l15] = acc[i][r];
    }
    __syncthreads();
    for (int e = tid; e < M * U; e += 256) {
        const int ml = e / U, u = e % U, j = u0 + u, m = r0 + ml;
        if (j >= H) continue;
        const float dh = a.dh_ext[(int64_t)m * H + j] + xch[(0 * MT * 16 + ml) * 17 + u] + xch[(1 * MT * 16 + ml) * 17 + u] +
                         xch[(2 * MT * 16 + ml) * 17 + u] + xch[(3 * MT * 16 + ml) * 17 + u];
        const bf16_t *ac = a.acts + (int64_t)m * a.ld4;
        const float f = (float)ac[j], i = (float)ac[H + j], o = (float)ac[2 * H + j], g = (float)ac[3 * H + j];
        const float tc = tanhf(a.c_new[(int64_t)m * H + j]);
        const float dov = dh * tc;
        const float dcv = a.dc[(int64_t)m * H + j] + dh * o * (1.0f - tc * tc);
        const float cp = a.c_prev ? a.c_prev[(int64_t)m * H + j] : 0.0f;
        bf16_t *z = a.dz_out + (int64_t)m * a.ld4;
        z[j] = (bf16_t)(dcv * cp * f * (1.0f - f));
        z[H + j] = (bf16_t)(dcv * g * i * (1.0f - i));
        z[2 * H + j] = (bf16_t)(dov * o * (1.0f - o));
        z[3 * H + j] = (bf16_t)(dcv * i * (1.0f - g * g));
        a.dc[(int64_t)m * H + j] = dcv * f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Second form of the forward step, for H <= 1024 (K <= 16 K-tiles).  The ring form is bound by the rate at which ONE CU takes in
// LDS-DMA bytes (guide: 68-90 GB/s per CU with four loading waves): a workgroup pulls 4 waves x 16 K-tiles x 6 KiB = 384 KiB, the
// h block four times over, and only 63 of the 256 CUs work -- 9.0 us per step at 32 rows.  Here a workgroup owns 8 hidden units
// (125 workgroups for H = 1000), the 32-row block of h[s-1] is staged ONCE and shared by the four gate waves (64 KiB for K = 1024),
// each wave adds the 8 Wh rows of its gate (16 KiB): 128 KiB per CU, every byte requested before the first wait, one barrier, then
// LDS reads + 64 MFMAs per wave (columns 8..15 of the 16-wide MFMA tile repeat columns 0..7 and are dropped).
// (Wh and h fragments straight from global memory into registers in MFMA operand layout -- 16 rows x 64 B per wave instruction --
// measured SLOWER than the ring: 10.1 vs 9.0 us forward, 17.0 vs 10.3 us backward.)
// Row blocks of 32 (grid.y = ceil(B / 32)).
// U = 12 (round 5): the same kernel with 12 units per workgroup -- 84 workgroups for H = 1000, 16 K-tiles x (4 KiB of h + 4 x 1.5 KiB of
// Wh rows) = exactly the CU's 160 KiB of LDS -- for the rank-of-8 training step, where the VGG forward's capped grid (160 workgroups, one
// per CU, persistent) leaves 96 CUs: 125 workgroups of the 8-unit form would need a second round, 84 fit in one, each pulling 160 KiB
// through its CU's LDS-DMA path where the ring form (63 workgroups of 16 units) pulls 384 KiB.  A wave's Wh rows of one K-tile are one
// full LDS-DMA piece (rows 0..7) + one half piece under an EXEC mask (rows 8..11: lanes 0..31 only, so nothing is written past the 12th
// row); columns 12..15 of the MFMA tile repeat columns 4..7 and are dropped.
constexpr int R2_ROWS = 32, R2_KT = 16;

template <int U> __global__ __launch_bounds__(256) void lstm_rec_fwd2_kernel(const RecFwdArgs a) {
    static_assert(U == 8 || U == 12, "8 or 12 hidden units per workgroup");
    constexpr int BROW = U * 128;  // bytes of one wave's Wh rows per K-tile
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = gate
    const int u0 = blockIdx.x * U, H = a.H;
    const int r0 = blockIdx.y * R2_ROWS;
    const int M = a.B - r0 < R2_ROWS ? a.B - r0 : R2_ROWS;
    const int valid = H - u0 < U ? H - u0 : U;
    const int KT = (int)(a.ldh / 64);  // <= R2_KT (launch check)
    // LDS: h block, K-tile kt = 32 rows x 128 B at kt * 4096; then per wave its Wh rows, K-tile kt = U rows x 128 B at kt * BROW.
    // 16-byte chunk j of row r sits at chunk j ^ ((r >> 1) & 7) (applied to the SOURCE address: LDS-DMA writes lanes in order).
    unsigned char *sB = smem + KT * (R2_ROWS * 128) + wave * (KT * BROW);
    {
        const bf16_t *Zp = reinterpret_cast<const bf16_t *>(a.zero_page) + (lane & 7) * 8;
        const int r8 = lane >> 3, row = wave * 8 + r8;  // h rows: wave w stages rows 8 w .. 8 w + 7 of every K-tile
        const bool aok = row < M, bok = r8 < valid;
        const bf16_t *asrc = aok ? a.h_prev + (int64_t)(r0 + row) * a.ldh + (((lane & 7) ^ ((row >> 1) & 7)) << 3) : Zp;
        const bf16_t *bsrc = bok ? a.Wh + (int64_t)(wave * H + u0 + r8) * a.ldh + (((lane & 7) ^ ((r8 >> 1) & 7)) << 3) : Zp;
        const int astep = aok ? 64 : 0, bstep = bok ? 64 : 0;
#pragma unroll
        for (int kt = 0; kt < R2_KT; ++kt)
            if (kt < KT) {
                __builtin_amdgcn_global_load_lds((glb_void *)(asrc + kt * astep), (lds_void *)(smem + kt * (R2_ROWS * 128) + wave * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_void *)(bsrc + kt * bstep), (lds_void *)(sB + kt * BROW), 16, 0, 0);
            }
        if constexpr (U == 12) {
            if (lane < 32) {  // rows 8..11: half a piece; the inactive lanes write nothing
                const int rb = 8 + r8;
                const bool ok2 = rb < valid;
                const bf16_t *b2 = ok2 ? a.Wh + (int64_t)(wave * H + u0 + rb) * a.ldh + (((lane & 7) ^ ((rb >> 1) & 7)) << 3) : Zp;
                const int b2step = ok2 ? 64 : 0;
#pragma unroll
                for (int kt = 0; kt < R2_KT; ++kt)
                    if (kt < KT) __builtin_amdgcn_global_load_lds((glb_void *)(b2 + kt * b2step), (lds_void *)(sB + kt * BROW + 1024), 16, 0, 0);
            }
        }
    }
    wait_vmcnt<0>();
    __syncthreads();
    f32x4v acc[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
    const int lb = U == 8 ? (l15 & 7) : (l15 < 12 ? l15 : l15 - 8);  // B row this lane's MFMA column reads (columns >= U repeat earlier ones)
    const int fa0 = l15 * 128 + (((0 + lq) ^ ((l15 >> 1) & 7)) << 4), fa1 = l15 * 128 + (((4 + lq) ^ ((l15 >> 1) & 7)) << 4);
    const int fb0 = lb * 128 + (((0 + lq) ^ ((lb >> 1) & 7)) << 4), fb1 = lb * 128 + (((4 + lq) ^ ((lb >> 1) & 7)) << 4);
#pragma unroll
    for (int kt = 0; kt < R2_KT; ++kt) {
        if (kt < KT) {
            const unsigned char *t = smem + kt * (R2_ROWS * 128), *tb = sB + kt * BROW;
            const uint4 b0 = *reinterpret_cast<const uint4 *>(tb + fb0), b1 = *reinterpret_cast<const uint4 *>(tb + fb1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint4 a0 = *reinterpret_cast<const uint4 *>(t + i * 2048 + fa0), a1 = *reinterpret_cast<const uint4 *>(t + i * 2048 + fa1);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), acc[i], 0, 0, 0);
            }
        }
    }
    __syncthreads();  // every wave is done with the staged operands: the exchange area may overwrite them
    float *xch = reinterpret_cast<float *>(smem);  // [gate][row][U + 1]
    if (l15 < U) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) xch[(wave * R2_ROWS + i * 16 + 4 * lq + r) * (U + 1) + l15] = acc[i][r];
    }
    __syncthreads();
    for (int e = tid; e < R2_ROWS * U; e += 256) {
        const int ml = e / U, u = e % U, j = u0 + u, m = r0 + ml;
        if (ml < M && j < H) {
            const float *gx = a.Gx + (int64_t)m * 4 * H;
            const float f = sigm_f(xch[(0 * R2_ROWS + ml) * (U + 1) + u] + gx[j]);
            const float i = sigm_f(xch[(1 * R2_ROWS + ml) * (U + 1) + u] + gx[H + j]);
            const float o = sigm_f(xch[(2 * R2_ROWS + ml) * (U + 1) + u] + gx[2 * H + j]);
            const float ch = tanhf(xch[(3 * R2_ROWS + ml) * (U + 1) + u] + gx[3 * H + j]);
            const float c = a.c_prev[(int64_t)m * H + j] * f + i * ch;
            const float h = o * tanhf(c);
            bf16_t *ac = a.acts + (int64_t)m * a.ld_a;
            ac[j] = (bf16_t)f;
            ac[H + j] = (bf16_t)i;
            ac[2 * H + j] = (bf16_t)o;
            ac[3 * H + j] = (bf16_t)ch;
            a.c_new[(int64_t)m * H + j] = c;
            a.h_new[(int64_t)m * a.ldh + j] = (bf16_t)h;
        }
    }
}

// The 8-units-per-workgroup forms are taken up to this many rows (LRCN_LSTM_REC2; 0 = never) and only when the LSTM step has the GPU to
// itself (training on precomputed features, the reference's default mode): measured ms per LSTM training step, 16-unit -> 8-unit
// forms, E = H = 1000, V = 10640, T = 11: 1.053 -> 0.955 at 32 rows, 1.284 -> 1.151 at 64, 1.456 -> 1.672 at 128 (two rounds of
// workgroups).  Beside the VGG forward of the two-stream step they LOSE (1.555 -> 1.685 ms per step at 32 rows, 2.30 -> 2.56 at 64):
// twice to four times as many workgroups, each holding 108-128 KiB of LDS, displace the convolution workgroups (one per CU, 128-160 KiB)
// from more CUs at every kernel boundary.
int rec2_max_batch() {
    const char *k = getenv("LRCN_LSTM_REC2");  // read per launch (the tests switch it inside one process)
    return k ? atoi(k) : 64;
}

// Beside the VGG forward (`alone` false): the 12-unit forms up to this many rows (LRCN_LSTM_REC3; 0 = never, the ring forms of 16 units).
int rec3_max_batch() {
    const char *k = getenv("LRCN_LSTM_REC3");
    return k ? atoi(k) : 32;
}

template <class K> hipError_t set_lds(K kern, int lds, LdsAttrMask &done) { return set_max_lds(reinterpret_cast<const void *>(kern), lds, done); }

}  // namespace

bool lstm_fused_eligible(int dtype, int B, int H, int64_t ldh, int64_t ld4) {
    return dtype == GEMM_T_BF16 && B >= 1 && B <= 1024 && H >= 16 && (ldh % 64) == 0 && (ld4 % 64) == 0 &&
           (int64_t)B * ld4 < (1ll << 31) && (int64_t)4 * H * ldh < (1ll << 31);
}

hipError_t launch_lstm_rec_fwd(hipStream_t st, const void *h_prev, int64_t ldh, const void *Wh, const float *Gx, const float *c_prev, int B,
                               int H, void *acts, int64_t ld_a, float *c_new, void *h_new, const void *zero_page, bool alone) {
    RecFwdArgs a{};
    a.h_prev = (const bf16_t *)h_prev; a.Wh = (const bf16_t *)Wh; a.Gx = Gx; a.c_prev = c_prev;
    a.acts = (bf16_t *)acts; a.c_new = c_new; a.h_new = (bf16_t *)h_new; a.zero_page = zero_page;
    a.ldh = ldh; a.ld_a = ld_a; a.B = B; a.H = H;
    static LdsAttrMask d2{0}, d4{0}, dr{0};
    hipError_t e;
    if (alone && B <= rec2_max_batch() && ldh / 64 <= R2_KT) {
        const int KT = (int)(ldh / 64), lds = KT * (R2_ROWS * 128 + 4 * 1024);  // >= the exchange area (4 x 32 x 9 floats)
        if ((e = set_lds(lstm_rec_fwd2_kernel<8>, R2_KT * (R2_ROWS * 128 + 4 * 1024), dr)) != hipSuccess) return e;
        hipLaunchKernelGGL(lstm_rec_fwd2_kernel<8>, dim3((H + 7) / 8, (B + R2_ROWS - 1) / R2_ROWS), dim3(256), lds, st, a);
        return hipGetLastError();
    }
    if (!alone && B <= rec3_max_batch() && ldh / 64 <= R2_KT) {
        static LdsAttrMask d3{0};
        const int KT = (int)(ldh / 64), lds = KT * (R2_ROWS * 128 + 4 * 1536) > 4 * R2_ROWS * 13 * 4 ? KT * (R2_ROWS * 128 + 4 * 1536) : 4 * R2_ROWS * 13 * 4;
        if ((e = set_lds(lstm_rec_fwd2_kernel<12>, R2_KT * (R2_ROWS * 128 + 4 * 1536), d3)) != hipSuccess) return e;
        hipLaunchKernelGGL(lstm_rec_fwd2_kernel<12>, dim3((H + 11) / 12, (B + R2_ROWS - 1) / R2_ROWS), dim3(256), lds, st, a);
        return hipGetLastError();
    }
    const dim3 grid((H + 15) / 16, B <= 32 ? 1 : (B + 63) / 64);
    if (B <= 32) {
        if ((e = set_lds(lstm_rec_fwd_kernel<2>, FusedGeom<2>::LDS, d2)) != hipSuccess) return e;
        hipLaunchKernelGGL(lstm_rec_fwd_kernel<2>, grid, dim3(256), FusedGeom<2>::LDS, st, a);
    } else {
        if ((e = set_lds(lstm_rec_fwd_kernel<4>, FusedGeom<4>::LDS, d4)) != hipSuccess) return e;
        hipLaunchKernelGGL(lstm_rec_fwd_kernel<4>, grid, dim3(256), FusedGeom<4>::LDS, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_lstm_rec_bwd(hipStream_t st, const void *dz_s, int64_t ld4, const void *WhT, const void *acts, const float *c_prev,
                               const float *c_new, const float *dh_ext, float *dc, int B, int H, void *dz_out, const void *zero_page,
                               bool alone) {
    RecBwdArgs a{};
    a.dz_s = (const bf16_t *)dz_s; a.WhT = (const bf16_t *)WhT; a.acts = (const bf16_t *)acts; a.c_prev = c_prev; a.c_new = c_new;
    a.dh_ext = dh_ext; a.dc = dc; a.dz_out = (bf16_t *)dz_out; a.zero_page = zero_page;
    a.ld4 = ld4; a.B = B; a.H = H;
    static LdsAttrMask d2{0}, d4{0};
    hipError_t e;
    if (alone && B <= rec2_max_batch()) {
        static LdsAttrMask d1{0};
        constexpr int lds = FusedGeom<1, 1>::LDS;
        auto kern = lstm_rec_bwd_kernel<1, 8>;
        if ((e = set_lds(kern, lds, d1)) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((H + 7) / 8, (B + 15) / 16), dim3(256), lds, st, a);
        return hipGetLastError();
    }
    if (!alone && B <= 32 && B <= rec3_max_batch()) {
        static LdsAttrMask d12{0};
        auto kern = lstm_rec_bwd_kernel<2, 12>;
        if ((e = set_lds(kern, FusedGeom<2>::LDS, d12)) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((H + 11) / 12, 1), dim3(256), FusedGeom<2>::LDS, st, a);
        return hipGetLastError();
    }
    const dim3 grid((H + 15) / 16, B <= 32 ? 1 : (B + 63) / 64);
    if (B <= 32) {
        if ((e = set_lds(lstm_rec_bwd_kernel<2>, FusedGeom<2>::LDS, d2)) != hipSuccess) return e;
        hipLaunchKernelGGL(lstm_rec_bwd_kernel<2>, grid, dim3(256), FusedGeom<2>::LDS, st, a);
    } else {
        if ((e = set_lds(lstm_rec_bwd_kernel<4>, FusedGeom<4>::LDS, d4)) != hipSuccess) return e;
        hipLaunchKernelGGL(lstm_rec_bwd_kernel<4>, grid, dim3(256), FusedGeom<4>::LDS, st, a);
    }
    return hipGetLastError();
}
