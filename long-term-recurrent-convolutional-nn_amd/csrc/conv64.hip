// conv64.hip -- 3x3 / pad 1 convolution + bias + ReLU (+ fused 2x2 max-pool) for the Cin = 64 layers of VGG-16
// (conv1_2: 64 -> 64 with pool, conv2_1: 64 -> 128), bf16, gfx950.                     (lrcn.jl:724-726 convx/relux/poolx)
//
// Why a second convolution kernel: in the implicit-GEMM formulation (gemm_8p.hip) every input pixel is pulled L2 -> LDS
// nine times (once per tap) and, with only 64..128 output channels to amortise it over, those layers are bound by the
// LDS-DMA fill rate (~50 FLOP per staged byte), not by MFMA.  Here the reuse is made explicit:
//   * a workgroup owns a 16 x 16 output-pixel tile x 64 output channels; the 18 x 18 x 64ch input PATCH (tile + halo,
//     40.5 KiB) is DMA'd to LDS once (global_load_lds_dwordx4, zero page outside the image) and serves all nine taps:
//     the A fragment of tap (kh,kw) is the same patch read at a shifted pixel -- 7x less staging traffic;
//   * the weights never touch LDS: each wave keeps the B fragments of ALL nine taps for its 32 output channels in
//     registers (144 VGPRs) for the whole persistent kernel, so LDS bandwidth is spent on A fragments only
//     (8 ds_read_b128 per 16 MFMAs per wave = 50 % of the LDS read rate at full MFMA rate);
//   * persistent workgroups walk the tiles; three patch buffers, the DMA of patch j+2 is issued while patch j is
//     multiplied (counted s_waitcnt vmcnt, raw s_barrier, two barriers per patch);
//   * no barrier inside the 9-tap loop (patch and weights are resident); waves 4..7 run one barrier (= half a patch)
//     behind waves 0..3, so one wave of every SIMD is in MFMAs while its partner does the epilogue;
//   * rows of a tile are in window-major order (DESIGN.md "conv rows"): with D = A x B the four accumulator registers of
//     a lane are the four pixels of one 2x2 pool window (pool = 3 v_max); without pool D = B x A gives a lane 8 consecutive
//     output channels of one pixel (one 16-byte store).  The channel <-> fragment-row permutation that makes those
//     stores contiguous is free: it only changes which weight row a lane loads.
// FUSE variant (conv1_2 fed by the mean-subtracted bf16 crops): the patch is not loaded but COMPUTED -- conv1_1 (3 -> 64
// channels, K = 27) runs inside the kernel: the 20 x 20 x 3 window of a tile is DMA'd to LDS (2.4 KB, zero page outside the
// image = conv1_1's zero padding), a lane's im2col fragment is ONE 16-byte LDS read (the 9 values (kh, c) of image row kw
// are contiguous in the crop: run kw = lane group lq < 3 takes the first 8, lane group 3 the three 9th values), 4 MFMAs per
// 16-pixel m-tile with the bias as accumulator input, ReLU, bf16, 8-byte writes into the swizzled patch.  The 1.64 GB
// conv1_1 activation tensor never exists in HBM.  (A first version converted uint8 bytes in the producer: 3x the VALU
// work, slower than two launches; the u8 -> bf16 - mean pass is now a 25 us elementwise pre-kernel.)
// LDS patch image: pixel q = py*18 + px at q*128 bytes, 16-byte chunk c at position c ^ g(py,px),
//   g = (((px >> 1) & 3) << 1) | (py & 1): conflict-free for the 16-lane groups of ds_read_b128 at every tap shift.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// ReLU on two packed bf16 values as ONE v_pk_max_i16: a negative float has the sign bit set, i.e. is a negative int16
// (-0 = 0x8000 included), and max(., 0) clears it; positive values are positive int16 and pass unchanged.  Half the vector
// ALU instructions of fmaxf-before-convert -- the epilogue / producer of this kernel run beside the partner wave's MFMAs
// and compete with them for the SIMD's issue port.
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu_bf16x2(unsigned w) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), s16x2{0, 0}));
}

constexpr int PATCH_BYTES = 48 * 1024;  // 48 DMA pieces of 1 KiB (41 carry pixels, the rest keep the per-wave count uniform)
constexpr int NPIECE = 6;               // pieces per wave per patch
constexpr int BIAS_OFF = 3 * PATCH_BYTES;
constexpr int LDS_BYTES = BIAS_OFF + 64 * 4;
// FUSE layout: 21 m-tiles (336 pixel rows) per patch, 4 raw-window buffers, conv1_1 weights and both bias vectors
constexpr int F_PATCH = 21 * 2048;
constexpr int F_RAW_OFF = 3 * F_PATCH;
constexpr int F_RAW_ROW = 144;                // pitch of a window row (128 bytes written, 120 + 2 read back)
constexpr int F_RAWB = 24 * F_RAW_ROW;        // the second copy of the window, shifted by one element (see issue_raw)
constexpr int F_RAW = 2 * F_RAWB;             // 24 window rows (20 real) x 2 copies per raw buffer
constexpr int F_W11_OFF = F_RAW_OFF + 4 * F_RAW;
constexpr int F_B11_OFF = F_W11_OFF + 64 * 64;
constexpr int F_BIAS_OFF = F_B11_OFF + 256;
constexpr int F_LDS_BYTES = F_BIAS_OFF + 256;
static_assert(F_LDS_BYTES <= 160 * 1024, "the fused kernel's LDS image must fit the 160 KiB of a gfx950 CU");

struct Conv64Args {
    const bf16_t *in;   // NHWC [N][H][W][64]
    const bf16_t *w;    // [Cout][9][64]   (tap = kh*3 + kw)
    const float *bias;  // [Cout]
    bf16_t *out;        // NHWC [N][H][W][Cout] or pooled [N][H/2][W/2][Cout]
    const void *zero_page;
    int N, H, W, Cout, relu;
    int tiles_y, tiles_x, ntiles;
    // FUSE only: mean-subtracted crops in a 2-pixel zero frame img16[n][S + 4][S + 4][3] bf16 (k_img_u8_to_bf16; [x + 2][y + 2] = pixel
    // (row x, col y)), conv1_1 weights [64][32] in k' order, bias
    const bf16_t *img16;
    const bf16_t *w11;
    const float *b11;
    float f8_inv_scale;  // > 0 (non-pool, non-fused kernel only): write OCP e4m3(relu(acc + b) * f8_inv_scale), 8 bytes per lane
    // kernel development (LRCN_STAMPS=1, tools/conv64_stamps.py): 8 x uint64 per (tile, wave group): shader clock at [0] patch start,
    // [1] first half-taps done, [2] mid barrier passed, [3] second half-taps done, [4] epilogue (+ producer) done, [5] end barrier passed;
    // [6] / [7] the 100 MHz wall counter at the end / start.  NULL = off.
    unsigned long long *stamps;
};

template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int OFF> __device__ __forceinline__ uint4 lds_read16(unsigned addr) {
    uint4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int OFF> __device__ __forceinline__ unsigned lds_read_u16(unsigned addr) {
    unsigned r;
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
__device__ __forceinline__ void lds_write8(unsigned addr, uint2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// fragment row j (0..15) of n-tile n (0..1) of a wave's 32 output channels -> channel offset within those 32
template <bool POOL> __device__ __forceinline__ int chan_of(int n, int j) {
    return POOL ? 2 * j + n                           // lane = channel pair (2 l15, 2 l15 + 1): one 4-byte store
                : (j >> 2) * 8 + n * 4 + (j & 3);     // lane (lq) = 8 consecutive channels: one 16-byte store
}

template <bool POOL, bool FUSE> __global__ __launch_bounds__(512) void conv64_kernel(const Conv64Args a) {
    constexpr int PB = FUSE ? F_PATCH : PATCH_BYTES;  // bytes per patch buffer
    constexpr int BOFF = FUSE ? F_BIAS_OFF : BIAS_OFF;
    constexpr int NST = POOL ? 4 : 4;
    // half-taps before the mid-patch barrier.  FUSE: ALL of them -- the second segment is epilogue + producer, so (waves 4..7
    // being one barrier behind) one wave of a SIMD multiplies while its partner produces the patch after next
#ifndef CONV64_FUSE_SPLIT
#define CONV64_FUSE_SPLIT 18
#endif
#ifndef CONV64_SPLIT
#define CONV64_SPLIT 15  // half-taps before the mid-patch barrier: 9 | 9 left the group that has no epilogue waiting at the barrier (tools/conv64_stamps.py); same-box cycles per patch 9 / 11 / 13 / 15: conv2_1 8072 / 7808 / 7532 / 7224, conv1_2 6856 / 6712 / 6616 / 6464
#endif
    constexpr int SPLIT = FUSE ? CONV64_FUSE_SPLIT : CONV64_SPLIT;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];  // K-half select is address ^ 64
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 3;   // pixel group: m-tiles 4 wp .. 4 wp + 3 of the 16 (window rows 2 wp, 2 wp + 1)
    const int wq = wave >> 2;  // channel group (32 channels) AND stagger group
    const int l15 = lane & 15, lq = lane >> 4;
    const int cc = blockIdx.y;  // 64-channel chunk of Cout
    const int H = a.H, W = a.W;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

    // ---- weights: B fragments of all 9 taps x 2 K-halves x 2 n-tiles, resident in registers ----
    uint4 breg[9][2][2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const bf16_t *wr = a.w + (size_t)(cc * 64 + wq * 32 + chan_of<POOL>(n, l15)) * 576 + lq * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) breg[t][s][n] = *reinterpret_cast<const uint4 *>(wr + t * 64 + s * 32);
    }
    if (tid < 64) reinterpret_cast<float *>(smem + BOFF)[tid] = a.bias ? a.bias[cc * 64 + tid] : 0.0f;

    // ---- A-fragment read addresses: lane l15 = (window w, dy, dx) of an m-tile, lq = 16-byte K chunk ----
    const int w_ = l15 >> 2, dy = (l15 >> 1) & 1, dx = l15 & 1, xl = 2 * w_ + dx;
    unsigned areg[2][3];  // [kh parity][kw], K half 0; K half 1 = address ^ 64
#pragma unroll
    for (int yp = 0; yp < 2; ++yp)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int gsw = ((((xl + kw) >> 1) & 3) << 1) | ((dy + yp) & 1);
            areg[yp][kw] = lds0 + ((4 * wp + dy) * 18 + xl) * 128 + ((lq ^ gsw) << 4);
        }

    // ---- DMA geometry: piece k = wave + 8 j covers patch pixels q = 8k .. 8k+7 (q = py*18 + px); lane -> (pixel, chunk) ----
    int doff[NPIECE];
    unsigned dflags = 0;  // 5 bits per piece: top, bottom, left, right halo row/column, not-a-pixel
    const bf16_t *Zp = reinterpret_cast<const bf16_t *>(a.zero_page) + (lane & 7) * 8;
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
        const int q = (wave + 8 * j) * 8 + (lane >> 3);
        const int py = q / 18, px = q - 18 * py;
        const int gsw = (((px >> 1) & 3) << 1) | (py & 1);
        doff[j] = ((py - 1) * W + (px - 1)) * 64 + (((lane & 7) ^ gsw) << 3);
        const unsigned f = (py == 0 ? 1u : 0u) | (py == 17 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == 17 ? 8u : 0u) | (q >= 324 ? 16u : 0u);
        dflags |= f << (5 * j);
    }
    auto issue_patch = [&](int tile, int buf) {
        // tile < 0: nothing left for this workgroup -- stage zeros so that every wave's vmcnt arithmetic stays uniform
        const bool live = tile >= 0;
        const int t = live ? tile : 0;
        const int per_img = a.tiles_y * a.tiles_x;
        const int n = t / per_img, r = t - n * per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int origin = ((n * H + ty * 16) * W + tx * 16) * 64;
        const unsigned edge = (ty == 0 ? 1u : 0u) | (ty == a.tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == a.tiles_x - 1 ? 8u : 0u) |
                              16u | (live ? 0u : 15u);
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) {
            const bool bad = ((dflags >> (5 * j)) & edge) != 0;
            const bf16_t *src = bad ? Zp : a.in + (origin + doff[j]);
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(smem + buf * PB + (wave + 8 * j) * 1024), 16, 0, 0);
        }
    };


    // ================= FUSE: raw-window DMA and the conv1_1 patch producer =================
    auto issue_raw = [&](int tile, int rbuf) {
        // six 128-byte DMAs per wave, lanes 0..31 only, one dword per lane: window row wrow = wave + 8 k (rows 20..23 are dummies that
        // keep the per-wave count uniform), each row TWICE: copy A = elements [E, E + 64) of the framed image row, copy B = [E + 1, E + 65),
        // i.e. the same data shifted by one bf16.  A pixel's 9-value run starts at element 3 py: on a dword boundary in copy A for even
        // py, in copy B for odd py -- so the producer reads it with ds_read2_b32 (4-byte alignment) instead of a 2-byte-aligned
        // ds_read_b128, which the LDS executes as a slow unaligned access: that cost 450 of the producer's 1330 read cycles AND 530
        // cycles of the partner wave's MFMA phase, whose fragment reads queue behind it (tools/conv64_stamps.py, round 3).
        // The crops carry a 2-pixel zero frame (k_img_u8_to_bf16), so conv1_1's zero padding is read as data: no in-image tests, and
        // no dword of either copy straddles the image edge.  Only elements < E + 61 are ever read back.
        // LDS rows are F_RAW_ROW = 144 bytes apart: the producer's 16 lanes of a group read 16 CONSECUTIVE window rows at one byte
        // offset, and 16 x 144 B covers all sixteen 16-byte bank slots of the 256-byte LDS line once.
        const bool live = tile >= 0;
        const int t = live ? tile : 0;
        const int per_img = a.tiles_y * a.tiles_x;
        const int n = t / per_img, r = t - n * per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int SP = H + 4;
        if (lane < 32) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int wrow = wave + 8 * k;
                const bool ok = live && wrow < 20;
                // image row 16 tx - 2 + wrow = framed row 16 tx + wrow; image column 16 ty - 2 = framed column 16 ty
                const bf16_t *srcA = a.img16 + (((size_t)(n * SP + 16 * tx + wrow) * SP + 16 * ty) * 3 + 2 * lane);
                const bf16_t *z = reinterpret_cast<const bf16_t *>(a.zero_page);
                __builtin_amdgcn_global_load_lds((glb_void *)(ok ? srcA : z), (lds_void *)(smem + F_RAW_OFF + rbuf * F_RAW + wrow * F_RAW_ROW), 4, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_void *)(ok ? srcA + 1 : z), (lds_void *)(smem + F_RAW_OFF + rbuf * F_RAW + F_RAWB + wrow * F_RAW_ROW),
                                                 4, 0, 0);
            }
        }
    };
    auto produce = [&](int tile, int pbuf, int rbuf) {
        if (tile < 0) return;
        const int per_img = a.tiles_y * a.tiles_x;
        const int n = tile / per_img, r = tile - n * per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        (void)n;
        const int S = H;
#ifdef CONV64_PRODUCER_PRIO_ALL
        __builtin_amdgcn_s_setprio(CONV64_PRODUCER_PRIO_ALL);
#endif
        const unsigned raw = lds0 + F_RAW_OFF + rbuf * F_RAW;
        const unsigned pat = lds0 + pbuf * F_PATCH;
        // everything below depends only on the lane and the tile: keep hipcc from hoisting it out of the patch loop (it would
        // pin ~30 registers across the MFMA body and spill)
        int l15v = l15, lqv = lq;
        asm volatile("" : "+v"(l15v), "+v"(lqv));
        const int lsel = lqv < 2 ? lqv : 2;
        const bool last = lqv == 3;
        // conv1_1 weight fragments: rows = channels nn*16 + l15 (64-byte rows: nn KiB apart), chunk lq swizzled by (row >> 2) & 3
        const unsigned wfa = lds0 + F_W11_OFF + l15v * 64 + ((lqv ^ ((l15v >> 2) & 3)) << 4);
        const unsigned bba = lds0 + F_B11_OFF + lqv * 16;  // bias of channels nn*16 + 4 lq .. + 3
        // All LDS reads of the call and their wait are ONE asm statement: for hipcc the outputs exist only after the wait.
        // (With separate statements it folded `phi(n9, 0) << 16` into the block that issues the read and consumed the
        // register before the data had landed.)  Waves 5..7 own two m-tiles; their third read set hits in-bounds garbage.
        const bool three = wave + 16 < 21;  // wave-uniform
        uint2 runl[3], runh[3];
        uint4 wf[4], bq[4];
        unsigned n9[3][3];
        int qv[3];
        unsigned rbs[3], rbl[3];
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const int q = (wave + 8 * sl) * 16 + l15v;
            const int py = q / 18, px = q - 18 * py;
            qv[sl] = q;
            // run kw of pixel (py, px): window row px + kw, 9 bf16 from byte 6 py of copy A = byte 6 py - 2 of copy B
            rbl[sl] = raw + px * F_RAW_ROW + 6 * py;                                    // copy A, 2-byte aligned: the ds_read_u16 of the 9th values
            rbs[sl] = rbl[sl] + lsel * F_RAW_ROW + ((py & 1) ? F_RAWB - 2 : 0);          // dword-aligned start of the run's first 8 values
        }
        asm volatile(
            "ds_read2_b32 %[l0], %[s0] offset1:1\n\tds_read2_b32 %[h0], %[s0] offset0:2 offset1:3\n\t"
            "ds_read_u16 %[a0], %[b0] offset:16\n\tds_read_u16 %[a1], %[b0] offset:160\n\tds_read_u16 %[a2], %[b0] offset:304\n\t"
            "ds_read2_b32 %[l1], %[s1] offset1:1\n\tds_read2_b32 %[h1], %[s1] offset0:2 offset1:3\n\t"
            "ds_read_u16 %[c0], %[b1] offset:16\n\tds_read_u16 %[c1], %[b1] offset:160\n\tds_read_u16 %[c2], %[b1] offset:304\n\t"
            "ds_read2_b32 %[l2], %[s2] offset1:1\n\tds_read2_b32 %[h2], %[s2] offset0:2 offset1:3\n\t"
            "ds_read_u16 %[d0], %[b2] offset:16\n\tds_read_u16 %[d1], %[b2] offset:160\n\tds_read_u16 %[d2], %[b2] offset:304\n\t"
            "ds_read_b128 %[w0], %[wa]\n\tds_read_b128 %[w1], %[wa] offset:1024\n\tds_read_b128 %[w2], %[wa] offset:2048\n\tds_read_b128 %[w3], %[wa] offset:3072\n\t"
            "ds_read_b128 %[q0], %[ba]\n\tds_read_b128 %[q1], %[ba] offset:64\n\tds_read_b128 %[q2], %[ba] offset:128\n\tds_read_b128 %[q3], %[ba] offset:192\n\t"
            "s_waitcnt lgkmcnt(0)"
            : [l0] "=&v"(runl[0]), [h0] "=&v"(runh[0]), [l1] "=&v"(runl[1]), [h1] "=&v"(runh[1]), [l2] "=&v"(runl[2]), [h2] "=&v"(runh[2]),
              [a0] "=&v"(n9[0][0]), [a1] "=&v"(n9[0][1]), [a2] "=&v"(n9[0][2]), [c0] "=&v"(n9[1][0]), [c1] "=&v"(n9[1][1]), [c2] "=&v"(n9[1][2]),
              [d0] "=&v"(n9[2][0]), [d1] "=&v"(n9[2][1]), [d2] "=&v"(n9[2][2]), [w0] "=&v"(wf[0]), [w1] "=&v"(wf[1]), [w2] "=&v"(wf[2]),
              [w3] "=&v"(wf[3]), [q0] "=&v"(bq[0]), [q1] "=&v"(bq[1]), [q2] "=&v"(bq[2]), [q3] "=&v"(bq[3])
            : [s0] "v"(rbs[0]), [s1] "v"(rbs[1]), [s2] "v"(rbs[2]), [b0] "v"(rbl[0]), [b1] "v"(rbl[1]), [b2] "v"(rbl[2]), [wa] "v"(wfa), [ba] "v"(bba)
            : "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (a.stamps && (tid & 255) == 0 && tile >= 2 * (int)gridDim.x)  // the patch being multiplied is two walks back (not in the prologue)
            a.stamps[((size_t)(tile - 2 * (int)gridDim.x) * 2 + wq) * 8 + 7] = __builtin_amdgcn_s_memtime();
        static_for<0, 3>([&](auto sc) {
            constexpr int sl = decltype(sc)::value;
            if (sl < 2 || three) {
                const int q = qv[sl];
                const int py = q / 18, px = q - 18 * py;
                const bool pv = (q < 324) & ((unsigned)(16 * ty - 1 + py) < (unsigned)S) & ((unsigned)(16 * tx - 1 + px) < (unsigned)S);
                uint4 afr;
                afr.x = last ? (n9[sl][0] | (n9[sl][1] << 16)) : runl[sl].x;
                afr.y = last ? n9[sl][2] : runl[sl].y;
                afr.z = last ? 0u : runh[sl].x;
                afr.w = last ? 0u : runh[sl].y;
                const bf16x8 av = __builtin_bit_cast(bf16x8, afr);
                const int gsw = (((px >> 1) & 3) << 1) | (py & 1);
                const unsigned pmask = pv ? 0xFFFFFFFFu : 0u;
                const unsigned wdst = pat + q * 128 + (lqv & 1) * 8;
                // the four channel blocks of an m-tile: all four MFMAs first (independent), then the conversions -- issued one by one, every
                // conversion waited out its own MFMA's latency (tools/conv64_stamps.py: 3600 of the producer's 5000 cycles were this chain)
                // the producer's four MFMAs overtake the partner wave's bursts of eight (priority 1): otherwise each batch queues behind a whole
                // burst and the conversions wait -- 9260 -> 8628 cycles per patch (priority 3: the same; the whole producer at 2: 8890)
#ifndef CONV64_PRODUCER_PRIO
#define CONV64_PRODUCER_PRIO 2
#endif
                __builtin_amdgcn_s_setprio(CONV64_PRODUCER_PRIO);
                const f32x4v d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[0]), av, __builtin_bit_cast(f32x4v, bq[0]), 0, 0, 0);
                const f32x4v d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[1]), av, __builtin_bit_cast(f32x4v, bq[1]), 0, 0, 0);
                const f32x4v d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[2]), av, __builtin_bit_cast(f32x4v, bq[2]), 0, 0, 0);
                const f32x4v d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[3]), av, __builtin_bit_cast(f32x4v, bq[3]), 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                // channels nn*16 + 4 lq .. + 3 of pixel q: chunk nn*2 + (lq >> 1), 8-byte half lq & 1.  (nn*2 + h) ^ gsw = (nn*2) ^ (gsw & 6) with
                // the low bit h ^ (gsw & 1) folded into the base; a VECTOR float -> bf16 conversion is two v_cvt_pk_bf16_f32 (element by element
                // hipcc emitted four of them plus two v_perm_b32)
                const unsigned wbase = wdst + ((((lqv >> 1) ^ gsw) & 1) << 4);
                const unsigned g6 = (gsw & 6) << 4;
#define CONV64_PRODUCE_STORE(D, NN)                                                    \
                {                                                                               \
                    uint2 ov = __builtin_bit_cast(uint2, __builtin_convertvector(D, bf16x4));   \
                    asm volatile("" : "+v"(ov.x), "+v"(ov.y)); /* keep the conversion packed */  \
                    ov.x = relu_bf16x2(ov.x) & pmask;                                           \
                    ov.y = relu_bf16x2(ov.y) & pmask;                                           \
                    lds_write8(wbase + (((NN) * 32) ^ g6), ov);                                 \
                }
                CONV64_PRODUCE_STORE(d0, 0)
                CONV64_PRODUCE_STORE(d1, 1)
                CONV64_PRODUCE_STORE(d2, 2)
                CONV64_PRODUCE_STORE(d3, 3)
#undef CONV64_PRODUCE_STORE
            }
        });
#ifdef CONV64_PRODUCER_PRIO_ALL
        __builtin_amdgcn_s_setprio(0);
#endif
        wait_lgkm<0>();
    };

    const int G = gridDim.x, b0 = blockIdx.x;
    const int my_tiles = (a.ntiles - b0 + G - 1) / G;  // tiles b0, b0 + G, ...  (>= 1: the launcher keeps G <= ntiles)
    auto tile_at = [&](int j) { return j < my_tiles ? b0 + j * G : -1; };

    f32x4v acc[4][2];
#ifndef CONV64_PF
#define CONV64_PF 1  // half-taps of fragment reads in flight ahead of the MFMAs (2: three fragment buffers)
#endif
    constexpr int NAF = CONV64_PF + 1;
    uint4 af[NAF][4];

    // ---- prologue: patches 0 and 1 ----
    if constexpr (FUSE) {
        {   // conv1_1 weights [64][32] bf16 -> LDS rows of 64 B (chunk c at c ^ ((row >> 2) & 3)); conv1_1 bias
            if (tid < 256) {
                const int r = tid >> 2, c = tid & 3;
                *reinterpret_cast<uint4 *>(smem + F_W11_OFF + r * 64 + ((c ^ ((r >> 2) & 3)) << 4)) =
                    *reinterpret_cast<const uint4 *>(a.w11 + r * 32 + c * 8);
            } else if (tid < 320) {
                reinterpret_cast<float *>(smem + F_B11_OFF)[tid - 256] = a.b11[tid - 256];
            }
        }
        issue_raw(tile_at(0), 0);
        issue_raw(tile_at(1), 1);
        issue_raw(tile_at(2), 2);
        issue_raw(tile_at(3), 3);
        wait_vmcnt<0>();
        __syncthreads();
        produce(tile_at(0), 0, 0);
        produce(tile_at(1), 1, 1);
        __syncthreads();
    } else {
        issue_patch(tile_at(0), 0);
        issue_patch(tile_at(1), 1);
        wait_vmcnt<0>();
        __syncthreads();                          // patches 0, 1 and the bias are visible (no DMA in flight here)
    }
    if (wq == 1) __builtin_amdgcn_s_barrier();    // stagger

    for (int j = 0; j < my_tiles; ++j) {
        const int buf = j % 3;
        const unsigned boff = buf * PB;
        auto stamp = [&](int k) {
            if (a.stamps && (tid & 255) == 0)
                a.stamps[((size_t)(b0 + j * G) * 2 + wq) * 8 + k] = (k >= 6 && !FUSE) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
        };
        if (!FUSE) stamp(7);
        stamp(0);
        unsigned ar[2][3][2];
#pragma unroll
        for (int yp = 0; yp < 2; ++yp)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                ar[yp][kw][0] = areg[yp][kw] + boff;
                ar[yp][kw][1] = ar[yp][kw][0] ^ 64u;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[i][n] = f32x4v{0.f, 0.f, 0.f, 0.f};

        // half-tap h = 2*tap + s; reads of h+1 are issued before the MFMAs of h
        auto read_half = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            constexpr int t = h >> 1, s = h & 1, kh = t / 3, kw = t % 3;
            const unsigned ad = ar[kh & 1][kw][s];
            static_for<0, 4>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                af[h % NAF][i] = lds_read16<((2 * (i / 2) + kh) * 18 + 8 * (i % 2) + kw) * 128>(ad);
            });
        };
        auto mfma_half = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            constexpr int t = h >> 1, s = h & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const bf16x8 av = __builtin_bit_cast(bf16x8, af[h % NAF][i]);
                    const bf16x8 bv = __builtin_bit_cast(bf16x8, breg[t][s][n]);
                    acc[i][n] = POOL ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i][n], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, av, acc[i][n], 0, 0, 0);
                }
        };
        auto run_halves = [&](auto lo, auto hi) {  // half-taps [lo, hi); the reads of `lo` are already in flight
            constexpr int LO = decltype(lo)::value, HI = decltype(hi)::value;
            static_for<LO, HI>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                if constexpr (h + CONV64_PF < 18) {
                    read_half(std::integral_constant<int, h + CONV64_PF>{});
                    wait_lgkm<4 * CONV64_PF>();
                } else {
                    wait_lgkm<4 * (17 - h)>();
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                mfma_half(hc);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        // ---------------- first half of the patch ----------------
        if (!FUSE && wq == 1) issue_patch(tile_at(j + 2), (j + 2) % 3);
        read_half(std::integral_constant<int, 0>{});
        if constexpr (CONV64_PF > 1) read_half(std::integral_constant<int, 1>{});
        run_halves(std::integral_constant<int, 0>{}, std::integral_constant<int, SPLIT>{});
        if (!FUSE && wq == 1) wait_vmcnt<NPIECE + NST>();  // retires this wave's pieces of patch j+1
        stamp(1);
#ifndef CONV64_FUSE_EPI_FIRST
#define CONV64_FUSE_EPI_FIRST 1
#endif
        // FUSE: the mid-patch barrier comes AFTER the epilogue (below) -- a patch is [18 half-taps + epilogue | barrier | producer | barrier]:
        // the epilogue touches no patch or raw-window buffer (registers -> global), so it may run on either side of the barrier, and the
        // two sides are then 3450 + 1100 and 4400 cycles instead of 3450 and 5600 (tools/conv64_stamps.py): the groups alternate sides,
        // so a patch costs twice the longer one
        if (!(FUSE && CONV64_FUSE_EPI_FIRST)) {
            __builtin_amdgcn_s_barrier();
            stamp(2);
        }
        // ---------------- second half ----------------
        if (!FUSE && wq == 0) issue_patch(tile_at(j + 2), (j + 2) % 3);
        run_halves(std::integral_constant<int, SPLIT>{}, std::integral_constant<int, 18>{});
        if (!FUSE && wq == 0) wait_vmcnt<NPIECE + NST>();
        if (!FUSE) stamp(3);

        // ---------------- epilogue: bias, ReLU, (pool), store ----------------
        {
            const int tile = b0 + j * G;
            const int per_img = a.tiles_y * a.tiles_x;
            const int n_img = tile / per_img, r = tile - n_img * per_img;
            const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
            if (POOL) {
                // lane: channels (2 l15, 2 l15 + 1) of the wave's 32; registers = the 4 pixels of window lq of m-tile 4 wp + i
                float b0v, b1v;
                {
                    const unsigned ba = lds0 + BOFF + (wq * 32 + 2 * l15) * 4;
                    uint2 bb;
                    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(bb) : "v"(ba) : "memory");
                    b0v = __builtin_bit_cast(float, bb.x);
                    b1v = __builtin_bit_cast(float, bb.y);
                }
                const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int widx = (4 * wp + i) * 4 + lq;  // window index in the tile (8 x 8 windows)
                    const int oy = ty * 8 + (widx >> 3), ox = tx * 8 + (widx & 7);
                    float v0 = fmaxf(fmaxf(acc[i][0][0], acc[i][0][1]), fmaxf(acc[i][0][2], acc[i][0][3])) + b0v;
                    float v1 = fmaxf(fmaxf(acc[i][1][0], acc[i][1][1]), fmaxf(acc[i][1][2], acc[i][1][3])) + b1v;
                    bf16x2 o;
                    o[0] = (bf16_t)v0;
                    o[1] = (bf16_t)v1;
                    unsigned ow = __builtin_bit_cast(unsigned, o);
                    if (a.relu) ow = relu_bf16x2(ow);
                    bf16_t *dst = a.out + ((size_t)(n_img * Ho + oy) * Wo + ox) * a.Cout + cc * 64 + wq * 32 + 2 * l15;
                    *reinterpret_cast<unsigned *>(dst) = ow;
                }
            } else {
                // lane: pixel l15 of m-tile 4 wp + i; registers of n-tile n = channels lq*8 + n*4 + (0..3)
                f32x4v bv0, bv1;
                {
                    const unsigned ba = lds0 + BOFF + (wq * 32 + lq * 8) * 4;
                    uint4 x0, x1;
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(x0), "=&v"(x1)
                                 : "v"(ba)
                                 : "memory");
                    bv0 = __builtin_bit_cast(f32x4v, x0);
                    bv1 = __builtin_bit_cast(f32x4v, x1);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int mt = 4 * wp + i;
                    const int y = ty * 16 + 2 * (mt >> 1) + dy, x = tx * 16 + 8 * (mt & 1) + xl;
                    f32x4v u0 = acc[i][0] + bv0, u1 = acc[i][1] + bv1;
                    typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
                    bf16x8v o;
#pragma unroll
                    for (int r2 = 0; r2 < 4; ++r2) {
                        o[r2] = (bf16_t)u0[r2];
                        o[4 + r2] = (bf16_t)u1[r2];
                    }
                    if (a.relu) {
                        typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
                        u32x4r w = __builtin_bit_cast(u32x4r, o);
#pragma unroll
                        for (int r2 = 0; r2 < 4; ++r2) w[r2] = relu_bf16x2(w[r2]);
                        o = __builtin_bit_cast(bf16x8v, w);
                    }
                    if constexpr (!FUSE) {
                        if (a.f8_inv_scale > 0.0f) {  // wave-uniform: the e4m3 input of the fp8 convolution stack (fp8.hip)
                            float q[8];
#pragma unroll
                            for (int r2 = 0; r2 < 4; ++r2) {
                                q[r2] = fminf(fmaxf(u0[r2], 0.0f) * a.f8_inv_scale, 448.0f);
                                q[4 + r2] = fminf(fmaxf(u1[r2], 0.0f) * a.f8_inv_scale, 448.0f);
                            }
                            uint2 o8;
                            o8.x = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
                            o8.x = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], o8.x, true);
                            o8.y = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], 0, false);
                            o8.y = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], o8.y, true);
                            unsigned char *d8 = reinterpret_cast<unsigned char *>(a.out) + ((size_t)(n_img * H + y) * W + x) * a.Cout + cc * 64 +
                                                wq * 32 + lq * 8;
                            *reinterpret_cast<uint2 *>(d8) = o8;
                            continue;
                        }
                    }
                    bf16_t *dst = a.out + ((size_t)(n_img * H + y) * W + x) * a.Cout + cc * 64 + wq * 32 + lq * 8;
                    *reinterpret_cast<bf16x8v *>(dst) = o;
                }
            }
        }
        if constexpr (FUSE) {
            stamp(3);
            if (CONV64_FUSE_EPI_FIRST) {
                __builtin_amdgcn_s_barrier();
                stamp(2);
            }
            // vm-op order of a wave: ... raw(j+3), stores(j), [here], raw(j+4) ...: vmcnt(NST) retires raw window j+3 (used by
            // the NEXT iteration's producer, after this iteration's barrier made every wave's piece visible)
            wait_vmcnt<NST>();
            stamp(6);
            produce(tile_at(j + 2), (j + 2) % 3, (j + 2) & 3);
            issue_raw(tile_at(j + 4), (j + 4) & 3);
        }
        stamp(4);
        __builtin_amdgcn_s_barrier();
        stamp(5);
        if (!FUSE) stamp(6);
    }
    if (wq == 0) __builtin_amdgcn_s_barrier();  // un-stagger
    wait_vmcnt<0>();
}

}  // namespace

bool conv64_eligible(int dtype, int Cin, int Cout, int H, int W) {
    return dtype == GEMM_T_BF16 && Cin == 64 && Cout % 64 == 0 && Cout >= 64 && H % 16 == 0 && W % 16 == 0 && H >= 16 && W >= 16;
}

hipError_t launch_conv64(hipStream_t stream, const void *in, const void *w, const float *bias, void *out, int N, int H, int W, int Cout,
                         int relu, int pool, const void *zero_page, float f8_inv_scale, int wg_cap, unsigned long long *stamps) {
    if (f8_inv_scale > 0.0f && (pool || !relu)) return hipErrorInvalidValue;  // e4m3 output: the non-pool ReLU epilogue only
    if (!conv64_eligible(GEMM_T_BF16, 64, Cout, H, W) || !in || !w || !out || !zero_page || N < 1) return hipErrorInvalidValue;
    if ((int64_t)N * H * W * 64 >= (1ll << 31)) return hipErrorInvalidValue;  // 32-bit element offsets into the input
    Conv64Args a{};
    a.in = reinterpret_cast<const bf16_t *>(in);
    a.w = reinterpret_cast<const bf16_t *>(w);
    a.bias = bias;
    a.out = reinterpret_cast<bf16_t *>(out);
    a.zero_page = zero_page;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.relu = relu;
    a.f8_inv_scale = f8_inv_scale;
    a.stamps = stamps;
    a.tiles_y = H / 16; a.tiles_x = W / 16; a.ntiles = N * a.tiles_y * a.tiles_x;
    const int chunks = Cout / 64;
    int gx = (wg_cap >= 8 ? wg_cap : 256) / chunks;  // one workgroup per CU (all of LDS); capped: leaves CUs to the other stream
    if (gx < 1) gx = 1;
    if (gx > a.ntiles) gx = a.ntiles;
    static LdsAttrMask attr_done[2] = {{0}, {0}};
    auto kern = pool ? conv64_kernel<true, false> : conv64_kernel<false, false>;
    if (hipError_t e = set_max_lds(reinterpret_cast<const void *>(kern), LDS_BYTES, attr_done[pool ? 1 : 0]); e != hipSuccess) return e;
    gemm_debug_note_route("conv64", -1);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)chunks), dim3(512), LDS_BYTES, stream, a);
    return hipGetLastError();
}

// conv1_1 + conv1_2 (+ pool) in one launch from the mean-subtracted bf16 crops img16[n][x][y][3] (k_img_u8_to_bf16): see the
// FUSE note at the top.  w11 = conv1_1 weights [64][32] in
// the k' order (k_repack_conv11_w_fused), S = crop size (multiple of 16), out = pooled NHWC [N][S/2][S/2][64].
hipError_t launch_conv64_fused11(hipStream_t stream, const void *img16, const void *w11, const float *b11, const void *w,
                                 const float *bias, void *out, int N, int S, const void *zero_page, int wg_cap, unsigned long long *stamps) {
    if (!img16 || !w11 || !b11 || !w || !out || !zero_page || N < 1 || S < 16 || (S % 16)) return hipErrorInvalidValue;
    if ((int64_t)N * (S + 4) * (S + 4) * 3 >= (1ll << 31)) return hipErrorInvalidValue;
    {
        const char *gen = getenv("LRCN_FUSE11_GEN");  // 1: this file's alternating kernel (round 3); default: conv64f.hip
        if (!(gen && gen[0] == '1')) return launch_conv64f(stream, img16, w11, w, bias, out, N, S, zero_page, wg_cap, stamps);
    }
    Conv64Args a{};
    a.w = reinterpret_cast<const bf16_t *>(w);
    a.bias = bias;
    a.out = reinterpret_cast<bf16_t *>(out);
    a.zero_page = zero_page;
    a.N = N; a.H = S; a.W = S; a.Cout = 64; a.relu = 1;
    a.tiles_y = S / 16; a.tiles_x = S / 16; a.ntiles = N * a.tiles_y * a.tiles_x;
    a.img16 = reinterpret_cast<const bf16_t *>(img16);
    a.w11 = reinterpret_cast<const bf16_t *>(w11);
    a.b11 = b11;
    a.stamps = stamps;
    int gx = wg_cap >= 8 ? wg_cap : 256;
    if (gx > a.ntiles) gx = a.ntiles;
    static LdsAttrMask attr_done{0};
    auto kern = conv64_kernel<true, true>;
    if (hipError_t e = set_max_lds(reinterpret_cast<const void *>(kern), F_LDS_BYTES, attr_done); e != hipSuccess) return e;
    gemm_debug_note_route("conv64-fused11", -1);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, 1), dim3(512), F_LDS_BYTES, stream, a);
    return hipGetLastError();
}
