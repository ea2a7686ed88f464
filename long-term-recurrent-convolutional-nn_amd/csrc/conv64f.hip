// conv64f.hip -- conv1_1 + conv1_2 + bias + ReLU + 2x2 max-pool of VGG-16 in ONE launch, bf16, gfx950: the second generation of
// conv64.hip's FUSE kernel.                                                              (lrcn.jl:724-726 convx/relux/poolx, twice)
//
// What the first generation (conv64_kernel<true,true>) ran into (tools/conv64_stamps.py): its two wave groups
// ALTERNATE -- one multiplies a patch while the other computes conv1_1 into the patch after next -- so one wave per SIMD feeds the
// matrix pipe at a time, and a wave issues in order: its 4 fragment reads, its wait and its priority switches sit BETWEEN bursts of 8
// MFMAs, not under them (171..218 cycles per half-tap of 128 MFMA cycles even with the SIMD to itself), both sides of the mid-patch
// barrier are ~4000 cycles and a patch costs 8630 cycles against ~5000 at full pipe rate.
// Here every wave does both jobs at once, instruction by instruction:
//   * what a patch is short of is the SIMD's vector ISSUE port, not the matrix pipe: a wave issues in order, every instruction of both
//     waves of a SIMD passes through that port, and the measured patch times of both generations equal the SUM of their issue costs
//     (MFMA 8 cycles whatever its shape, vector ALU 4..7, LDS / wait / s_nop ~4, an LDS-DMA piece ~100).  So:
//     conv1_2 runs on v_mfma_f32_32x32x16_bf16 (half the MFMA instructions of 16x16x32 for the same pipe cycles: a wave's 64 pixels x
//     32 channels = 2 m-tiles of 32 pixels (2 image rows x 16 columns) x 1 n-tile, K = 16 per instruction), and whatever else a wave
//     has to issue is cut into pieces of <= 3..4 instructions and PINNED (sched_barrier) behind the MFMAs: a patch is 72 sub-steps
//     per wave (36 K-steps x 2 m-tiles) of [wait for the m-tile's A fragment | MFMA | LDS read of the same m-tile's fragment two K-steps
//     ahead | one micro-slice of the conv1_1 producer of the NEXT patch];
//   * the fragment is single-buffered (16 registers; the read goes into the registers the MFMA has just consumed) -- the registers
//     that frees are what the producer's rolling state lives in (the conv1_2 weights of all nine taps stay resident: 144);
//   * the producer of an m-tile (16 patch pixels x 64 channels, K = 27 -> 32) is 17 micro-slices over 24 sub-steps: raw-window reads (2) |
//     conv1_1 weight reads | store address + border mask | im2col fragment (2) | MFMA | MFMA + weight reads | convert + ReLU (2) |
//     16-byte store | MFMA | MFMA | the next slot's table read | convert + ReLU (2) | 16-byte store.  Everything that depends only on the
//     lane (which patch pixel, where its raw run starts, where its 16 bytes go in the swizzled patch, which tile borders would zero it)
//     comes from a per-lane TABLE in LDS, filled once per launch: an LDS read instead of ~30 vector-ALU instructions per m-tile.  No inline
//     asm in a slice but the LDS operations themselves, no compare into an SGPR pair, conversions >= 4 sub-steps behind their MFMA: each
//     of those drew an s_nop, and a pad inside an MFMA-paced stream costs 17..43 cycles, not 4.  A wave owns 3 of a patch's 21 m-tiles
//     (waves 5..7: two, and repeat one -- the LDS-operation count per sub-step is what the counted s_waitcnt lgkmcnt(N) of every
//     sub-step is computed from, so it must not depend on the wave);
//   * LDS operations return in order: an operation issued in sub-step k has landed once sub-step k + 5 has waited for ITS fragment
//     (issued in k + 1, behind it), so micro-slices are simply placed >= 5 sub-steps after the reads they consume: no extra waits.
//     Every asm read must be CONSUMED: a result hipcc sees as dead leaves its registers free for other values, and the data lands in them;
//   * conv1_1's bias rides in the K padding: k' = 27, 28, 29 of the im2col fragment hold 1.0 and the weight rows hold the f32 bias cut
//     into three bf16 pieces (hi + mid + lo = the f32 value exactly), so the accumulator input is the constant 0, no bias registers
//     or LDS reads, and a patch pixel outside the image (conv1_2's zero padding) is ONE mask on the fragment: 0 x w + 0 x b = 0;
//   * two patch buffers (multiply j, produce j + 1), ONE barrier per patch; raw windows are DMA'd three patches ahead (4 buffers).
//   * LDS bank behaviour (simulated per instruction for every tap, wave and slot -- tools/conv64f_banks.py -- and visible in the stamps): the
//     patch image is pixel q = py*18 + px at q*128 bytes, 16-byte chunk c (8 channels) at position c ^ ((px & 7) ^ (py & 1)): conflict-free
//     for the 16-lane groups of the consumer's ds_read_b128 at every tap shift AND for the producer's stores, which are ds_write_b128 of
//     8 consecutive channels (the conv1_1 weight rows are permuted so that a lane's results of MFMA 2P and 2P + 1 are channels
//     32 P + 8 lq .. + 7: one 16-byte store instead of two 8-byte ones that conflicted 4-way under conv64.hip's swizzle).
// Raw-window double copy and k' order of the conv1_1 weights: as conv64.hip (FUSE notes there).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // (HIP's uint4 / uint2 are structs: no tied asm operands)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned relu_bf16x2(unsigned w) {  // see conv64.hip
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), s16x2{0, 0}));
}

constexpr int P_BYTES = 21 * 2048;        // 336 pixel rows of 128 B (324 used)
constexpr int P1_OFF = 65536;             // the second patch buffer
constexpr int RAW_ROW = 260;              // a raw-window row: copy A (128 B written, 120 + 2 read back), copy B = the same shifted by one element at + 128, 4 B pad:
constexpr int RAWB = 128;                 //   65 dwords per row put the 16 rows a read instruction touches on 16 different banks (pitch 144 / 272: 2-way conflicts);
constexpr int RAW = 24 * RAW_ROW;         //   one 256-byte LDS-DMA piece fills both copies of a row (lanes 0..31 | 32..63)
constexpr int RAW_OFF0 = P_BYTES;         // raw buffers 0..2 between the patch buffers (43008 + 3 x 6240 <= 65536)
constexpr int RAW3_OFF = P1_OFF + P_BYTES;
constexpr int W11_OFF = RAW3_OFF + RAW;   // conv1_1 weights: 64 rows of 64 B (row = MFMA nn * 16 + fragment row), chunk c at c ^ ((row >> 1) & 3)
constexpr int BIAS_OFF = W11_OFF + 64 * 64;
constexpr int TAB_OFF = BIAS_OFF + 256;   // per-lane producer table: [slot 0..2][thread 0..511] x 16 bytes
constexpr int TAB_SLOT = 512 * 16;
constexpr int LDS_BYTES = TAB_OFF + 3 * TAB_SLOT;
static_assert(RAW_OFF0 + 3 * RAW <= P1_OFF && LDS_BYTES <= 160 * 1024, "LDS image");

struct Conv64fArgs {
    const bf16_t *img16;  // mean-subtracted crops in a 2-pixel zero frame [n][S + 4][S + 4][3] (k_img_u8_to_bf16)
    const bf16_t *w11;    // conv1_1 weights [64][32] in k' order with the bias pieces at k' = 27..29 (k_repack_conv11_w_fused)
    const bf16_t *w;      // conv1_2 weights [64][9][64]
    const float *bias;    // conv1_2 bias
    bf16_t *out;          // pooled NHWC [N][S/2][S/2][64]
    const void *zero_page;
    int N, S, tiles, ntiles;
    unsigned inv_per_img, inv_tiles;  // fastdiv_inv(tiles * tiles), fastdiv_inv(tiles): tile -> (image, ty, tx) without dividing
    unsigned long long *stamps;       // LRCN_STAMPS=f: 8 x uint64 per (tile, wave group), tools/conv64_stamps.py
};

template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int OFF> __device__ __forceinline__ u32x4 lds_read16(unsigned addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
#define PIN() __builtin_amdgcn_sched_barrier(0)

// ---- the static schedule of a patch: 72 sub-steps s = 2 ksn + mi (K-step ksn = 4 tap + kk of 36, m-tile mi of the wave's two) ----
constexpr int NSUB = 72;
// producer micro-slice of sub-step s: m-tile slot s / 24, event r = s % 24
constexpr int R_RAW2 = 0, R_RAW16 = 1, R_W01 = 2, R_WBASE = 3, R_AV0 = 6, R_AV1 = 7, R_MM0 = 8, R_MM1 = 9, R_CV0 = 13, R_CV1 = 14, R_ST01 = 15, R_MM2 = 16,
              R_MM3 = 17, R_TAB = 18, R_CV2 = 21, R_CV3 = 22, R_ST23 = 23;
constexpr int np_of(int s) {  // LDS operations the micro-slice of sub-step s issues
    const int r = s % 24;
    return r == R_RAW2 ? 2 : r == R_RAW16 ? 3 : (r == R_W01 || r == R_MM1) ? 2 : (r == R_ST01 || r == R_ST23 || r == R_TAB) ? 1 : 0;
}
// LDS operations issued after the fragment read that sub-step s consumes (program order of a sub-step: [wait | MFMA | read for s + 4 |
// micro-slice]: everything sub-step s - 4 issued after its own read, then three whole sub-steps)
constexpr int wait_of(int s) {
    int n = 0;
    if (s < 4) {
        n = 3 - s;  // the patch prologue issues the fragment reads of sub-steps 0..3 back to back
        for (int k = 0; k < s; ++k) n += 1 + np_of(k);
    } else {
        n = np_of(s - 4);
        for (int k = s - 3; k < s; ++k) n += (k + 4 < NSUB ? 1 : 0) + np_of(k);
    }
    return n > 15 ? 15 : n;
}
// a micro-slice's LDS operations have landed once sub-step k + 5 has waited for ITS fragment (issued in k + 1, behind them): consumers sit
// >= 5 sub-steps behind their reads (the table entry of the next slot is read in R_TAB); conversions sit >= 4 sub-steps behind their MFMA
static_assert(R_AV0 - R_RAW16 >= 5 && R_MM0 - R_W01 >= 5 && R_MM2 - R_MM1 >= 5 && 24 - R_TAB >= 5 && R_TAB > R_MM1, "micro-slices too close to their reads");
static_assert(R_CV0 - R_MM0 >= 4 && R_CV1 - R_MM1 >= 4 && R_CV2 - R_MM2 >= 4 && R_CV3 - R_MM3 >= 4, "conversions too close to their MFMAs");

template <bool STAMPS> __global__ __launch_bounds__(512) void conv64f_kernel(const Conv64fArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 3;   // pixel group: image rows 4 wp .. 4 wp + 3 of the 16 x 16 tile (window rows 2 wp, 2 wp + 1)
    const int wq = wave >> 2;  // channel group (32 channels)
    const int S = a.S, So = S >> 1;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

    // ---- conv1_2 weights: B fragments (32 channels x K 16) of all 9 taps x 4 K-steps, resident in registers (lane = channel l31, K half lh) ----
    uint4 breg[9][4];
    unsigned ar[3];     // A-fragment read addresses in patch buffer 0, [kw], kh even, K-step 0 (kh odd: ^ 16, K-step kk: ^ 32 kk)
    unsigned out_lane;  // byte offset of this lane's pooled output inside a tile's 8 x 8 windows x 64 channels
    unsigned tab_addr;  // this thread's entry of table slot 0
    unsigned lastm;     // all ones in the lanes of K chunk lq = 3 (the 9th values + the bias ones of the im2col fragment)
    {
        const int lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
        const bf16_t *wr = a.w + (size_t)(wq * 32 + l31) * 576 + lh * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) breg[t][kk] = *reinterpret_cast<const uint4 *>(wr + t * 64 + kk * 16);
        if (tid < 64) reinterpret_cast<float *>(smem + BIAS_OFF)[tid] = a.bias ? a.bias[tid] : 0.0f;
        if (tid >= 256) {  // conv1_1 weights -> LDS rows of 64 B: row nn*16 + r = channel 32 (nn >> 1) + 8 (r >> 2) + 4 (nn & 1) + (r & 3)
            const int R = (tid - 256) >> 2, c = tid & 3, nn = R >> 4, r = R & 15;
            const int ch = 32 * (nn >> 1) + 8 * (r >> 2) + 4 * (nn & 1) + (r & 3);
            *reinterpret_cast<uint4 *>(smem + W11_OFF + R * 64 + ((c ^ ((R >> 1) & 3)) << 4)) = *reinterpret_cast<const uint4 *>(a.w11 + ch * 32 + c * 8);
        }
        // A rows of an m-tile (32 pixels = 2 image rows x 16 columns = 8 pool windows): row l31 = 4 win + 2 dy + dx -- the four accumulator
        // registers r of a group g are rows 8 g + 4 lh + r = ONE window (pool = max of 4 registers).  Window win sits at window column
        // kWinPos[win]: the LDS serves a ds_read_b128 in the lane groups {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} (+ 32), i.e. windows
        // {0, 3, 5, 6} and {1, 2, 4, 7}; each group gets 8 adjacent columns x 2 rows, which the patch swizzle spreads over all 16 bank slots
        // at every tap shift.  (kWinPos[2 g + 1] = kWinPos[2 g] ^ 4: the epilogue's store offset of lane half lh is one XOR.)
        const int win = l31 >> 2, dy = (l31 >> 1) & 1, dx = l31 & 1;
        const int wpos = (0x73261540 >> (4 * win)) & 7;  // kWinPos[win] = {0, 4, 5, 1, 6, 2, 3, 7}
        const int x = 2 * wpos + dx;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int gsw = ((x + kw) & 7) ^ (dy & 1);
            ar[kw] = lds0 + ((4 * wp + dy) * 18 + x) * 128 + ((lh ^ gsw) << 4);
        }
        // epilogue: lane = channel l31; accumulator group g of m-tile mi = window 2 g + lh of window row 2 wp + mi
        out_lane = (unsigned)((wq * 32 + l31) * 2) | ((unsigned)lh << 9);  // (lh << 9) = 4 window columns x 128 B, XORed in per group
        // ---- the producer's per-lane table: slot sl -> m-tile wave + 8 sl (waves 5..7, slot 2: m-tile wave + 8 once more) ----
        const int l15 = lane & 15, lq = lane >> 4;  // the PRODUCER's lane roles (16x16x32: pixel l15 of an m-tile of 16, K chunk lq)
        tab_addr = lds0 + TAB_OFF + tid * 16;
        lastm = lq == 3 ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const int mt = (wave + 8 * sl < 21) ? wave + 8 * sl : wave + 8;
            const int q = mt * 16 + l15;
            const int py = q / 18, px = q - 18 * py;
            const int lsel = lq < 2 ? lq : 2;
            // run kw of pixel (py, px): window row px + kw, 9 bf16 from byte 6 py of copy A = byte 6 py - 2 of copy B (dword-aligned for odd py)
            const unsigned rbl = px * RAW_ROW + 6 * py;
            const unsigned rbs = rbl + lsel * RAW_ROW + ((py & 1) ? RAWB - 2 : 0);
            const int gsw = (px & 7) ^ (py & 1);
            // a lane's results of MFMA 2P, 2P + 1 = channels 32 P + 8 lq .. + 7 of pixel q = chunk 4 P + lq, at position chunk ^ gsw
            const unsigned wb = q * 128 + ((lq ^ gsw) << 4);
            const unsigned flags = (py == 0 ? 1u : 0u) | (py == 17 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == 17 ? 8u : 0u) | (q >= 324 ? 16u : 0u);
            u32x4 e;
            e.x = rbs;
            e.y = rbl;
            e.z = wb | (flags << 20);  // wb < 2^16
            e.w = lds0 + W11_OFF + l15 * 64 + ((lq ^ ((l15 >> 1) & 3)) << 4);  // conv1_1 weight fragment rows nn*16 + l15 (nn KiB apart)
            *reinterpret_cast<u32x4 *>(smem + TAB_OFF + sl * TAB_SLOT + tid * 16) = e;
        }
    }

    const int G = gridDim.x, b0 = blockIdx.x;
    const int my_tiles = (a.ntiles - b0 + G - 1) / G;  // tiles b0, b0 + G, ...  (>= 1: the launcher keeps G <= ntiles)
    auto tile_at = [&](int j) { return j < my_tiles ? b0 + j * G : -1; };
    const int per_img = a.tiles * a.tiles;
    struct TileXY {
        int n, ty, tx;
    };
    auto decode_tile = [&](int t) {  // scalar: s_mul_hi_u32 by host-made reciprocals (0 = divisor 1)
        TileXY d;
        d.n = a.inv_per_img ? (int)__umulhi((unsigned)t, a.inv_per_img) : t;
        const int r = t - d.n * per_img;
        d.ty = a.inv_tiles ? (int)__umulhi((unsigned)r, a.inv_tiles) : r;
        d.tx = r - d.ty * a.tiles;
        return d;
    };
    auto raw_off = [&](int rbuf) { return rbuf < 3 ? RAW_OFF0 + rbuf * RAW : RAW3_OFF; };
    // ---- raw-window DMA: three 256-byte pieces per wave: window row wrow = wave + 8 k (rows 20..23 are dummies that keep the per-wave
    // count uniform), lanes 0..31 = elements [E, E + 64) of the framed image row (copy A), lanes 32..63 = [E + 1, E + 65) (copy B: the same
    // data shifted by one bf16, so that a run starting at an odd element is dword-aligned there; conv64.hip issue_raw).  The crops carry a
    // 2-pixel zero frame (k_img_u8_to_bf16): conv1_1's zero padding is read as data ----
    // (addresses: ONE scalar element offset per tile, a scalar step per row, the lane's 4 bytes as a 32-bit vector offset -- three 64-bit
    // multiply-adds per piece were a fifth of a patch's scalar instructions)
    const unsigned raw_voff = (unsigned)((2 * (tid & 31) + ((tid >> 5) & 1)) * 2);
    auto issue_raw = [&](int tile, int rbuf) {
        const bool live = tile >= 0;
        const TileXY d = decode_tile(live ? tile : 0);
        const int SP = S + 4;
        const int ro = raw_off(rbuf);
        // image row 16 tx - 2 + wrow = framed row 16 tx + wrow; image column 16 ty - 2 = framed column 16 ty
        const unsigned e0 = ((unsigned)(d.n * SP + 16 * d.tx + wave) * (unsigned)SP + 16u * (unsigned)d.ty) * 3u;  // < 2^31 (launcher)
        const char *base = reinterpret_cast<const char *>(a.img16) + (size_t)e0 * 2;
        const size_t step = (size_t)(8 * SP * 3 * 2);  // 8 window rows further
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int wrow = wave + 8 * k;
            const bool ok = live && wrow < 20;
            const char *src = ok ? base + k * step : reinterpret_cast<const char *>(a.zero_page);  // wave-uniform
            __builtin_amdgcn_global_load_lds((glb_void *)(src + raw_voff), (lds_void *)(smem + ro + wrow * RAW_ROW), 4, 0, 0);
        }
    };

    // ---- the conv1_1 producer, in micro-slices.  No inline asm but the LDS operations themselves, no compare into an SGPR pair, every
    // conversion >= 4 sub-steps behind its MFMA: each of those made hipcc pad with an s_nop, and a pad inside an MFMA-paced stream costs
    // 17..43 cycles of the wave's time (MI355X_MICROARCH.md), not 4 ----
    u32x4 p_tab;                    // this slot's table entry
    u32x2 p_runl, p_runh;           // the first 8 values of the lane's 9-value run
    unsigned p_n9a, p_n9b, p_n9c;   // the 9th values of the three runs
    u32x4 p_av, p_w0, p_w1, p_w2, p_w3;
    f32x4v p_d0, p_d1, p_d2, p_d3;
    u32x4 p_ov;                     // 8 converted channels on their way to LDS
    unsigned p_rbl = 0, p_wbase = 0, p_mask = 0;
    unsigned p_raw = 0, p_pat = 0;  // raw window and patch buffer of the patch being produced (LDS byte addresses; wave-uniform)
    unsigned p_edge = 0;            // which table flags zero a pixel of this tile (<< 20)
    TileXY t_cur = {0, 0, 0}, t_nxt = {0, 0, 0};  // the tile being multiplied / produced (decoded once: the epilogue uses t_cur)
    auto p_set_tile = [&](int tile, int rbuf, unsigned pat) {
        const TileXY d = decode_tile(tile >= 0 ? tile : 0);  // no tile left: produce garbage nobody reads (keeps the LDS-operation count uniform)
        t_cur = t_nxt;
        t_nxt = d;
        p_edge = ((d.ty == 0 ? 1u : 0u) | (d.ty == a.tiles - 1 ? 2u : 0u) | (d.tx == 0 ? 4u : 0u) | (d.tx == a.tiles - 1 ? 8u : 0u) | 16u) << 20;
        p_raw = lds0 + raw_off(rbuf);
        p_pat = pat;
    };
    auto p_tab_read = [&](auto slc) { p_tab = lds_read16<decltype(slc)::value * TAB_SLOT>(tab_addr); };
    auto p_raw2 = [&]() {
        const unsigned rbs = p_raw + p_tab.x;
        p_rbl = p_raw + p_tab.y;
        asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3" : "=&v"(p_runl), "=&v"(p_runh) : "v"(rbs) : "memory");
    };
    auto p_raw16 = [&]() {
        asm volatile("ds_read_u16 %0, %3 offset:16\n\tds_read_u16 %1, %3 offset:276\n\tds_read_u16 %2, %3 offset:536"
                     : "=&v"(p_n9a), "=&v"(p_n9b), "=&v"(p_n9c)
                     : "v"(p_rbl)
                     : "memory");
    };
    auto p_w01_reads = [&]() { asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(p_w0), "=&v"(p_w1) : "v"(p_tab.w) : "memory"); };
    auto p_w23_reads = [&]() {
        asm volatile("ds_read_b128 %0, %2 offset:2048\n\tds_read_b128 %1, %2 offset:3072" : "=&v"(p_w2), "=&v"(p_w3) : "v"(p_tab.w) : "memory");
    };
    auto p_wbase_mask = [&]() {
        p_wbase = p_pat + (p_tab.z & 0xFFFFu);
        p_mask = (p_tab.z & p_edge) ? 0u : 0xFFFFFFFFu;
    };
    // the im2col fragment of 16 pixels (lane group lq < 3: the first 8 values of run kw = lq; 3: the 9th values + the bias ones):
    // (lastm & special) | (~lastm & run) is one v_bfi_b32 each
    auto bfi = [&](unsigned m, unsigned x, unsigned y) { return (m & x) | (~m & y); };
    auto p_av0 = [&]() {
        p_av.x = bfi(lastm, p_n9a | (p_n9b << 16), p_runl.x) & p_mask;
        p_av.y = bfi(lastm, p_n9c | 0x3F800000u, p_runl.y) & p_mask;
    };
    auto p_av1 = [&]() {
        p_av.z = bfi(lastm, 0x3F803F80u, p_runh.x) & p_mask;
        p_av.w = p_runh.y & ~lastm & p_mask;
    };
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
    auto p_mm = [&](const u32x4 &w) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, p_av), zero4, 0, 0, 0); };
    auto cvt2 = [&](float lo, float hi) {  // one v_cvt_pk_bf16_f32 + ReLU on the packed pair (one v_pk_max_i16)
        return relu_bf16x2(__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{lo, hi}, bf16x2)));
    };
    auto p_cvt_lo = [&](const f32x4v &d) {
        p_ov.x = cvt2(d[0], d[1]);
        p_ov.y = cvt2(d[2], d[3]);
    };
    auto p_cvt_hi = [&](const f32x4v &d) {
        p_ov.z = cvt2(d[0], d[1]);
        p_ov.w = cvt2(d[2], d[3]);
    };
    auto p_store = [&](int P) {  // 16 bytes = channels 32 P + 8 lq .. + 7 into the swizzled patch: chunk 4 P + lq, i.e. position ^ 4 for P = 1
        const unsigned ad = P ? (p_wbase ^ 64u) : p_wbase;
#ifdef CONV64F_NOSTORE  // timing experiment only (wrong results)
        asm volatile("" ::"v"(ad), "v"(p_ov));
#else
        asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(p_ov) : "memory");
#endif
    };
    auto produce_all = [&]() {  // the whole patch, one slice after the other (prologue only)
        static_for<0, 3>([&](auto slc) {
            p_tab_read(slc);
            wait_lgkm<0>();
            PIN();  // (the arithmetic on the loaded values must not be scheduled above the wait: for hipcc they exist since the asm)
            p_raw2();
            p_raw16();
            p_w01_reads();
            p_w23_reads();
            p_wbase_mask();
            wait_lgkm<0>();
            PIN();
            p_av0();
            p_av1();
            p_d0 = p_mm(p_w0);
            p_d1 = p_mm(p_w1);
            p_d2 = p_mm(p_w2);
            p_d3 = p_mm(p_w3);
            p_cvt_lo(p_d0);
            p_cvt_hi(p_d1);
            p_store(0);
            p_cvt_lo(p_d2);
            p_cvt_hi(p_d3);
            p_store(1);
        });
        wait_lgkm<0>();
    };

    // ---- prologue: raw windows 0..2, patch 0 ----
    issue_raw(tile_at(0), 0);
    issue_raw(tile_at(1), 1);
    issue_raw(tile_at(2), 2);
    wait_vmcnt<0>();
    __syncthreads();  // raw windows, conv1_1 weights, the bias and the table are visible
    p_set_tile(tile_at(0), 0, lds0);
    produce_all();
    __syncthreads();

    // slot 0's table entry of the first patch; later ones are read in the last slot of the patch before.  (Every asm read must be
    // CONSUMED: a result hipcc sees as dead leaves its registers free for other values, and the data lands in them later.)
    p_tab_read(std::integral_constant<int, 0>{});
    wait_lgkm<0>();
    f32x16 acc[2];
    u32x4 af[4];
    unsigned toggle = P1_OFF;  // added to the fragment addresses after every patch: buffer 0 -> 1 -> 0 ...
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int j = 0; j < my_tiles; ++j) {
        const int tile = b0 + j * G;
        auto stamp = [&](int k) {
            if (STAMPS && (tid & 255) == 0)  // (a template parameter: the tests of a run-time pointer were 50 instructions per patch)
                a.stamps[((size_t)tile * 2 + wq) * 8 + k] = k >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
        };
        stamp(7);
        stamp(0);
        issue_raw(tile_at(j + 3), (j + 3) & 3);
        p_set_tile(tile_at(j + 1), (j + 1) & 3, lds0 + ((j & 1) ? 0 : P1_OFF));

        unsigned ar1 = 0;
        auto read_frag = [&](auto sc) {  // the A fragment sub-step s consumes: m-tile s & 1 of K-step s >> 1
            constexpr int s = decltype(sc)::value;
            constexpr int mi = s & 1, ksn = s >> 1, t = ksn >> 2, kk = ksn & 3, kh = t / 3, kw = t % 3;
            constexpr unsigned flip = (unsigned)(16 * (kh & 1) + 32 * kk);
            if constexpr (flip != 0 && mi == 0) ar1 = ar[kw] ^ flip;  // one XOR per K-step, not one per read
            af[s & 3] = lds_read16<((2 * mi + kh) * 18 + kw) * 128>(flip ? ar1 : ar[kw]);
        };
        read_frag(std::integral_constant<int, 0>{});
        read_frag(std::integral_constant<int, 1>{});
        read_frag(std::integral_constant<int, 2>{});
        read_frag(std::integral_constant<int, 3>{});
        PIN();

        static_for<0, NSUB>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int mi = s & 1, ksn = s >> 1, t = ksn >> 2, kk = ksn & 3;
#ifdef CONV64F_NOWAIT  // timing experiment only (wrong results): what the counted waits cost
            wait_lgkm<15>();
#else
            wait_lgkm<wait_of(s)>();
#endif
            PIN();  // nothing moves above the wait (tying af[] to the asm instead makes hipcc pad every MFMA behind it with an s_nop)
            acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[s & 3]), __builtin_bit_cast(bf16x8, breg[t][kk]),
                                                              ksn == 0 ? zero16 : acc[mi], 0, 0, 0);
            PIN();
            if constexpr (s + 4 < NSUB) read_frag(std::integral_constant<int, s + 4>{});  // into the registers the MFMA has just read
            PIN();  // (a producer MFMA directly behind the big one would be padded with an s_nop)
#ifdef CONV64F_NOPROD  // timing experiment only: the K loop without the producer
            constexpr int r = -1, sl = 0;
            (void)sl;
#else
            constexpr int r = s % 24, sl = s / 24;
#endif
            if constexpr (r == R_RAW2) p_raw2();
            if constexpr (r == R_RAW16) p_raw16();
            if constexpr (r == R_W01) p_w01_reads();
            if constexpr (r == R_WBASE) p_wbase_mask();
            if constexpr (r == R_AV0) p_av0();
            if constexpr (r == R_AV1) p_av1();
            if constexpr (r == R_MM0) p_d0 = p_mm(p_w0);
            if constexpr (r == R_MM1) {
                p_d1 = p_mm(p_w1);
                p_w23_reads();
            }
            if constexpr (r == R_CV0) p_cvt_lo(p_d0);
            if constexpr (r == R_CV1) p_cvt_hi(p_d1);
            if constexpr (r == R_ST01) p_store(0);
            if constexpr (r == R_MM2) p_d2 = p_mm(p_w2);
            if constexpr (r == R_MM3) p_d3 = p_mm(p_w3);
            if constexpr (r == R_TAB) p_tab_read(std::integral_constant<int, (sl < 2 ? sl + 1 : 0)>{});  // slot 2: slot 0's entry for the NEXT patch
            if constexpr (r == R_CV2) p_cvt_lo(p_d2);
            if constexpr (r == R_CV3) p_cvt_hi(p_d3);
            if constexpr (r == R_ST23) p_store(1);
            if constexpr (s == 35) stamp(1);
            PIN();
        });
        stamp(2);
        stamp(3);
        wait_vmcnt<0>();  // this wave's pieces of raw window j + 3 (issued a patch ago) and the stores of patch j - 1: long done

        // ---------------- epilogue: pool, bias, ReLU, store ----------------
        {
            const TileXY d = t_cur;
            // lane: channel l31 of the wave's 32; accumulator registers 4 g .. 4 g + 3 of m-tile mi = the 4 pixels of window 2 g + lh
            float bv;
            {
                const unsigned ba = lds0 + BIAS_OFF + wq * 128 + ((out_lane >> 1) & 31) * 4;  // (channel l31 from the lane's output offset)
                unsigned bb;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(bb) : "v"(ba) : "memory");
                bv = __builtin_bit_cast(float, bb);
            }
            // this tile's 8 x 8 windows x 64 channels; window row 2 wp + mi as a scalar base, window column kWinPos[2 g + lh] = kWinPos[2 g] ^ 4 lh
            unsigned char *tile0 = reinterpret_cast<unsigned char *>(a.out + ((size_t)(d.n * So + d.ty * 8 + 2 * wp) * So + d.tx * 8) * 64);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                unsigned char *row = tile0 + (size_t)mi * So * 128;
#pragma unroll
                for (int g = 0; g < 4; g += 2) {  // two windows per conversion: v_max3 + v_max + v_add each, ONE v_cvt_pk_bf16_f32 and ONE packed ReLU for both
                    constexpr int kPos2g[4] = {0, 5, 6, 3};  // kWinPos[2 g]
                    const float v0 = fmaxf(fmaxf(fmaxf(acc[mi][4 * g], acc[mi][4 * g + 1]), acc[mi][4 * g + 2]), acc[mi][4 * g + 3]) + bv;
                    const float v1 = fmaxf(fmaxf(fmaxf(acc[mi][4 * g + 4], acc[mi][4 * g + 5]), acc[mi][4 * g + 6]), acc[mi][4 * g + 7]) + bv;
                    const unsigned ow = relu_bf16x2(__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{v0, v1}, bf16x2)));
                    // (out_lane's channel part is < 128: the XOR touches the window-column bits only)
                    *reinterpret_cast<unsigned short *>(row + (out_lane ^ (unsigned)(kPos2g[g] * 128))) = (unsigned short)ow;
                    *reinterpret_cast<unsigned short *>(row + (out_lane ^ (unsigned)(kPos2g[g + 1] * 128))) = (unsigned short)(ow >> 16);
                }
            }
        }
        // next patch: the other buffer
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) ar[kw] += toggle;
        toggle = 0u - toggle;
        wait_lgkm<0>();  // this wave's slices of patch j + 1 are in LDS
        stamp(4);
        __builtin_amdgcn_s_barrier();  // every wave has produced its slices of patch j + 1 and finished reading patch j
        stamp(5);
        stamp(6);
    }
    wait_vmcnt<0>();
}

}  // namespace

// conv1_1 + conv1_2 + pool in one launch from the mean-subtracted bf16 crops; arguments as launch_conv64_fused11 (conv64.hip), w11 from
// k_repack_conv11_w_fused WITH the bias (the bias pieces ride in the K padding).
hipError_t launch_conv64f(hipStream_t stream, const void *img16, const void *w11, const void *w, const float *bias, void *out, int N, int S,
                          const void *zero_page, int wg_cap, unsigned long long *stamps) {
    if (!img16 || !w11 || !w || !out || !zero_page || N < 1 || S < 16 || (S % 16)) return hipErrorInvalidValue;
    if ((int64_t)N * (S + 4) * (S + 4) * 3 >= (1ll << 31)) return hipErrorInvalidValue;
    Conv64fArgs a{};
    a.img16 = reinterpret_cast<const bf16_t *>(img16);
    a.w11 = reinterpret_cast<const bf16_t *>(w11);
    a.w = reinterpret_cast<const bf16_t *>(w);
    a.bias = bias;
    a.out = reinterpret_cast<bf16_t *>(out);
    a.zero_page = zero_page;
    a.N = N; a.S = S; a.tiles = S / 16; a.ntiles = N * a.tiles * a.tiles;
    a.inv_per_img = fastdiv_inv((unsigned)(a.tiles * a.tiles));
    a.inv_tiles = fastdiv_inv((unsigned)a.tiles);
    a.stamps = stamps;
    int gx = wg_cap >= 8 ? wg_cap : 256;  // one workgroup per CU; capped: leaves CUs to the other stream
    if (gx > a.ntiles) gx = a.ntiles;
    static LdsAttrMask attr_done[2] = {{0}, {0}};
    auto kern = stamps ? conv64f_kernel<true> : conv64f_kernel<false>;
    if (hipError_t e = set_max_lds(reinterpret_cast<const void *>(kern), LDS_BYTES, attr_done[stamps ? 1 : 0]); e != hipSuccess) return e;
    gemm_debug_note_route("conv64-fused11", -1);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, 1), dim3(512), LDS_BYTES, stream, a);
    return hipGetLastError();
}
