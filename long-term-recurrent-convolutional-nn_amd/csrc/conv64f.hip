// conv64f.hip -- conv1_1 + conv1_2 + bias + ReLU + 2x2 max-pool of VGG-16 in ONE launch, bf16, gfx950: the second generation of
// conv64.hip's FUSE kernel.                                                              (lrcn.jl:724-726 convx/relux/poolx, twice)
//
// What the first generation (conv64_kernel<true,true>) ran into (tools/conv64_stamps.py): its two wave groups
// ALTERNATE -- one multiplies a patch while the other computes conv1_1 into the patch after next -- so one wave per SIMD feeds the
// matrix pipe at a time, and a wave issues in order: its 4 fragment reads, its wait and its priority switches sit BETWEEN bursts of 8
// MFMAs, not under them (171..218 cycles per half-tap of 128 MFMA cycles even with the SIMD to itself), both sides of the mid-patch
// barrier are ~4000 cycles and a patch costs 8630 cycles against ~5000 at full pipe rate.
// Here every wave does both jobs at once, instruction by instruction:
//   * a v_mfma_f32_16x16x32_bf16 holds the SIMD's vector issue for 8 of its 16 cycles; whatever a wave has to issue besides MFMAs is
//     cut into pieces of <= 3..4 instructions and PINNED (sched_barrier) into those gaps: a patch is 72 sub-steps per wave
//     (18 half-taps x 4 m-tiles) of [wait for the m-tile's A fragment | MFMA n-tile 0 | one micro-slice of the conv1_1 producer of the
//     NEXT patch | MFMA n-tile 1 | LDS read of the same m-tile's fragment of the NEXT half-tap];
//   * the fragment is single-buffered (16 registers; the read goes into the registers the two MFMAs just consumed) -- the registers
//     that frees are what the producer's rolling state lives in (the conv1_2 weights of all nine taps stay resident: 144);
//   * the producer of an m-tile (16 patch pixels x 64 channels, K = 27 -> 32) is 20 micro-slices: table read | raw-window reads |
//     conv1_1 weight reads | store address | border mask | im2col fragment (2) | MFMA | MFMA + weight reads | 2 x (convert + ReLU |
//     address + store) | MFMA | MFMA | 2 x (...).  Everything that depends only on the lane (which patch pixel, where its raw run
//     starts, where its 8 bytes go in the swizzled patch, which tile borders would zero it) comes from a per-lane TABLE in LDS, filled
//     once per launch: an LDS read instead of ~30 vector-ALU instructions per m-tile -- the vector issue port, not the matrix pipe, is
//     what a patch is short of.  A wave owns 3 of a patch's 21 m-tiles (waves 5..7: two, and repeat one -- the LDS-operation count per
//     sub-step is what the counted s_waitcnt lgkmcnt(N) of every sub-step is computed from, so it must not depend on the wave);
//   * LDS operations return in order: an operation issued in sub-step k has landed once sub-step k + 4 has waited for ITS fragment
//     (issued at the end of k), so micro-slices are simply placed >= 5 sub-steps after the reads they consume: no extra waits;
//   * conv1_1's bias rides in the K padding: k' = 27, 28, 29 of the im2col fragment hold 1.0 and the weight rows hold the f32 bias cut
//     into three bf16 pieces (hi + mid + lo = the f32 value exactly), so the accumulator input is the constant 0, no bias registers
//     or LDS reads, and a patch pixel outside the image (conv1_2's zero padding) is ONE mask on the fragment: 0 x w + 0 x b = 0;
//   * two patch buffers (multiply j, produce j + 1), ONE barrier per patch; raw windows are DMA'd three patches ahead (4 buffers).
// LDS image of a patch, raw-window double copy, k' order of the conv1_1 weights: as conv64.hip (FUSE notes there).
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
constexpr int RAW_ROW = 144;              // pitch of a raw-window row (conv64.hip issue_raw)
constexpr int RAWB = 24 * RAW_ROW;        // the copy shifted by one element
constexpr int RAW = 2 * RAWB;             // 6912
constexpr int RAW_OFF0 = P_BYTES;         // raw buffers 0..2 between the patch buffers (43008 + 3 x 6912 = 63744 <= 65536)
constexpr int RAW3_OFF = P1_OFF + P_BYTES;
constexpr int W11_OFF = RAW3_OFF + RAW;   // conv1_1 weights [64][32] bf16, 64-byte rows, chunk c at c ^ ((row >> 2) & 3)
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
__device__ __forceinline__ void lds_write8(unsigned addr, u32x2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
#define PIN() __builtin_amdgcn_sched_barrier(0)

// ---- the static schedule of a patch: 72 sub-steps s = 4 h + i (half-tap h, m-tile i of the wave) ----
constexpr int NSUB = 72;
// producer micro-slice of sub-step s: m-tile slot s / 24, event r = s % 24
constexpr int R_RAW2 = 0, R_RAW16 = 1, R_W01 = 2, R_WBASE = 3, R_MASK = 4, R_AV0 = 6, R_AV1 = 7, R_MM0 = 8, R_MM1 = 9, R_ST0A = 10, R_ST0B = 11,
              R_ST1A = 12, R_ST1B = 13, R_MM2 = 14, R_MM3 = 15, R_ST2A = 16, R_ST2B = 17, R_ST3A = 18, R_ST3B = 19;
constexpr int np_of(int s) {  // LDS operations the micro-slice of sub-step s issues
    const int r = s % 24;
    return r == R_RAW2 ? 2 : r == R_RAW16 ? 3 : (r == R_W01 || r == R_MM1) ? 2 : (r == R_ST0B || r == R_ST1B || r == R_ST2B || r == R_ST3B) ? 1 : r == R_ST3A ? 1 : 0;
}
// LDS operations issued after the fragment read that sub-step s consumes (program order of a sub-step: [wait | MFMA | micro-slice | MFMA |
// read for s + 4]: the read of s was the LAST operation of sub-step s - 4)
constexpr int wait_of(int s) {
    int n = 0;
    if (s < 4) {
        n = 3 - s;  // the patch prologue issues the fragment reads of sub-steps 0..3 back to back
        for (int k = 0; k < s; ++k) n += 1 + np_of(k);
    } else {
        for (int k = s - 3; k < s; ++k) n += (k + 4 < NSUB ? 1 : 0) + np_of(k);
    }
    return n > 15 ? 15 : n;
}
// every micro-slice consumes reads issued >= 5 sub-steps earlier (see the header); the table of the next slot is read in R_ST3A
static_assert(R_AV0 - R_RAW16 >= 5 && R_MM0 - R_W01 >= 5 && R_MM2 - R_MM1 >= 5 && 24 - R_ST3A >= 5, "micro-slices too close to their reads");

__global__ __launch_bounds__(512) void conv64f_kernel(const Conv64fArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 3;   // pixel group: m-tiles 4 wp .. 4 wp + 3 of the 16 (window rows 2 wp, 2 wp + 1)
    const int wq = wave >> 2;  // channel group (32 channels)
    const int S = a.S, So = S >> 1;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

    // ---- conv1_2 weights: B fragments of all 9 taps x 2 K-halves x 2 n-tiles, resident in registers (lane = channel pair 2 l15 + n) ----
    uint4 breg[9][2][2];
    unsigned ar[3];     // A-fragment read addresses in patch buffer 0, [kw], kh even, K half 0 (kh odd: ^ 16, K half 1: ^ 64)
    unsigned out_lane;  // byte offset of this lane's pooled output inside a tile's 8 x 8 windows x 64 channels (m-tile 0)
    {
        const int lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const bf16_t *wr = a.w + (size_t)(wq * 32 + 2 * l15 + n) * 576 + lq * 8;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) breg[t][s][n] = *reinterpret_cast<const uint4 *>(wr + t * 64 + s * 32);
        }
        if (tid < 64) reinterpret_cast<float *>(smem + BIAS_OFF)[tid] = a.bias ? a.bias[tid] : 0.0f;
        if (tid >= 256) {  // conv1_1 weights -> LDS rows of 64 B
            const int r = (tid - 256) >> 2, c = tid & 3;
            *reinterpret_cast<uint4 *>(smem + W11_OFF + r * 64 + ((c ^ ((r >> 2) & 3)) << 4)) = *reinterpret_cast<const uint4 *>(a.w11 + r * 32 + c * 8);
        }
        // lane l15 = (window w, dy, dx) of an m-tile, lq = 16-byte K chunk
        const int w_ = l15 >> 2, dy = (l15 >> 1) & 1, dx = l15 & 1, xl = 2 * w_ + dx;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int gsw = ((((xl + kw) >> 1) & 3) << 1) | (dy & 1);
            ar[kw] = lds0 + ((4 * wp + dy) * 18 + xl) * 128 + ((lq ^ gsw) << 4);
        }
        // epilogue: registers of m-tile 4 wp + i = window lq of that m-tile = window (row 2 wp + (i >> 1), column 4 (i & 1) + lq) of the tile
        out_lane = (unsigned)(((2 * wp) * So + lq) * 64 + wq * 32 + 2 * l15) * 2u;
        // ---- the producer's per-lane table: slot sl -> m-tile wave + 8 sl (waves 5..7, slot 2: m-tile wave + 8 once more) ----
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const int mt = (wave + 8 * sl < 21) ? wave + 8 * sl : wave + 8;
            const int q = mt * 16 + l15;
            const int py = q / 18, px = q - 18 * py;
            const int lsel = lq < 2 ? lq : 2;
            // run kw of pixel (py, px): window row px + kw, 9 bf16 from byte 6 py of copy A = byte 6 py - 2 of copy B (dword-aligned for odd py)
            const unsigned rbl = px * RAW_ROW + 6 * py;
            const unsigned rbs = rbl + lsel * RAW_ROW + ((py & 1) ? RAWB - 2 : 0);
            const int gsw = (((px >> 1) & 3) << 1) | (py & 1);
            // channels nn*16 + 4 lq .. + 3 of pixel q: chunk nn*2 + (lq >> 1) at position chunk ^ gsw, 8-byte half lq & 1
            const unsigned wb = q * 128 + (lq & 1) * 8 + ((((lq >> 1) ^ gsw) & 1) << 4);
            const unsigned flags = (py == 0 ? 1u : 0u) | (py == 17 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == 17 ? 8u : 0u) | (q >= 324 ? 16u : 0u);
            u32x4 e;
            e.x = rbs;
            e.y = rbl;
            e.z = wb | ((unsigned)(gsw >> 1) & 3u) | (flags << 20);  // wb < 2^16 and a multiple of 8
            e.w = lds0 + W11_OFF + l15 * 64 + ((lq ^ ((l15 >> 2) & 3)) << 4);  // conv1_1 weight rows = channels nn*16 + l15 (nn KiB apart)
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
    auto lane_of = [&]() {  // lane-only values are re-derived where they are used, or they would sit in registers across the whole patch loop
        int tv = tid;
        asm volatile("" : "+v"(tv));
        return tv;
    };

    // ---- raw-window DMA (conv64.hip issue_raw): six 128-byte pieces per wave, lanes 0..31, each window row twice ----
    auto issue_raw = [&](int tile, int rbuf) {
        const bool live = tile >= 0;
        const TileXY d = decode_tile(live ? tile : 0);
        const int SP = S + 4;
        const int ro = raw_off(rbuf);
        const int lanev = lane_of() & 63;
        if (lanev < 32) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int wrow = wave + 8 * k;
                const bool ok = live && wrow < 20;
                const bf16_t *srcA = a.img16 + (((size_t)(d.n * SP + 16 * d.tx + wrow) * SP + 16 * d.ty) * 3 + 2 * lanev);
                const bf16_t *z = reinterpret_cast<const bf16_t *>(a.zero_page);
                __builtin_amdgcn_global_load_lds((glb_void *)(ok ? srcA : z), (lds_void *)(smem + ro + wrow * RAW_ROW), 4, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_void *)(ok ? srcA + 1 : z), (lds_void *)(smem + ro + RAWB + wrow * RAW_ROW), 4, 0, 0);
            }
        }
    };

    // ---- the conv1_1 producer, in micro-slices ----
    u32x4 p_tab;                    // this slot's table entry
    u32x2 p_runl, p_runh;           // the first 8 values of the lane's 9-value run
    unsigned p_n9a, p_n9b, p_n9c;   // the 9th values of the three runs
    u32x4 p_av, p_w0, p_w1, p_w2, p_w3;
    f32x4v p_d0, p_d1, p_d2, p_d3;
    u32x2 p_ov;                     // a converted result on its way to LDS
    unsigned p_rbl = 0, p_wbase = 0, p_g6 = 0, p_mask = 0;
    unsigned p_raw = 0, p_pat = 0;  // raw window and patch buffer of the patch being produced (LDS byte addresses; wave-uniform)
    unsigned p_edge = 0;            // which table flags zero a pixel of this tile (<< 20)
    auto p_set_tile = [&](int tile, int rbuf, unsigned pat) {
        const TileXY d = decode_tile(tile >= 0 ? tile : 0);  // no tile left: produce garbage nobody reads (keeps the LDS-operation count uniform)
        p_edge = ((d.ty == 0 ? 1u : 0u) | (d.ty == a.tiles - 1 ? 2u : 0u) | (d.tx == 0 ? 4u : 0u) | (d.tx == a.tiles - 1 ? 8u : 0u) | 16u) << 20;
        p_raw = lds0 + raw_off(rbuf);
        p_pat = pat;
    };
    auto p_tab_read = [&](int sl) {
        const unsigned ta = lds0 + TAB_OFF + sl * TAB_SLOT + (unsigned)lane_of() * 16u;
        asm volatile("ds_read_b128 %0, %1" : "=v"(p_tab) : "v"(ta) : "memory");
    };
    auto p_raw2 = [&]() {
        const unsigned rbs = p_raw + p_tab.x;
        p_rbl = p_raw + p_tab.y;
        asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3" : "=&v"(p_runl), "=&v"(p_runh) : "v"(rbs) : "memory");
    };
    auto p_raw16 = [&]() {
        asm volatile("ds_read_u16 %0, %3 offset:16\n\tds_read_u16 %1, %3 offset:160\n\tds_read_u16 %2, %3 offset:304"
                     : "=&v"(p_n9a), "=&v"(p_n9b), "=&v"(p_n9c)
                     : "v"(p_rbl)
                     : "memory");
    };
    auto p_w01_reads = [&]() { asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(p_w0), "=&v"(p_w1) : "v"(p_tab.w) : "memory"); };
    auto p_w23_reads = [&]() {
        asm volatile("ds_read_b128 %0, %2 offset:2048\n\tds_read_b128 %1, %2 offset:3072" : "=&v"(p_w2), "=&v"(p_w3) : "v"(p_tab.w) : "memory");
    };
    auto p_wbase_g6 = [&]() {
        p_wbase = p_pat + (p_tab.z & 0xFFF8u);
        p_g6 = (p_tab.z << 5) & 0x60u;
    };
    auto p_mask_calc = [&]() { p_mask = (p_tab.z & p_edge) ? 0u : 0xFFFFFFFFu; };
    // the im2col fragment of 16 pixels (lane group lq < 3: the first 8 values of run kw = lq; 3: the 9th values + the bias ones)
    auto p_av0 = [&]() {
        const bool last = (lane_of() & 48) == 48;
        p_av.x = (last ? (p_n9a | (p_n9b << 16)) : p_runl.x) & p_mask;
        p_av.y = (last ? (p_n9c | 0x3F800000u) : p_runl.y) & p_mask;
    };
    auto p_av1 = [&]() {
        const bool last = (lane_of() & 48) == 48;
        p_av.z = (last ? 0x3F803F80u : p_runh.x) & p_mask;
        p_av.w = (last ? 0u : p_runh.y) & p_mask;
    };
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
    auto p_mm = [&](const u32x4 &w) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, p_av), zero4, 0, 0, 0); };
    auto p_cvt = [&](const f32x4v &d) {  // ReLU on the packed bf16 pair (two v_cvt_pk_bf16_f32 + two v_pk_max_i16)
        u32x2 ov = __builtin_bit_cast(u32x2, __builtin_convertvector(d, bf16x4));
        asm volatile("" : "+v"(ov));  // keep the conversion packed
        p_ov.x = relu_bf16x2(ov.x);
        p_ov.y = relu_bf16x2(ov.y);
    };
    auto p_store = [&](int nn) {  // 8 bytes into the swizzled patch: chunk position (nn*2 + (lq >> 1)) ^ gsw
        unsigned ad;
        asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(ad) : "v"(p_g6), "v"((unsigned)(nn * 32)), "v"(p_wbase));
        lds_write8(ad, p_ov);
    };
    auto produce_all = [&]() {  // the whole patch, one slice after the other (prologue only)
        for (int sl = 0; sl < 3; ++sl) {
            p_tab_read(sl);
            wait_lgkm<0>();
            PIN();  // (the arithmetic on the loaded values must not be scheduled above the wait: for hipcc they exist since the asm)
            p_raw2();
            p_raw16();
            p_w01_reads();
            p_w23_reads();
            p_wbase_g6();
            p_mask_calc();
            wait_lgkm<0>();
            PIN();
            p_av0();
            p_av1();
            p_d0 = p_mm(p_w0);
            p_d1 = p_mm(p_w1);
            p_d2 = p_mm(p_w2);
            p_d3 = p_mm(p_w3);
            p_cvt(p_d0);
            p_store(0);
            p_cvt(p_d1);
            p_store(1);
            p_cvt(p_d2);
            p_store(2);
            p_cvt(p_d3);
            p_store(3);
        }
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
    p_tab_read(0);
    wait_lgkm<0>();
    f32x4v acc[4][2];
    u32x4 af[4];
    unsigned toggle = P1_OFF;  // added to the fragment addresses after every patch: buffer 0 -> 1 -> 0 ...

    for (int j = 0; j < my_tiles; ++j) {
        const int tile = b0 + j * G;
        auto stamp = [&](int k) {
            if (a.stamps && (tid & 255) == 0)
                a.stamps[((size_t)tile * 2 + wq) * 8 + k] = k >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
        };
        stamp(7);
        stamp(0);
        issue_raw(tile_at(j + 3), (j + 3) & 3);
        p_set_tile(tile_at(j + 1), (j + 1) & 3, lds0 + ((j & 1) ? 0 : P1_OFF));

        unsigned ar1 = 0;
        auto read_frag = [&](auto sc) {  // the A fragment sub-step s consumes
            constexpr int s = decltype(sc)::value;
            constexpr int h = s >> 2, i = s & 3, t = h >> 1, ks = h & 1, kh = t / 3, kw = t % 3;
            constexpr unsigned flip = (unsigned)(16 * (kh & 1) + 64 * ks);
            if constexpr (flip != 0 && i == 0) ar1 = ar[kw] ^ flip;  // one XOR per half-tap, not one per read
            af[i] = lds_read16<((2 * (i / 2) + kh) * 18 + 8 * (i % 2) + kw) * 128>(flip ? ar1 : ar[kw]);
        };
        read_frag(std::integral_constant<int, 0>{});
        read_frag(std::integral_constant<int, 1>{});
        read_frag(std::integral_constant<int, 2>{});
        read_frag(std::integral_constant<int, 3>{});
        PIN();

        static_for<0, NSUB>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int h = s >> 2, i = s & 3, t = h >> 1, ks = h & 1;
#ifdef CONV64F_NOWAIT  // timing experiment only (wrong results): what the counted waits cost
            wait_lgkm<15>();
#else
            wait_lgkm<wait_of(s)>();
#endif
            PIN();  // nothing moves above the wait (tying af[i] to the asm instead makes hipcc pad every MFMA behind it with an s_nop)
            const bf16x8 av = __builtin_bit_cast(bf16x8, af[i]);
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, breg[t][ks][0]), h == 0 ? zero4 : acc[i][0], 0, 0, 0);
            PIN();
#ifdef CONV64F_NOPROD  // timing experiment only: the half-tap loop without the producer
            constexpr int r = -1, sl = 0;
            (void)sl;
#else
            constexpr int r = s % 24, sl = s / 24;
#endif
            if constexpr (r == R_RAW2) p_raw2();
            if constexpr (r == R_RAW16) p_raw16();
            if constexpr (r == R_W01) p_w01_reads();
            if constexpr (r == R_WBASE) p_wbase_g6();
            if constexpr (r == R_MASK) p_mask_calc();
            if constexpr (r == R_AV0) p_av0();
            if constexpr (r == R_AV1) p_av1();
            if constexpr (r == R_MM0) p_d0 = p_mm(p_w0);
            if constexpr (r == R_MM1) {
                p_d1 = p_mm(p_w1);
                p_w23_reads();
            }
            if constexpr (r == R_ST0A) p_cvt(p_d0);
            if constexpr (r == R_ST0B) p_store(0);
            if constexpr (r == R_ST1A) p_cvt(p_d1);
            if constexpr (r == R_ST1B) p_store(1);
            if constexpr (r == R_MM2) p_d2 = p_mm(p_w2);
            if constexpr (r == R_MM3) p_d3 = p_mm(p_w3);
            if constexpr (r == R_ST2A) p_cvt(p_d2);
            if constexpr (r == R_ST2B) p_store(2);
            if constexpr (r == R_ST3A) {
                p_cvt(p_d3);
                p_tab_read(sl < 2 ? sl + 1 : 0);  // slot 2: slot 0's entry for the NEXT patch
            }
            if constexpr (r == R_ST3B) p_store(3);
            PIN();
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, breg[t][ks][1]), h == 0 ? zero4 : acc[i][1], 0, 0, 0);
            PIN();
            if constexpr (s + 4 < NSUB) read_frag(std::integral_constant<int, s + 4>{});  // into the registers both MFMAs have just read
            if constexpr (s == 35) stamp(1);
            PIN();
        });
        stamp(2);
        stamp(3);
        wait_vmcnt<0>();  // this wave's pieces of raw window j + 3 (issued a patch ago) and the stores of patch j - 1: long done

        // ---------------- epilogue: bias, ReLU, pool, store ----------------
        {
            const TileXY d = decode_tile(tile);
            // lane: channels (2 l15, 2 l15 + 1) of the wave's 32; registers = the 4 pixels of window lq of m-tile 4 wp + i
            float b0v, b1v;
            {
                const int tv = lane_of();
                const unsigned ba = lds0 + BIAS_OFF + (wq * 32 + 2 * (tv & 15)) * 4;
                uint2 bb;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(bb) : "v"(ba) : "memory");
                b0v = __builtin_bit_cast(float, bb.x);
                b1v = __builtin_bit_cast(float, bb.y);
            }
            // this tile's 8 x 8 windows: window row 0 and row 1 of the wave's pair as two scalar bases, the lane offset in one register
            unsigned char *row0 = reinterpret_cast<unsigned char *>(a.out + ((size_t)(d.n * So + d.ty * 8) * So + d.tx * 8) * 64);
            unsigned char *row1 = row0 + (size_t)So * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v0 = fmaxf(fmaxf(acc[i][0][0], acc[i][0][1]), fmaxf(acc[i][0][2], acc[i][0][3])) + b0v;
                const float v1 = fmaxf(fmaxf(acc[i][1][0], acc[i][1][1]), fmaxf(acc[i][1][2], acc[i][1][3])) + b1v;
                const unsigned ow = relu_bf16x2(__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{v0, v1}, bf16x2)));
                *reinterpret_cast<unsigned *>(((i >> 1) ? row1 : row0) + out_lane + (i & 1) * 512) = ow;
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
    static LdsAttrMask attr_done{0};
    if (hipError_t e = set_max_lds(reinterpret_cast<const void *>(conv64f_kernel), LDS_BYTES, attr_done); e != hipSuccess) return e;
    gemm_debug_note_route("conv64-fused11", -1);
    hipLaunchKernelGGL(conv64f_kernel, dim3((unsigned)gx, 1), dim3(512), LDS_BYTES, stream, a);
    return hipGetLastError();
}
