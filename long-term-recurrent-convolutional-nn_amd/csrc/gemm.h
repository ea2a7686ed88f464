// gemm.h -- the one contraction engine of liblrcn_hip: C[M][N] (+)= A[M][K] * B[N][K]^T  ("NT", both operands
// K-contiguous), MFMA on gfx950, with an implicit-GEMM 3x3 convolution A-operand mode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { GEMM_A_PLAIN = 0, GEMM_A_CONV3 = 1 };
enum { GEMM_OUT_PLAIN = 0, GEMM_OUT_CONV = 1, GEMM_OUT_POOL = 2, GEMM_OUT_LSTM_FWD = 3, GEMM_OUT_LSTM_BWD = 4, GEMM_OUT_SMAX_TOPK = 5 };

// GEMM_OUT_SMAX_TOPK (gemm_8p.hip, bf16, 256 x 256 tiles; round 6): the logits GEMM of a batched beam-decode step whose C never reaches HBM.
// softmax + sortperm of lrcn.jl:652-656 need, per row, max / sum-exp over all V columns and the K best columns; each tile reduces ITS 256
// columns (+ bias) to two records per row -- 128 columns each, as alternating runs of 32: {max, sum exp(x - max), the SMAX_KC largest logits and their column ids}
// -- and softmax_topk_merge_kernel (kernels.hip) combines the V / 128 records of a row.  218 MB of f32 logits per step at 5120 x 10640 become
// 28 MB of records.  A record is SMAX_REC floats (64 bytes): [0] max, [1] sum, [2 .. 2+KC) values, [8 .. 8+KC) column ids (int bits).
enum { SMAX_KC = 6, SMAX_REC = 16 };
struct SmaxEpi {
    float *part;   // [M][nrec][SMAX_REC]
    int nrec;      // records per row = 2 * ceil(N / 256)
};

// Epilogue operands of the two LSTM out-modes of gemm_8p.hip (bf16 only): the recurrent GEMM of a timestep with the cell update
// (forward) / the cell backward of the previous step (backward) computed in the accumulator registers -- no f32 round trip of the
// pre-activations / dh and no separate cell launch.  Same arithmetic as lstm_fwd_kernel / lstm_bwd_kernel (kernels.hip).
//   LSTM_FWD: C columns are (unit, gate)-interleaved: B = Wh rows in the order u*4 + gate (prepare_weights' "gi" copy), N = 4H.
//             acc(row, 4u + gate) + Gx[row][gate*H + u] -> f, i, o, g -> c = c_prev f + i g, h = o tanh(c)        (lrcn.jl:529-536)
//   LSTM_BWD: C columns are hidden units, N = H, A = dZ of step s, B = Wh^T: dh = acc + dh_ext -> dZ of step s-1, dc   
struct LstmEpi {
    int H;
    int64_t ld_a, ld_h;     // leading dimensions of acts / dz (elements) and of h_new
    const float *Gx;        // FWD: [M][4H] input-side pre-activations (+ bias) of this step
    const float *c_prev;    // [M][H] or NULL (first step of the sequence)
    const int *c_prev_idx;  // FWD, optional: row r reads c_prev[c_prev_idx[r]] (the beam decode: a hypothesis continues its PARENT's cell state; then
                            // c_out must not alias c_prev)
    const float *c_new;     // BWD: [M][H] cell state of step s-1
    float *c_out;           // FWD: [M][H]
    void *acts;             // FWD: out (NULL: not kept -- a decode step has no backward pass), BWD: in -- activated gates [M][ld_a], columns [f | i | o | g]
    void *h_new;            // FWD: [M][ld_h]
    int gx_bcast;           // FWD: 1 = Gx is ONE row [4H] added to every row (the bias of a decode step whose GEMM contracts [x | h] itself)
    const int *gx_idx;      // FWD, optional (gx_bcast = 0): row r adds Gx row gx_idx[r] -- a TABLE of input-side pre-activations (per token, per image)
    float *h_f32;           // FWD, optional: [M][H] f32 copy of h (the ABI's state arrays)
    const float *dh_ext;    // BWD: [M][H] dh of step s-1 from the layer above / the loss
    float *dc;              // BWD: [M][H] in/out
    void *dz_out;           // BWD: [M][ld_a] dZ of step s-1
};
enum { GEMM_T_F32 = 0, GEMM_T_BF16 = 1, GEMM_T_F8 = 2 };  // F8: OCP e4m3 A, B and C (gemm_8p.hip CONV3 only)

struct GemmArgs {
    int dtype;      // GEMM_T_*: element type of A and B
    const void *A;  // PLAIN: [M][lda]; CONV3: NHWC activations [n][H][W][Cin]
    int64_t lda;    // elements (PLAIN only)
    const void *B;  // [N][ldb], K-contiguous (weights)
    int64_t ldb;
    void *C;        // [M][ldc] (or conv/pool mapped), float if c_f32 else same type as A
    int64_t ldc;
    int M, N, K;    // CONV3: M = n*H*W output pixels (window-major order), K = 9*Cin
    const float *bias;  // per output column, or NULL
    const float *scale; // GEMM_T_F8: per-output-column multiplier applied to the accumulator before the bias (dequant x requant)
    int c_f32;
    int beta;       // 1: C += result (reads C)
    int c_is_zero;  // caller guarantees C holds zeros: a split-K-by-atomics kernel may skip its own memset
    int relu;
    int a_mode, out_mode;
    int H, W, Cin;  // conv geometry (square-agnostic; H, W even)
    unsigned inv_w2, inv_h2;  // gemm_8p.hip: reciprocals of W / 2 and H / 2 for decode_pixel_fast (set by launch_gemm_8p; 0 = divide)
    int wg_cap;             // > 0: at most this many workgroups (gemm_8p.hip / conv64.hip walk the tiles persistently)
    LstmEpi lstm;           // out_mode GEMM_OUT_LSTM_FWD / _BWD only
    SmaxEpi smax;           // out_mode GEMM_OUT_SMAX_TOPK only (bias = the logits' bias row; C unused)
    int cfg_pref;           // gemm_8p.hip: 0 = the dispatcher's tile menu, 2 = prefer the 256 x 128 tile (set by the bg_cus route)
    int free_cus;           // > 0 (rows per GPU below the bg_cus route's threshold): this many CUs are free beside the capped convolution grids --
                            // the split-K planner cuts K so that tiles x slices fit them in ONE round (round 5)
    int bg_cus;             // > 0: this contraction runs BESIDE the capped persistent convolution grids of another stream and will
                            // find about this many free CUs (lrcn_api.hip sets it for the LSTM GEMMs when lrcn_vgg_set_wg_cap is
                            // active): the dispatcher then prefers a route whose workgroups fit them in one round
    int *tile_ctr;          // capped grids only: 8 zeroed ints = per-XCD work queues -- workgroups PULL tiles (8 i + queue) instead of
                            // walking a fixed share, so one that starts late (its CU still busy with another stream's kernel) does
                            // fewer tiles instead of stretching the whole launch; NULL = static round-robin walk
    unsigned long long *stamps;  // kernel-development: per-tile s_memtime stamps of gemm8p_tile's segments (8 per tile), or NULL
    int splitk_forced;      // launch_gemm_8p(.., splitk): take the caller's slice count as it is (gemm.hip's 8p-bg-splitk route)
    int splitk_no_reduce;   // ... and leave the f32 slabs [slices][M][N] in ws for the caller's next kernel to sum (no reduce launch, C untouched)
    int deterministic;      // 1: no float-atomic split-K (gemm_glds.hip takes one K range per tile; gemm_8p's slab split-K is ordered anyway)
    int dbg;                // kernel-development ablation flags (LRCN_DBG env): 8 = gemm_8p.hip does not issue the next tile's first K-tile early
    const void *zero_page;  // >= 256 zero bytes, 16-byte aligned (source of padding rows for the direct-to-LDS path) or NULL
    void *ws;               // split-K workspace (f32 slabs [slices][M][N]) or NULL: enables gemm_8p's split-K form
    size_t ws_bytes;
};

// Requirements (checked): A/B base 16-byte aligned, lda/ldb multiples of the 16-byte chunk (4 f32 / 8 bf16),
// CONV3: Cin a multiple of 32 (f32) / 64 (bf16).  Returns hipSuccess or an error; never faults on bad shapes.
hipError_t launch_gemm(hipStream_t stream, const GemmArgs &g);

// Test / development aid: the kernel family the most recent launch on this thread took ("8p:0" = 256x256 tile, "8p:1" = 256x128,
// "8p:2" = 512x128, "8p-splitk:N", "glds", "skinny", "gemm_nt", "conv64", "conv64-fused11", ...).  lrcn_debug_last_route() returns it.
void gemm_debug_note_route(const char *route, int cfg);
const char *gemm_debug_last_route();

// bf16 direct-to-LDS variant (gemm_glds.hip): 256-row tiles, global_load_lds staging, XOR-swizzled LDS.
// launch_gemm routes to it when gemm_glds_eligible(g) and the grid is large enough to fill the chip.
bool gemm_glds_eligible(const GemmArgs &g);
hipError_t launch_gemm_glds(hipStream_t stream, const GemmArgs &g);
int64_t gemm_glds_blocks(const GemmArgs &g);  // workgroups the direct-to-LDS path would launch (0 = no config fits)

// Phase-interleaved variant (gemm_8p.hip): 256 x 256 / 256 x 128 tiles, v_mfma_f32_16x16x32_bf16, two wave groups one
// barrier apart.  gemm_8p_config returns the tile config (>= 0) and the grid size, or -1 when the problem does not fit it.
int gemm_8p_config(const GemmArgs &g, int64_t *blocks);
int gemm_8p_splitk(const GemmArgs &g, int64_t *blocks);  // split-K slices for skinny problems (0 = not applicable)
hipError_t launch_gemm_8p(hipStream_t stream, const GemmArgs &g, int splitk = 1);
hipError_t launch_splitk_reduce(hipStream_t stream, const GemmArgs &g, int splits);

// Skinny-M weight-streaming variant (gemm_skinny.hip): M <= 256 rows, B fragments straight from HBM/L2 to registers.
bool gemm_skinny_eligible(const GemmArgs &g);
hipError_t launch_gemm_skinny(hipStream_t stream, const GemmArgs &g);

// Halo-patch convolution for the Cin = 64 layers (conv64.hip): NHWC bf16 in/out, weights [Cout][9][64], H and W multiples
// of 16, Cout a multiple of 64; bias + optional ReLU + optional fused 2x2 max-pool.
bool conv64_eligible(int dtype, int Cin, int Cout, int H, int W);
// f8_inv_scale > 0 (ReLU, no pool): the output is OCP e4m3(relu(.) * f8_inv_scale) instead of bf16
hipError_t launch_conv64(hipStream_t stream, const void *in, const void *w, const float *bias, void *out, int N, int H, int W, int Cout,
                         int relu, int pool, const void *zero_page, float f8_inv_scale = 0.0f, int wg_cap = 0, unsigned long long *stamps = nullptr);
// conv1_1 + conv1_2 (+ pool) fused, from mean-subtracted bf16 crops (conv64.hip FUSE): w11 from k_repack_conv11_w_fused
hipError_t launch_conv64_fused11(hipStream_t stream, const void *img16, const void *w11, const float *b11, const void *w,
                                 const float *bias, void *out, int N, int S, const void *zero_page, int wg_cap = 0, unsigned long long *stamps = nullptr);
// the second generation of that launch (conv64f.hip: producer slices inside every wave's half-tap loop, one barrier per patch); conv1_1's
// bias comes in w11's K padding.  launch_conv64_fused11 routes here unless LRCN_FUSE11_GEN=1.
hipError_t launch_conv64f(hipStream_t stream, const void *img16, const void *w11, const void *w, const float *bias, void *out, int N, int S,
                          const void *zero_page, int wg_cap = 0, unsigned long long *stamps = nullptr);
