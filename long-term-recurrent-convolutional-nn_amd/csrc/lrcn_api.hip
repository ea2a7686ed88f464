// lrcn_api.hip -- context, workspace and the C ABI of liblrcn_hip.so (include/lrcn.h).
//
// Orchestration of the hot path on one gfx950 device.  The reference runs lrcn() once per timestep with ~50 tiny
// kernels and a blocking D2H per step (lrcn.jl:560-570); here everything that does not feed back through the
// recurrence is time-batched over all S = T+1 steps (M = S*B rows, row m = s*B + b):
//   forward : embedding gather (+dropout) -> input-side gate GEMM for all steps -> S x [recurrent GEMM (beta=1) + fused
//             cell] -> projection GEMM -> concat/dropout -> LSTM-2 likewise -> ONE logits GEMM for all steps -> fused
//             log-softmax / NLL / dlogits with on-device double accumulation (no per-step D2H).
//   backward: dWout/dH2 GEMMs for all steps -> reverse recurrence (fused cell backward + one GEMM per step) ->
//             time-batched weight-gradient GEMMs (K = S*B) -> projection/x_cnn/embedding duals.
// Internal activations are row-major [row][feature] (K-contiguous) in the context's arithmetic type T (f32 or bf16);
// the column-major f32 arrays of the ABI are converted at the boundary.  Every contraction is the NT MFMA kernel of
// gemm.hip; operands that the math wants transposed are materialised K-contiguous by k_transpose.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/lrcn.h"
#include "comm.h"
#include "common.h"
#include "gemm.h"
#include "kernels.h"

static thread_local std::string g_create_err;

struct VggLayer {
    void *w = nullptr;    // [Cout][9*Cin] T (conv) ; conv1_1: [64][32]
    void *w_fused = nullptr;  // conv1_1 only (bf16): [64][32] in the K order of the fused conv1_1+conv1_2 kernel
    float *b = nullptr;   // [Cout] f32
    int Cin = 0, Cout = 0, S = 0, pool = 0;
    // LRCN_FP8 (layers conv2_2 .. conv5_3): e4m3 weights [Cout][9*Cin], per-channel weight scale, effective epilogue scale/bias
    void *w8 = nullptr;
    float *sw = nullptr, *escale = nullptr, *ebias = nullptr;
};
constexpr int kFp8First = 3;  // conv2_2: the first layer with Cin % 128 == 0

struct lrcn_ctx {
    lrcn_config cfg{};
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<void *> allocs;
    int dt = 0, vdt = 0;
    size_t esz = 4, vesz = 4;
    int E = 0, H1 = 0, H2 = 0, h = 0, V = 0, maxB = 0, maxS = 0;
    int nl = 2;             // LSTM layers: 2 = the reference's LRCN-2f (lrcn.jl:540-551), 1 = LRCN-1f (BASELINE configs[1])
    int X1 = 0;             // input width of LSTM-1: E (2 layers) or E + h = [embedding | x_cnn] (1 layer)
    int64_t ldX1 = 0;
    int64_t ldE = 0, ldH1 = 0, ldH2 = 0, ldh = 0, ld4H1 = 0, ld4H2 = 0, ldV = 0, ldM = 0, ldB = 0;
    // shadow weights (T)
    void *W1x = nullptr, *W1h = nullptr, *W1xT = nullptr, *W1hT = nullptr;
    // decode only: [x | h] concatenated along K -- weights [4H][ldXH] and the step inputs [B][ldXH]: one gate GEMM per LSTM
    void *W1cat = nullptr, *W2cat = nullptr, *st_xh1 = nullptr, *st_xh2 = nullptr;
    int64_t ldXH1 = 0, ldXH2 = 0;
    void *W2x = nullptr, *W2h = nullptr, *W2xT = nullptr, *W2hT = nullptr;
    // batched decode with input-projection TABLES (round 6; decode_tables_on): T1 [V][4H1] = Wembed W1x + b1 per TOKEN, U2 [images][4H2] =
    // x_cnn W2x(right half) + b2 per IMAGE (f32, gate-block columns), the gate GEMMs' operands A1 = h1[parent] and A2 = [h1 Wproj | h2[parent]],
    // W2's matching weights (x_cnn columns left out, rows interleaved) and the image of every hypothesis row; all lazily allocated
    float *dec_T1 = nullptr, *dec_U2 = nullptr;
    void *dec_A1 = nullptr, *dec_A2 = nullptr, *dec_W2c = nullptr, *dec_Aimg = nullptr;
    int32_t *dec_img = nullptr;
    float *smax_part = nullptr;             // [maxB][2 ceil(V / 256)][SMAX_REC]: the logits GEMM's softmax / top-K records of a batched decode step (round 6), lazily
    void *alt_gi[2] = {nullptr, nullptr};   // LRCN_OPT_FUSED_UPDATE: the second set's gate-interleaved copies (round 6), written by the Adam kernel
    bool gi_live = false;                   // a training call has taken the cell-epilogue route: the fused update keeps the interleaved copies current
    bool shadow_has_gi = false;             // ... and the current set's were made by it
    void *W1h_gi = nullptr, *W2h_gi = nullptr;  // recurrent weights with (unit, gate)-interleaved rows (gemm_8p.hip LSTM_FWD epilogue), lazily
    void *Wpd = nullptr, *WpT = nullptr, *Wcd = nullptr, *WeT = nullptr, *Wod = nullptr, *WoT = nullptr;
    // LRCN_OPT_FUSED_UPDATE: the second set of the 14 training shadows above.  The Adam kernel of a train step writes the NEXT step's
    // shadows into it while (per-group pipeline) the backward pass may still be reading the current set; then the two sets swap roles.
    void *alt[14] = {};
    bool opt_fused = false, opt_det = false;
    int64_t conv_chunk_bytes = 0;           // LRCN_OPT_CONV_CHUNK_BYTES (0 = default)
    unsigned fused_groups = 0;              // gradient groups whose fused Adam has been issued in the current step (bit per group)
    int fused_step = 0;                     // the `step` those bits belong to: a call with another step starts a new mask
    unsigned refresh_groups = 0;            // lrcn_refresh_shadows_group: groups whose shadows of the NEXT step have been issued
    bool shadow_valid = false;              // the current set holds the shadows (direct AND transposed) of the parameters at shadow_p
    const float *shadow_p[9] = {};
    float *dWe_rm = nullptr;                // [V][ldE] f32, all zero between calls: row-major staging of the embedding gradient
    unsigned long long *sort_keys = nullptr;  // [maxS * maxB] (token, row) keys of the ordered embedding-gradient sums
    double *logp_rows = nullptr;            // [maxS * maxB] per-row log p(target): the ordered loss sum of LRCN_OPT_DETERMINISTIC
    // sparse exchange of the embedding gradient (lrcn_set_embed_rows_buffer): lossgradient writes its (T+1) B rows of d(x_lstm) and their
    // token ids HERE instead of scattering them into the dense gradient; lrcn_embed_grad_from_rows sums the rows of all ranks in a fixed order
    float *emb_rows_out = nullptr;
    int32_t *emb_tok_out = nullptr;
    int emb_rows_cap = 0;
    unsigned long long *imp_keys = nullptr;  // [8192] sort keys of lrcn_embed_grad_from_rows
    // activations
    int32_t *tok = nullptr, *tok_in = nullptr, *tok_tgt = nullptr;
    void *F = nullptr, *FT = nullptr;
    float *xcnn = nullptr;
    void *Xemb = nullptr, *A1 = nullptr, *H1all = nullptr, *X2 = nullptr, *A2 = nullptr, *H2all = nullptr;
    float *G1 = nullptr, *C1 = nullptr, *G2 = nullptr, *C2 = nullptr, *Logits = nullptr;
    void *dLog = nullptr, *dZ1 = nullptr, *dZ2 = nullptr, *dX2 = nullptr;
    float *dH1all = nullptr, *dH2all = nullptr, *dXemb = nullptr, *dhrec = nullptr, *dc = nullptr, *dxcnn = nullptr;
    void *TA = nullptr, *TB = nullptr;  // transposed-operand scratch: up to [max(4H,V)][ldM] and [max(2*H2, E+H1)][ldM]
    void *dxcT = nullptr;
    double *logp = nullptr;
    void *zero_page = nullptr;
    hipEvent_t grad_ev[LRCN_GRAD_GROUPS] = {};  // recorded when the gradients of a group are final (lrcn_grad_group_wait)
    void *gemm_ws = nullptr;  // split-K slabs of gemm_8p / gemm_skinny (LSTM side)
    void *vgg_ws = nullptr;   // same for fc6/fc7: the VGG forward may run on another stream, concurrently with the LSTM step
    size_t gemm_ws_bytes = 0;
    int last_norm = 1, last_S = 1;
    int cur_B = 0;  // rows of the loss / lossgradient call in flight (the "beside the convolutions" GEMM hints apply from 256 rows)
    // single-step scratch (lrcn_lstm / lrcn_step / beam search), row-major
    float *st_f32[4] = {nullptr, nullptr, nullptr, nullptr};   // h1,c1,h2,c2 [B][H]
    float *st2_f32[4] = {nullptr, nullptr, nullptr, nullptr};  // ping-pong for the beam gather
    void *st_h1 = nullptr, *st_h2 = nullptr, *st_x = nullptr, *st_x2 = nullptr, *st_a = nullptr;
    float *st_g = nullptr, *st_logits = nullptr, *st_prob = nullptr, *st_io = nullptr, *st_topv = nullptr;
    int32_t *st_topi = nullptr, *st_parent = nullptr;
    // batched beam search (lrcn_beam_search_batch): token histories (ping-pong), bookkeeping, results -- all on the device
    int32_t *bs_seq[2] = {nullptr, nullptr}, *bs_last = nullptr, *bs_done = nullptr, *bs_ndone = nullptr, *bs_res_tok = nullptr,
            *bs_res_len = nullptr;
    float *bs_p = nullptr, *bs_res_p = nullptr;
    // VGG
    int vgg_wg_cap = 0;  // > 0: cap on the convolution grids (lrcn_vgg_set_wg_cap)
    bool vgg_loaded = false;
    bool vgg_fp8 = false, fp8_ready = false;  // LRCN_FP8: conv2_2..conv5_3 in e4m3 once lrcn_vgg_calibrate has run
    float *amax_dev = nullptr;                // [13] per-layer output amax collected by the calibration pass
    float act_scale[13] = {};                 // sa of layer l's output (l = 2..12)
    VggLayer conv[13];
    void *fc6w = nullptr, *fc7w = nullptr;
    float *fc6b = nullptr, *fc7b = nullptr;
    void *actA = nullptr, *actB = nullptr, *im2col = nullptr, *f6 = nullptr, *img16 = nullptr;
    float *featsRM = nullptr;  // [N][4096] f32 row-major
    // live timing of the dominant kernel (the 12 implicit-GEMM conv launches conv1_2..conv5_3), see lrcn_profile*
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
    size_t prof_used = 0;
    double prof_ms = 0.0;
    int64_t prof_launches = 0;
    // level 2 (lrcn_profile(ctx, 2)): event pairs around the HBM-bound segments of SURVEY 8(d), see lrcn_profile_segment
    int prof_level = 0;
    struct SegProf {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
        size_t used = 0;
        double ms = 0.0, bytes = 0.0;
        int64_t n = 0;
    } seg[LRCN_SEG_COUNT];
    std::string vgg_routes;  // kernel family per layer of the most recent VGG forward (lrcn_debug_route)
    // image front end: the full averageImage (lrcn_set_average_image), per-batch image descriptors, float scratch of the unfused path
    float *avg_img = nullptr;
    bool avg_on = false;
    void *img_meta = nullptr;
    int img_meta_cap = 0;
    float *pre_f32 = nullptr;
    // data parallelism: RCCL communicator (lrcn_comm_init) and one stream per gradient group for [all-reduce -> Adam]
    // weight-gradient stream: the dW / db GEMMs of lossgradient feed nothing but update!, so they run on their own stream beside the
    // reverse recurrences (which are chains of small launches that leave CUs idle); own split-K workspace, fork / join by events
    hipStream_t wg_stream = nullptr;
    bool wg_stream_owned = true;   // false: handed in through lrcn_set_wg_stream (not destroyed here)
    hipEvent_t wg_fork[4] = {}, wg_done = nullptr;
    hipEvent_t xc_fork = nullptr, xc_done = nullptr;  // the image-embedding GEMM of the forward pass on the weight-gradient stream (loss_impl)
    void *wg_ws = nullptr;
    void *pin = nullptr;      // pinned host staging for results larger than HIP's fast pageable-copy path (lrcn_beam_search_batch)
    size_t pin_bytes = 0;
    unsigned long long *stamps = nullptr;  // kernel-development: per-tile segment stamps (LRCN_STAMPS=1, lrcn_debug_stamps)
    int64_t stamps_n = 0;
    int *tile_ctr = nullptr;  // per-layer work queues of the capped persistent convolution grids (GemmArgs::tile_ctr)
    // input feed (lrcn_upload_crops): uint8 crops travel host -> HBM on the context's own copy stream into one of kStage staging buffers,
    // beside the running step; a VGG forward that is handed a staging buffer waits (on the device) for its upload, and the upload into a
    // staging buffer waits for the one kernel of the forward that last read it (the crops are consumed by the forward's FIRST kernel).
    // The copy stream never carries a device-side wait for a read that has not happened yet.  Measured (bench.py, host 4 steps ahead of the
    // device, which is where it runs when nothing holds it back): a hipStreamWaitEvent on an event one or two steps in the device's future
    // is a barrier packet at the head of a HARDWARE queue that the copy stream shares with compute streams (HIP maps its streams onto a few
    // hardware queues) -- kernels queued behind it stall, and the step ran 8.5 instead of 7.0 ms until the host happened to fall back.  So an
    // upload whose staging buffer is still unread BLOCKS THE CALLING THREAD (hipEventSynchronize) and then queues a copy with no dependency.
    // With kStage buffers that happens only when the host is more than kStage - 1 steps ahead of the device: a bound on the run-ahead.
    static constexpr int kStage = 3;
    hipStream_t copy_stream = nullptr;
    uint8_t *stage[kStage] = {};
    hipEvent_t up_done[kStage] = {}, rd_done[kStage] = {};
    bool stage_full[kStage] = {};   // holds crops that no forward has been issued on yet
    bool stage_read[kStage] = {};   // rd_done[j] has been recorded at least once
    int stage_next = 0;
    LrcnComm *comm = nullptr;
    hipStream_t comm_stream = nullptr;  // every collective of the communicator is issued on this ONE stream, in group order
    bool comm_stream_owned = false;     // created here (destroyed here), or handed in through lrcn_comm_set_stream
    // bucket[g] == comm_stream for every g since round 4: the groups become final in order, so one stream runs [wait, all-reduce, Adam] of
    // group after group and loses nothing, while five streams on HIP's four hardware queues meant that one of them shared a queue with the
    // VGG side stream and its Adam waited for the whole forward (dp.py streams_share_a_queue)
    hipStream_t bucket[LRCN_GRAD_GROUPS] = {};
    hipEvent_t ar_done[LRCN_GRAD_GROUPS] = {};
    hipEvent_t bucket_done[LRCN_GRAD_GROUPS] = {};
    bool bucket_pending[LRCN_GRAD_GROUPS] = {};
};

#define FAIL(ctx, code, ...)                          \
    do {                                              \
        char _b[512];                                 \
        snprintf(_b, sizeof(_b), __VA_ARGS__);        \
        (ctx)->err = _b;                              \
        return (code);                                \
    } while (0)
#define HIPCHK(ctx, expr)                                                                        \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) FAIL(ctx, LRCN_EHIP, "%s: %s", #expr, hipGetErrorString(_e));      \
    } while (0)
#define KCHK(ctx, what)                                                                          \
    do {                                                                                         \
        hipError_t _e = hipGetLastError();                                                       \
        if (_e != hipSuccess) FAIL(ctx, LRCN_EHIP, "%s: %s", what, hipGetErrorString(_e));       \
    } while (0)

namespace {

// Every entry point that takes a context runs on the context's device, whatever device the calling thread had selected,
// and restores the caller's selection on return (allocations, null-stream work and hipFuncSetAttribute are per device).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const lrcn_ctx *c) {
        if (!c) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != c->cfg.device) switched = hipSetDevice(c->cfg.device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

template <class P> int dalloc(lrcn_ctx *c, P *&p, size_t bytes) {
    void *q = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
        c->err = std::string("hipMalloc failed: ") + hipGetErrorString(e);
        return LRCN_ENOMEM;
    }
    c->allocs.push_back(q);
    p = reinterpret_cast<P *>(q);
    // K-padding columns must hold zeros (never NaN) from the start.  The fill runs on the NULL stream and a device-memory hipMemset may
    // return before it has executed; work that the caller then queues on a NON-BLOCKING stream (torch's side streams, the context's
    // weight-gradient / group streams) is not ordered behind the null stream -- a buffer allocated lazily inside a step could be
    // zeroed AFTER its first kernel had written it (found with tools/fake_multi_check.py: the second shadow set, allocated by the
    // first fused update, lost what the group streams' Adam kernels had just written).  Drain the null stream before handing it out.
    if (hipMemset(q, 0, bytes) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
        c->err = "hipMemset failed";
        return LRCN_EHIP;
    }
    return LRCN_OK;
}
#define DALLOC(c, p, bytes)                         \
    do {                                            \
        int _r = dalloc(c, p, (size_t)(bytes));     \
        if (_r) return _r;                          \
    } while (0)

// leading dimensions: whole 64-element K-steps, so the direct-to-LDS GEMM can run with K rounded up (pads are zero)
inline int64_t ld8(int64_t n) { return round_up64(n, 64); }
inline char *boff(void *p, int64_t elems, size_t esz) { return reinterpret_cast<char *>(p) + elems * (int64_t)esz; }
inline const char *boff(const void *p, int64_t elems, size_t esz) {
    return reinterpret_cast<const char *>(p) + elems * (int64_t)esz;
}

// C[M][N] (+)= A[M][K] * B[N][K]^T
int gemm(lrcn_ctx *c, int dtype, const void *A, int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc, int M, int N,
         int K, const float *bias, bool c_f32, bool beta = false, bool relu = false, bool c_is_zero = false, bool on_wg_stream = false) {
    GemmArgs g{};
    g.dtype = dtype;
    g.A = A;
    g.lda = lda;
    g.B = B;
    g.ldb = ldb;
    g.C = C;
    g.ldc = ldc;
    g.M = M;
    g.N = N;
    // bf16: K rounded up to whole 128-byte K-steps.  Every internal operand has ld >= that and zero (weights: written
    // zeros; activations: zero or stale-but-finite values that meet a zero on the other side) in the padding.
    g.K = (dtype == GEMM_T_BF16 && lda >= round_up64(K, 64) && ldb >= round_up64(K, 64)) ? (int)round_up64(K, 64) : K;
    g.bias = bias;
    g.c_f32 = c_f32;
    g.beta = beta;
    g.c_is_zero = c_is_zero;
    g.relu = relu;
    g.a_mode = GEMM_A_PLAIN;
    g.out_mode = GEMM_OUT_PLAIN;
    g.zero_page = c->zero_page;
    g.deterministic = c->opt_det;
    g.ws = on_wg_stream ? c->wg_ws : c->gemm_ws;  // one split-K workspace per stream
    g.ws_bytes = c->gemm_ws_bytes;
    {   // the LSTM GEMMs of a two-stream training step run beside the capped convolution grids (LRCN_BG_ROUTE=0 turns the hint off)
        static const char *kb = getenv("LRCN_BG_ROUTE");
        // from 256 rows per GPU only: below, the VGG forward's own grids are small, more CUs are free, and the LSTM chain is the critical
        // path -- the hints measured 1.64 -> 1.79 ms/step at 32 rows, 2.39 -> 2.43 at 64, 4.11 -> 4.12 at 128, 7.31 -> 7.20 at 256
        static const char *kmb = getenv("LRCN_BG_MINB");  // kernel-development knob: rows per GPU from which the hints apply (default 256)
        static const char *kfc = getenv("LRCN_FREE_CUS_HINT");  // 0: the split-K planner assumes the whole chip, as before round 5
        if (c->vgg_wg_cap >= 8 && c->vgg_loaded && c->cur_B < (kmb ? atoi(kmb) : 256) && !(kfc && kfc[0] == '0')) {
            static int ncu1 = 0;
            if (!ncu1) {
                hipDeviceProp_t pr;
                ncu1 = (hipGetDeviceProperties(&pr, c->cfg.device) == hipSuccess) ? pr.multiProcessorCount : 256;
            }
            g.free_cus = ncu1 - c->vgg_wg_cap > 0 ? ncu1 - c->vgg_wg_cap : 0;
        }
        if (c->vgg_wg_cap >= 8 && c->vgg_loaded && c->cur_B >= (kmb ? atoi(kmb) : 256) && !(kb && kb[0] == '0')) {
            static int ncu = 0;
            if (!ncu) {
                hipDeviceProp_t pr;
                ncu = (hipGetDeviceProperties(&pr, c->cfg.device) == hipSuccess) ? pr.multiProcessorCount : 256;
            }
            g.bg_cus = ncu - c->vgg_wg_cap > 0 ? ncu - c->vgg_wg_cap : 0;
            // ... and the large time-batched GEMMs walk their tiles persistently on that many workgroups instead of queueing hundreds
            // of them: they get the same CUs either way, but no longer take a convolution workgroup's CU at a kernel boundary
            static const char *kc = getenv("LRCN_BG_CAP");  // workgroups per free CU (0 = uncapped).  Round 3, three same-box rounds of
            const int capmul = kc ? atoi(kc) : 4;           // 2 / 4 / 6 / 8: 7.56 / 7.44 / 7.45 / 7.48 ms per step (round 2 had 2 ahead)
            if (capmul > 0 && g.bg_cus >= 8) g.wg_cap = g.bg_cus * capmul;
        }
    }
    hipError_t e = launch_gemm(on_wg_stream ? c->wg_stream : c->stream, g);
    if (e != hipSuccess) FAIL(c, LRCN_EHIP, "gemm M=%d N=%d K=%d: %s", M, N, K, hipGetErrorString(e));
    {   // kernel-development aid: LRCN_TRACE_ROUTES=1 prints which kernel family every contraction of a call took
        static const char *kt = getenv("LRCN_TRACE_ROUTES");
        if (kt && kt[0] == '1') fprintf(stderr, "gemm M=%d N=%d K=%d %s-> %s\n", M, N, K, on_wg_stream ? "(wg stream) " : "", gemm_debug_last_route());
    }
    return LRCN_OK;
}
#define GEMM(...)                     \
    do {                              \
        int _r = gemm(__VA_ARGS__);   \
        if (_r) return _r;            \
    } while (0)

DropSpec make_drop(const lrcn_dropout *d, int which) {
    DropSpec s{};
    s.which = which;
    if (d) {
        s.p = d->pdrop;
        s.seed = d->seed;
        s.mask = which == 1 ? d->mask1 : d->mask2;
        if (s.mask) s.p = 0.0f;
    }
    return s;
}

void ctx_sizes(const lrcn_ctx *c, int64_t sz[9]);
// the 14 training shadows as an array, in a fixed order (lrcn_ctx::alt mirrors it)
struct ShadowSet {
    void *W1x, *W1h, *W1xT, *W1hT, *W2x, *W2h, *W2xT, *W2hT, *Wpd, *WpT, *Wcd, *WeT, *Wod, *WoT;
};
ShadowSet cur_shadows(const lrcn_ctx *c) {
    return ShadowSet{c->W1x, c->W1h, c->W1xT, c->W1hT, c->W2x, c->W2h, c->W2xT, c->W2hT, c->Wpd, c->WpT, c->Wcd, c->WeT, c->Wod, c->WoT};
}
ShadowSet alt_shadows(const lrcn_ctx *c) {
    void *const *a = c->alt;
    return ShadowSet{a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13]};
}
void swap_shadow_sets(lrcn_ctx *c) {
    void **cur[14] = {&c->W1x, &c->W1h, &c->W1xT, &c->W1hT, &c->W2x, &c->W2h, &c->W2xT, &c->W2hT, &c->Wpd, &c->WpT, &c->Wcd, &c->WeT, &c->Wod, &c->WoT};
    for (int i = 0; i < 14; ++i) std::swap(*cur[i], c->alt[i]);
    if (c->alt_gi[0]) {
        std::swap(c->W1h_gi, c->alt_gi[0]);
        std::swap(c->W2h_gi, c->alt_gi[1]);
    }
}
int ensure_alt_shadows(lrcn_ctx *c) {
    if (c->alt[0]) return LRCN_OK;
    const size_t es = c->esz;
    const int E = c->E, H1 = c->H1, H2 = c->H2, h = c->h, V = c->V, X1 = c->X1;
    const size_t bytes[14] = {es * 4 * H1 * c->ldX1, es * 4 * H1 * c->ldH1, es * X1 * c->ld4H1, es * H1 * c->ld4H1,
                              es * 4 * H2 * c->ldH2, es * 4 * H2 * c->ldH2, es * H2 * c->ld4H2, es * H2 * c->ld4H2,
                              es * h * c->ldH1, es * H1 * c->ldh, es * h * LRCN_CNNOUT, es * V * c->ldE, es * V * c->ldH2, es * H2 * c->ldV};
    (void)E;
    for (int i = 0; i < 14; ++i) DALLOC(c, c->alt[i], bytes[i]);  // zero-filled: the K padding must hold zeros
    return LRCN_OK;
}
// the gate-interleaved recurrent weights of BOTH sets (the cell-epilogue route of a training step under LRCN_OPT_FUSED_UPDATE)
int ensure_gi_sets(lrcn_ctx *c) {
    const bool two = c->nl == 2;
    if (!c->W1h_gi) DALLOC(c, c->W1h_gi, c->esz * 4 * c->H1 * c->ldH1);
    if (two && !c->W2h_gi) DALLOC(c, c->W2h_gi, c->esz * 4 * c->H2 * c->ldH2);
    if (c->opt_fused) {
        if (!c->alt_gi[0]) DALLOC(c, c->alt_gi[0], c->esz * 4 * c->H1 * c->ldH1);
        if (two && !c->alt_gi[1]) DALLOC(c, c->alt_gi[1], c->esz * 4 * c->H2 * c->ldH2);
    }
    return LRCN_OK;
}

// the six parameter matrices of the model -> descriptors of their shadows in `w` (memory images: see the comments per line)
void plan_matrices(const lrcn_ctx *c, const float *const p[9], const ShadowSet &w, bool b, PrepPlan &plan, int only_a = -1, int only_b = -1,
                   void *const *gi = nullptr) {   // gi: {W1h, W2h} destinations with (unit, gate)-interleaved rows, or NULL
    const int E = c->E, H1 = c->H1, H2 = c->H2, h = c->h, V = c->V, X1 = c->X1;
    const bool two = c->nl == 2;
    auto add = [&](int k, int R, int C, int cs, void *dA, int64_t ldA, void *dB, int64_t ldB, void *tA, int64_t ldtA, void *tB, int64_t ldtB) {
        if (only_a >= 0 && k != only_a && k != only_b) return;
        PrepDesc &d = plan.d[plan.n++];
        d = PrepDesc{};
        d.src = p[k]; d.R = R; d.C = C; d.cs = cs;
        d.dA = dA; d.ldA = ldA; d.dB = dB; d.ldB = ldB;
        d.tA = tA; d.ldtA = ldtA; d.tB = tB; d.ldtB = ldtB;
    };
    (void)E;
    // W1: memory [4H1][X1 + H1] -> W1x | W1h (and their transposes [X1][ld4H1] | [H1][ld4H1] for the backward dX GEMMs)
    auto with_gi = [&](int k, void *dst, int64_t ld, int H) {   // the descriptor just added (if `only` kept it) also writes the interleaved copy
        if (!dst || (only_a >= 0 && k != only_a && k != only_b)) return;
        PrepDesc &d = plan.d[plan.n - 1];
        d.dG = dst; d.ldG = ld; d.giH = H;
    };
    add(0, 4 * H1, X1 + H1, X1, w.W1x, c->ldX1, w.W1h, c->ldH1, b ? w.W1xT : nullptr, c->ld4H1, b ? w.W1hT : nullptr, c->ld4H1);
    with_gi(0, gi ? gi[0] : nullptr, c->ldH1, H1);
    if (two) {
        add(2, 4 * H2, 2 * H2, H2, w.W2x, c->ldH2, w.W2h, c->ldH2, b ? w.W2xT : nullptr, c->ld4H2, b ? w.W2hT : nullptr, c->ld4H2);
        with_gi(2, gi ? gi[1] : nullptr, c->ldH2, H2);
        add(4, h, H1, H1, w.Wpd, c->ldH1, nullptr, 0, b ? w.WpT : nullptr, c->ldh, nullptr, 0);  // Wproj (H1 x h): memory [h][H1]
    }
    add(5, h, LRCN_CNNOUT, LRCN_CNNOUT, w.Wcd, LRCN_CNNOUT, nullptr, 0, nullptr, 0, nullptr, 0);   // Wcnn: memory [h][4096]
    add(6, E, V, V, nullptr, 0, nullptr, 0, w.WeT, c->ldE, nullptr, 0);                          // Wembed (V x E): memory [E][V] -> [V][ldE]
    add(7, V, H2, H2, w.Wod, c->ldH2, nullptr, 0, b ? w.WoT : nullptr, c->ldV, nullptr, 0);   // Wout (H2 x V): memory [V][H2]
}

// f32 column-major params -> K-contiguous shadows in T (direct and transposed).  See DESIGN.md "shadow weights".
int prepare_weights(lrcn_ctx *c, const float *const p[9], bool need_bwd, bool cat = false, bool gi = false, bool cat_perm = false,
                    bool dec_tables = false) {   // dec_tables: the interleaved recurrent copies + W2 without its x_cnn columns (a decode call: not sticky)
    const int dt = c->dt, H1 = c->H1, H2 = c->H2, X1 = c->X1;
    if (!p[0] || !p[1] || !p[5] || !p[6] || !p[7] || !p[8] || (c->nl == 2 && (!p[2] || !p[3] || !p[4]))) FAIL(c, LRCN_EINVAL, "null parameter tensor");
    hipStream_t st = c->stream;
    const bool two = c->nl == 2;
    // LRCN_OPT_FUSED_UPDATE: the previous train step's Adam kernel already wrote this set from these very parameters
    if (gi || dec_tables) {
        int rg = ensure_gi_sets(c);
        if (rg) return rg;
        if (gi) c->gi_live = true;   // from now on the fused update writes the interleaved copies with the other shadows
    }
    if (dec_tables) gi = true;
    if (c->opt_fused && c->shadow_valid && !cat && (!gi || c->shadow_has_gi)) {
        bool same = true;
        for (int k = 0; k < 9; ++k) same = same && c->shadow_p[k] == p[k];
        if (same) return LRCN_OK;
    }
    c->shadow_valid = false;
    c->refresh_groups = 0;  // a full shadow pass supersedes a per-group refresh sequence that was left unfinished
    PrepPlan plan{};
    void *const gi_cur[2] = {c->W1h_gi, c->W2h_gi};
    plan_matrices(c, p, cur_shadows(c), need_bwd, plan, -1, -1, gi ? gi_cur : nullptr);
    if (dec_tables && c->nl == 2) {
        // W2 (memory [4H2][2 H2]: columns [h1 Wproj (h) | x_cnn (h) | h2 (H2)]) -> dec_W2c [4H2][ldh + ldH2] = [proj columns | h2 columns], rows
        // (unit, gate)-interleaved: two descriptors over the same source, each with one live side
        const int64_t ld = c->ldh + c->ldH2;
        PrepDesc &a = plan.d[plan.n++];
        a = PrepDesc{};
        a.src = p[2]; a.R = 4 * H2; a.C = 2 * H2; a.cs = c->h; a.dA = c->dec_W2c; a.ldA = ld; a.permH = H2;
        PrepDesc &b = plan.d[plan.n++];
        b = PrepDesc{};
        b.src = p[2]; b.R = 4 * H2; b.C = 2 * H2; b.cs = H2; b.dB = boff(c->dec_W2c, c->ldh, c->esz); b.ldB = ld; b.permH = H2;
    }
    if (cat) {  // batched decode: W1 / W2 with the x and h column blocks each padded to whole K-steps, side by side
        // cat_perm: the rows in (unit, gate)-interleaved order, for the decode step with the cell math in the GEMM's epilogue
        auto add = [&](const float *src, int R, int C, int cs, void *dA, int64_t ldA, void *dB, int64_t ldB, int permH) {
            PrepDesc &d = plan.d[plan.n++];
            d = PrepDesc{};
            d.src = src; d.R = R; d.C = C; d.cs = cs; d.dA = dA; d.ldA = ldA; d.dB = dB; d.ldB = ldB;
            d.permH = cat_perm ? permH : 0;
        };
        add(p[0], 4 * H1, X1 + H1, X1, c->W1cat, c->ldXH1, boff(c->W1cat, c->ldX1, c->esz), c->ldXH1, H1);
        if (two) add(p[2], 4 * H2, 2 * H2, H2, c->W2cat, c->ldXH2, boff(c->W2cat, c->ldH2, c->esz), c->ldXH2, H2);
    }
    k_prepare_weights(st, dt, plan);
    KCHK(c, "prepare_weights");
    return LRCN_OK;
}

// update! (lrcn.jl:394) of the tensors of gradient group `group` (-1: all nine) fused with the NEXT step's shadow pass (LRCN_OPT_FUSED_UPDATE):
// the kernel writes the not-current shadow set; the caller swaps the sets once every group has been issued.
int adam_fused(lrcn_ctx *c, float *const p[9], const float *const g[9], float *const m[9], float *const v[9], int group, int step, float lr,
               float b1, float b2, float eps, hipStream_t st) {
    static const int kGroup[LRCN_GRAD_GROUPS][2] = {{7, 8}, {2, 3}, {4, 5}, {0, 1}, {6, 6}};
    int r = ensure_alt_shadows(c);
    if (r) return r;
    int64_t sz[9];
    ctx_sizes(c, sz);
    PrepPlan plan{};
    const int ka = group < 0 ? -1 : kGroup[group][0], kb = group < 0 ? -1 : kGroup[group][1];
    if (c->gi_live && (r = ensure_gi_sets(c))) return r;
    plan_matrices(c, p, alt_shadows(c), true, plan, ka, kb, c->gi_live ? c->alt_gi : nullptr);
    for (int i = 0; i < plan.n; ++i) {
        PrepDesc &d = plan.d[i];
        int k = 0;
        while (k < 9 && p[k] != d.src) ++k;
        d.g = g[k]; d.m = m[k]; d.v = v[k];
    }
    for (int k : {1, 3, 8}) {  // the biases: plain Adam, one "row" of n elements
        if (sz[k] == 0 || (group >= 0 && k != ka && k != kb)) continue;
        PrepDesc &d = plan.d[plan.n++];
        d = PrepDesc{};
        d.src = p[k]; d.g = g[k]; d.m = m[k]; d.v = v[k];
        d.R = 1; d.C = (int)sz[k]; d.cs = (int)sz[k];
    }
    // A group's update issued on its own stream runs BESIDE the rest of the backward pass (the per-group pipeline).  Kernel-development knob
    // LRCN_ADAM_GROUP_WGS = n > 0: only n persistent workgroups walk the tiles of groups 0..3, so that the update does not take every CU from
    // the recurrence's latency-bound kernels.  MEASURED AND LEFT OFF (emulated rank of 8, per-group pipeline, ms per step): uncapped 1.40 /
    // 1.41, 192 workgroups 1.44 / 1.46, 128: 1.50, 96: 1.63, 64: 1.96, 32: 2.93 -- the capped stream of Wout's 340 MB is still running when
    // the step joins its update stream; the groups' updates are on the critical path, not beside it.
    {
        static const char *kg = getenv("LRCN_ADAM_GROUP_WGS");
        const int cap = kg ? atoi(kg) : 0;
        plan.grid_cap = (group >= 0 && group < LRCN_GRAD_GROUPS - 1 && st != c->stream) ? cap : 0;
    }
    k_adam_shadows(st, c->dt, plan, step, lr, b1, b2, eps);
    KCHK(c, "adam (fused with the shadow pass)");
    return LRCN_OK;
}
void fused_update_done(lrcn_ctx *c, float *const p[9]) {  // every tensor's Adam has been issued: the written set becomes the current one
    swap_shadow_sets(c);
    for (int k = 0; k < 9; ++k) c->shadow_p[k] = p[k];
    c->shadow_valid = true;
    c->shadow_has_gi = c->gi_live && c->alt_gi[0] != nullptr;
}

// One LSTM layer over all S steps.  Gx f32 [M][4H] holds the input-side pre-activations (+bias) on entry and the full
// pre-activations on exit; acts/Call/Hall receive the per-step results.  (lrcn.jl:528-538, time-batched)
// nothing runs beside the LSTM step: no VGG forward with capped grids on another stream (what the two-stream trainer sets up)
bool lstm_alone(const lrcn_ctx *c) { return !(c->vgg_wg_cap >= 8 && c->vgg_loaded); }

// A pair of HIP events around one segment of a call, on the stream its work is launched on (lrcn_profile level 2; a no-op otherwise).
struct SegScope {
    lrcn_ctx *c;
    hipStream_t st;
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
    SegScope(lrcn_ctx *c_, int seg, hipStream_t st_, double bytes) : c(c_), st(st_) {
        if (c->prof_level < 2) return;
        auto &sp = c->seg[seg];
        if (sp.used == sp.ev.size()) {
            std::pair<hipEvent_t, hipEvent_t> e;
            if (hipEventCreate(&e.first) != hipSuccess || hipEventCreate(&e.second) != hipSuccess) return;
            sp.ev.push_back(e);
        }
        ev = &sp.ev[sp.used++];
        sp.bytes += bytes;
        sp.n += 1;
        (void)hipEventRecord(ev->first, st);
    }
    ~SegScope() {
        if (ev) (void)hipEventRecord(ev->second, st);
    }
};
bool lstm_fused_on(lrcn_ctx *c, int B, int H, int64_t ldH, int64_t ld4H) {
    const char *k = getenv("LRCN_LSTM_FUSED");  // LRCN_LSTM_FUSED=0: GEMM + cell as separate launches at every batch size
    const char *mb = getenv("LRCN_LSTM_FUSED_MAXB");  // kernel-development knob: largest batch routed to the fused step kernels
    return !(k && k[0] == '0') && B <= (mb ? atoi(mb) : 128) && lstm_fused_eligible(c->dt, B, H, ldH, ld4H);
}
// The recurrent GEMM with the cell math in its epilogue (gemm_8p.hip GEMM_OUT_LSTM_*), for the two-stream training step at 256..512
// rows per GPU: one launch of 32 (forward) / 8 (backward) workgroups per timestep instead of GEMM + cell kernel.  LRCN_LSTM_EPI=f turns
// the forward one on, =1 both (tests/test_gpu_lstm_parity.py checks them against the CPU oracle).  OFF BY DEFAULT, with numbers:
//   round 2: one unit per epilogue thread -- per timestep beside the VGG forward, forward 45 us fused vs 27 + 9.6 us as two launches,
//            backward 87 us vs 55 + 8.7 us; training step 7.49 vs 7.22 ms.
//   round 6: the forward epilogue rewritten (four units per thread, c_prev / Gx requested before the staging, 16- / 8-byte accesses) and
//            its gate-interleaved weight copy kept current by the fused update: the forward recurrence's segment goes 1.00 -> 0.83 ms
//            per step (18.8 us per step and layer: FASTER than the two launches), and the training step does not move or gets slower:
//            five same-box pairs 6.858 -> 6.942 ms mean (profiles/r06_ab_c4_step_forward_cell_epilogue.txt), with the shader clock the
//            package holds in the timed region 2.163 -> 2.130 GHz median (bench line hw_held_in_timed_region) -- the convolution launches
//            beside it slow by what the chain gained, once more.  The backward epilogue stays at 8 workgroups (N = H:
//            eight 128-column tiles) and 1.94 vs 1.72 ms per step; a 32-workgroup form needs split-K, whose partial sums can only be
//            combined by a second launch or a grid barrier.
// The same forward epilogue IS the default of the batched beam decode (decode_gates_epi below), where its GEMMs fill the chip.
bool lstm_epi_on(lrcn_ctx *c, int B) {
    const char *k = getenv("LRCN_LSTM_EPI"), *kb = getenv("LRCN_BG_ROUTE");
    return c->dt == GEMM_T_BF16 && c->vgg_wg_cap >= 8 && c->vgg_loaded && B >= 256 && B <= 512 && !(c->H1 & 3) && !(c->H2 & 3) &&
           (k && (k[0] == '1' || k[0] == 'f')) && !(kb && kb[0] == '0');
}
// LRCN_LSTM_EPI=f: the forward recurrence only (its launch has 32 workgroups -- one per free CU; the backward dh GEMM has N = H: 8 tiles)
bool lstm_epi_bwd_on(lrcn_ctx *c, int B, int H) {
    const char *k = getenv("LRCN_LSTM_EPI");
    return lstm_epi_on(c, B) && k && k[0] == '1' && H >= 128;   // its GEMM has N = H columns: at least one 128-column tile
}
int lstm_layer_fwd(lrcn_ctx *c, int S, int B, int H, int64_t ldH, int64_t ld4H, float *Gx, const void *Wh, void *acts,
                   float *Call, void *Hall, const void *Wh_gi = nullptr) {
    const int dt = c->dt;
    const bool fused = lstm_fused_on(c, B, H, ldH, ld4H);
    const bool epi = !fused && Wh_gi && lstm_epi_on(c, B);
    SegScope seg(c, LRCN_SEG_REC_FWD, c->stream, (double)(S - 1) * 4.0 * H * H * c->esz);  // one read of Wh (4H x H) per recurrent step
    for (int s = 0; s < S; ++s) {
        float *G = Gx + (int64_t)s * B * 4 * H;
        if (s > 0 && epi) {
            GemmArgs g{};
            g.dtype = dt;
            g.A = boff(Hall, (int64_t)(s - 1) * B * ldH, c->esz); g.lda = ldH;
            g.B = Wh_gi; g.ldb = ldH;
            g.M = B; g.N = 4 * H; g.K = (int)ldH;
            g.a_mode = GEMM_A_PLAIN;
            g.out_mode = GEMM_OUT_LSTM_FWD;
            g.zero_page = c->zero_page;
            g.lstm.H = H; g.lstm.ld_a = ld4H; g.lstm.ld_h = ldH;
            g.lstm.Gx = G;
            g.lstm.c_prev = Call + (int64_t)(s - 1) * B * H;
            g.lstm.c_out = Call + (int64_t)s * B * H;
            g.lstm.acts = boff(acts, (int64_t)s * B * ld4H, c->esz);
            g.lstm.h_new = boff(Hall, (int64_t)s * B * ldH, c->esz);
            hipError_t e = launch_gemm_8p(c->stream, g);
            if (e != hipSuccess) FAIL(c, LRCN_EHIP, "lstm fwd step (GEMM + cell epilogue): %s", hipGetErrorString(e));
            continue;
        }
        if (s > 0 && fused) {  // recurrent GEMM + cell in one launch (small batches: launch-latency bound otherwise)
            hipError_t e = launch_lstm_rec_fwd(c->stream, boff(Hall, (int64_t)(s - 1) * B * ldH, c->esz), ldH, Wh, G,
                                               Call + (int64_t)(s - 1) * B * H, B, H, boff(acts, (int64_t)s * B * ld4H, c->esz), ld4H,
                                               Call + (int64_t)s * B * H, boff(Hall, (int64_t)s * B * ldH, c->esz), c->zero_page, lstm_alone(c));
            if (e != hipSuccess) FAIL(c, LRCN_EHIP, "lstm_rec_fwd: %s", hipGetErrorString(e));
            continue;
        }
        if (s > 0)
            GEMM(c, dt, boff(Hall, (int64_t)(s - 1) * B * ldH, c->esz), ldH, Wh, ldH, G, 4 * H, B, 4 * H, H, nullptr, true,
                 true);
        k_lstm_fwd(c->stream, dt, G, 4 * H, s ? Call + (int64_t)(s - 1) * B * H : nullptr, B, H,
                   boff(acts, (int64_t)s * B * ld4H, c->esz), ld4H, Call + (int64_t)s * B * H,
                   boff(Hall, (int64_t)s * B * ldH, c->esz), ldH, nullptr);
    }
    KCHK(c, "lstm_layer_fwd");
    return LRCN_OK;
}

// Reverse recurrence of one layer: dHall f32 [M][H] (external dh per step) -> dZ (T) [M][ld4H].
int lstm_layer_bwd(lrcn_ctx *c, int S, int B, int H, int64_t ld4H, const void *acts, const float *Call, const float *dHall,
                   const void *WhT, void *dZ) {
    const int dt = c->dt;
    SegScope seg(c, LRCN_SEG_REC_BWD, c->stream, (double)(S - 1) * 4.0 * H * H * c->esz);
    if (!lstm_fused_on(c, B, H, round_up64(H, 64), ld4H) && lstm_epi_bwd_on(c, B, H)) {
        // cell backward of the last step, then one launch per step: dh_rec = dZ[s] Wh with the cell backward of s-1 in its epilogue
        k_lstm_bwd(c->stream, dt, boff(acts, (int64_t)(S - 1) * B * ld4H, c->esz), ld4H, S > 1 ? Call + (int64_t)(S - 2) * B * H : nullptr,
                   Call + (int64_t)(S - 1) * B * H, dHall + (int64_t)(S - 1) * B * H, H, nullptr, 0, c->dc, 1, B, H,
                   boff(dZ, (int64_t)(S - 1) * B * ld4H, c->esz), ld4H);
        for (int s = S - 1; s >= 1; --s) {
            GemmArgs g{};
            g.dtype = dt;
            g.A = boff(dZ, (int64_t)s * B * ld4H, c->esz); g.lda = ld4H;
            g.B = WhT; g.ldb = ld4H;
            g.M = B; g.N = H; g.K = (int)ld4H;
            g.a_mode = GEMM_A_PLAIN;
            g.out_mode = GEMM_OUT_LSTM_BWD;
            g.zero_page = c->zero_page;
            g.lstm.H = H; g.lstm.ld_a = ld4H;
            g.lstm.acts = const_cast<char *>(boff(acts, (int64_t)(s - 1) * B * ld4H, c->esz));
            g.lstm.c_prev = s > 1 ? Call + (int64_t)(s - 2) * B * H : nullptr;
            g.lstm.c_new = Call + (int64_t)(s - 1) * B * H;
            g.lstm.dh_ext = dHall + (int64_t)(s - 1) * B * H;
            g.lstm.dc = c->dc;
            g.lstm.dz_out = boff(dZ, (int64_t)(s - 1) * B * ld4H, c->esz);
            hipError_t e = launch_gemm_8p(c->stream, g);
            if (e != hipSuccess) FAIL(c, LRCN_EHIP, "lstm bwd step (GEMM + cell epilogue): %s", hipGetErrorString(e));
        }
        KCHK(c, "lstm_layer_bwd (epilogue)");
        return LRCN_OK;
    }
    if (lstm_fused_on(c, B, H, round_up64(H, 64), ld4H)) {
        // cell backward of the last step, then one launch per step: dh_rec = dZ[s] Wh fused with the cell backward of s-1
        k_lstm_bwd(c->stream, dt, boff(acts, (int64_t)(S - 1) * B * ld4H, c->esz), ld4H, S > 1 ? Call + (int64_t)(S - 2) * B * H : nullptr,
                   Call + (int64_t)(S - 1) * B * H, dHall + (int64_t)(S - 1) * B * H, H, nullptr, 0, c->dc, 1, B, H,
                   boff(dZ, (int64_t)(S - 1) * B * ld4H, c->esz), ld4H);
        for (int s = S - 1; s >= 1; --s) {
            hipError_t e = launch_lstm_rec_bwd(c->stream, boff(dZ, (int64_t)s * B * ld4H, c->esz), ld4H, WhT,
                                               boff(acts, (int64_t)(s - 1) * B * ld4H, c->esz), s > 1 ? Call + (int64_t)(s - 2) * B * H : nullptr,
                                               Call + (int64_t)(s - 1) * B * H, dHall + (int64_t)(s - 1) * B * H, c->dc, B, H,
                                               boff(dZ, (int64_t)(s - 1) * B * ld4H, c->esz), c->zero_page, lstm_alone(c));
            if (e != hipSuccess) FAIL(c, LRCN_EHIP, "lstm_rec_bwd: %s", hipGetErrorString(e));
        }
        KCHK(c, "lstm_layer_bwd (fused)");
        return LRCN_OK;
    }
    // Beside the capped convolution grids at 256..512 rows the dh GEMM (M = B, N = H, K = 4H) has EIGHT 256 x 128 tiles: 8 of the 32 free CUs,
    // 3 MB of operand ingest each (55 us per timestep).  LRCN_BWD_SLABS=n (2..8; round 6): n K-slices per tile = 8 n workgroups, each writing
    // its partial tile to an f32 slab; the NEXT cell kernel sums the slabs (no reduce launch, fixed order: deterministic).  With n = 4, four
    // same-box pairs (profiles/r06_ab_bwd_slabs.txt): the backward recurrence's segment 1.86 -> 0.87 ms per step, the step 7.270 -> 7.241 ms
    // (-0.4 %: inside a lease's spread), the convolution launches beside the busier chain 0.571 -> 0.587 ms (+2.8 %, roofline.frac -0.013).
    // OFF BY DEFAULT like the forward cell epilogue: the LSTM chain is not what bounds the step, and a shorter chain is returned as a lower
    // clock for the convolutions; the route is kept, tested against the CPU oracle, for a configuration where the chain matters.
    {
        const char *ksl = getenv("LRCN_BWD_SLABS");   // read per call (the tests switch it inside one process)
        const int nsl = ksl ? atoi(ksl) : 0;
        const int Kp = (int)round_up64(4 * H, 64);
        if (nsl >= 2 && nsl <= 8 && dt == GEMM_T_BF16 && c->vgg_wg_cap >= 8 && c->vgg_loaded && B >= 256 && B <= 512 && !(H & 3) && H >= 128 &&
            Kp / 64 >= 8 * nsl && Kp <= ld4H && (size_t)nsl * B * H * sizeof(float) <= c->gemm_ws_bytes && c->gemm_ws) {
            float *slabs = reinterpret_cast<float *>(c->gemm_ws);
            for (int s = S - 1; s >= 0; --s) {
                k_lstm_bwd(c->stream, dt, boff(acts, (int64_t)s * B * ld4H, c->esz), ld4H, s ? Call + (int64_t)(s - 1) * B * H : nullptr,
                           Call + (int64_t)s * B * H, dHall + (int64_t)s * B * H, H, slabs, s < S - 1, c->dc, s == S - 1, B, H,
                           boff(dZ, (int64_t)s * B * ld4H, c->esz), ld4H, nsl);
                if (s > 0) {
                    GemmArgs g{};
                    g.dtype = dt;
                    g.A = boff(dZ, (int64_t)s * B * ld4H, c->esz); g.lda = ld4H;
                    g.B = WhT; g.ldb = ld4H;
                    g.M = B; g.N = H; g.K = Kp;
                    g.C = slabs; g.ldc = H; g.c_f32 = 1;   // (unused: the slabs are the output)
                    g.a_mode = GEMM_A_PLAIN; g.out_mode = GEMM_OUT_PLAIN;
                    g.zero_page = c->zero_page;
                    g.ws = c->gemm_ws; g.ws_bytes = c->gemm_ws_bytes;
                    g.cfg_pref = 2;
                    g.splitk_forced = 1; g.splitk_no_reduce = 1;
                    hipError_t e = launch_gemm_8p(c->stream, g, nsl);
                    if (e != hipSuccess) FAIL(c, LRCN_EHIP, "lstm bwd step (split-K slabs): %s", hipGetErrorString(e));
                }
            }
            KCHK(c, "lstm_layer_bwd (K slices summed by the cell kernel)");
            return LRCN_OK;
        }
    }
    for (int s = S - 1; s >= 0; --s) {
        k_lstm_bwd(c->stream, dt, boff(acts, (int64_t)s * B * ld4H, c->esz), ld4H, s ? Call + (int64_t)(s - 1) * B * H : nullptr,
                   Call + (int64_t)s * B * H, dHall + (int64_t)s * B * H, H, c->dhrec, s < S - 1, c->dc, s == S - 1, B, H,
                   boff(dZ, (int64_t)s * B * ld4H, c->esz), ld4H);
        if (s > 0)  // dh_prev = dZ[s] * Wh'   (Wh' K-contiguous = WhT [H][ld4H]); dhrec was zeroed by the cell kernel above
            GEMM(c, dt, boff(dZ, (int64_t)s * B * ld4H, c->esz), ld4H, WhT, ld4H, c->dhrec, H, B, H, 4 * H, nullptr, true, false, false, true);
    }
    KCHK(c, "lstm_layer_bwd");
    return LRCN_OK;
}

int check_shapes(lrcn_ctx *c, int T, int B, int norm_B) {
    if (T < 0 || T + 1 > c->maxS) FAIL(c, LRCN_EINVAL, "T=%d outside [0,%d]", T, c->maxS - 1);
    if (B < 1 || B > c->maxB) FAIL(c, LRCN_EINVAL, "B=%d outside [1,%d]", B, c->maxB);
    if (norm_B < 1) FAIL(c, LRCN_EINVAL, "norm_B=%d must be >= 1", norm_B);
    return LRCN_OK;
}

// loss / lossgradient on internal buffers. feats: B x 4096 column-major f32 (device).
int loss_impl(lrcn_ctx *c, const float *const p[9], const float *feats, const int32_t *tokens, int T, int B, int norm_B,
              const lrcn_dropout *drop, float *const grads[9], float *logits_out) {
    int r = check_shapes(c, T, B, norm_B);
    if (r) return r;
    if (drop && (drop->pdrop < 0.0f || drop->pdrop >= 1.0f)) FAIL(c, LRCN_EINVAL, "pdrop=%g outside [0,1)", drop->pdrop);
    if (drop && c->nl == 2 && ((drop->mask1 == nullptr) != (drop->mask2 == nullptr))) FAIL(c, LRCN_EINVAL, "mask1/mask2 must both be set");
    const int dt = c->dt, E = c->E, H1 = c->H1, H2 = c->H2, h = c->h, V = c->V, X1 = c->X1;
    const int S = T + 1, M = S * B;
    const size_t es = c->esz;
    hipStream_t st = c->stream;
    const bool bwd = grads != nullptr, two = c->nl == 2;
    if (bwd && (!grads[0] || !grads[1] || !grads[5] || !grads[6] || !grads[7] || !grads[8] || (two && (!grads[2] || !grads[3] || !grads[4]))))
        FAIL(c, LRCN_EINVAL, "null gradient tensor");
    const DropSpec d1 = make_drop(drop, 1), d2 = make_drop(drop, 2);
    const DropSpec none{};

    c->cur_B = B;
    const bool epi = lstm_epi_on(c, B) && !lstm_fused_on(c, B, H1, c->ldH1, c->ld4H1);
    r = prepare_weights(c, p, bwd, false, epi);
    if (r) return r;
    k_build_tokens(st, tokens, T, B, V, c->tok_in, c->tok_tgt, c->logp);  // reads the caller's (T, B) ids once (T = 0: never)
    // input = input * param[end-3]   lrcn.jl:558.  The two-layer model needs x_cnn only at LSTM-2's input, after the whole first
    // recurrence: its three launches (transpose, split-K GEMM, reduce: ~20 us at 32 rows) run on the weight-gradient stream beside that
    // chain and are joined before the concat (round 5).  Not beside the capped VGG forward from 256 rows (as the weight gradients:
    // there a second stream of LSTM-side workgroups takes CUs from the convolutions); LRCN_WG_STREAM=0 / 1 forces it off / on.
    const char *kwgf = getenv("LRCN_WG_STREAM");
    static const char *kxf = getenv("LRCN_XCNN_FORK");  // development knob: 0 keeps the image embedding on the main stream
    const bool xfork = two && !(kxf && kxf[0] == '0') && (kwgf ? kwgf[0] != '0' : !(c->vgg_wg_cap >= 8 && c->vgg_loaded && B >= 256));
    auto image_embedding = [&](hipStream_t s_, bool on_wg) -> int {
        // feats (B x 4096 column-major = memory [4096][B]) -> F [B][4096] (T)
        k_transpose(s_, dt, 1, feats, B, LRCN_CNNOUT, B, c->F, LRCN_CNNOUT, 0);
        return gemm(c, dt, c->F, LRCN_CNNOUT, c->Wcd, LRCN_CNNOUT, c->xcnn, c->ldh, B, h, LRCN_CNNOUT, nullptr, true, false, false, false, on_wg);
    };
    int rx = LRCN_OK;
    if (xfork) {
        HIPCHK(c, hipEventRecord(c->xc_fork, st));
        HIPCHK(c, hipStreamWaitEvent(c->wg_stream, c->xc_fork, 0));
        rx = image_embedding(c->wg_stream, true);
        (void)hipEventRecord(c->xc_done, c->wg_stream);
    } else {
        rx = image_embedding(st, false);
        if (rx) return rx;
    }
    auto first_layer = [&]() -> int {
        // embeddings of [bos, tokens...] with the :542 dropout.  LRCN-1f: the LSTM input is dropout(hcat(embedding, x_cnn)) -- the
        // gather fills columns [0, E), the concat kernel appends x_cnn and applies the one mask over all E + h columns
        {
            SegScope seg(c, LRCN_SEG_EMBED_GATHER, st, 2.0 * M * E * es);  // (T+1) B rows of E elements read and written
            k_embed_gather(st, dt, c->WeT, c->ldE, c->tok_in, S, B, E, two ? d1 : none, c->Xemb, c->ldX1);
        }
        if (!two) k_concat_x2(st, dt, c->Xemb, c->ldX1, c->xcnn, c->ldh, S, B, E, h, d1);
        // LSTM 1
        GEMM(c, dt, c->Xemb, c->ldX1, c->W1x, c->ldX1, c->G1, 4 * H1, M, 4 * H1, X1, p[1], true);
        return lstm_layer_fwd(c, S, B, H1, c->ldH1, c->ld4H1, c->G1, c->W1h, c->A1, c->C1, c->H1all, epi ? c->W1h_gi : nullptr);
    };
    r = first_layer();
    if (xfork) {  // joined on EVERY exit: the side chain reads the caller's feats
        if (hipStreamWaitEvent(st, c->xc_done, 0) != hipSuccess) (void)hipStreamSynchronize(c->wg_stream);
        if (rx) return rx;
    }
    if (r) return r;
    const void *Htop = c->H1all;  // the hidden states the logits are computed from
    if (two) {
        // x = s[1]*w[end-4]; x = hcat(x, x_cnn); x = dropout(x)    lrcn.jl:544-547
        GEMM(c, dt, c->H1all, c->ldH1, c->Wpd, c->ldH1, c->X2, c->ldH2, M, h, H1, nullptr, false);
        k_concat_x2(st, dt, c->X2, c->ldH2, c->xcnn, c->ldh, S, B, h, h, d2);
        // LSTM 2
        GEMM(c, dt, c->X2, c->ldH2, c->W2x, c->ldH2, c->G2, 4 * H2, M, 4 * H2, H2, p[3], true);
        r = lstm_layer_fwd(c, S, B, H2, c->ldH2, c->ld4H2, c->G2, c->W2h, c->A2, c->C2, c->H2all, epi ? c->W2h_gi : nullptr);
        if (r) return r;
        Htop = c->H2all;
    }
    // logits for all steps: x * w[end-1] .+ w[end]   lrcn.jl:550
    GEMM(c, dt, Htop, c->ldH2, c->Wod, c->ldH2, c->Logits, c->ldV, M, V, H2, p[8], true);
    if (logits_out) {  // (T+1) blocks of B x V column-major: block s memory [V][B]
        for (int s = 0; s < S; ++s)
            k_transpose_f32(st, c->Logits + (int64_t)s * B * c->ldV, c->ldV, B, V, logits_out + (int64_t)s * B * V, B);
    }
    const float scale = (float)(1.0 / ((double)norm_B * (double)S));
    if (c->opt_det && !c->logp_rows) DALLOC(c, c->logp_rows, sizeof(double) * (size_t)c->maxS * c->maxB);
    k_softmax_xent(st, dt, c->Logits, c->ldV, c->tok_tgt, M, V, scale, c->logp, bwd ? c->dLog : nullptr, c->ldV, c->opt_det ? c->logp_rows : nullptr);
    c->last_norm = norm_B;
    c->last_S = S;
    KCHK(c, "forward");
    if (!bwd) return LRCN_OK;

    const int64_t ldM = ld8(M), ldB = ld8(B);
    // Transposed operands of the weight-gradient GEMMs (contraction over the M = S*B rows) are materialised K-contiguous,
    // several per launch; the x- and h-side inputs of one LSTM share one stacked buffer so that dW = dZ' [x | h_prev] is
    // one GEMM per layer.
    auto tr = [&](TrPlan &pl, const void *src, int64_t ld_src, int R, int C, void *dst, int shift) {
        TrDesc &d = pl.d[pl.n++];
        d.src = src; d.ld_src = ld_src; d.R = R; d.C = C; d.dst = dst; d.ld_dst = ldM; d.shift = shift;
    };
    // The weight / bias gradients feed nothing but update!: they run on the context's weight-gradient stream (sw), forked from the
    // main chain by an event each time their operands are final, while the main stream goes on with the reverse recurrences --
    // chains of small launches that leave most CUs idle (alone on the chip) or some of the 32 free ones (beside the VGG forward).
    // Both streams only READ shared activations; TA / TB / dxcT / FT and the second split-K workspace belong to sw alone.
    // grad_ev[g] is recorded on whichever stream finalises group g; the main stream joins sw before the call returns.
    // Not from 256 rows per GPU beside the capped VGG forward: there the convolutions are the critical path and a second stream
    // of LSTM-side workgroups takes CUs from them at every kernel boundary (measured on one box, ms/step off -> on: LSTM step
    // alone 2.117 -> 2.052 at 256 rows, 1.086 -> 1.072 at 32; two-stream step 1.679 -> 1.611 at 32 but 7.38 -> 7.54 at 256).
    // LRCN_WG_STREAM=0 / 1 forces it off / on.
    const char *kwg = getenv("LRCN_WG_STREAM");
    const bool beside_vgg = c->vgg_wg_cap >= 8 && c->vgg_loaded && B >= 256;
    const bool par = kwg ? kwg[0] != '0' : !beside_vgg;
    hipStream_t sw = par ? c->wg_stream : st;
    int nfork = 0;
    auto fork = [&]() -> int {  // sw waits for everything issued on the main stream so far
        if (!par) return LRCN_OK;
        HIPCHK(c, hipEventRecord(c->wg_fork[nfork], st));
        HIPCHK(c, hipStreamWaitEvent(sw, c->wg_fork[nfork], 0));
        ++nfork;
        return LRCN_OK;
    };
#define FORK()                  \
    do {                        \
        int _r = fork();        \
        if (_r) return _r;      \
    } while (0)
    // Everything from the first fork on runs inside one scope whose every exit -- the normal one and each early error return --
    // is followed by the join below: an error after a fork must not leave the weight-gradient stream writing the caller's grads[]
    // (and reading the caller's feats) after the call has returned.
    auto backward = [&]() -> int {
    // ---- logits layer: dWout, dbout (sw) | dH2 (main) ----
        FORK();
        {
            TrPlan pl{};
            tr(pl, c->dLog, c->ldV, M, V, c->TA, 0);      // dLog^T [V][ldM]
            tr(pl, Htop, c->ldH2, M, H2, c->TB, 0);       // H2all^T [H2][ldM]
            k_transpose_multi(sw, dt, pl);
        }
        GEMM(c, dt, c->TA, ldM, c->TB, ldM, grads[7], H2, V, H2, M, nullptr, true, false, false, false, par);
        k_colsum(sw, dt, c->dLog, c->ldV, M, V, grads[8], c->opt_det);
        HIPCHK(c, hipEventRecord(c->grad_ev[0], sw));  // group 0: Wout, bout
        GEMM(c, dt, c->dLog, c->ldV, c->WoT, c->ldV, two ? c->dH2all : c->dH1all, H2, M, H2, V, nullptr, true);
        if (two) {
            // ---- LSTM 2 ----
            r = lstm_layer_bwd(c, S, B, H2, c->ld4H2, c->A2, c->C2, c->dH2all, c->W2hT, c->dZ2);
            if (r) return r;
            FORK();
            {
                TrPlan pl{};
                tr(pl, c->dZ2, c->ld4H2, M, 4 * H2, c->TA, 0);                              // dZ2^T [4H2][ldM]
                tr(pl, c->X2, c->ldH2, M, H2, c->TB, 0);                                     // X2^T [2h][ldM]
                tr(pl, c->H2all, c->ldH2, M - B, H2, boff(c->TB, (int64_t)H2 * ldM, es), M > B ? B : 0);  // h2_prev^T (one step later)
                k_transpose_multi(sw, dt, pl);
            }
            GEMM(c, dt, c->TA, ldM, c->TB, ldM, grads[2], 2 * H2, 4 * H2, 2 * H2, M, nullptr, true, false, false, false, par);
            k_colsum(sw, dt, c->dZ2, c->ld4H2, M, 4 * H2, grads[3], c->opt_det);
        }
        HIPCHK(c, hipEventRecord(c->grad_ev[1], sw));  // group 1: W2, b2
        if (two) {
            GEMM(c, dt, c->dZ2, c->ld4H2, c->W2xT, c->ld4H2, c->dX2, c->ldH2, M, H2, 4 * H2, nullptr, false);
            k_dx2_mask_reduce(st, dt, c->dX2, c->ldH2, S, B, h, h, d2, c->dxcnn, c->ldh);
            // ---- projection and image embedding: dWproj, dWcnn (sw) | dH1 (main) ----
            FORK();
            {
                TrPlan pl{};
                tr(pl, c->dX2, c->ldH2, M, h, c->TA, 0);      // dP^T [h][ldM]
                tr(pl, c->H1all, c->ldH1, M, H1, c->TB, 0);   // H1all^T [H1][ldM]
                k_transpose_multi(sw, dt, pl);
            }
            GEMM(c, dt, c->TA, ldM, c->TB, ldM, grads[4], H1, h, H1, M, nullptr, true, false, false, false, par);
            GEMM(c, dt, c->dX2, c->ldH2, c->WpT, c->ldh, c->dH1all, H1, M, H1, h, nullptr, true);
            k_transpose(sw, dt, 1, c->dxcnn, c->ldh, B, h, c->dxcT, ldB, 0);     // dxcnn^T [h][ldB]
            k_cast_rows(sw, dt, feats, B, LRCN_CNNOUT, B, c->FT, ldB);           // feats^T [4096][ldB] (it already is, in memory)
            GEMM(c, dt, c->dxcT, ldB, c->FT, ldB, grads[5], LRCN_CNNOUT, h, LRCN_CNNOUT, B, nullptr, true, false, false, false, par);
            HIPCHK(c, hipEventRecord(c->grad_ev[2], sw));  // group 2: Wproj, Wcnn
        }
        // ---- LSTM 1 ----
        r = lstm_layer_bwd(c, S, B, H1, c->ld4H1, c->A1, c->C1, c->dH1all, c->W1hT, c->dZ1);
        if (r) return r;
        FORK();
        {
            TrPlan pl{};
            tr(pl, c->dZ1, c->ld4H1, M, 4 * H1, c->TA, 0);
            tr(pl, c->Xemb, c->ldX1, M, X1, c->TB, 0);
            tr(pl, c->H1all, c->ldH1, M - B, H1, boff(c->TB, (int64_t)X1 * ldM, es), M > B ? B : 0);
            k_transpose_multi(sw, dt, pl);
        }
        GEMM(c, dt, c->TA, ldM, c->TB, ldM, grads[0], X1 + H1, 4 * H1, X1 + H1, M, nullptr, true, false, false, false, par);
        k_colsum(sw, dt, c->dZ1, c->ld4H1, M, 4 * H1, grads[1], c->opt_det);
        HIPCHK(c, hipEventRecord(c->grad_ev[3], sw));  // group 3: W1, b1
        GEMM(c, dt, c->dZ1, c->ld4H1, c->W1xT, c->ld4H1, c->dXemb, c->ldX1, M, X1, 4 * H1, nullptr, true);
        if (!two) {
            // LRCN-1f: d[embedding | x_cnn] -- mask all E + h columns in place, sum the right h columns over the steps -> d x_cnn,
            // then the image-embedding gradient exactly as in the two-layer model (on sw, after the dW1 GEMM that shares its scratch)
            k_dx2_mask_reduce(st, GEMM_T_F32, c->dXemb, c->ldX1, S, B, E, h, d1, c->dxcnn, c->ldh);
            FORK();
            k_transpose(sw, dt, 1, c->dxcnn, c->ldh, B, h, c->dxcT, ldB, 0);
            k_cast_rows(sw, dt, feats, B, LRCN_CNNOUT, B, c->FT, ldB);
            GEMM(c, dt, c->dxcT, ldB, c->FT, ldB, grads[5], LRCN_CNNOUT, h, LRCN_CNNOUT, B, nullptr, true, false, false, false, par);
            HIPCHK(c, hipEventRecord(c->grad_ev[2], sw));  // group 2: Wcnn
        }
        SegScope seg_eg(c, LRCN_SEG_EMBED_GRAD, st, 4.0 * M * E + 4.0 * (double)V * E);  // (T+1) B rows of E f32 in, dense V x E f32 out
        if (c->emb_rows_out) {
            // data-parallel host with the sparse exchange on: hand out this rank's rows and ids; grads[6] is NOT written by this call
            if (M > c->emb_rows_cap) FAIL(c, LRCN_EINVAL, "embedding-row buffer holds %d rows, this call has %d", c->emb_rows_cap, M);
            k_embed_rows_export(st, c->dXemb, c->ldX1, S, B, E, two ? d1 : none, c->emb_rows_out);
            HIPCHK(c, hipMemcpyAsync(c->emb_tok_out, c->tok_in, sizeof(int32_t) * (size_t)M, hipMemcpyDeviceToDevice, st));
        } else {
            // dWembed: per-token sums in an E-contiguous staging array, then one transpose into the column-major gradient (kernels.hip).
            // LRCN_EMBED_SCATTER=0: the direct scatter (one float atomic per element, 64 cache lines per wave instruction: 99 vs ~25 us).
            static const char *ks = getenv("LRCN_EMBED_SCATTER");
            bool done = false;
            if (!(ks && ks[0] == '0')) {
                if (!c->dWe_rm) DALLOC(c, c->dWe_rm, sizeof(float) * (size_t)V * c->ldE);
                unsigned long long *keys = nullptr;
                if (c->opt_det) {
                    if (!c->sort_keys) DALLOC(c, c->sort_keys, sizeof(unsigned long long) * (size_t)c->maxS * c->maxB);
                    keys = c->sort_keys;
                }
                done = k_embed_scatter_rm(st, c->dXemb, c->ldX1, c->tok_in, S, B, E, V, two ? d1 : none, c->dWe_rm, c->ldE, grads[6], keys);
                if (!done && c->opt_det) FAIL(c, LRCN_EINVAL, "LRCN_OPT_DETERMINISTIC supports (T+1)*B <= 8192 rows per call (got %d)", M);
            }
            if (!done) {
                HIPCHK(c, hipMemsetAsync(grads[6], 0, sizeof(float) * (size_t)V * E, st));
                k_embed_scatter(st, c->dXemb, c->ldX1, c->tok_in, S, B, E, V, two ? d1 : none, grads[6]);
            }
        }
        HIPCHK(c, hipEventRecord(c->grad_ev[4], st));  // group 4: Wembed
        return LRCN_OK;
    };
    r = backward();
    if (par && nfork > 0) {  // join: whatever follows on the main stream (update!, the next call's scratch reuse) comes after the weight gradients
        const hipError_t e1 = hipEventRecord(c->wg_done, sw);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(st, c->wg_done, 0) : e1;
        if (e2 != hipSuccess) {
            (void)hipStreamSynchronize(sw);  // the event path failed: fall back to a host-side join rather than return unjoined
            if (!r) FAIL(c, LRCN_EHIP, "joining the weight-gradient stream: %s", hipGetErrorString(e2));
        }
    }
    if (r) return r;
#undef FORK
    KCHK(c, "backward");
    return LRCN_OK;
}

// logp[0] = running sum of log p(target); logp[1] = sticky "token id outside [0, V)" flag raised by build_tokens_kernel
int fetch_loss(lrcn_ctx *c, double *out) {
    double s[2] = {0.0, 0.0};
    HIPCHK(c, hipMemcpyAsync(s, c->logp, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (s[1] != 0.0) {
        HIPCHK(c, hipMemsetAsync(c->logp + 1, 0, sizeof(double), c->stream));
        FAIL(c, LRCN_EINVAL, "a token id was outside [0, V=%d) (ids are 0-based at the ABI: eos=0, bos=1, unk=2; the reference raises BoundsError, lrcn.jl:556/569)", c->V);
    }
    if (out) *out = -s[0] / ((double)c->last_norm * (double)c->last_S);
    return LRCN_OK;
}

// lrcn() on internal single-step buffers: state st_f32 (f32 row-major), inputs st_x (T [B][ldX1]: the embedding in columns
// [0, E); LRCN-1f appends x_cnn here) and xcnn (f32 [B][ldh]).
// d2: dropout of the concatenated input (LSTM-2's in the two-layer model, LSTM-1's in LRCN-1f). Leaves logits in st_logits [B][ldV].
int step_internal(lrcn_ctx *c, const float *const p[9], int B, const DropSpec &d2, bool h_ready = false) {
    const int dt = c->dt, E = c->E, H1 = c->H1, H2 = c->H2, h = c->h, V = c->V, X1 = c->X1;
    hipStream_t st = c->stream;
    const bool two = c->nl == 2;
    if (!two) k_concat_x2(st, dt, c->st_x, c->ldX1, c->xcnn, c->ldh, 1, B, E, h, d2);  // x = dropout(hcat(x_lstm, x_cnn))
    // LSTM 1: gates = x*W1x' + h1*W1h' + b1
    if (!h_ready) k_cast_rows(st, dt, c->st_f32[0], H1, B, H1, c->st_h1, c->ldH1);  // h_ready: st_h1 / st_h2 already hold T(h)
    GEMM(c, dt, c->st_x, c->ldX1, c->W1x, c->ldX1, c->st_g, 4 * H1, B, 4 * H1, X1, p[1], true);
    GEMM(c, dt, c->st_h1, c->ldH1, c->W1h, c->ldH1, c->st_g, 4 * H1, B, 4 * H1, H1, nullptr, true, true);
    k_lstm_fwd(st, dt, c->st_g, 4 * H1, c->st_f32[1], B, H1, c->st_a, c->ld4H1, c->st_f32[1], c->st_h1, c->ldH1, c->st_f32[0]);
    if (!two) {
        GEMM(c, dt, c->st_h1, c->ldH1, c->Wod, c->ldH2, c->st_logits, c->ldV, B, V, H2, p[8], true);
        KCHK(c, "step (1 layer)");
        return LRCN_OK;
    }
    // projection + concat + dropout
    GEMM(c, dt, c->st_h1, c->ldH1, c->Wpd, c->ldH1, c->st_x2, c->ldH2, B, h, H1, nullptr, false);
    k_concat_x2(st, dt, c->st_x2, c->ldH2, c->xcnn, c->ldh, 1, B, h, h, d2);
    // LSTM 2
    if (!h_ready) k_cast_rows(st, dt, c->st_f32[2], H2, B, H2, c->st_h2, c->ldH2);
    GEMM(c, dt, c->st_x2, c->ldH2, c->W2x, c->ldH2, c->st_g, 4 * H2, B, 4 * H2, H2, p[3], true);
    GEMM(c, dt, c->st_h2, c->ldH2, c->W2h, c->ldH2, c->st_g, 4 * H2, B, 4 * H2, H2, nullptr, true, true);
    k_lstm_fwd(st, dt, c->st_g, 4 * H2, c->st_f32[3], B, H2, c->st_a, c->ld4H2, c->st_f32[3], c->st_h2, c->ldH2, c->st_f32[2]);
    GEMM(c, dt, c->st_h2, c->ldH2, c->Wod, c->ldH2, c->st_logits, c->ldV, B, V, H2, p[8], true);
    KCHK(c, "step");
    return LRCN_OK;
}

// The same step for the batched beam decode, on the concatenated buffers: st_xh1 = [x | h1], st_xh2 = [x2 | h2] (T, the
// h blocks already hold this step's input states), one GEMM per LSTM against W1cat / W2cat.  LRCN-1f: st_xh1 = [emb | x_cnn | h1]
// (the caller wrote the x_cnn columns once: they do not change during a decode).
// The batched decode step with the cell math in the gate GEMM's epilogue (gemm_8p.hip GEMM_OUT_LSTM_FWD; round 5): from 256 hypotheses
// the gate GEMM is a chip-filling contraction (5120 x 4000 x 2048 at 1024 images x 5 beams), and the f32 pre-activations it used to write
// for a separate cell kernel -- 82 MB out and back per layer and step, plus the kernel -- never leave the workgroup.  The concatenated
// weights are then made with (unit, gate)-interleaved rows (prepare_weights cat_perm), the bias rides in as a broadcast row, the
// activated gates are not kept (no backward pass).  LRCN_DECODE_EPI=0: GEMM + cell kernel as before.
bool decode_epi_on(const lrcn_ctx *c, int B) {
    const char *k = getenv("LRCN_DECODE_EPI");  // read per call (the tests switch it inside one process)
    return !(k && k[0] == '0') && c->dt == GEMM_T_BF16 && B >= 256 && !(c->H1 & 3) && !(c->H2 & 3);
}
int decode_gates_epi(lrcn_ctx *c, const void *xh, int64_t ldxh, const void *Wcat, int K, const float *bias, int B, int H, const float *c_prev,
                     const int32_t *c_prev_idx, float *c_out, void *h_out, int64_t ld_h_out, const int32_t *gx_idx = nullptr) {
    // bias: ONE row [4H] for every row, or -- gx_idx given -- a table of input-side pre-activations of which row r adds row gx_idx[r]
    // h_out must NOT be the h columns of `xh`: every tile of this launch reads them as A-operand columns, and tiles of one row block run in
    // different rounds (2.5 rounds of 256 x 128 tiles at 5120 x 4000), so an in-place h(t) would reach tiles that still need h(t-1).
    GemmArgs g{};
    g.dtype = c->dt;
    g.A = xh; g.lda = ldxh;
    g.B = Wcat; g.ldb = ldxh;
    g.M = B; g.N = 4 * H;
    g.K = (int)round_up64(K, 64);  // whole 128-byte K-steps: both operands carry zeros in the padding (as gemm() does)
    if (g.K > ldxh) FAIL(c, LRCN_EINVAL, "decode step: K = %d exceeds the operand rows (%lld)", g.K, (long long)ldxh);
    g.a_mode = GEMM_A_PLAIN;
    g.out_mode = GEMM_OUT_LSTM_FWD;
    g.zero_page = c->zero_page;
    g.lstm.H = H; g.lstm.ld_a = 4 * H; g.lstm.ld_h = ld_h_out;
    g.lstm.Gx = bias; g.lstm.gx_bcast = gx_idx ? 0 : 1; g.lstm.gx_idx = gx_idx;
    // the cell state of row r continues its PARENT hypothesis' (lrcn.jl:673-676): read through c_prev_idx (round 6; NULL = the first step,
    // zero state) into the other buffer of the pair -- no gather launch between the steps
    g.lstm.c_prev = c_prev; g.lstm.c_prev_idx = c_prev_idx; g.lstm.c_out = c_out;
    g.lstm.acts = nullptr;
    g.lstm.h_new = h_out;
    g.lstm.h_f32 = nullptr;
    hipError_t e = launch_gemm_8p(c->stream, g);
    if (e != hipSuccess) FAIL(c, LRCN_EHIP, "decode step (gate GEMM + cell epilogue): %s", hipGetErrorString(e));
    return LRCN_OK;
}

// The logits GEMM of a batched decode step with softmax + top-K in its epilogue (gemm_8p.hip GEMM_OUT_SMAX_TOPK; round 6): x * w[end-1] .+ w[end]
// (lrcn.jl:550) is reduced tile by tile to per-row records and merged by k_softmax_topk_merge -- the B x V f32 logits (218 MB per step at
// 5120 x 10640) are never written, and softmax_topk_rows_kernel's pass over them disappears.  LRCN_DECODE_SMAX=0: GEMM + that kernel.
bool decode_smax_on(const lrcn_ctx *c, int B, int K) {
    const char *k = getenv("LRCN_DECODE_SMAX");  // read per call (the tests switch it inside one process)
    return !(k && k[0] == '0') && c->dt == GEMM_T_BF16 && B >= 256 && K < SMAX_KC && c->V >= 256 && !(c->V & 3) && c->H2 > 64;  // (>= 2 K-tiles)
}
int decode_logits_smax(lrcn_ctx *c, const void *hT, int64_t ldh, const float *bias, int B, int K) {
    const int V = c->V, H2 = c->H2, nrec = 2 * ((V + 255) / 256);
    if (!c->smax_part) DALLOC(c, c->smax_part, sizeof(float) * (size_t)c->maxB * nrec * SMAX_REC);
    GemmArgs g{};
    g.dtype = c->dt;
    g.A = hT; g.lda = ldh;
    g.B = c->Wod; g.ldb = c->ldH2;
    g.M = B; g.N = V;
    g.K = (int)round_up64(H2, 64);
    if (g.K > ldh || g.K > c->ldH2) FAIL(c, LRCN_EINVAL, "decode logits: K = %d exceeds the operand rows", g.K);
    g.bias = bias;
    g.a_mode = GEMM_A_PLAIN;
    g.out_mode = GEMM_OUT_SMAX_TOPK;
    g.zero_page = c->zero_page;
    g.smax.part = c->smax_part; g.smax.nrec = nrec;
    hipError_t e = launch_gemm_8p(c->stream, g);
    if (e != hipSuccess) FAIL(c, LRCN_EHIP, "decode step (logits GEMM + softmax / top-K epilogue): %s", hipGetErrorString(e));
    if (!k_softmax_topk_merge(c->stream, c->smax_part, nrec, B, K, c->st_topi, c->st_topv)) FAIL(c, LRCN_EINVAL, "softmax / top-K merge: K = %d, %d records", K, nrec);
    return LRCN_OK;
}

// The batched decode step with input-projection TABLES (round 6).  [x | h] W of lrcn.jl:529 is x Wx + h Wh, and in a decode x is not free:
// LSTM-1's x is the embedding of one of V tokens, LSTM-2's is [h1 Wproj | x_cnn] with x_cnn fixed per image (lrcn.jl:546, :611).  So
// T1 = Wembed W1x + b1 (V x 4H1: 85 GFLOP once per call -- what ONE step spent on it for its 5120 rows) and U2 = x_cnn W2x[right half] + b2
// (one row per image) are made once, each step's gate GEMMs contract h (K = 1024) resp. [h1 Wproj | h2] (K = 1536) instead of 2048, and the
// cell epilogue adds row last_token / row image of the tables (LstmEpi::gx_idx).  63 of a step's 282 GFLOP at 5120 hypotheses are not done,
// the embedding gather and the concat launch disappear.  Same products, f32 accumulation in two chains instead of one.  Memory for FLOPs:
// T1 is 170 MB of the 288 GB.  LRCN_DECODE_TABLES=0: the [x | h] form.
bool decode_tables_on(const lrcn_ctx *c, int B) {
    const char *k = getenv("LRCN_DECODE_TABLES");  // read per call (the tests switch it inside one process)
    return !(k && k[0] == '0') && c->nl == 2 && decode_epi_on(c, B) && c->H1 > 64 && c->H2 > 64;
}
int decode_tables_alloc(lrcn_ctx *c) {
    const size_t es = c->esz;
    if (!c->dec_T1) DALLOC(c, c->dec_T1, sizeof(float) * (size_t)c->V * 4 * c->H1);
    if (!c->dec_U2) DALLOC(c, c->dec_U2, sizeof(float) * (size_t)c->maxB * 4 * c->H2);
    if (!c->dec_A1) DALLOC(c, c->dec_A1, es * (size_t)c->maxB * c->ldH1);
    if (!c->dec_A2) DALLOC(c, c->dec_A2, es * (size_t)c->maxB * (c->ldh + c->ldH2));
    if (!c->dec_W2c) DALLOC(c, c->dec_W2c, es * (size_t)4 * c->H2 * (c->ldh + c->ldH2));
    if (!c->dec_Aimg) DALLOC(c, c->dec_Aimg, es * (size_t)c->maxB * c->ldH2);
    if (!c->dec_img) DALLOC(c, c->dec_img, sizeof(int32_t) * (size_t)c->maxB);
    return LRCN_OK;
}
// once per decode call, after prepare_weights(dec_tables) and the image embedding (dxcnn [N][ldh] f32): the two tables and the row -> image map
int decode_tables_build(lrcn_ctx *c, const float *const p[9], int N, int K) {
    const int dt = c->dt, E = c->E, H1 = c->H1, H2 = c->H2, h = c->h, V = c->V;
    hipStream_t st = c->stream;
    GEMM(c, dt, c->WeT, c->ldE, c->W1x, c->ldX1, c->dec_T1, 4 * H1, V, 4 * H1, E, p[1], true);                 // per token
    HIPCHK(c, hipMemsetAsync(c->dec_Aimg, 0, c->esz * (size_t)N * c->ldH2, st));
    DropSpec none{};
    k_concat_x2(st, dt, c->dec_Aimg, c->ldH2, c->dxcnn, c->ldh, 1, N, h, h, none);                              // [0 | x_cnn] per image
    GEMM(c, dt, c->dec_Aimg, c->ldH2, c->W2x, c->ldH2, c->dec_U2, 4 * H2, N, 4 * H2, H2, p[3], true);           // per image
    k_row_div(st, c->dec_img, N * K, K);
    HIPCHK(c, hipMemsetAsync(c->dec_A1, 0, c->esz * (size_t)N * K * c->ldH1, st));                              // zero initial h1 / h2 and K padding
    HIPCHK(c, hipMemsetAsync(c->dec_A2, 0, c->esz * (size_t)N * K * (c->ldh + c->ldH2), st));
    KCHK(c, "decode tables");
    return LRCN_OK;
}
int step_decode_tables(lrcn_ctx *c, const float *const p[9], int B, const int32_t *parent, bool first, int smax_K) {
    const int dt = c->dt, H1 = c->H1, H2 = c->H2, h = c->h, V = c->V;
    const int64_t ldA2 = c->ldh + c->ldH2;
    if (!first) k_decode_prep_h(c->stream, parent, B, c->st_h1, c->ldH1, H1, c->st_h2, c->ldH2, H2, c->dec_A1, c->ldH1, c->dec_A2, ldA2, c->ldh);
    int r = decode_gates_epi(c, c->dec_A1, c->ldH1, c->W1h_gi, H1, c->dec_T1, B, H1, first ? nullptr : c->st_f32[1], parent, c->st2_f32[1], c->st_h1,
                             c->ldH1, c->bs_last);
    if (r) return r;
    GEMM(c, dt, c->st_h1, c->ldH1, c->Wpd, c->ldH1, c->dec_A2, ldA2, B, h, H1, nullptr, false);   // x = s[1] * w[end-4] (lrcn.jl:544) into A2's left block
    r = decode_gates_epi(c, c->dec_A2, ldA2, c->dec_W2c, (int)ldA2, c->dec_U2, B, H2, first ? nullptr : c->st_f32[3], parent, c->st2_f32[3], c->st_h2,
                         c->ldH2, c->dec_img);
    if (r) return r;
    if (smax_K > 0) {
        if ((r = decode_logits_smax(c, c->st_h2, c->ldH2, p[8], B, smax_K))) return r;
    } else {
        GEMM(c, dt, c->st_h2, c->ldH2, c->Wod, c->ldH2, c->st_logits, c->ldV, B, V, H2, p[8], true);
    }
    KCHK(c, "step_decode (tables)");
    return LRCN_OK;
}

int step_decode(lrcn_ctx *c, const float *const p[9], int B, const DropSpec &d2, bool epi = false, const int32_t *parent = nullptr, bool first = false,
                int smax_K = 0) {   // smax_K > 0 (epi only): the logits end as st_topi / st_topv (decode_logits_smax) instead of st_logits
    const int dt = c->dt, H1 = c->H1, H2 = c->H2, h = c->h, V = c->V;
    hipStream_t st = c->stream;
    void *h1T = boff(c->st_xh1, c->ldX1, c->esz), *h2T = boff(c->st_xh2, c->ldH2, c->esz);
    if (epi) {
        // the cell epilogue writes h(t) to st_h1 / st_h2 (never into the [x | h] operand it is still reading); k_gather_state rebuilds
        // the h blocks of st_xh1 / st_xh2 from the f32 states for the next step
        // (epi: cell states ping-pong st_f32[1] -> st2_f32[1] / st_f32[3] -> st2_f32[3], read through `parent`; the caller swaps the pairs)
        int r = decode_gates_epi(c, c->st_xh1, c->ldXH1, c->W1cat, (int)c->ldX1 + H1, p[1], B, H1, first ? nullptr : c->st_f32[1], parent, c->st2_f32[1],
                                 c->st_h1, c->ldH1);
        if (r) return r;
        if (c->nl == 1) {
            if (smax_K > 0) {
                if ((r = decode_logits_smax(c, c->st_h1, c->ldH1, p[8], B, smax_K))) return r;
            } else {
                GEMM(c, dt, c->st_h1, c->ldH1, c->Wod, c->ldH2, c->st_logits, c->ldV, B, V, H2, p[8], true);
            }
            KCHK(c, "step_decode (1 layer, cell epilogue)");
            return LRCN_OK;
        }
        GEMM(c, dt, c->st_h1, c->ldH1, c->Wpd, c->ldH1, c->st_xh2, c->ldXH2, B, h, H1, nullptr, false);
        k_concat_x2(st, dt, c->st_xh2, c->ldXH2, c->xcnn, c->ldh, 1, B, h, h, d2);
        r = decode_gates_epi(c, c->st_xh2, c->ldXH2, c->W2cat, (int)c->ldH2 + H2, p[3], B, H2, first ? nullptr : c->st_f32[3], parent, c->st2_f32[3],
                             c->st_h2, c->ldH2);
        if (r) return r;
        if (smax_K > 0) {
            if ((r = decode_logits_smax(c, c->st_h2, c->ldH2, p[8], B, smax_K))) return r;
        } else {
            GEMM(c, dt, c->st_h2, c->ldH2, c->Wod, c->ldH2, c->st_logits, c->ldV, B, V, H2, p[8], true);
        }
        KCHK(c, "step_decode (cell epilogue)");
        return LRCN_OK;
    }
    GEMM(c, dt, c->st_xh1, c->ldXH1, c->W1cat, c->ldXH1, c->st_g, 4 * H1, B, 4 * H1, (int)c->ldX1 + H1, p[1], true);
    k_lstm_fwd(st, dt, c->st_g, 4 * H1, c->st_f32[1], B, H1, c->st_a, c->ld4H1, c->st_f32[1], h1T, c->ldXH1, c->st_f32[0]);
    if (c->nl == 1) {
        GEMM(c, dt, h1T, c->ldXH1, c->Wod, c->ldH2, c->st_logits, c->ldV, B, V, H2, p[8], true);
        KCHK(c, "step_decode (1 layer)");
        return LRCN_OK;
    }
    GEMM(c, dt, h1T, c->ldXH1, c->Wpd, c->ldH1, c->st_xh2, c->ldXH2, B, h, H1, nullptr, false);
    k_concat_x2(st, dt, c->st_xh2, c->ldXH2, c->xcnn, c->ldh, 1, B, h, h, d2);
    GEMM(c, dt, c->st_xh2, c->ldXH2, c->W2cat, c->ldXH2, c->st_g, 4 * H2, B, 4 * H2, (int)c->ldH2 + H2, p[3], true);
    k_lstm_fwd(st, dt, c->st_g, 4 * H2, c->st_f32[3], B, H2, c->st_a, c->ld4H2, c->st_f32[3], h2T, c->ldXH2, c->st_f32[2]);
    GEMM(c, dt, h2T, c->ldXH2, c->Wod, c->ldH2, c->st_logits, c->ldV, B, V, H2, p[8], true);
    KCHK(c, "step_decode");
    return LRCN_OK;
}

// element counts of the context's 9 tensors (0 for the slots its model does not have)
void ctx_sizes(const lrcn_ctx *c, int64_t sz[9]) { lrcn_param_sizes_n(c->nl, c->E, c->H1, c->H2, c->V, sz); }

}  // namespace

// =====================================================================================================
extern "C" {

const char *lrcn_version(void) { return "lrcn-hip 0.4 (gfx950)"; }
int lrcn_abi_version(void) { return LRCN_ABI_VERSION; }

int lrcn_set_option(lrcn_ctx *c, int option, int64_t value) {
    if (!c) return LRCN_EINVAL;
    switch (option) {
    case LRCN_OPT_FUSED_UPDATE:
        if (value != 0 && value != 1) FAIL(c, LRCN_EINVAL, "LRCN_OPT_FUSED_UPDATE takes 0 or 1");
        c->opt_fused = value != 0;
        c->shadow_valid = false;
        c->fused_groups = 0;
        if (c->opt_fused) {  // the second shadow set is allocated here, not inside the first update (no allocation in a step)
            DeviceGuard dg(c);
            return ensure_alt_shadows(c);
        }
        return LRCN_OK;
    case LRCN_OPT_DETERMINISTIC:
        if (value != 0 && value != 1) FAIL(c, LRCN_EINVAL, "LRCN_OPT_DETERMINISTIC takes 0 or 1");
        c->opt_det = value != 0;
        return LRCN_OK;
    case LRCN_OPT_CONV_CHUNK_BYTES:
        if (value < 0) FAIL(c, LRCN_EINVAL, "LRCN_OPT_CONV_CHUNK_BYTES must be >= 0");
        c->conv_chunk_bytes = value;
        return LRCN_OK;
    default:
        FAIL(c, LRCN_EINVAL, "unknown option %d", option);
    }
}

int lrcn_params_touched(lrcn_ctx *c) {
    if (!c) return LRCN_EINVAL;
    c->shadow_valid = false;
    c->fused_groups = 0;  // a per-group update that stopped partway must not complete a later step's mask
    c->refresh_groups = 0;
    return LRCN_OK;
}

const char *lrcn_last_error(const lrcn_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int lrcn_param_sizes_n(int n_layers, int E, int H1, int H2, int V, int64_t s[9]) {
    if (E < 1 || H1 < 1 || H2 < 2 || (H2 & 1) || V < 3 || !s) return LRCN_EINVAL;
    if (n_layers != 0 && n_layers != 1 && n_layers != 2) return LRCN_EINVAL;
    const int h = H2 / 2;
    if (n_layers == 1) {  // LRCN-1f: one LSTM over [embedding | x_cnn]; W2, b2, Wproj do not exist
        if (H1 != H2) return LRCN_EINVAL;
        s[0] = (int64_t)(E + h + H1) * 4 * H1;
        s[2] = s[3] = s[4] = 0;
    } else {
        s[0] = (int64_t)(E + H1) * 4 * H1;
        s[2] = (int64_t)(2 * H2) * 4 * H2;
        s[3] = 4 * H2;
        s[4] = (int64_t)H1 * h;
    }
    s[1] = 4 * H1;
    s[5] = (int64_t)LRCN_CNNOUT * h;
    s[6] = (int64_t)V * E;
    s[7] = (int64_t)H2 * V;
    s[8] = V;
    return LRCN_OK;
}
int lrcn_param_sizes(int E, int H1, int H2, int V, int64_t s[9]) { return lrcn_param_sizes_n(2, E, H1, H2, V, s); }

int lrcn_malloc(void **p, size_t bytes) { return hipMalloc(p, bytes ? bytes : 16) == hipSuccess ? LRCN_OK : LRCN_ENOMEM; }
int lrcn_free(void *p) { return hipFree(p) == hipSuccess ? LRCN_OK : LRCN_EHIP; }
int lrcn_memcpy_h2d(void *d, const void *s, size_t n) { return hipMemcpy(d, s, n, hipMemcpyHostToDevice) == hipSuccess ? LRCN_OK : LRCN_EHIP; }
int lrcn_memcpy_d2h(void *d, const void *s, size_t n) { return hipMemcpy(d, s, n, hipMemcpyDeviceToHost) == hipSuccess ? LRCN_OK : LRCN_EHIP; }

void lrcn_destroy(lrcn_ctx *c) {
    if (!c) return;
    DeviceGuard dg(c);
    (void)hipDeviceSynchronize();
    for (void *p : c->allocs) (void)hipFree(p);
    for (auto &e : c->grad_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->wg_fork)
        if (e) (void)hipEventDestroy(e);
    if (c->wg_done) (void)hipEventDestroy(c->wg_done);
    if (c->xc_fork) (void)hipEventDestroy(c->xc_fork);
    if (c->xc_done) (void)hipEventDestroy(c->xc_done);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->wg_stream && c->wg_stream_owned) (void)hipStreamDestroy(c->wg_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (int j = 0; j < lrcn_ctx::kStage; ++j) {
        if (c->up_done[j]) (void)hipEventDestroy(c->up_done[j]);
        if (c->rd_done[j]) (void)hipEventDestroy(c->rd_done[j]);
    }
    comm_destroy(c->comm);
    for (auto &e : c->bucket_done)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ar_done)
        if (e) (void)hipEventDestroy(e);
    if (c->comm_stream && c->comm_stream_owned) (void)hipStreamDestroy(c->comm_stream);
    for (auto &sp : c->seg)
        for (auto &e : sp.ev) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    for (auto &e : c->prof_ev) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    delete c;
}

int lrcn_create(const lrcn_config *cfg, lrcn_ctx **out) {
    if (!cfg || !out) {
        g_create_err = "null argument";
        return LRCN_EINVAL;
    }
    *out = nullptr;
    int64_t sz[9];
    if (lrcn_param_sizes_n(cfg->n_layers, cfg->E, cfg->H1, cfg->H2, cfg->V, sz) != LRCN_OK || cfg->max_B < 1 || cfg->max_T < 0 ||
        cfg->max_T > LRCN_MAX_T || (cfg->lstm_dtype != LRCN_F32 && cfg->lstm_dtype != LRCN_BF16) ||
        (cfg->vgg_dtype != LRCN_F32 && cfg->vgg_dtype != LRCN_BF16 && cfg->vgg_dtype != LRCN_FP8) || cfg->max_images < 0) {
        g_create_err = "invalid lrcn_config (need E,H1>=1, even H2>=2, V>=3, max_B>=1, 0<=max_T<=28, lstm_dtype in {F32,BF16}, vgg_dtype in {F32,BF16,FP8}, n_layers in {0,1,2} with H1 == H2 when n_layers == 1)";
        return LRCN_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) {
        g_create_err = "no such HIP device (is a GPU visible?)";
        return LRCN_EHIP;
    }
    lrcn_ctx *c = new lrcn_ctx();
    c->cfg = *cfg;
    DeviceGuard dg(c);  // allocate on cfg->device, then restore the caller's selection
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != cfg->device) {
            g_create_err = "hipSetDevice failed";
            delete c;
            return LRCN_EHIP;
        }
    }
    {   // environment defaults of the options (lrcn_set_option overrides): LRCN_DETERMINISTIC=1
        const char *kd = getenv("LRCN_DETERMINISTIC");
        c->opt_det = kd && kd[0] == '1';
    }
    c->dt = cfg->lstm_dtype == LRCN_BF16 ? GEMM_T_BF16 : GEMM_T_F32;
    c->vdt = cfg->vgg_dtype == LRCN_F32 ? GEMM_T_F32 : GEMM_T_BF16;  // LRCN_FP8: bf16 everywhere outside conv2_2..conv5_3
    c->vgg_fp8 = cfg->vgg_dtype == LRCN_FP8;
    c->esz = c->dt == GEMM_T_BF16 ? 2 : 4;
    c->vesz = c->vdt == GEMM_T_BF16 ? 2 : 4;
    c->E = cfg->E; c->H1 = cfg->H1; c->H2 = cfg->H2; c->h = cfg->H2 / 2; c->V = cfg->V;
    c->nl = cfg->n_layers == 1 ? 1 : 2;
    c->X1 = c->nl == 1 ? c->E + c->h : c->E;
    c->maxB = cfg->max_B; c->maxS = cfg->max_T + 1;
    const int E = c->E, H1 = c->H1, H2 = c->H2, h = c->h, V = c->V, B = c->maxB, S = c->maxS;
    const int64_t M = (int64_t)S * B;
    const int X1 = c->X1;
    c->ldX1 = ld8(X1);
    c->ldE = ld8(E); c->ldH1 = ld8(H1); c->ldH2 = ld8(H2); c->ldh = ld8(h); c->ld4H1 = ld8(4 * H1); c->ld4H2 = ld8(4 * H2);
    c->ldV = ld8(V); c->ldM = ld8(M); c->ldB = ld8(B);
    const size_t es = c->esz;
    int rc = [&]() -> int {
        c->ldXH1 = c->ldX1 + c->ldH1; c->ldXH2 = 2 * c->ldH2;
        DALLOC(c, c->W1cat, es * 4 * H1 * c->ldXH1); DALLOC(c, c->W2cat, es * 4 * H2 * c->ldXH2);
        DALLOC(c, c->st_xh1, es * B * c->ldXH1);     DALLOC(c, c->st_xh2, es * B * c->ldXH2);
        DALLOC(c, c->W1x, es * 4 * H1 * c->ldX1);  DALLOC(c, c->W1h, es * 4 * H1 * c->ldH1);
        DALLOC(c, c->W1xT, es * X1 * c->ld4H1);    DALLOC(c, c->W1hT, es * H1 * c->ld4H1);
        DALLOC(c, c->W2x, es * 4 * H2 * c->ldH2);  DALLOC(c, c->W2h, es * 4 * H2 * c->ldH2);
        DALLOC(c, c->W2xT, es * H2 * c->ld4H2);    DALLOC(c, c->W2hT, es * H2 * c->ld4H2);
        DALLOC(c, c->Wpd, es * h * c->ldH1);       DALLOC(c, c->WpT, es * H1 * c->ldh);
        DALLOC(c, c->Wcd, es * h * LRCN_CNNOUT);   DALLOC(c, c->WeT, es * V * c->ldE);
        DALLOC(c, c->Wod, es * V * c->ldH2);       DALLOC(c, c->WoT, es * H2 * c->ldV);
        DALLOC(c, c->tok, sizeof(int32_t) * M);    DALLOC(c, c->tok_in, sizeof(int32_t) * M);
        DALLOC(c, c->tok_tgt, sizeof(int32_t) * M);
        DALLOC(c, c->F, es * B * LRCN_CNNOUT);     DALLOC(c, c->FT, es * LRCN_CNNOUT * c->ldB);
        DALLOC(c, c->xcnn, sizeof(float) * B * c->ldh);
        DALLOC(c, c->Xemb, es * M * c->ldX1);
        DALLOC(c, c->G1, sizeof(float) * M * 4 * H1); DALLOC(c, c->A1, es * M * c->ld4H1);
        DALLOC(c, c->C1, sizeof(float) * M * H1);     DALLOC(c, c->H1all, es * M * c->ldH1);
        DALLOC(c, c->X2, es * M * c->ldH2);
        DALLOC(c, c->G2, sizeof(float) * M * 4 * H2); DALLOC(c, c->A2, es * M * c->ld4H2);
        DALLOC(c, c->C2, sizeof(float) * M * H2);     DALLOC(c, c->H2all, es * M * c->ldH2);
        DALLOC(c, c->Logits, sizeof(float) * M * c->ldV);
        DALLOC(c, c->dLog, es * M * c->ldV);
        DALLOC(c, c->dZ1, es * M * c->ld4H1);      DALLOC(c, c->dZ2, es * M * c->ld4H2);
        DALLOC(c, c->dX2, es * M * c->ldH2);
        DALLOC(c, c->dH1all, sizeof(float) * M * H1); DALLOC(c, c->dH2all, sizeof(float) * M * H2);
        DALLOC(c, c->dXemb, sizeof(float) * M * c->ldX1);
        const int Hm = H1 > H2 ? H1 : H2;
        DALLOC(c, c->dhrec, sizeof(float) * B * Hm); DALLOC(c, c->dc, sizeof(float) * B * Hm);
        DALLOC(c, c->dxcnn, sizeof(float) * B * c->ldh); DALLOC(c, c->dxcT, es * h * c->ldB);
        int64_t ra = 4 * Hm; if (V > ra) ra = V;
        int64_t rb = 2 * H2; if (X1 + H1 > rb) rb = X1 + H1;  // stacked [x | h_prev]^T of one LSTM
        DALLOC(c, c->TA, es * ra * c->ldM);        DALLOC(c, c->TB, es * rb * c->ldM);
        DALLOC(c, c->logp, sizeof(double) * 2);
        DALLOC(c, c->zero_page, 256);
        if (hipMemset(c->zero_page, 0, 256) != hipSuccess) return LRCN_EHIP;
        for (auto &e : c->grad_ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return LRCN_EHIP;
        c->gemm_ws_bytes = 48u << 20;
        DALLOC(c, c->gemm_ws, c->gemm_ws_bytes);
        DALLOC(c, c->wg_ws, c->gemm_ws_bytes);
        if (hipStreamCreateWithFlags(&c->wg_stream, hipStreamNonBlocking) != hipSuccess) return LRCN_EHIP;
        for (auto &e : c->wg_fork)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return LRCN_EHIP;
        if (hipEventCreateWithFlags(&c->wg_done, hipEventDisableTiming) != hipSuccess) return LRCN_EHIP;
        if (hipEventCreateWithFlags(&c->xc_fork, hipEventDisableTiming) != hipSuccess) return LRCN_EHIP;
        if (hipEventCreateWithFlags(&c->xc_done, hipEventDisableTiming) != hipSuccess) return LRCN_EHIP;
        if (cfg->max_images > 0) DALLOC(c, c->vgg_ws, c->gemm_ws_bytes);
        for (int i = 0; i < 4; ++i) {
            DALLOC(c, c->st_f32[i], sizeof(float) * B * Hm);
            DALLOC(c, c->st2_f32[i], sizeof(float) * B * Hm);
        }
        DALLOC(c, c->st_h1, es * B * c->ldH1);     DALLOC(c, c->st_h2, es * B * c->ldH2);
        DALLOC(c, c->st_x, es * B * c->ldX1);      DALLOC(c, c->st_x2, es * B * c->ldH2);
        DALLOC(c, c->st_a, es * B * (c->ld4H1 > c->ld4H2 ? c->ld4H1 : c->ld4H2));
        DALLOC(c, c->st_g, sizeof(float) * B * 4 * Hm);
        DALLOC(c, c->st_logits, sizeof(float) * B * c->ldV); DALLOC(c, c->st_prob, sizeof(float) * B * c->ldV);
        int64_t io = (int64_t)B * (V > 4 * Hm ? V : 4 * Hm); if (io < (int64_t)B * (X1 + Hm)) io = (int64_t)B * (X1 + Hm);
        DALLOC(c, c->st_io, sizeof(float) * io);
        DALLOC(c, c->st_topi, sizeof(int32_t) * B * 32); DALLOC(c, c->st_topv, sizeof(float) * B * 32);
        DALLOC(c, c->st_parent, sizeof(int32_t) * B);
        for (int i = 0; i < 2; ++i) DALLOC(c, c->bs_seq[i], sizeof(int32_t) * B * LRCN_BEAM_MAXLEN);
        DALLOC(c, c->bs_last, sizeof(int32_t) * B);      DALLOC(c, c->bs_done, sizeof(int32_t) * B);
        DALLOC(c, c->bs_ndone, sizeof(int32_t) * 4);     DALLOC(c, c->bs_res_tok, sizeof(int32_t) * B * LRCN_BEAM_MAXLEN);
        DALLOC(c, c->bs_res_len, sizeof(int32_t) * B);   DALLOC(c, c->bs_p, sizeof(float) * B);
        DALLOC(c, c->bs_res_p, sizeof(float) * B);
        if (cfg->max_images > 0) {
            const int64_t N = cfg->max_images;
            const size_t ve = c->vesz;
            if (c->vdt != GEMM_T_BF16) DALLOC(c, c->im2col, ve * N * 224 * 224 * 32);  // bf16 fuses conv1_1's im2col
            DALLOC(c, c->actA, ve * N * 224 * 224 * 64);
            if (c->vdt == GEMM_T_BF16) DALLOC(c, c->img16, 2 * (N * 228 * 228 * 3 + 8));  // mean-subtracted crops in a 2-pixel zero frame (fused conv1_1+conv1_2)
            DALLOC(c, c->actB, ve * N * 112 * 112 * 128);  // largest tensor ever written to the second buffer (pool1 out = N*112*112*64; conv2_1 out = N*112*112*128)
            DALLOC(c, c->f6, ve * N * 4096);
            DALLOC(c, c->featsRM, sizeof(float) * N * 4096);
            if (c->vgg_fp8) DALLOC(c, c->amax_dev, sizeof(float) * 16);
        }
        return LRCN_OK;
    }();
    if (rc) {
        g_create_err = c->err;
        lrcn_destroy(c);
        return rc;
    }
    *out = c;
    return LRCN_OK;
}

int lrcn_vgg_set_wg_cap(lrcn_ctx *c, int cap) {
    DeviceGuard dg(c);
    if (!c) return LRCN_EINVAL;
    if (cap < 0 || (cap > 0 && cap < 8)) FAIL(c, LRCN_EINVAL, "wg_cap=%d must be 0 (off) or >= 8", cap);
    c->vgg_wg_cap = cap;
    return LRCN_OK;
}

int lrcn_set_stream(lrcn_ctx *c, void *s) {
    DeviceGuard dg(c);
    if (!c) return LRCN_EINVAL;
    c->stream = reinterpret_cast<hipStream_t>(s);
    return LRCN_OK;
}
int lrcn_set_wg_stream(lrcn_ctx *c, void *s) {
    DeviceGuard dg(c);
    if (!c || !s) return LRCN_EINVAL;
    if (c->wg_stream) {
        HIPCHK(c, hipStreamSynchronize(c->wg_stream));
        if (c->wg_stream_owned) (void)hipStreamDestroy(c->wg_stream);
    }
    c->wg_stream = reinterpret_cast<hipStream_t>(s);
    c->wg_stream_owned = false;
    return LRCN_OK;
}
int lrcn_sync(lrcn_ctx *c) {
    if (!c) return LRCN_EINVAL;
    DeviceGuard dg(c);
    return fetch_loss(c, nullptr);  // synchronises, and reports a pending out-of-range-token error
}

int lrcn_init_weights(lrcn_ctx *c, float *const p[9], uint64_t seed) {
    DeviceGuard dg(c);
    if (!c || !p) return LRCN_EINVAL;
    int64_t sz[9];
    ctx_sizes(c, sz);
    const int E = c->E, H1 = c->H1, H2 = c->H2, h = c->h, V = c->V;
    const int rows[9] = {c->X1 + H1, 1, 2 * H2, 1, H1, LRCN_CNNOUT, V, H2, 1};
    const int cols[9] = {4 * H1, 4 * H1, 4 * H2, 4 * H2, h, h, E, V, V};
    for (int k = 0; k < 9; ++k) {
        if (sz[k] == 0) continue;  // LRCN-1f has no W2 / b2 / Wproj
        if (!p[k]) FAIL(c, LRCN_EINVAL, "null parameter tensor %d", k);
        if (k == 1 || k == 3 || k == 8) {
            k_fill(c->stream, p[k], sz[k], 0.0f);
            if (k != 8) k_fill(c->stream, p[k], k == 1 ? H1 : H2, 1.0f);  // forget-gate bias = 1 (lrcn.jl:501)
        } else {
            k_init_uniform(c->stream, p[k], sz[k], (float)std::sqrt(2.0 / ((double)rows[k] + (double)cols[k])), seed, k);
        }
    }
    KCHK(c, "init_weights");
    c->shadow_valid = false;
    c->fused_groups = 0;
    return LRCN_OK;
}

int lrcn_loss(lrcn_ctx *c, const float *const p[9], const float *feats, const int32_t *tokens, int T, int B, int norm_B,
              const lrcn_dropout *drop, double *loss_host) {
    DeviceGuard dg(c);
    if (!c || !p || !feats || (!tokens && T > 0)) return LRCN_EINVAL;
    int r = loss_impl(c, p, feats, tokens, T, B, norm_B, drop, nullptr, nullptr);
    if (r) return r;
    return loss_host ? fetch_loss(c, loss_host) : LRCN_OK;
}

int lrcn_loss_grad(lrcn_ctx *c, const float *const p[9], const float *feats, const int32_t *tokens, int T, int B, int norm_B,
                   const lrcn_dropout *drop, float *const grads[9], double *loss_host) {
    DeviceGuard dg(c);
    if (!c || !p || !feats || (!tokens && T > 0) || !grads) return LRCN_EINVAL;
    int r = loss_impl(c, p, feats, tokens, T, B, norm_B, drop, grads, nullptr);
    if (r) return r;
    return loss_host ? fetch_loss(c, loss_host) : LRCN_OK;
}

int lrcn_grad_group_wait(lrcn_ctx *c, int group, void *stream) {
    DeviceGuard dg(c);
    if (!c || group < 0 || group >= LRCN_GRAD_GROUPS) return LRCN_EINVAL;
    HIPCHK(c, hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), c->grad_ev[group], 0));
    return LRCN_OK;
}

int lrcn_last_loss(lrcn_ctx *c, double *loss_host) {
    DeviceGuard dg(c);
    if (!c || !loss_host) return LRCN_EINVAL;
    return fetch_loss(c, loss_host);
}

int lrcn_forward_logits(lrcn_ctx *c, const float *const p[9], const float *feats, const int32_t *tokens, int T, int B,
                        float *logits_out) {
    DeviceGuard dg(c);
    if (!c || !p || !feats || (!tokens && T > 0) || !logits_out) return LRCN_EINVAL;
    return loss_impl(c, p, feats, tokens, T, B, B, nullptr, nullptr, logits_out);
}

int lrcn_adam_update(lrcn_ctx *c, float *const p[9], const float *const g[9], float *const m[9], float *const v[9], int step,
                     float lr, float b1, float b2, float eps) {
    DeviceGuard dg(c);
    if (!c || !p || !g || !m || !v || step < 1) return LRCN_EINVAL;
    c->fused_groups = 0;  // the whole-model update supersedes any per-group sequence left unfinished (an error between two groups)
    int64_t nparam = 0;
    {
        int64_t szp[9];
        ctx_sizes(c, szp);
        for (int k = 0; k < 9; ++k) nparam += szp[k];
    }
    SegScope seg(c, LRCN_SEG_UPDATE, c->stream, 28.0 * (double)nparam);  // w, m, v read + written, g read: 28 B per parameter
    if (c->opt_fused) {  // LRCN_OPT_FUSED_UPDATE: the same update, and the next step's shadow weights in the same pass
        c->shadow_valid = false;
        int r = adam_fused(c, p, g, m, v, -1, step, lr, b1, b2, eps, c->stream);
        if (r) return r;
        fused_update_done(c, p);
        return LRCN_OK;
    }
    c->shadow_valid = false;
    AdamTensors t;
    int64_t sz[9];
    ctx_sizes(c, sz);
    for (int k = 0; k < 9; ++k) {
        t.w[k] = p[k];
        t.g[k] = g[k];
        t.m[k] = m[k];
        t.v[k] = v[k];
        t.n[k] = sz[k];
    }
    k_adam(c->stream, t, step, lr, b1, b2, eps);
    KCHK(c, "adam");
    return LRCN_OK;
}

int lrcn_adam_update_group(lrcn_ctx *c, float *const p[9], const float *const g[9], float *const m[9], float *const v[9], int group,
                           int step, float lr, float b1, float b2, float eps, void *stream) {
    DeviceGuard dg(c);
    if (!c || !p || !g || !m || !v || step < 1) return LRCN_EINVAL;
    if (group < 0 || group >= LRCN_GRAD_GROUPS) FAIL(c, LRCN_EINVAL, "group=%d outside [0,%d)", group, LRCN_GRAD_GROUPS);
    static const int kGroup[LRCN_GRAD_GROUPS][2] = {{7, 8}, {2, 3}, {4, 5}, {0, 1}, {6, 6}};  // order of the grad_ev records
    int64_t szg[9];
    ctx_sizes(c, szg);
    SegScope seg(c, LRCN_SEG_UPDATE, stream ? reinterpret_cast<hipStream_t>(stream) : c->stream,
                 28.0 * (double)(szg[kGroup[group][0]] + (kGroup[group][1] != kGroup[group][0] ? szg[kGroup[group][1]] : 0)));
    if (c->opt_fused) {
        // fused with the shadow pass; the written set becomes current once all five groups of this step have been issued (they are
        // issued in any order, each exactly once per step, with the same `step`)
        if (step != c->fused_step) {  // first group of a new step: bits left by a step that never completed do not count
            c->fused_groups = 0;
            c->fused_step = step;
        }
        c->shadow_valid = false;
        int r = adam_fused(c, p, g, m, v, group, step, lr, b1, b2, eps, stream ? reinterpret_cast<hipStream_t>(stream) : c->stream);
        if (r) {
            c->fused_groups = 0;
            return r;
        }
        c->fused_groups |= 1u << group;
        if (c->fused_groups == (1u << LRCN_GRAD_GROUPS) - 1) {
            c->fused_groups = 0;
            fused_update_done(c, p);
        }
        return LRCN_OK;
    }
    c->shadow_valid = false;
    AdamTensors t;
    int64_t sz[9];
    ctx_sizes(c, sz);
    for (int k = 0; k < 9; ++k) {
        const bool in = k == kGroup[group][0] || k == kGroup[group][1];
        t.w[k] = p[k];
        t.g[k] = g[k];
        t.m[k] = m[k];
        t.v[k] = v[k];
        t.n[k] = in ? sz[k] : 0;
    }
    k_adam(stream ? reinterpret_cast<hipStream_t>(stream) : c->stream, t, step, lr, b1, b2, eps);
    KCHK(c, "adam (group)");
    return LRCN_OK;
}

int lrcn_refresh_shadows_group(lrcn_ctx *c, const float *const p[9], int group, void *stream) {
    DeviceGuard dg(c);
    if (!c || !p) return LRCN_EINVAL;
    if (group < 0 || group >= LRCN_GRAD_GROUPS) FAIL(c, LRCN_EINVAL, "group=%d outside [0,%d)", group, LRCN_GRAD_GROUPS);
    if (!c->opt_fused) FAIL(c, LRCN_ESTATE, "lrcn_refresh_shadows_group needs LRCN_OPT_FUSED_UPDATE = 1 (the second shadow set)");
    static const int kGroup[LRCN_GRAD_GROUPS][2] = {{7, 8}, {2, 3}, {4, 5}, {0, 1}, {6, 6}};
    int r = ensure_alt_shadows(c);
    if (r) return r;
    if (c->refresh_groups == 0) c->shadow_valid = false;  // first group of a step: the current set describes the OLD parameters from now on
    PrepPlan plan{};
    if (c->gi_live && (r = ensure_gi_sets(c))) return r;
    plan_matrices(c, p, alt_shadows(c), true, plan, kGroup[group][0], kGroup[group][1], c->gi_live ? c->alt_gi : nullptr);
    if (plan.n > 0) {  // LRCN-1f has no W2 / Wproj: an empty group is only counted
        k_prepare_weights(stream ? reinterpret_cast<hipStream_t>(stream) : c->stream, c->dt, plan);
        KCHK(c, "refresh_shadows_group");
    }
    c->refresh_groups |= 1u << group;
    if (c->refresh_groups == (1u << LRCN_GRAD_GROUPS) - 1) {
        c->refresh_groups = 0;
        float *pp[9];
        for (int k = 0; k < 9; ++k) pp[k] = const_cast<float *>(p[k]);
        fused_update_done(c, pp);
    }
    return LRCN_OK;
}

int lrcn_adam_update_flat(lrcn_ctx *c, float *w, const float *g, float *m, float *v, int64_t n, int step, float lr, float b1, float b2,
                          float eps, void *stream) {
    DeviceGuard dg(c);
    if (!c || step < 1 || n < 0) return LRCN_EINVAL;
    if (n == 0) return LRCN_OK;
    if (!w || !g || !m || !v) return LRCN_EINVAL;
    c->shadow_valid = false;  // some parameter changed: the next call makes its shadow weights afresh
    SegScope seg(c, LRCN_SEG_UPDATE, stream ? reinterpret_cast<hipStream_t>(stream) : c->stream, 28.0 * (double)n);
    AdamTensors t{};
    t.w[0] = w; t.g[0] = g; t.m[0] = m; t.v[0] = v; t.n[0] = n;
    k_adam(stream ? reinterpret_cast<hipStream_t>(stream) : c->stream, t, step, lr, b1, b2, eps);
    KCHK(c, "adam (flat)");
    return LRCN_OK;
}

int lrcn_train_step(lrcn_ctx *c, float *const p[9], float *const g[9], float *const m[9], float *const v[9], const float *feats,
                    const int32_t *tokens, int T, int B, int norm_B, const lrcn_dropout *drop, int step, float lr, float b1,
                    float b2, float eps, double *loss_host) {
    DeviceGuard dg(c);
    if (!c || !p || !g || !m || !v) return LRCN_EINVAL;
    int r = lrcn_loss_grad(c, p, feats, tokens, T, B, norm_B, drop, g, nullptr);
    if (r) return r;
    r = lrcn_adam_update(c, p, g, m, v, step, lr, b1, b2, eps);
    if (r) return r;
    return loss_host ? fetch_loss(c, loss_host) : LRCN_OK;
}

// ------------------------------------------------------------------------------------------- data parallelism
namespace {
const int kGradGroup[LRCN_GRAD_GROUPS][2] = {{7, 8}, {2, 3}, {4, 5}, {0, 1}, {6, 6}};  // order of the grad_ev records in loss_impl

// LRCN_DP_FORCE_PIPELINE=1: run the per-group [all-reduce -> Adam] pipeline (and the collectives) even on a one-rank communicator,
// so that a single-GPU box exercises exactly the code N > 1 runs (tests)
bool dp_force_pipeline() {
    const char *k = getenv("LRCN_DP_FORCE_PIPELINE");
    return k && k[0] == '1';
}

int ensure_buckets(lrcn_ctx *c) {
    if (!c->comm_stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
        c->comm_stream_owned = true;
    }
    for (int g = 0; g < LRCN_GRAD_GROUPS; ++g) {
        c->bucket[g] = c->comm_stream;
        if (!c->bucket_done[g]) HIPCHK(c, hipEventCreateWithFlags(&c->bucket_done[g], hipEventDisableTiming));
        if (!c->ar_done[g]) HIPCHK(c, hipEventCreateWithFlags(&c->ar_done[g], hipEventDisableTiming));
    }
    return LRCN_OK;
}

// The communicator's stream waits for the group's gradient-ready event and all-reduces the group's tensors in place (one collective
// when they are adjacent in memory, which they are in a flat gradient buffer); the group's own stream -- on which the caller may
// queue that group's Adam -- waits for the collective.  One stream for all collectives: the same issue order on every rank, no
// concurrent use of one communicator from several streams.
int allreduce_group(lrcn_ctx *c, float *const grads[9], int group) {
    int64_t sz[9];
    ctx_sizes(c, sz);
    hipStream_t s = c->comm_stream;
    HIPCHK(c, hipStreamWaitEvent(s, c->grad_ev[group], 0));
    if (c->comm && (comm_world(c->comm) > 1 || dp_force_pipeline())) {
        char err[256] = "";
        const int k0 = kGradGroup[group][0], k1 = kGradGroup[group][1];
        int rc = 0;
        if (k0 == k1 || sz[k1] == 0) {
            rc = comm_allreduce_f32(c->comm, grads[k0], (size_t)sz[k0], s, err, sizeof(err));
        } else if (sz[k0] == 0) {
            rc = comm_allreduce_f32(c->comm, grads[k1], (size_t)sz[k1], s, err, sizeof(err));
        } else if (grads[k0] + sz[k0] == grads[k1]) {
            rc = comm_allreduce_f32(c->comm, grads[k0], (size_t)(sz[k0] + sz[k1]), s, err, sizeof(err));
        } else {
            comm_group_begin(c->comm);
            rc = comm_allreduce_f32(c->comm, grads[k0], (size_t)sz[k0], s, err, sizeof(err));
            if (!rc) rc = comm_allreduce_f32(c->comm, grads[k1], (size_t)sz[k1], s, err, sizeof(err));
            comm_group_end(c->comm);
        }
        if (rc) FAIL(c, LRCN_EHIP, "%s", err);
    }
    HIPCHK(c, hipEventRecord(c->ar_done[group], s));
    HIPCHK(c, hipStreamWaitEvent(c->bucket[group], c->ar_done[group], 0));
    c->bucket_pending[group] = true;
    return LRCN_OK;
}

int join_buckets(lrcn_ctx *c) {
    for (int g = 0; g < LRCN_GRAD_GROUPS; ++g)
        if (c->bucket_pending[g]) {
            HIPCHK(c, hipEventRecord(c->bucket_done[g], c->bucket[g]));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->bucket_done[g], 0));
            c->bucket_pending[g] = false;
        }
    return LRCN_OK;
}
}  // namespace

int lrcn_comm_unique_id(void *id_out) {
    if (!id_out) return LRCN_EINVAL;
    char err[256] = "";
    if (comm_unique_id(id_out, err, sizeof(err))) {
        g_create_err = err;
        return LRCN_EHIP;
    }
    return LRCN_OK;
}

int lrcn_comm_probe(lrcn_ctx *c) {
    if (!c) return LRCN_EINVAL;
    if (c->comm) FAIL(c, LRCN_ESTATE, "the context already has a communicator");
    char err[256] = "";
    if (comm_available(err, sizeof(err))) FAIL(c, LRCN_EHIP, "%s", err);
    return LRCN_OK;
}

int lrcn_comm_init(lrcn_ctx *c, int world, int rank, const void *unique_id) {
    DeviceGuard dg(c);
    if (!c) return LRCN_EINVAL;
    if (world < 1 || rank < 0 || rank >= world || !unique_id) FAIL(c, LRCN_EINVAL, "comm_init: world=%d rank=%d", world, rank);
    if (c->comm) FAIL(c, LRCN_ESTATE, "the context already has a communicator");
    char err[256] = "";
    c->comm = comm_create(world, rank, unique_id, err, sizeof(err));
    if (!c->comm) FAIL(c, LRCN_EHIP, "%s", err);
    return ensure_buckets(c);
}

int lrcn_set_embed_rows_buffer(lrcn_ctx *c, float *rows, int32_t *tok, int capacity_rows) {
    if (!c) return LRCN_EINVAL;
    if ((rows == nullptr) != (tok == nullptr) || capacity_rows < 0) FAIL(c, LRCN_EINVAL, "rows and tok must both be given (capacity >= 0) or both NULL");
    c->emb_rows_out = rows;
    c->emb_tok_out = tok;
    c->emb_rows_cap = rows ? capacity_rows : 0;
    return LRCN_OK;
}

int lrcn_embed_grad_from_rows(lrcn_ctx *c, const float *rows, const int32_t *tok, int n_rows, float *grad_wembed, void *stream) {
    DeviceGuard dg(c);
    if (!c || !rows || !tok || !grad_wembed) return LRCN_EINVAL;
    if (n_rows < 1 || n_rows > 8192) FAIL(c, LRCN_EINVAL, "n_rows=%d outside [1, 8192] (the ordered sum sorts its keys in one workgroup)", n_rows);
    if (!c->dWe_rm) DALLOC(c, c->dWe_rm, sizeof(float) * (size_t)c->V * c->ldE);
    if (!c->imp_keys) DALLOC(c, c->imp_keys, sizeof(unsigned long long) * 8192);
    hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : c->stream;
    DropSpec none{};
    // ordered: every rank sums the same rows in the same order -> bit-identical dense gradients (what an all-reduce guarantees)
    if (!k_embed_scatter_rm(st, rows, c->E, tok, n_rows, 1, c->E, c->V, none, c->dWe_rm, c->ldE, grad_wembed, c->imp_keys))
        FAIL(c, LRCN_EINVAL, "embed_grad_from_rows: too many rows (%d)", n_rows);
    KCHK(c, "embed_grad_from_rows");
    return LRCN_OK;
}

int lrcn_comm_set_stream(lrcn_ctx *c, void *hip_stream) {
    DeviceGuard dg(c);
    if (!c || !hip_stream) return LRCN_EINVAL;
    for (bool p : c->bucket_pending)
        if (p) FAIL(c, LRCN_ESTATE, "a gradient exchange is in flight: call lrcn_comm_join first");
    if (c->comm_stream && c->comm_stream_owned) {
        HIPCHK(c, hipStreamSynchronize(c->comm_stream));
        (void)hipStreamDestroy(c->comm_stream);
    }
    c->comm_stream = reinterpret_cast<hipStream_t>(hip_stream);
    c->comm_stream_owned = false;
    return ensure_buckets(c);
}

int lrcn_comm_destroy(lrcn_ctx *c) {
    DeviceGuard dg(c);
    if (!c) return LRCN_EINVAL;
    HIPCHK(c, hipDeviceSynchronize());
    comm_destroy(c->comm);
    c->comm = nullptr;
    return LRCN_OK;
}

int lrcn_allreduce_grads(lrcn_ctx *c, float *const grads[9], int group) {
    DeviceGuard dg(c);
    if (!c || !grads) return LRCN_EINVAL;
    if (group < -1 || group >= LRCN_GRAD_GROUPS) FAIL(c, LRCN_EINVAL, "group=%d outside [-1,%d)", group, LRCN_GRAD_GROUPS);
    int r = ensure_buckets(c);
    if (r) return r;
    for (int g = (group < 0 ? 0 : group); g < (group < 0 ? LRCN_GRAD_GROUPS : group + 1); ++g) {
        r = allreduce_group(c, grads, g);
        if (r) return r;
    }
    return LRCN_OK;
}

int lrcn_comm_join(lrcn_ctx *c) {
    DeviceGuard dg(c);
    if (!c) return LRCN_EINVAL;
    return join_buckets(c);
}

int lrcn_lstm(lrcn_ctx *c, const float *W, const float *b, int X, int H, int B, const float *x, const float *h, const float *cc,
              float *h_out, float *c_out) {
    DeviceGuard dg(c);
    if (!c || !W || !b || !x || !h || !cc || !h_out || !c_out) return LRCN_EINVAL;
    void *Wx, *Wh, *xb, *hb;
    if (X == c->X1 && H == c->H1) {
        Wx = c->W1x; Wh = c->W1h; xb = c->st_x; hb = c->st_h1;
    } else if (c->nl == 2 && X == c->H2 && H == c->H2) {
        Wx = c->W2x; Wh = c->W2h; xb = c->st_x2; hb = c->st_h2;
    } else {
        FAIL(c, LRCN_EINVAL, "lrcn_lstm: (X=%d,H=%d) must be the context's LSTM-1 (%d,%d) or (two layers) LSTM-2 (%d,%d)", X, H, c->X1, c->H1,
             c->H2, c->H2);
    }
    if (B < 1 || B > c->maxB) FAIL(c, LRCN_EINVAL, "B=%d outside [1,%d]", B, c->maxB);
    const int dt = c->dt;
    hipStream_t st = c->stream;
    const int64_t ldX = ld8(X), ldH = ld8(H);
    // shadows of this W: memory [4H][X+H]
    c->shadow_valid = false;
    k_cast_rows(st, dt, W, X + H, 4 * H, X, Wx, ldX);
    k_cast_rows(st, dt, W + X, X + H, 4 * H, H, Wh, ldH);
    // x (B x X column-major = memory [X][B]) -> [B][ldX] T ; h likewise ; c -> f32 row-major
    k_transpose(st, dt, 1, x, B, X, B, xb, ldX, 0);
    k_transpose(st, dt, 1, h, B, H, B, hb, ldH, 0);
    k_transpose_f32(st, cc, B, H, B, c->st_f32[1], H);
    GEMM(c, dt, xb, ldX, Wx, ldX, c->st_g, 4 * H, B, 4 * H, X, b, true);
    GEMM(c, dt, hb, ldH, Wh, ldH, c->st_g, 4 * H, B, 4 * H, H, nullptr, true, true);
    k_lstm_fwd(st, dt, c->st_g, 4 * H, c->st_f32[1], B, H, c->st_a, ld8(4 * H), c->st_f32[1], hb, ldH, c->st_f32[0]);
    k_transpose_f32(st, c->st_f32[0], H, B, H, h_out, B);
    k_transpose_f32(st, c->st_f32[1], H, B, H, c_out, B);
    KCHK(c, "lrcn_lstm");
    return LRCN_OK;
}

int lrcn_step(lrcn_ctx *c, const float *const p[9], float *const state[4], int B, const float *x_cnn, const float *x_lstm,
              const float *mask1, const float *mask2, float *logits) {
    DeviceGuard dg(c);
    if (!c || !p || !state || !x_cnn || !x_lstm || !logits) return LRCN_EINVAL;
    if (B < 1 || B > c->maxB) FAIL(c, LRCN_EINVAL, "B=%d outside [1,%d]", B, c->maxB);
    const int dt = c->dt, E = c->E, H1 = c->H1, H2 = c->H2, h = c->h, V = c->V;
    hipStream_t st = c->stream;
    int r = prepare_weights(c, p, false);
    if (r) return r;
    const int Hs[4] = {H1, H1, H2, H2};
    const int ns = c->nl == 1 ? 2 : 4;  // LRCN-1f: state = {h, c}
    for (int i = 0; i < ns; ++i) {
        if (!state[i]) FAIL(c, LRCN_EINVAL, "null state tensor %d", i);
        k_transpose_f32(st, state[i], B, Hs[i], B, c->st_f32[i], Hs[i]);
    }
    k_transpose_f32(st, x_cnn, B, h, B, c->xcnn, c->ldh);
    // two layers: x = dropout(x_lstm)  (lrcn.jl:542): both arrays are B x E column-major, multiply first, then lay out [B][ldE] (T);
    // mask2 (B x H2) multiplies the concatenated LSTM-2 input (:547).  LRCN-1f: mask1 (B x (E+h)) multiplies hcat(x_lstm, x_cnn).
    DropSpec d2{};
    d2.which = c->nl == 1 ? 1 : 2;
    d2.mask = c->nl == 1 ? mask1 : mask2;
    const float *xsrc = x_lstm;
    if (mask1 && c->nl == 2) {
        k_mul_f32(st, x_lstm, mask1, (int64_t)B * E, c->st_io);
        xsrc = c->st_io;
    }
    k_transpose(st, dt, 1, xsrc, B, E, B, c->st_x, c->ldX1, 0);
    r = step_internal(c, p, B, d2);
    if (r) return r;
    for (int i = 0; i < ns; ++i) k_transpose_f32(st, c->st_f32[i], Hs[i], B, Hs[i], state[i], B);
    k_transpose_f32(st, c->st_logits, c->ldV, B, V, logits, B);
    KCHK(c, "lrcn_step");
    return LRCN_OK;
}

int lrcn_beam_search(lrcn_ctx *c, const float *const p[9], const float *feat, int K, int nword, int32_t *out_tokens, int *out_len,
                     float *out_prob) {
    DeviceGuard dg(c);
    if (!c || !p || !feat || !out_tokens || !out_len) return LRCN_EINVAL;
    if (K < 1 || K > 32 || K > c->maxB || K > c->V) FAIL(c, LRCN_EINVAL, "beam width K=%d must be in [1, min(32, max_B=%d, V=%d)]", K, c->maxB, c->V);
    if (nword < 1 || nword > 256) FAIL(c, LRCN_EINVAL, "nword=%d outside [1,256]", nword);
    const int dt = c->dt, E = c->E, H1 = c->H1, H2 = c->H2, h = c->h, V = c->V;
    hipStream_t st = c->stream;
    int r = prepare_weights(c, p, false);
    if (r) return r;
    // input = input * param[end-3]  (lrcn.jl:611), replicated to K rows
    k_cast_rows(st, dt, feat, LRCN_CNNOUT, 1, LRCN_CNNOUT, c->F, LRCN_CNNOUT);
    for (int i = 1; i < K; ++i)
        HIPCHK(c, hipMemcpyAsync(boff(c->F, (int64_t)i * LRCN_CNNOUT, c->esz), c->F, c->esz * LRCN_CNNOUT, hipMemcpyDeviceToDevice, st));
    GEMM(c, dt, c->F, LRCN_CNNOUT, c->Wcd, LRCN_CNNOUT, c->xcnn, c->ldh, K, h, LRCN_CNNOUT, nullptr, true);
    const int Hs[4] = {H1, H1, H2, H2};
    for (int i = 0; i < 4; ++i) HIPCHK(c, hipMemsetAsync(c->st_f32[i], 0, sizeof(float) * (size_t)K * Hs[i], st));
    struct Hyp {
        std::vector<int32_t> seq;
        float p;
    };
    std::vector<Hyp> x(K);
    for (auto &hy : x) {
        hy.seq = {LRCN_BOS};
        hy.p = 1.0f;
    }
    std::vector<int32_t> last(K), topi((size_t)K * K), parent(K);
    std::vector<float> topv((size_t)K * K);
    DropSpec none{};
    for (int current = 1;; ++current) {
        for (int i = 0; i < K; ++i) last[i] = x[i].seq.back();
        HIPCHK(c, hipMemcpyAsync(c->st_parent, last.data(), sizeof(int32_t) * K, hipMemcpyHostToDevice, st));
        k_embed_gather(st, dt, c->WeT, c->ldE, c->st_parent, 1, K, E, none, c->st_x, c->ldX1);  // lrcn.jl:650
        r = step_internal(c, p, K, none);                                                    // lrcn.jl:651 (K hypotheses batched)
        if (r) return r;
        if (!k_softmax_topk_rows(st, c->st_logits, c->ldV, K, V, K, c->st_topi, c->st_topv)) {  // :652, :655-656 on device
            k_softmax_rows(st, c->st_logits, c->ldV, K, V, c->st_prob, c->ldV);
            k_topk_rows(st, c->st_prob, c->ldV, K, V, K, c->st_topi, c->st_topv);
        }
        HIPCHK(c, hipMemcpyAsync(topi.data(), c->st_topi, sizeof(int32_t) * K * K, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(topv.data(), c->st_topv, sizeof(float) * K * K, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        // candidates (lrcn.jl:657-664): step 1 expands hypothesis 1 only
        const int nexp = current == 1 ? 1 : K;
        std::vector<Hyp> cand;
        std::vector<int> cparent;
        for (int i = 0; i < nexp; ++i)
            for (int j = 0; j < K; ++j) {
                Hyp hy;
                hy.seq = x[i].seq;
                hy.seq.push_back(topi[(size_t)i * K + j]);
                hy.p = topv[(size_t)i * K + j] * x[i].p;
                cand.push_back(std::move(hy));
                cparent.push_back(i);
            }
        // stable descending sort by probability (lrcn.jl:667)
        std::vector<int> order(cand.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cand[a].p > cand[b].p; });
        std::vector<Hyp> xs(K);
        for (int i = 0; i < K; ++i) xs[i] = cand[order[i]];
        const bool done = xs[0].seq.back() == LRCN_EOS || current > nword;  // :670
        if (done) {
            x.swap(xs);
            break;
        }
        for (int i = 0; i < K; ++i) parent[i] = cparent[order[i]];  // :673-676
        HIPCHK(c, hipMemcpyAsync(c->st_parent, parent.data(), sizeof(int32_t) * K, hipMemcpyHostToDevice, st));
        for (int i = 0; i < 4; ++i) {
            k_gather_rows_f32(st, c->st_f32[i], Hs[i], c->st_parent, K, Hs[i], c->st2_f32[i]);
            std::swap(c->st_f32[i], c->st2_f32[i]);
        }
        HIPCHK(c, hipStreamSynchronize(st));  // parent/last host vectors are reused next iteration
        x.swap(xs);
    }
    const int n = (int)x[0].seq.size();
    memcpy(out_tokens, x[0].seq.data(), sizeof(int32_t) * n);
    *out_len = n;
    if (out_prob) *out_prob = x[0].p;
    return LRCN_OK;
}

// generate/beam_search for N images at once (lrcn.jl:585-678 per image; the reference decodes one image at a time with K
// sequential batch-1 lrcn() calls and a device->host copy of V floats per hypothesis per step).  Here the N*K hypotheses of all
// images are the rows of ONE batched lrcn() step; softmax, top-K, candidate ordering, history update and the stop test
// run on the device (beam_update_kernel); the host only polls a done-counter every few steps.  Per image the result is what
// lrcn_beam_search returns (tests/test_gpu_lstm_parity.py).  feats: N x 4096 column-major; out_tokens: [N][nword + 2]
// (bos first), out_len[N], out_prob[N] (may be NULL) on the HOST.
int lrcn_beam_search_batch(lrcn_ctx *c, const float *const p[9], const float *feats, int N, int K, int nword, int32_t *out_tokens,
                           int *out_len, float *out_prob) {
    DeviceGuard dg(c);
    if (!c || !p || !feats || !out_tokens || !out_len) return LRCN_EINVAL;
    if (K < 1 || K > 32 || K > c->V) FAIL(c, LRCN_EINVAL, "beam width K=%d must be in [1, min(32, V=%d)]", K, c->V);
    if (N < 1 || (int64_t)N * K > c->maxB) FAIL(c, LRCN_EINVAL, "N*K = %d*%d exceeds max_B = %d", N, K, c->maxB);
    if (nword < 1 || nword + 2 > LRCN_BEAM_MAXLEN) FAIL(c, LRCN_EINVAL, "nword=%d outside [1,%d]", nword, LRCN_BEAM_MAXLEN - 2);
    const int dt = c->dt, E = c->E, H1 = c->H1, H2 = c->H2, h = c->h, V = c->V;
    const int R = N * K, Lh = nword + 2;
    hipStream_t st = c->stream;
    const bool epi = decode_epi_on(c, R);
    const bool smax = epi && decode_smax_on(c, R, K);
    const bool tables = epi && decode_tables_on(c, R);
    int r = tables ? decode_tables_alloc(c) : LRCN_OK;
    if (r) return r;
    r = prepare_weights(c, p, false, !tables, false, epi, tables);
    if (r) return r;
    // input = input * param[end-3] per image (lrcn.jl:611), each row repeated for the image's K hypotheses
    k_transpose(st, dt, 1, feats, N, LRCN_CNNOUT, N, c->F, LRCN_CNNOUT, 0);
    GEMM(c, dt, c->F, LRCN_CNNOUT, c->Wcd, LRCN_CNNOUT, c->dxcnn, c->ldh, N, h, LRCN_CNNOUT, nullptr, true);
    k_repeat_rows(st, GEMM_T_F32, c->dxcnn, c->ldh, N, K, h, c->xcnn);
    const int Hs[4] = {H1, H1, H2, H2};
    for (int i = 0; i < 4; ++i) HIPCHK(c, hipMemsetAsync(c->st_f32[i], 0, sizeof(float) * (size_t)R * Hs[i], st));
    HIPCHK(c, hipMemsetAsync(c->st_xh1, 0, c->esz * (size_t)R * c->ldXH1, st));  // zero initial h1 / h2 (T copies) and K padding
    HIPCHK(c, hipMemsetAsync(c->st_xh2, 0, c->esz * (size_t)R * c->ldXH2, st));
    if (c->nl == 1) {  // LRCN-1f: the x_cnn columns of [emb | x_cnn | h1] are constant over the decode (a hypothesis never changes image)
        DropSpec nd{};
        k_concat_x2(st, dt, c->st_xh1, c->ldXH1, c->xcnn, c->ldh, 1, R, E, h, nd);
    }
    HIPCHK(c, hipMemsetAsync(c->bs_done, 0, sizeof(int32_t) * N, st));
    HIPCHK(c, hipMemsetAsync(c->bs_ndone, 0, sizeof(int32_t), st));
    k_beam_init(st, c->bs_seq[0], c->bs_last, c->bs_p, R, Lh, LRCN_BOS);  // histories = [bos], probabilities 1, next input = bos
    DropSpec none{};
    int cur = 0;
    if (tables && (r = decode_tables_build(c, p, N, K))) return r;
    for (int current = 1; current <= nword + 1; ++current) {
        if (tables) {
            r = step_decode_tables(c, p, R, current > 1 ? c->st_parent : nullptr, current == 1, smax ? K : 0);
            if (r) return r;
            std::swap(c->st_f32[1], c->st2_f32[1]);
            std::swap(c->st_f32[3], c->st2_f32[3]);
        } else if (epi) {
            // one launch: embedding of every hypothesis' last token (lrcn.jl:650) + h1 / h2 of its parent (:673-676) into the [x | h] operands
            const bool two = c->nl == 2;
            k_decode_prep(st, c->WeT, c->ldE, c->bs_last, current > 1 ? c->st_parent : nullptr, R, E, c->st_h1, c->ldH1, H1, two ? c->st_h2 : nullptr,
                          c->ldH2, H2, c->st_xh1, c->ldXH1, c->ldX1, two ? c->st_xh2 : nullptr, c->ldXH2, c->ldH2);
            r = step_decode(c, p, R, none, true, current > 1 ? c->st_parent : nullptr, current == 1, smax ? K : 0);   // :651, all N*K hypotheses batched
            if (r) return r;
            std::swap(c->st_f32[1], c->st2_f32[1]);
            if (two) std::swap(c->st_f32[3], c->st2_f32[3]);
        } else {
            k_embed_gather(st, dt, c->WeT, c->ldE, c->bs_last, 1, R, E, none, c->st_xh1, c->ldXH1);  // lrcn.jl:650
            r = step_decode(c, p, R, none, false);                                          // :651, all N*K hypotheses batched
            if (r) return r;
        }
        if (smax) {
            // :652, :655-656 happened in the logits GEMM's epilogue + merge
        } else if (!k_softmax_topk_rows(st, c->st_logits, c->ldV, R, V, K, c->st_topi, c->st_topv)) {  // :652, :655-656 in one pass
            k_softmax_rows(st, c->st_logits, c->ldV, R, V, c->st_prob, c->ldV);
            k_topk_rows(st, c->st_prob, c->ldV, R, V, K, c->st_topi, c->st_topv);
        }
        k_beam_update(st, c->st_topi, c->st_topv, c->bs_seq[cur], c->bs_seq[cur ^ 1], c->bs_p, c->st_parent, c->bs_last, c->bs_done,
                      c->bs_ndone, c->bs_res_tok, c->bs_res_len, c->bs_res_p, N, K, Lh, current, nword, LRCN_EOS);
        cur ^= 1;
        if (!epi) {   // :673-676: the four states follow their parents; the T copies of h1 / h2 for the next step's GEMMs ride along
            void *const hT[4] = {boff(c->st_xh1, c->ldX1, c->esz), nullptr, boff(c->st_xh2, c->ldH2, c->esz), nullptr};
            const int64_t ldT[4] = {c->ldXH1, 0, c->ldXH2, 0};
            k_gather_state(st, dt, c->st_f32, c->st2_f32, hT, ldT, Hs, c->st_parent, R);
            for (int i = 0; i < 4; ++i) std::swap(c->st_f32[i], c->st2_f32[i]);
        }
        if ((current & 3) == 0 && current <= nword) {  // every image finished early?
            int32_t nd = 0;
            HIPCHK(c, hipMemcpyAsync(&nd, c->bs_ndone, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            if (nd >= N) break;
        }
    }
    KCHK(c, "beam_search_batch");
    // results through a PINNED staging buffer of the context: a device -> pageable-host copy above 64 KB takes HIP's pin-on-the-fly
    // path (measured: +16 ms per decode from 512 images, whose token block is 67 KB -- more than the 12.9 ms of kernels)
    const size_t nb_tok = sizeof(int32_t) * (size_t)N * Lh, nb_n = sizeof(int32_t) * (size_t)N;
    const size_t need = nb_tok + 2 * nb_n;
    if (need > c->pin_bytes) {
        if (c->pin) (void)hipHostFree(c->pin);
        c->pin = nullptr;
        c->pin_bytes = 0;
        if (hipHostMalloc(&c->pin, need, hipHostMallocDefault) != hipSuccess) FAIL(c, LRCN_ENOMEM, "hipHostMalloc(%zu) failed", need);
        c->pin_bytes = need;
    }
    unsigned char *pin = reinterpret_cast<unsigned char *>(c->pin);
    HIPCHK(c, hipMemcpyAsync(pin, c->bs_res_tok, nb_tok, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(pin + nb_tok, c->bs_res_len, nb_n, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(pin + nb_tok + nb_n, c->bs_res_p, nb_n, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    memcpy(out_tokens, pin, nb_tok);
    memcpy(out_len, pin + nb_tok, nb_n);
    if (out_prob) memcpy(out_prob, pin + nb_tok + nb_n, nb_n);
    return LRCN_OK;
}

// ------------------------------------------------------------------------------------------- VGG
static const int kVggCout[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
static const int kVggPool[13] = {0, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1};

int lrcn_vgg_load(lrcn_ctx *c, const float *const cw[13], const float *const cb[13], const float *fc6_w, const float *fc6_b,
                  const float *fc7_w, const float *fc7_b) {
    DeviceGuard dg(c);
    if (!c || !cw || !cb || !fc6_w || !fc6_b || !fc7_w || !fc7_b) return LRCN_EINVAL;
    if (c->cfg.max_images < 1) FAIL(c, LRCN_ESTATE, "context was created with max_images = 0");
    if (c->vgg_loaded) FAIL(c, LRCN_ESTATE, "VGG weights already loaded");
    const int vdt = c->vdt;
    const size_t ve = c->vesz;
    hipStream_t st = c->stream;
    int Cin = 3, S = 224;
    for (int l = 0; l < 13; ++l) {
        VggLayer &L = c->conv[l];
        L.Cin = Cin;
        L.Cout = kVggCout[l];
        L.S = S;
        L.pool = kVggPool[l];
        DALLOC(c, L.b, sizeof(float) * L.Cout);
        HIPCHK(c, hipMemcpyAsync(L.b, cb[l], sizeof(float) * L.Cout, hipMemcpyDeviceToDevice, st));
        if (l == 0) {
            DALLOC(c, L.w, ve * 64 * 32);
            k_repack_conv11_w(st, vdt, cw[0], 64, L.w, 32);
            if (vdt == GEMM_T_BF16) {
                DALLOC(c, L.w_fused, 2 * 64 * 32);
                k_repack_conv11_w_fused(st, cw[0], cb[0], L.w_fused);
            }
        } else {
            DALLOC(c, L.w, ve * (size_t)L.Cout * 9 * Cin);
            k_repack_conv_w(st, vdt, cw[l], Cin, L.Cout, Cin, L.w);
            if (c->vgg_fp8 && l >= kFp8First) {
                DALLOC(c, L.w8, (size_t)L.Cout * 9 * Cin);
                DALLOC(c, L.sw, sizeof(float) * L.Cout);
                DALLOC(c, L.escale, sizeof(float) * L.Cout);
                DALLOC(c, L.ebias, sizeof(float) * L.Cout);
                k_quant_conv_w_fp8(st, cw[l], Cin, L.Cout, L.w8, L.sw);
            }
        }
        Cin = L.Cout;
        if (L.pool) S /= 2;
    }
    DALLOC(c, c->fc6w, ve * 4096ull * 25088ull);
    DALLOC(c, c->fc7w, ve * 4096ull * 4096ull);
    DALLOC(c, c->fc6b, sizeof(float) * 4096);
    DALLOC(c, c->fc7b, sizeof(float) * 4096);
    k_repack_fc6_w(st, vdt, fc6_w, c->fc6w);
    k_transpose(st, vdt, 1, fc7_w, 4096, 4096, 4096, c->fc7w, 4096, 0);  // (o,k) at o + 4096k -> [o][k]
    HIPCHK(c, hipMemcpyAsync(c->fc6b, fc6_b, sizeof(float) * 4096, hipMemcpyDeviceToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->fc7b, fc7_b, sizeof(float) * 4096, hipMemcpyDeviceToDevice, st));
    KCHK(c, "vgg_load");
    HIPCHK(c, hipStreamSynchronize(st));
    c->vgg_loaded = true;
    return LRCN_OK;
}

namespace {
bool conv64_enabled() {
    const char *k = getenv("LRCN_CONV64");  // LRCN_CONV64=0 routes the Cin = 64 layers back to the implicit-GEMM kernels
    return !(k && k[0] == '0');
}
// f8_inv_scale > 0: write e4m3(out * f8_inv_scale) if the layer's kernel can (returns *wrote_f8), else bf16 as usual
constexpr int kTileCtrStride = 8 + 2 * 512;  // ints per layer: 8 queue heads + two hand-off slots per workgroup (<= 512 workgroups)
// An implicit-GEMM convolution of N images (g.M = N * H * W rows, operands and output of `es` bytes per element), cut into launches
// of whole images whose input stays below the 4 GiB that the direct-to-LDS kernels address with 32-bit offsets (bf16 conv2_2 from
// 1171 images, conv3_1 from 5349; images are independent, so the cut costs nothing but the tail of one more launch).
// Without it a larger batch fell through to the register-staged kernel: 2048 images 72.8 ms per forward, 28 k images/s.
hipError_t launch_conv_chunked(hipStream_t st, const GemmArgs &g0, int N, int es, int64_t limit_bytes = 0) {
    const int64_t per_img = (int64_t)g0.H * g0.W * g0.Cin * es;
    // limit_bytes: LRCN_OPT_CONV_CHUNK_BYTES (tests force several chunks at a handful of images; at least one image per launch)
    int64_t cap = ((limit_bytes > 0 ? limit_bytes : 0xF0000000ll) - (limit_bytes > 0 ? 0 : (int64_t)(g0.W + 1) * g0.Cin * es)) / per_img;
    if (limit_bytes > 0 && cap < 1) cap = 1;
    if (N <= cap || cap < 1) return launch_gemm(st, g0);
    const int nch = (int)((N + cap - 1) / cap), per = (N + nch - 1) / nch;
    const int64_t out_img = (int64_t)(g0.out_mode == GEMM_OUT_POOL ? (g0.H / 2) * (g0.W / 2) : g0.H * g0.W) * g0.ldc * es;
    for (int n0 = 0; n0 < N; n0 += per) {
        GemmArgs g = g0;
        const int n = N - n0 < per ? N - n0 : per;
        g.A = reinterpret_cast<const unsigned char *>(g0.A) + (int64_t)n0 * per_img;
        g.C = reinterpret_cast<unsigned char *>(g0.C) + (int64_t)n0 * out_img;
        g.M = n * g0.H * g0.W;
        g.tile_ctr = nullptr;  // one set of tile queues per launch
        if (hipError_t e = launch_gemm(st, g); e != hipSuccess) return e;
    }
    return hipSuccess;
}

int conv_layer(lrcn_ctx *c, int dtype, const void *in, const VggLayer &L, int N, void *out, float f8_inv_scale = 0.0f, bool *wrote_f8 = nullptr,
               int *tile_ctr = nullptr) {
    if (wrote_f8) *wrote_f8 = false;
    if (conv64_enabled() && conv64_eligible(dtype, L.Cin, L.Cout, L.S, L.S)) {
        const bool f8 = f8_inv_scale > 0.0f && !L.pool;
        if (wrote_f8) *wrote_f8 = f8;
        unsigned long long *stamps = nullptr;
        if (getenv("LRCN_STAMPS")) {  // kernel development (tools/conv64_stamps.py): 16 stamps per 16 x 16 tile per 64-channel chunk 0
            const int64_t need = (int64_t)N * (L.S / 16) * (L.S / 16) * 16;
            if (need > c->stamps_n) {
                c->stamps = nullptr;
                DALLOC(c, c->stamps, sizeof(unsigned long long) * (size_t)need);
                c->stamps_n = need;
            }
            stamps = c->stamps;
        }
        hipError_t e = launch_conv64(c->stream, in, L.w, L.b, out, N, L.S, L.S, L.Cout, 1, L.pool, c->zero_page, f8 ? f8_inv_scale : 0.0f, c->vgg_wg_cap, stamps);
        if (e != hipSuccess) FAIL(c, LRCN_EHIP, "conv64 layer S=%d Cout=%d: %s", L.S, L.Cout, hipGetErrorString(e));
        return LRCN_OK;
    }
    GemmArgs g{};
    g.dtype = dtype;
    g.A = in;
    g.B = L.w;
    g.ldb = 9 * L.Cin;
    g.C = out;
    g.ldc = L.Cout;
    g.M = N * L.S * L.S;
    g.N = L.Cout;
    g.K = 9 * L.Cin;
    g.bias = L.b;
    g.relu = 1;
    g.a_mode = GEMM_A_CONV3;
    g.out_mode = L.pool ? GEMM_OUT_POOL : GEMM_OUT_CONV;
    g.H = g.W = L.S;
    g.Cin = L.Cin;
    g.zero_page = c->zero_page;
    g.ws = c->vgg_ws;  // one split-K workspace per stream: the VGG forward may run beside the LSTM step (gemm_ws)
    g.ws_bytes = c->vgg_ws ? c->gemm_ws_bytes : 0;
    g.wg_cap = c->vgg_wg_cap;
    g.tile_ctr = (c->vgg_wg_cap >= 8 && c->vgg_wg_cap <= 512) ? tile_ctr : nullptr;
    if (getenv("LRCN_STAMPS")) {  // kernel-development (include/lrcn.h lrcn_debug_stamps)
        const int64_t need = ((int64_t)g.M / 256 + 1) * ((int64_t)g.N / 128 + 1) * 8;
        if (need > c->stamps_n) {
            c->stamps = nullptr;  // the previous, smaller buffer stays on the context's allocation list until lrcn_destroy
            DALLOC(c, c->stamps, sizeof(unsigned long long) * (size_t)need);
            c->stamps_n = need;
        }
        g.stamps = c->stamps;
    }
    hipError_t e = launch_conv_chunked(c->stream, g, N, dtype == GEMM_T_BF16 ? 2 : 4, c->conv_chunk_bytes);
    if (e != hipSuccess) FAIL(c, LRCN_EHIP, "conv layer S=%d Cin=%d Cout=%d: %s", L.S, L.Cin, L.Cout, hipGetErrorString(e));
    return LRCN_OK;
}

// e4m3 in -> e4m3 out (lrcn.jl:724-728 conv4 .+ b, relu, pool at reduced precision; scales from lrcn_vgg_calibrate)
int conv_layer_fp8(lrcn_ctx *c, const void *in, const VggLayer &L, int N, void *out, int *tile_ctr = nullptr) {
    GemmArgs g{};
    g.dtype = GEMM_T_F8;
    g.A = in;
    g.B = L.w8;
    g.ldb = 9 * L.Cin;
    g.C = out;
    g.ldc = L.Cout;
    g.M = N * L.S * L.S;
    g.N = L.Cout;
    g.K = 9 * L.Cin;
    g.bias = L.ebias;
    g.scale = L.escale;
    g.relu = 1;
    g.a_mode = GEMM_A_CONV3;
    g.out_mode = L.pool ? GEMM_OUT_POOL : GEMM_OUT_CONV;
    g.H = g.W = L.S;
    g.Cin = L.Cin;
    g.zero_page = c->zero_page;
    g.wg_cap = c->vgg_wg_cap;
    g.tile_ctr = (c->vgg_wg_cap >= 8 && c->vgg_wg_cap <= 512) ? tile_ctr : nullptr;
    // the e4m3 kernel addresses its A operand with SIGNED 32-bit element offsets (gemm_8p_f8_ok: M * Cin < 2^31), half of what the bf16 / f32
    // descriptors reach: cut at 2 GiB minus a margin (round 6: 1536 and 2048 images failed at conv2_2 -- 3.3 GB of e4m3 input in one launch)
    hipError_t e = launch_conv_chunked(c->stream, g, N, 1, c->conv_chunk_bytes > 0 ? c->conv_chunk_bytes : 0x7F000000ll);
    if (e != hipSuccess) FAIL(c, LRCN_EHIP, "fp8 conv layer S=%d Cin=%d Cout=%d: %s", L.S, L.Cin, L.Cout, hipGetErrorString(e));
    return LRCN_OK;
}

// source image (uint8 crops or the preprocessed float tensor) -> featsRM [N][4096] f32
// calibrate: run every layer in bf16 and collect the output amax of conv2_1 .. conv5_3 (post-pool) into amax_dev
int vgg_body(lrcn_ctx *c, int N, const void *src, bool src_u8, const float *mean, bool calibrate = false) {
    const bool fp8 = c->vgg_fp8 && !calibrate;
    if (fp8 && !c->fp8_ready) FAIL(c, LRCN_ESTATE, "vgg_dtype = LRCN_FP8: call lrcn_vgg_calibrate before the first forward");
    const int vdt = c->vdt;
    const float m0 = mean ? mean[0] : 0.f, m1 = mean ? mean[1] : 0.f, m2 = mean ? mean[2] : 0.f;
    const char *kf = getenv("LRCN_FUSE11");  // LRCN_FUSE11=0: conv1_1 and conv1_2 as two launches
    const bool fuse11 = vdt == GEMM_T_BF16 && src_u8 && c->conv[0].w_fused && conv64_enabled() && !(kf && kf[0] == '0');
    const float *avg = (src_u8 && c->avg_on) ? c->avg_img : nullptr;
    // crops that arrived through lrcn_upload_crops: the forward's stream waits for the upload; the staging buffer is free again as soon as the
    // ONE kernel below that reads the uint8 source has run (recorded right after it)
    int staged = -1;
    if (src_u8)
        for (int j = 0; j < lrcn_ctx::kStage; ++j)
            if (c->stage[j] && src == c->stage[j]) staged = j;
    if (staged >= 0) HIPCHK(c, hipStreamWaitEvent(c->stream, c->up_done[staged], 0));
    auto crops_consumed = [&]() -> int {
        if (staged < 0) return LRCN_OK;
        HIPCHK(c, hipEventRecord(c->rd_done[staged], c->stream));
        c->stage_read[staged] = true;
        c->stage_full[staged] = false;
        staged = -1;
        return LRCN_OK;
    };
    if (avg && !fuse11) {
        // full averageImage outside the fused path: read_image_data's arithmetic as its own pass into a float tensor (lrcn.jl:770-771),
        // then the float-input route
        if (!c->pre_f32) DALLOC(c, c->pre_f32, sizeof(float) * (size_t)c->cfg.max_images * 224 * 224 * 3);
        k_preprocess_u8(c->stream, reinterpret_cast<const uint8_t *>(src), N, 224, 0.f, 0.f, 0.f, avg, c->pre_f32);
        if (int r = crops_consumed()) return r;
        src = c->pre_f32;
        src_u8 = false;
    }
    c->vgg_routes.clear();
    auto note = [&](const char *r) {
        if (!c->vgg_routes.empty()) c->vgg_routes += ',';
        c->vgg_routes += r;
    };
    if (fuse11) {
        // read_image_data's arithmetic as an elementwise pass (38 MB -> 77 MB at N = 256); conv1_1 itself runs inside conv1_2's launch
        SegScope seg_pp(c, LRCN_SEG_PREPROCESS, c->stream, 3.0 * N * 224 * 224 * 3);  // 1 B in, one bf16 out per pixel value
        k_img_u8_to_bf16(c->stream, reinterpret_cast<const uint8_t *>(src), (int64_t)N * 224 * 224 * 3, m0, m1, m2, avg, 224, c->img16);
    } else if (vdt == GEMM_T_BF16) {
        // conv1_1 fused with the preprocessing arithmetic (conv11.hip): HBM-bound, no im2col in memory
        k_conv11_fused(c->stream, src_u8 ? 1 : 0, src, N, 224, m0, m1, m2, c->conv[0].w, c->conv[0].b, c->actA);
    } else {
        // f32: conv1_1 as a plain GEMM over an explicit im2col (K = 27), scattered to NHWC
        if (src_u8)
            k_im2col11_u8(c->stream, vdt, reinterpret_cast<const uint8_t *>(src), N, 224, m0, m1, m2, c->im2col, 32);
        else
            k_im2col11_f32(c->stream, vdt, reinterpret_cast<const float *>(src), N, 224, c->im2col, 32);
        GemmArgs g{};
        g.dtype = vdt;
        g.A = c->im2col;
        g.lda = 32;
        g.B = c->conv[0].w;
        g.ldb = 32;
        g.C = c->actA;
        g.ldc = 64;
        g.M = N * 224 * 224;
        g.N = 64;
        g.K = 27;
        g.bias = c->conv[0].b;
        g.relu = 1;
        g.a_mode = GEMM_A_PLAIN;
        g.out_mode = GEMM_OUT_CONV;
        g.H = g.W = 224;
        g.zero_page = c->zero_page;
        hipError_t e = launch_gemm(c->stream, g);
        if (e != hipSuccess) FAIL(c, LRCN_EHIP, "conv1_1: %s", hipGetErrorString(e));
    }
    // (f32: the im2col pass above was the reader and the GEMM after it does not touch the crops -- recording behind it only delays the release)
    if (int r = crops_consumed()) return r;
    if (!fuse11) note(vdt == GEMM_T_BF16 ? "conv11" : gemm_debug_last_route());
    void *cur = c->actA, *nxt = c->actB;
    // capped persistent grids (the two-stream training step): LRCN_DYN_TILES=1 makes the workgroups of a layer PULL their tiles
    // from per-XCD queues instead of walking static round-robin shares.  Measured and left off: the hypothesis was that a
    // workgroup starting late (its CU still held by an LSTM-stream kernel) stretches the whole launch; pulling costs 2 % alone
    // (6.59 -> 6.74 ms per forward at cap 224) and gains nothing in the step (7.54 -> 7.64 ms) -- the contention is not tail imbalance.
    int *ctr = nullptr;
    {
        static const char *kd = getenv("LRCN_DYN_TILES");
        if (c->vgg_wg_cap >= 8 && kd && kd[0] == '1') {
            if (!c->tile_ctr) DALLOC(c, c->tile_ctr, sizeof(int) * 13 * kTileCtrStride);
            HIPCHK(c, hipMemsetAsync(c->tile_ctr, 0, sizeof(int) * 13 * kTileCtrStride, c->stream));
            ctr = c->tile_ctr;
        }
    }
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
    if (c->prof) {
        if (c->prof_used == c->prof_ev.size()) {
            std::pair<hipEvent_t, hipEvent_t> e;
            HIPCHK(c, hipEventCreate(&e.first));
            HIPCHK(c, hipEventCreate(&e.second));
            c->prof_ev.push_back(e);
        }
        ev = &c->prof_ev[c->prof_used++];
        HIPCHK(c, hipEventRecord(ev->first, c->stream));
    }
    int l0 = 1;
    if (fuse11) {  // conv1_1 + conv1_2 + pool1 in one launch, straight from the uint8 crops: actA is never written
        unsigned long long *stamps = nullptr;
        if (getenv("LRCN_STAMPS") && getenv("LRCN_STAMPS")[0] == 'f') {  // LRCN_STAMPS=f: stamp the fused conv1 kernel of a VGG forward
            const int64_t need = (int64_t)N * 14 * 14 * 16;
            if (need > c->stamps_n) {
                c->stamps = nullptr;
                DALLOC(c, c->stamps, sizeof(unsigned long long) * (size_t)need);
                c->stamps_n = need;
            }
            stamps = c->stamps;
        }
        hipError_t e = launch_conv64_fused11(c->stream, c->img16, c->conv[0].w_fused, c->conv[0].b, c->conv[1].w, c->conv[1].b, nxt, N, 224,
                                             c->zero_page, c->vgg_wg_cap, stamps);
        if (e != hipSuccess) FAIL(c, LRCN_EHIP, "fused conv1_1+conv1_2: %s", hipGetErrorString(e));
        note(gemm_debug_last_route());
        std::swap(cur, nxt);
        l0 = 2;
    }
    auto out_count = [&](int l) {
        const VggLayer &L = c->conv[l];
        const int64_t So = L.pool ? L.S / 2 : L.S;
        return (int64_t)N * So * So * L.Cout;
    };
    bool in_is_f8 = false;
    for (int l = l0; l < 13; ++l) {
        if (fp8 && l == kFp8First && !in_is_f8) {  // conv2_1 ran on a kernel without the e4m3 epilogue: one elementwise pass
            k_cast_bf16_fp8(c->stream, cur, out_count(l - 1), 1.0f / c->act_scale[l - 1], nxt);
            std::swap(cur, nxt);
        }
        int r;
        if (fp8 && l >= kFp8First)
            r = conv_layer_fp8(c, cur, c->conv[l], N, nxt, ctr ? ctr + l * kTileCtrStride : nullptr);
        else  // conv2_1 writes the e4m3 input of conv2_2 directly when it runs on conv64.hip
            r = conv_layer(c, vdt, cur, c->conv[l], N, nxt, (fp8 && l == kFp8First - 1) ? 1.0f / c->act_scale[l] : 0.0f, &in_is_f8,
                           ctr ? ctr + l * kTileCtrStride : nullptr);
        if (r) return r;
        note(gemm_debug_last_route());
        std::swap(cur, nxt);
        if (calibrate && l >= kFp8First - 1) k_amax(c->stream, 0, cur, out_count(l), c->amax_dev + l);
    }
    if (fp8) {  // pool5 e4m3 -> bf16 for fc6
        k_cast_fp8_bf16(c->stream, cur, out_count(12), c->act_scale[12], nxt);
        std::swap(cur, nxt);
    }
    if (ev) HIPCHK(c, hipEventRecord(ev->second, c->stream));
    // cur = pool5 output [N][7*7*512]; fc6 + relu6; fc7 (no relu7: lrcn.jl:717)
    GemmArgs g{};
    g.dtype = vdt;
    g.A = cur;
    g.lda = 25088;
    g.B = c->fc6w;
    g.ldb = 25088;
    g.M = N;
    g.N = 4096;
    g.K = 25088;
    g.zero_page = c->zero_page;
    g.ws = c->vgg_ws;
    g.ws_bytes = c->gemm_ws_bytes;
    g.C = c->f6;
    g.ldc = 4096;
    g.bias = c->fc6b;
    g.relu = 1;
    hipError_t e = launch_gemm(c->stream, g);  // N = 256 images: 205 MB of weights through 32 tiles -> gemm_8p's split-K form
    if (e != hipSuccess) FAIL(c, LRCN_EHIP, "fc6: %s", hipGetErrorString(e));
    note(gemm_debug_last_route());
    g.A = c->f6;
    g.lda = 4096;
    g.B = c->fc7w;
    g.ldb = 4096;
    g.C = c->featsRM;
    g.K = 4096;
    g.bias = c->fc7b;
    g.relu = 0;
    g.c_f32 = 1;
    e = launch_gemm(c->stream, g);
    if (e != hipSuccess) FAIL(c, LRCN_EHIP, "fc7: %s", hipGetErrorString(e));
    note(gemm_debug_last_route());
    return LRCN_OK;
}
int vgg_check(lrcn_ctx *c, int N) {
    if (!c->vgg_loaded) FAIL(c, LRCN_ESTATE, "lrcn_vgg_load has not been called");
    if (N < 1 || N > c->cfg.max_images) FAIL(c, LRCN_EINVAL, "N=%d outside [1,%d]", N, c->cfg.max_images);
    return LRCN_OK;
}
}  // namespace

int lrcn_train_step_dp(lrcn_ctx *c, float *const p[9], float *const g[9], float *const m[9], float *const v[9], const uint8_t *img_u8,
                       const float mean[3], int normalize, float *feats, const int32_t *tokens, int T, int B, int norm_B,
                       const lrcn_dropout *drop, int step, float lr, float b1, float b2, float eps, double *loss_host) {
    DeviceGuard dg(c);
    if (!c || !p || !g || !m || !v || !feats || step < 1) return LRCN_EINVAL;
    int r;
    if (img_u8) {  // [VGG forward of this rank's crops] -> feats
        if (!mean && !c->avg_on) FAIL(c, LRCN_EINVAL, "img_u8 given without channel means or an averageImage");
        r = vgg_check(c, B);
        if (r) return r;
        r = vgg_body(c, B, img_u8, true, mean);
        if (r) return r;
        k_transpose_f32(c->stream, c->featsRM, 4096, B, 4096, feats, B);
        if (normalize) k_normalize_rows(c->stream, feats, B, LRCN_CNNOUT);
        KCHK(c, "train_step_dp (vgg)");
    }
    r = loss_impl(c, p, feats, tokens, T, B, norm_B, drop, g, nullptr);
    if (r) return r;
    if (!c->comm || (comm_world(c->comm) == 1 && !dp_force_pipeline())) {  // one rank: nothing to hide the update behind -- one Adam launch
        r = lrcn_adam_update(c, p, g, m, v, step, lr, b1, b2, eps);
        if (r) return r;
        return loss_host ? fetch_loss(c, loss_host) : LRCN_OK;
    }
    // per gradient group, on its own stream, while the rest of the backward pass is still running on the context's stream:
    // [wait for the group's gradients] -> [all-reduce over xGMI] -> [Adam of that group].  The backward reads the shadows made
    // at the start of the step, never the f32 parameters, so updating a group early is safe.
    r = ensure_buckets(c);
    if (r) return r;
    for (int grp = 0; grp < LRCN_GRAD_GROUPS; ++grp) {
        r = allreduce_group(c, g, grp);
        if (r) return r;
        r = lrcn_adam_update_group(c, p, g, m, v, grp, step, lr, b1, b2, eps, c->bucket[grp]);
        if (r) return r;
    }
    r = join_buckets(c);  // the next step's shadow-weight pass reads the updated parameters
    if (r) return r;
    return loss_host ? fetch_loss(c, loss_host) : LRCN_OK;
}

int lrcn_debug_stamps(lrcn_ctx *c, unsigned long long *host_out, int64_t n) {
    DeviceGuard dg(c);
    if (!c || !host_out || n < 1) return LRCN_EINVAL;
    if (!c->stamps || n > c->stamps_n) FAIL(c, LRCN_ESTATE, "no stamps recorded (LRCN_STAMPS=1 and lrcn_bench_conv first), or n > %lld", (long long)c->stamps_n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(host_out, c->stamps, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost));
    return LRCN_OK;
}

const char *lrcn_debug_route(lrcn_ctx *c, int which) {
    if (which == 1) return c ? c->vgg_routes.c_str() : "";
    return gemm_debug_last_route();
}

int lrcn_profile(lrcn_ctx *c, int enable) {
    DeviceGuard dg(c);
    if (!c) return LRCN_EINVAL;
    c->prof = enable != 0;
    c->prof_level = enable;
    c->prof_used = 0;
    c->prof_ms = 0.0;
    c->prof_launches = 0;
    for (auto &sp : c->seg) {
        sp.used = 0;
        sp.ms = sp.bytes = 0.0;
        sp.n = 0;
    }
    return LRCN_OK;
}

int lrcn_profile_segment(lrcn_ctx *c, int segment, double *ms, int64_t *brackets, double *bytes) {
    DeviceGuard dg(c);
    if (!c || !ms || !brackets || !bytes || segment < 0 || segment >= LRCN_SEG_COUNT) return LRCN_EINVAL;
    HIPCHK(c, hipDeviceSynchronize());  // the segments live on several streams (context, copy, caller-given update streams)
    auto &sp = c->seg[segment];
    for (size_t i = 0; i < sp.used; ++i) {
        float t = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&t, sp.ev[i].first, sp.ev[i].second));
        sp.ms += t;
    }
    sp.used = 0;
    *ms = sp.ms;
    *brackets = sp.n;
    *bytes = sp.bytes;
    return LRCN_OK;
}

int lrcn_avg_loss_batch(lrcn_ctx *c, const float *const p[9], const float *feats, const int32_t *tokens, int T, int B, double *loss_host) {
    // average_loss's loop body (lrcn.jl:452-475): pdrop 0, normalised by the batch's own size (lrcn.jl:412)
    return lrcn_loss(c, p, feats, tokens, T, B, B, nullptr, loss_host);
}

int lrcn_profile_get(lrcn_ctx *c, double *conv_ms, int64_t *conv_launches) {
    DeviceGuard dg(c);
    if (!c || !conv_ms || !conv_launches) return LRCN_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < c->prof_used; ++i) {
        float ms = 0.0f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->prof_ev[i].first, c->prof_ev[i].second));
        c->prof_ms += ms;
        c->prof_launches += 12;
    }
    c->prof_used = 0;
    *conv_ms = c->prof_ms;
    *conv_launches = c->prof_launches;
    return LRCN_OK;
}

// Diagnostic: time one bf16 implicit-GEMM convolution layer (random data) in isolation: avg ms over `iters` launches.
int lrcn_bench_conv(lrcn_ctx *c, int N, int S, int Cin, int Cout, int pool, int iters, double *ms_out) {
    DeviceGuard dg(c);
    if (!c || !ms_out || N < 1 || S < 2 || (S & 1) || Cin % 64 || Cout < 1 || iters < 1) return LRCN_EINVAL;
    const size_t in_e = (size_t)N * S * S * Cin, w_e = (size_t)Cout * 9 * Cin, out_e = (size_t)N * S * S * Cout;
    void *in = nullptr, *w = nullptr, *out = nullptr;
    float *tmp = nullptr, *bias = nullptr;
    hipEvent_t e0, e1;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(in); (void)hipFree(w); (void)hipFree(out); (void)hipFree(tmp); (void)hipFree(bias);
    };
    const size_t big = in_e > w_e ? in_e : w_e;
    if (hipMalloc(&in, 2 * in_e) != hipSuccess || hipMalloc(&w, 2 * w_e) != hipSuccess || hipMalloc(&out, 2 * out_e) != hipSuccess ||
        hipMalloc((void **)&tmp, 4 * big) != hipSuccess || hipMalloc((void **)&bias, 4 * Cout) != hipSuccess) {
        cleanup();
        FAIL(c, LRCN_ENOMEM, "bench_conv scratch");
    }
    k_init_uniform(c->stream, tmp, (int64_t)in_e, 1.0f, 11, 0);
    // cast in row chunks of Cin (k_cast_rows works row-wise)
    k_cast_rows(c->stream, GEMM_T_BF16, tmp, Cin, (int)(in_e / Cin), Cin, in, Cin);
    k_init_uniform(c->stream, tmp, (int64_t)w_e, (float)std::sqrt(2.0 / (9.0 * Cin)), 12, 1);
    k_cast_rows(c->stream, GEMM_T_BF16, tmp, 9 * Cin, Cout, 9 * Cin, w, 9 * Cin);
    k_fill(c->stream, bias, Cout, 0.01f);
    if (getenv("LRCN_BENCH_ZERO")) {  // all-zero operands: the same instruction stream at the clock the chip holds WITHOUT data toggling
        (void)hipMemsetAsync(in, 0, 2 * in_e, c->stream);   // (MI355X_MICROARCH.md, DVFS give-back): separates issue efficiency from power
        (void)hipMemsetAsync(w, 0, 2 * w_e, c->stream);
    }
    VggLayer L;
    L.w = w; L.b = bias; L.Cin = Cin; L.Cout = Cout; L.S = S; L.pool = pool;
    int r = conv_layer(c, GEMM_T_BF16, in, L, N, out);  // warm-up
    if (r) { cleanup(); return r; }
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, c->stream);
    for (int i = 0; i < iters && !r; ++i) r = conv_layer(c, GEMM_T_BF16, in, L, N, out);
    (void)hipEventRecord(e1, c->stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    cleanup();
    if (r) return r;
    *ms_out = ms / iters;
    return LRCN_OK;
}

// Diagnostic: time one bf16 NT GEMM C[M][N] = A[M][K] B[N][K]^T (random data, bf16 output) through launch_gemm.
int lrcn_bench_gemm(lrcn_ctx *c, int M, int N, int K, int iters, double *ms_out) {
    DeviceGuard dg(c);
    if (!c || !ms_out || M < 1 || N < 8 || K < 64 || (K % 64) || (N % 8) || iters < 1) return LRCN_EINVAL;
    void *A = nullptr, *B = nullptr, *C = nullptr;
    float *tmp = nullptr;
    const size_t ae = (size_t)M * K, be = (size_t)N * K, ce = (size_t)M * N;
    const size_t big = ae > be ? ae : be;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(A); (void)hipFree(B); (void)hipFree(C); (void)hipFree(tmp);
    };
    if (hipMalloc(&A, 2 * ae) != hipSuccess || hipMalloc(&B, 2 * be) != hipSuccess || hipMalloc(&C, 2 * ce) != hipSuccess ||
        hipMalloc((void **)&tmp, 4 * big) != hipSuccess) {
        cleanup();
        FAIL(c, LRCN_ENOMEM, "bench_gemm scratch");
    }
    k_init_uniform(c->stream, tmp, (int64_t)ae, 1.0f, 21, 0);
    k_cast_rows(c->stream, GEMM_T_BF16, tmp, K, M, K, A, K);
    k_init_uniform(c->stream, tmp, (int64_t)be, 1.0f, 22, 1);
    k_cast_rows(c->stream, GEMM_T_BF16, tmp, K, N, K, B, K);
    int r = gemm(c, GEMM_T_BF16, A, K, B, K, C, N, M, N, K, nullptr, false);
    if (r) { cleanup(); return r; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, c->stream);
    for (int i = 0; i < iters && !r; ++i) r = gemm(c, GEMM_T_BF16, A, K, B, K, C, N, M, N, K, nullptr, false);
    (void)hipEventRecord(e1, c->stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    cleanup();
    if (r) return r;
    *ms_out = ms / iters;
    return LRCN_OK;
}

// ---- input feed (rev 4) ----
int lrcn_host_alloc(void **host_ptr, size_t bytes) {
    if (!host_ptr) return LRCN_EINVAL;
    *host_ptr = nullptr;
    return hipHostMalloc(host_ptr, bytes ? bytes : 16, hipHostMallocDefault) == hipSuccess ? LRCN_OK : LRCN_ENOMEM;
}
int lrcn_host_free(void *host_ptr) { return hipHostFree(host_ptr) == hipSuccess ? LRCN_OK : LRCN_EHIP; }

int lrcn_upload_crops(lrcn_ctx *c, const uint8_t *host_u8, int N, const uint8_t **dev_out) {
    DeviceGuard dg(c);
    if (!c || !host_u8 || !dev_out) return LRCN_EINVAL;
    *dev_out = nullptr;
    if (c->cfg.max_images < 1) FAIL(c, LRCN_ESTATE, "context was created with max_images = 0");
    if (N < 1 || N > c->cfg.max_images) FAIL(c, LRCN_EINVAL, "N=%d outside [1,%d]", N, c->cfg.max_images);
    const size_t per = (size_t)224 * 224 * 3;
    if (!c->copy_stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int j = 0; j < lrcn_ctx::kStage; ++j) {
            HIPCHK(c, hipEventCreateWithFlags(&c->up_done[j], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->rd_done[j], hipEventDisableTiming));
            DALLOC(c, c->stage[j], per * (size_t)c->cfg.max_images);
        }
    }
    const int j = c->stage_next;
    if (c->stage_full[j])
        FAIL(c, LRCN_ESTATE, "all %d staging buffers hold crops that no VGG forward has been issued on yet (upload at most %d batches ahead)",
             lrcn_ctx::kStage, lrcn_ctx::kStage);
    // the forward that last read this buffer: normally long finished; otherwise wait for it HERE, on the host (see lrcn_ctx::kStage)
    if (c->stage_read[j] && hipEventQuery(c->rd_done[j]) != hipSuccess) HIPCHK(c, hipEventSynchronize(c->rd_done[j]));
    {
        SegScope seg_up(c, LRCN_SEG_UPLOAD, c->copy_stream, (double)per * N);
        HIPCHK(c, hipMemcpyAsync(c->stage[j], host_u8, per * (size_t)N, hipMemcpyHostToDevice, c->copy_stream));
    }
    HIPCHK(c, hipEventRecord(c->up_done[j], c->copy_stream));
    c->stage_full[j] = true;
    c->stage_next = (j + 1) % lrcn_ctx::kStage;
    *dev_out = c->stage[j];
    return LRCN_OK;
}

int lrcn_upload_wait(lrcn_ctx *c) {
    DeviceGuard dg(c);
    if (!c) return LRCN_EINVAL;
    if (c->copy_stream) HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    return LRCN_OK;
}

int lrcn_vgg_forward(lrcn_ctx *c, const float *x, int N, float *feats) {
    DeviceGuard dg(c);
    if (!c || !x || !feats) return LRCN_EINVAL;
    int r = vgg_check(c, N);
    if (r) return r;
    r = vgg_body(c, N, x, false, nullptr);
    if (r) return r;
    k_transpose_f32(c->stream, c->featsRM, 4096, N, 4096, feats, N);  // return transpose(xs): N x 4096 column-major
    KCHK(c, "vgg_forward");
    return LRCN_OK;
}

int lrcn_vgg_forward_u8(lrcn_ctx *c, const uint8_t *img, int N, const float mean[3], float *feats) {
    DeviceGuard dg(c);
    if (!c || !img || !feats || (!mean && !c->avg_on)) return LRCN_EINVAL;
    int r = vgg_check(c, N);
    if (r) return r;
    r = vgg_body(c, N, img, true, mean);
    if (r) return r;
    k_transpose_f32(c->stream, c->featsRM, 4096, N, 4096, feats, N);
    KCHK(c, "vgg_forward_u8");
    return LRCN_OK;
}

int lrcn_vgg_forward_u8_blocks(lrcn_ctx *c, const uint8_t *img, int N, const float mean[3], int block_rows, int normalize, float *feats) {
    DeviceGuard dg(c);
    if (!c || !img || !feats || (!mean && !c->avg_on)) return LRCN_EINVAL;
    int r = vgg_check(c, N);
    if (r) return r;
    if (block_rows < 1 || N % block_rows) FAIL(c, LRCN_EINVAL, "block_rows=%d must divide N=%d", block_rows, N);
    r = vgg_body(c, N, img, true, mean);
    if (r) return r;
    for (int b = 0; b < N / block_rows; ++b) {  // block b: rows [b block_rows, (b+1) block_rows) as its own block_rows x 4096 column-major array
        float *dst = feats + (int64_t)b * block_rows * LRCN_CNNOUT;
        k_transpose_f32(c->stream, c->featsRM + (int64_t)b * block_rows * LRCN_CNNOUT, 4096, block_rows, 4096, dst, block_rows);
        if (normalize) k_normalize_rows(c->stream, dst, block_rows, LRCN_CNNOUT);
    }
    KCHK(c, "vgg_forward_u8_blocks");
    return LRCN_OK;
}

int lrcn_preprocess_u8(lrcn_ctx *c, const uint8_t *img, int N, const float mean[3], float *out) {
    DeviceGuard dg(c);
    if (!c || !img || !out || (!mean && !c->avg_on) || N < 1) return LRCN_EINVAL;
    k_preprocess_u8(c->stream, img, N, 224, mean ? mean[0] : 0.f, mean ? mean[1] : 0.f, mean ? mean[2] : 0.f, c->avg_on ? c->avg_img : nullptr, out);
    KCHK(c, "preprocess_u8");
    return LRCN_OK;
}

int lrcn_set_average_image(lrcn_ctx *c, const float *avg) {
    DeviceGuard dg(c);
    if (!c) return LRCN_EINVAL;
    if (!avg) {
        c->avg_on = false;
        return LRCN_OK;
    }
    if (!c->avg_img) DALLOC(c, c->avg_img, sizeof(float) * 224 * 224 * 3);
    HIPCHK(c, hipMemcpyAsync(c->avg_img, avg, sizeof(float) * 224 * 224 * 3, hipMemcpyDeviceToDevice, c->stream));
    c->avg_on = true;
    return LRCN_OK;
}

int lrcn_resize_crop_u8(lrcn_ctx *c, const uint8_t *src, const int64_t *offsets, const int *heights, const int *widths, const int *channels,
                        int N, uint8_t *out) {
    DeviceGuard dg(c);
    if (!c || !src || !offsets || !heights || !widths || !channels || !out) return LRCN_EINVAL;
    if (N < 1 || N > 65536) FAIL(c, LRCN_EINVAL, "N=%d outside [1,65536]", N);
    struct Meta {
        int64_t off;
        int h, w, ch, pad;
    };
    std::vector<Meta> m(N);
    for (int n = 0; n < N; ++n) {
        if (heights[n] < 1 || widths[n] < 1 || heights[n] > 32768 || widths[n] > 32768 || (channels[n] != 1 && channels[n] != 3 && channels[n] != 4) ||
            offsets[n] < 0)
            FAIL(c, LRCN_EINVAL, "image %d: %d x %d x %d at offset %lld (need 1..32768 pixels per side, 1, 3 or 4 channels)", n, heights[n],
                 widths[n], channels[n], (long long)offsets[n]);
        m[n] = Meta{offsets[n], heights[n], widths[n], channels[n], 0};
    }
    if (N > c->img_meta_cap) {
        void *p = nullptr;
        const int cap = N < 256 ? 256 : N;
        if (hipMalloc(&p, sizeof(Meta) * (size_t)cap) != hipSuccess) FAIL(c, LRCN_ENOMEM, "image descriptors");
        c->allocs.push_back(p);  // the old (smaller) buffer stays owned by the context until lrcn_destroy
        c->img_meta = p;
        c->img_meta_cap = cap;
    }
    HIPCHK(c, hipMemcpyAsync(c->img_meta, m.data(), sizeof(Meta) * (size_t)N, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // m goes out of scope
    k_resize_crop_u8(c->stream, src, c->img_meta, N, 224, out);
    KCHK(c, "resize_crop_u8");
    return LRCN_OK;
}

int lrcn_normalize_features(lrcn_ctx *c, float *feats, int N) {
    DeviceGuard dg(c);
    if (!c || !feats || N < 1) return LRCN_EINVAL;
    k_normalize_rows(c->stream, feats, N, LRCN_CNNOUT);
    KCHK(c, "normalize_features");
    return LRCN_OK;
}

int lrcn_conv3x3(lrcn_ctx *c, const float *x, int W, int H, int Cin, int N, const float *w, const float *b, int Cout, int relu,
                 int pool, float *y) {
    DeviceGuard dg(c);
    if (!c || !x || !w || !b || !y) return LRCN_EINVAL;
    if (W < 2 || H < 2 || (W & 1) || (H & 1) || Cin < 1 || Cout < 1 || N < 1) FAIL(c, LRCN_EINVAL, "conv3x3: W,H must be even, sizes positive");
    const int vdt = c->vdt;
    const size_t ve = c->vesz;
    const int bk = vdt == GEMM_T_BF16 ? 64 : 32;
    const int Cp = (int)round_up64(Cin, bk);
    void *xin = nullptr, *wp = nullptr, *out = nullptr;
    float *bd = nullptr;
    const int Wo = pool ? W / 2 : W, Ho = pool ? H / 2 : H;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(xin);
        (void)hipFree(wp);
        (void)hipFree(out);
        (void)hipFree(bd);
    };
    if (hipMalloc(&xin, ve * (size_t)N * H * W * Cp) != hipSuccess || hipMalloc(&wp, ve * (size_t)Cout * 9 * Cp) != hipSuccess ||
        hipMalloc(&out, ve * (size_t)N * Ho * Wo * Cout) != hipSuccess || hipMalloc((void **)&bd, sizeof(float) * Cout) != hipSuccess) {
        cleanup();
        FAIL(c, LRCN_ENOMEM, "conv3x3 scratch");
    }
    (void)hipMemcpyAsync(bd, b, sizeof(float) * Cout, hipMemcpyDeviceToDevice, c->stream);
    k_ref_to_nhwc(c->stream, vdt, x, W, H, Cin, N, xin, Cp);
    k_repack_conv_w(c->stream, vdt, w, Cin, Cout, Cp, wp);
    if (conv64_enabled() && conv64_eligible(vdt, Cp, Cout, H, W)) {
        hipError_t e = launch_conv64(c->stream, xin, wp, bd, out, N, H, W, Cout, relu, pool, c->zero_page);
        if (e == hipSuccess) {
            k_nhwc_to_ref(c->stream, vdt, out, Wo, Ho, Cout, N, Cout, y);
            e = hipGetLastError();
        }
        cleanup();
        if (e != hipSuccess) FAIL(c, LRCN_EHIP, "conv3x3 (conv64): %s", hipGetErrorString(e));
        return LRCN_OK;
    }
    GemmArgs g{};
    g.dtype = vdt;
    g.A = xin;
    g.B = wp;
    g.ldb = 9 * Cp;
    g.C = out;
    g.ldc = Cout;
    g.M = N * H * W;
    g.N = Cout;
    g.K = 9 * Cp;
    g.bias = bd;
    g.relu = relu;
    g.a_mode = GEMM_A_CONV3;
    g.out_mode = pool ? GEMM_OUT_POOL : GEMM_OUT_CONV;
    g.H = H;
    g.W = W;
    g.Cin = Cp;
    g.zero_page = c->zero_page;
    g.ws = c->gemm_ws;
    g.ws_bytes = c->gemm_ws_bytes;
    hipError_t e = launch_gemm(c->stream, g);
    if (e == hipSuccess) {
        k_nhwc_to_ref(c->stream, vdt, out, Wo, Ho, Cout, N, Cout, y);
        e = hipGetLastError();
    }
    cleanup();
    if (e != hipSuccess) FAIL(c, LRCN_EHIP, "conv3x3: %s", hipGetErrorString(e));
    return LRCN_OK;
}

int lrcn_conv1_fused(lrcn_ctx *c, const uint8_t *img, int N, int S, const float mean[3], const float *w11, const float *b11, const float *w12,
                     const float *b12, float *y) {
    DeviceGuard dg(c);
    if (!c || !img || !mean || !w11 || !b11 || !w12 || !b12 || !y) return LRCN_EINVAL;
    if (N < 1 || S < 16 || (S % 16) || (int64_t)N * (S + 4) * (S + 4) * 3 >= (1ll << 31)) FAIL(c, LRCN_EINVAL, "conv1_fused: S must be a multiple of 16, N >= 1");
    void *img16 = nullptr, *wf = nullptr, *wp = nullptr, *out = nullptr;
    float *bd = nullptr;
    const int So = S / 2;
    const size_t img16_bytes = 2 * ((size_t)N * (S + 4) * (S + 4) * 3 + 8);
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(img16);
        (void)hipFree(wf);
        (void)hipFree(wp);
        (void)hipFree(out);
        (void)hipFree(bd);
    };
    if (hipMalloc(&img16, img16_bytes) != hipSuccess || hipMalloc(&wf, 2 * 64 * 32) != hipSuccess || hipMalloc(&wp, 2 * (size_t)64 * 9 * 64) != hipSuccess ||
        hipMalloc(&out, 2 * (size_t)N * So * So * 64) != hipSuccess || hipMalloc((void **)&bd, sizeof(float) * 128) != hipSuccess) {
        cleanup();
        FAIL(c, LRCN_ENOMEM, "conv1_fused scratch");
    }
    (void)hipMemsetAsync(img16, 0, img16_bytes, c->stream);  // the 2-pixel frame is conv1_1's zero padding
    (void)hipMemcpyAsync(bd, b11, sizeof(float) * 64, hipMemcpyDeviceToDevice, c->stream);
    (void)hipMemcpyAsync(bd + 64, b12, sizeof(float) * 64, hipMemcpyDeviceToDevice, c->stream);
    k_img_u8_to_bf16(c->stream, img, (int64_t)N * S * S * 3, mean[0], mean[1], mean[2], nullptr, S, img16);
    k_repack_conv11_w_fused(c->stream, w11, bd, wf);
    k_repack_conv_w(c->stream, GEMM_T_BF16, w12, 64, 64, 64, wp);
    hipError_t e = launch_conv64_fused11(c->stream, img16, wf, bd, wp, bd + 64, out, N, S, c->zero_page, c->vgg_wg_cap);
    if (e == hipSuccess) {
        k_nhwc_to_ref(c->stream, GEMM_T_BF16, out, So, So, 64, N, 64, y);
        e = hipGetLastError();
    }
    cleanup();
    if (e != hipSuccess) FAIL(c, LRCN_EHIP, "conv1_fused: %s", hipGetErrorString(e));
    return LRCN_OK;
}

int lrcn_vgg_calibrate(lrcn_ctx *c, const uint8_t *img, int N, const float mean[3], float margin) {
    DeviceGuard dg(c);
    if (!c || !img || (!mean && !c->avg_on)) return LRCN_EINVAL;
    if (!c->vgg_fp8) FAIL(c, LRCN_ESTATE, "lrcn_vgg_calibrate needs a context created with vgg_dtype = LRCN_FP8");
    if (!(margin >= 1.0f) || margin > 16.0f) FAIL(c, LRCN_EINVAL, "margin=%g outside [1,16]", margin);
    int r = vgg_check(c, N);
    if (r) return r;
    HIPCHK(c, hipMemsetAsync(c->amax_dev, 0, sizeof(float) * 16, c->stream));
    r = vgg_body(c, N, img, true, mean, true);
    if (r) return r;
    float amax[16];
    HIPCHK(c, hipMemcpyAsync(amax, c->amax_dev, sizeof(float) * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int l = kFp8First - 1; l < 13; ++l) {
        if (!(amax[l] > 0.0f) || !std::isfinite(amax[l])) FAIL(c, LRCN_ESTATE, "calibration: layer %d output amax = %g", l, amax[l]);
        c->act_scale[l] = margin * amax[l] / 448.0f;
    }
    for (int l = kFp8First; l < 13; ++l) {
        const VggLayer &L = c->conv[l];
        k_fp8_epilogue_params(c->stream, L.b, L.sw, L.Cout, c->act_scale[l - 1], c->act_scale[l], L.escale, L.ebias);
    }
    KCHK(c, "vgg_calibrate");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->fp8_ready = true;
    return LRCN_OK;
}

int lrcn_conv3x3_fp8(lrcn_ctx *c, const float *x, int W, int H, int Cin, int N, const float *w, const float *b, int Cout, int relu,
                     int pool, float sa_in, float sa_out, float *y, float *sw_out) {
    DeviceGuard dg(c);
    if (!c || !x || !w || !b || !y) return LRCN_EINVAL;
    if (W < 2 || H < 2 || (W & 1) || (H & 1) || Cin < 128 || (Cin % 128) || Cout < 128 || (Cout % 16) || N < 1 || (int64_t)N * W * H < 256 ||
        !(sa_in > 0.0f) || !(sa_out > 0.0f))
        FAIL(c, LRCN_EINVAL, "conv3x3_fp8: need even W,H, Cin %% 128 == 0, Cout >= 128 and %% 16 == 0, N*W*H >= 256, positive scales");
    void *xin = nullptr, *wp = nullptr, *out = nullptr;
    float *f = nullptr;  // b, sw, escale, ebias
    const int Wo = pool ? W / 2 : W, Ho = pool ? H / 2 : H;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(xin);
        (void)hipFree(wp);
        (void)hipFree(out);
        (void)hipFree(f);
    };
    if (hipMalloc(&xin, (size_t)N * H * W * Cin) != hipSuccess || hipMalloc(&wp, (size_t)Cout * 9 * Cin) != hipSuccess ||
        hipMalloc(&out, (size_t)N * Ho * Wo * Cout) != hipSuccess || hipMalloc((void **)&f, sizeof(float) * 4 * Cout) != hipSuccess) {
        cleanup();
        FAIL(c, LRCN_ENOMEM, "conv3x3_fp8 scratch");
    }
    float *bd = f, *sw = f + Cout, *es = f + 2 * Cout, *eb = f + 3 * Cout;
    (void)hipMemcpyAsync(bd, b, sizeof(float) * Cout, hipMemcpyDeviceToDevice, c->stream);
    k_ref_to_nhwc_fp8(c->stream, x, W, H, Cin, N, 1.0f / sa_in, xin);
    k_quant_conv_w_fp8(c->stream, w, Cin, Cout, wp, sw);
    k_fp8_epilogue_params(c->stream, bd, sw, Cout, sa_in, sa_out, es, eb);
    if (sw_out) (void)hipMemcpyAsync(sw_out, sw, sizeof(float) * Cout, hipMemcpyDeviceToDevice, c->stream);
    GemmArgs g{};
    g.dtype = GEMM_T_F8;
    g.A = xin;
    g.B = wp;
    g.ldb = 9 * Cin;
    g.C = out;
    g.ldc = Cout;
    g.M = N * H * W;
    g.N = Cout;
    g.K = 9 * Cin;
    g.bias = eb;
    g.scale = es;
    g.relu = relu;
    g.a_mode = GEMM_A_CONV3;
    g.out_mode = pool ? GEMM_OUT_POOL : GEMM_OUT_CONV;
    g.H = H;
    g.W = W;
    g.Cin = Cin;
    g.zero_page = c->zero_page;
    hipError_t e = launch_gemm(c->stream, g);
    if (e == hipSuccess) {
        k_nhwc_fp8_to_ref(c->stream, out, Wo, Ho, Cout, N, sa_out, y);
        e = hipGetLastError();
    }
    cleanup();
    if (e != hipSuccess) FAIL(c, LRCN_EHIP, "conv3x3_fp8: %s", hipGetErrorString(e));
    return LRCN_OK;
}

}  // extern "C"
