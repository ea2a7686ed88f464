# UNTESTED: no Julia in the image this repository is built and tested in (SURVEY.md section 0.2) -- nothing below has ever been run.
#
# LRCNHip.jl -- the binding a maintainer of lrcn.jl would add to route the hot path through liblrcn_hip.so.  It is written
# for Julia >= 1.6 with plain `ccall`; arrays are device buffers owned by the caller (e.g. AMDGPU.jl ROCArray{Float32}
# or raw pointers from lrcn_malloc).  Julia arrays are column-major, which is exactly what include/lrcn.h expects, so
# nothing is copied or transposed.  Token ids: the reference is 1-based (eos/bos/unk = 1/2/3, lrcn.jl:248-255); the ABI
# is 0-based, so the shim subtracts 1.
module LRCNHip

const lib = get(ENV, "LRCN_HIP_LIB", "liblrcn_hip.so")
const ABI_VERSION = 5   # include/lrcn.h LRCN_ABI_VERSION: the revision the struct layouts below were written against (rev 5 added entry
                        # points only -- lrcn_avg_loss_batch is wrapped below -- the struct layouts are those of rev 2..4)
function __init__()
    v = ccall((:lrcn_abi_version, lib), Cint, ())
    v == ABI_VERSION || error("liblrcn_hip implements ABI revision $v, LRCNHip.jl was written against $ABI_VERSION")
end

struct Config
    device::Cint; E::Cint; H1::Cint; H2::Cint; V::Cint
    max_B::Cint; max_T::Cint; lstm_dtype::Cint; vgg_dtype::Cint; max_images::Cint
    n_layers::Cint      # 0 / 2: lrcn.jl's two-layer model; 1: LRCN-1f (include/lrcn.h)
end
struct Dropout
    pdrop::Cfloat; seed::UInt64; mask1::Ptr{Cfloat}; mask2::Ptr{Cfloat}
end
const F32 = Cint(0); const BF16 = Cint(1)

mutable struct Context
    h::Ptr{Cvoid}
end

function check(ctx, rc)
    rc == 0 && return
    msg = unsafe_string(ccall((:lrcn_last_error, lib), Cstring, (Ptr{Cvoid},), ctx === nothing ? C_NULL : ctx.h))
    error("liblrcn_hip: $msg")          # the reference's convention: Julia exceptions (lrcn.jl:395, 603)
end

function Context(; device=0, embed=1000, hidden=[1000, 1000], vocab, batchsize=25, maxlen=28, dtype=BF16, images=0)
    cfg = Ref(Config(device, embed, hidden[1], hidden[end], vocab, batchsize, maxlen, dtype, dtype, images, length(hidden)))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(nothing, ccall((:lrcn_create, lib), Cint, (Ref{Config}, Ref{Ptr{Cvoid}}), cfg, out))
    ctx = Context(out[])
    finalizer(c -> ccall((:lrcn_destroy, lib), Cvoid, (Ptr{Cvoid},), c.h), ctx)
    ctx
end

ptrs(v) = Ptr{Cfloat}[Ptr{Cfloat}(pointer(a)) for a in v]      # model / grads: Vector of 9 device arrays

# lstm(weight,bias,hidden,cell,input)                                                    lrcn.jl:528-538
function lstm(ctx, weight, bias, hidden, cell, input)
    B, X = size(input); H = size(hidden, 2)
    h2 = similar(hidden); c2 = similar(cell)
    check(ctx, ccall((:lrcn_lstm, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Cint, Cint, Cint, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}),
        ctx.h, pointer(weight), pointer(bias), X, H, B, pointer(input), pointer(hidden), pointer(cell), pointer(h2), pointer(c2)))
    (h2, c2)
end

# lrcn(w, s, x_cnn, x_lstm; pdrop) -- dropout enters as explicit mask arrays               lrcn.jl:540-551
function lrcn(ctx, w, s, x_cnn, x_lstm; mask1=nothing, mask2=nothing)
    B = size(x_lstm, 1); V = size(w[end], 2)
    logits = similar(x_lstm, B, V)
    m1 = mask1 === nothing ? Ptr{Cfloat}(C_NULL) : Ptr{Cfloat}(pointer(mask1))
    m2 = mask2 === nothing ? Ptr{Cfloat}(C_NULL) : Ptr{Cfloat}(pointer(mask2))
    check(ctx, ccall((:lrcn_step, lib), Cint,
        (Ptr{Cvoid}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Cint, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}),
        ctx.h, ptrs(w), ptrs(s), B, pointer(x_cnn), pointer(x_lstm), m1, m2, pointer(logits)))
    logits
end

# tokens: the reference's sequence[range] as a device Int32 array [T][B] (already minus 1)
function loss(ctx, param, input, tokens, T, B; batchsize=B, pdrop=0.0, seed=0)                 # lrcn.jl:553-581
    d = Ref(Dropout(pdrop, seed, C_NULL, C_NULL)); out = Ref{Cdouble}(0)
    check(ctx, ccall((:lrcn_loss, lib), Cint,
        (Ptr{Cvoid}, Ptr{Ptr{Cfloat}}, Ptr{Cfloat}, Ptr{Int32}, Cint, Cint, Cint, Ref{Dropout}, Ref{Cdouble}),
        ctx.h, ptrs(param), pointer(input), pointer(tokens), T, B, batchsize, d, out))
    out[]
end

# the body of average_loss's batch loop: pdrop 0, divided by the batch's own size                lrcn.jl:452-475 (batch size from the data, :412)
function avg_loss_batch(ctx, param, input, tokens, T, B)
    out = Ref{Cdouble}(0)
    check(ctx, ccall((:lrcn_avg_loss_batch, lib), Cint,
        (Ptr{Cvoid}, Ptr{Ptr{Cfloat}}, Ptr{Cfloat}, Ptr{Int32}, Cint, Cint, Ref{Cdouble}),
        ctx.h, ptrs(param), pointer(input), pointer(tokens), T, B, out))
    out[]
end

function lossgradient(ctx, param, input, tokens, T, B, grads; batchsize=B, pdrop=0.0, seed=0)  # lrcn.jl:583
    d = Ref(Dropout(pdrop, seed, C_NULL, C_NULL))
    check(ctx, ccall((:lrcn_loss_grad, lib), Cint,
        (Ptr{Cvoid}, Ptr{Ptr{Cfloat}}, Ptr{Cfloat}, Ptr{Int32}, Cint, Cint, Cint, Ref{Dropout}, Ptr{Ptr{Cfloat}}, Ptr{Cdouble}),
        ctx.h, ptrs(param), pointer(input), pointer(tokens), T, B, batchsize, d, ptrs(grads), C_NULL))
    grads
end

# update!(param, gloss, optim): optim = (m, v, t) with Knet's Adam defaults                lrcn.jl:394, 399-405
function update!(ctx, param, grads, m, v, t; lr=1f-3, beta1=0.9f0, beta2=0.999f0, eps=1f-8)
    check(ctx, ccall((:lrcn_adam_update, lib), Cint,
        (Ptr{Cvoid}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Cint, Cfloat, Cfloat, Cfloat, Cfloat),
        ctx.h, ptrs(param), ptrs(grads), ptrs(m), ptrs(v), t, lr, beta1, beta2, eps))
end

# convnet(xs): (224,224,3,N) -> N x 4096                                                   lrcn.jl:733-748
function convnet(ctx, xs)
    N = size(xs, 4); feats = similar(xs, N, 4096)
    check(ctx, ccall((:lrcn_vgg_forward, lib), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Cint, Ptr{Cfloat}), ctx.h, pointer(xs), N, pointer(feats)))
    feats
end

# beam_search as generate drives it: returns 1-based ids after bos, up to eos              lrcn.jl:585-678
function beam_search(ctx, param, feat, beam_width, nword)
    out = Vector{Int32}(undef, nword + 3); n = Ref{Cint}(0); p = Ref{Cfloat}(0)
    check(ctx, ccall((:lrcn_beam_search, lib), Cint,
        (Ptr{Cvoid}, Ptr{Ptr{Cfloat}}, Ptr{Cfloat}, Cint, Cint, Ptr{Int32}, Ref{Cint}, Ref{Cfloat}),
        ctx.h, ptrs(param), pointer(feat), beam_width, nword, out, n, p))
    (out[1:n[]] .+ Int32(1), p[])
end

# beam_search for N images at once (feats N x 4096): Vector of (1-based ids incl. bos, probability)   (new entry point)
function beam_search_batch(ctx, param, feats, beam_width, nword)
    N = size(feats, 1); L = nword + 2
    out = Matrix{Int32}(undef, L, N); n = Vector{Cint}(undef, N); p = Vector{Cfloat}(undef, N)
    check(ctx, ccall((:lrcn_beam_search_batch, lib), Cint,
        (Ptr{Cvoid}, Ptr{Ptr{Cfloat}}, Ptr{Cfloat}, Cint, Cint, Cint, Ptr{Int32}, Ptr{Cint}, Ptr{Cfloat}),
        ctx.h, ptrs(param), pointer(feats), N, beam_width, nword, out, n, p))
    [(out[1:n[i], i] .+ Int32(1), p[i]) for i in 1:N]
end

# data parallelism: make `stream` (a hipStream_t) wait until gradient group g (0-based, see include/lrcn.h) of the last
# lossgradient is final -- the host then starts that group's all-reduce while the rest of the backward pass runs
grad_group_wait(ctx, g, stream) = check(ctx, ccall((:lrcn_grad_group_wait, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), ctx.h, g, stream))

# update! for one gradient group on `stream`, right after that group's all-reduce (new entry point)
adam_update_group(ctx, param, grads, mom, var, group, step; lr = 0.001f0, b1 = 0.9f0, b2 = 0.999f0, eps = 1f-8, stream = C_NULL) =
    check(ctx, ccall((:lrcn_adam_update_group, lib), Cint,
        (Ptr{Cvoid}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Cint, Cint, Cfloat, Cfloat, Cfloat, Cfloat, Ptr{Cvoid}),
        ctx.h, ptrs(param), ptrs(grads), ptrs(mom), ptrs(var), group, step, lr, b1, b2, eps, stream))

# vgg_dtype = LRCN_FP8 contexts: fix the e4m3 activation scales from one bf16 pass over uint8 crops img[c, col, row, n]   (new entry point)
vgg_calibrate(ctx, img, mean::Vector{Cfloat}; margin = 1.25f0) = check(ctx, ccall((:lrcn_vgg_calibrate, lib), Cint,
    (Ptr{Cvoid}, Ptr{UInt8}, Cint, Ptr{Cfloat}, Cfloat), ctx.h, pointer(img), size(img, 4), mean, margin))

# ---- image front end (lrcn.jl:750-773): decoded images of any size -> uint8 crops [3, 224, 224, N] on the device ----
# src: device buffer with the N images back to back (row-major [h][w][c] bytes); offsets / heights / widths / channels: host vectors
resize_crop(ctx, src, offsets::Vector{Int64}, heights::Vector{Cint}, widths::Vector{Cint}, channels::Vector{Cint}, out) =
    check(ctx, ccall((:lrcn_resize_crop_u8, lib), Cint,
        (Ptr{Cvoid}, Ptr{UInt8}, Ptr{Int64}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Cint, Ptr{UInt8}),
        ctx.h, pointer(src), offsets, heights, widths, channels, length(offsets), pointer(out)))
# the full averageImage of lrcn.jl:113 (device array (224,224,3)); afterwards the *_u8 calls take mean = C_NULL
set_average_image(ctx, avg) = check(ctx, ccall((:lrcn_set_average_image, lib), Cint, (Ptr{Cvoid}, Ptr{Cfloat}), ctx.h, pointer(avg)))
# input / sum(input) per row (lrcn.jl:595-597), in place
normalize_features(ctx, feats) = check(ctx, ccall((:lrcn_normalize_features, lib), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Cint), ctx.h, pointer(feats), size(feats, 1)))
# parity probe of the bf16 stack's first launch (mean subtraction, conv1_1 + ReLU, conv1_2 + ReLU, pool in one kernel): img = N decoded
# crops [n][S][S][3] uint8 on the device, w11 (3,3,3,64), w12 (3,3,64,64) as in the vgg16 .mat file -> y (S/2,S/2,64,N)   lrcn.jl:770, 724-726
conv1_fused(ctx, img, N, S, mean::Vector{Cfloat}, w11, b11, w12, b12, y) = check(ctx, ccall((:lrcn_conv1_fused, lib), Cint,
        (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}),
        ctx.h, pointer(img), N, S, mean, pointer(w11), pointer(b11), pointer(w12), pointer(b12), pointer(y)))

# ---- data parallelism over the GPUs of a node: one Julia process (or task) + one Context per GPU ----
# rank 0: id = comm_unique_id(); ship the 128 bytes to the other ranks (Distributed.jl, a file, MPI ...); every rank: comm_init
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    rc = ccall((:lrcn_comm_unique_id, lib), Cint, (Ptr{UInt8},), id)
    rc == 0 || check(nothing, rc)
    id
end
# local (non-collective) check; every rank probes, the ranks agree, and only then enter the collective comm_init
comm_probe(ctx) = ccall((:lrcn_comm_probe, lib), Cint, (Ptr{Cvoid},), ctx.h) == 0
set_option(ctx, option::Integer, value::Integer) = check(ctx, ccall((:lrcn_set_option, lib), Cint, (Ptr{Cvoid}, Cint, Int64), ctx.h, option, value))
# Adam on one flat run of n floats (a rank's 1/N slice under a sharded data-parallel update)
update_flat!(ctx, w, g, m, v, n, step; lr=1f-3, beta1=0.9f0, beta2=0.999f0, eps=1f-8, stream=C_NULL) =
    check(ctx, ccall((:lrcn_adam_update_flat, lib), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Cint, Cfloat, Cfloat, Cfloat, Cfloat, Ptr{Cvoid}),
                ctx.h, w, g, m, v, n, step, lr, beta1, beta2, eps, stream))
params_touched(ctx) = check(ctx, ccall((:lrcn_params_touched, lib), Cint, (Ptr{Cvoid},), ctx.h))
comm_init(ctx, world, rank, id::Vector{UInt8}) = check(ctx, ccall((:lrcn_comm_init, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), ctx.h, world, rank, id))
allreduce_grads(ctx, grads; group = -1) = check(ctx, ccall((:lrcn_allreduce_grads, lib), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cfloat}}, Cint), ctx.h, ptrs(grads), group))
comm_join(ctx) = check(ctx, ccall((:lrcn_comm_join, lib), Cint, (Ptr{Cvoid},), ctx.h))

# The body of train1's loop (lrcn.jl:369-394) on this rank's rows in ONE call: [VGG-16 forward of img (uint8 crops) -> feats]
# + lossgradient + per-group all-reduce over the ranks + per-group Adam.  batchsize = the GLOBAL batch (lrcn.jl:564-568).
function train_step_dp(ctx, param, grads, mom, var, feats, tokens, T, B, step; img = nothing, mean = Cfloat[123.68, 116.779, 103.939],
                       normalize = true, batchsize = B, pdrop = 0.4, seed = step, lr = 1f-3, beta1 = 0.9f0, beta2 = 0.999f0, eps = 1f-8)
    d = Ref(Dropout(pdrop, seed, C_NULL, C_NULL)); out = Ref{Cdouble}(0)
    check(ctx, ccall((:lrcn_train_step_dp, lib), Cint,
        (Ptr{Cvoid}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Ptr{UInt8}, Ptr{Cfloat}, Cint, Ptr{Cfloat}, Ptr{Int32},
         Cint, Cint, Cint, Ref{Dropout}, Cint, Cfloat, Cfloat, Cfloat, Cfloat, Ref{Cdouble}),
        ctx.h, ptrs(param), ptrs(grads), ptrs(mom), ptrs(var), img === nothing ? Ptr{UInt8}(C_NULL) : Ptr{UInt8}(pointer(img)), mean,
        normalize ? 1 : 0, pointer(feats), pointer(tokens), T, B, batchsize, d, step, lr, beta1, beta2, eps, out))
    out[]
end

# ---- ABI revision 4: the input feed, several batches per VGG forward, the communicator's stream ----
# page-locked host memory for the crops a loader decodes into (the source of an asynchronous upload)
function host_alloc(bytes::Integer)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:lrcn_host_alloc, lib), Cint, (Ref{Ptr{Cvoid}}, Csize_t), p, bytes)
    rc == 0 || error("lrcn_host_alloc failed ($rc)")
    p[]
end
host_free(p) = ccall((:lrcn_host_free, lib), Cint, (Ptr{Cvoid},), p)
# the per-batch host -> device copy of lrcn.jl:369-376 on the context's copy stream; returns device crops for convnet_u8 / train_step_dp
function upload_crops(ctx, host_u8::Ptr{UInt8}, N::Integer)
    dev = Ref{Ptr{UInt8}}(C_NULL)
    check(ctx, ccall((:lrcn_upload_crops, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Ref{Ptr{UInt8}}), ctx.h, host_u8, N, dev))
    dev[]
end
upload_wait(ctx) = check(ctx, ccall((:lrcn_upload_wait, lib), Cint, (Ptr{Cvoid},), ctx.h))
# VGG forward for the crops of N / block_rows consecutive batches; feats: N * 4096 floats = one block_rows x 4096 array per batch
convnet_u8_blocks(ctx, img::Ptr{UInt8}, N::Integer, block_rows::Integer, feats; mean = Cfloat[123.68, 116.779, 103.939], normalize = true) =
    check(ctx, ccall((:lrcn_vgg_forward_u8_blocks, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Ptr{Cfloat}, Cint, Cint, Ptr{Cfloat}),
                ctx.h, img, N, mean, block_rows, normalize ? 1 : 0, pointer(feats)))
# the stream (hipStream_t) on which the context issues its collectives and per-group updates
comm_set_stream(ctx, stream::Ptr{Cvoid}) = check(ctx, ccall((:lrcn_comm_set_stream, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.h, stream))

set_wg_stream(ctx, stream::Ptr{Cvoid}) = check(ctx, ccall((:lrcn_set_wg_stream, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.h, stream))
# sparse exchange of the embedding gradient: lossgradient writes its (T+1) B rows of d(x_lstm) + token ids into (rows, tok) instead of grads[7];
# all-gather them over the ranks (rank order), then embed_grad_from_rows! on every rank rebuilds the dense V x E gradient in one fixed order
set_embed_rows_buffer(ctx, rows, tok, capacity::Integer) =
    check(ctx, ccall((:lrcn_set_embed_rows_buffer, lib), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Int32}, Cint), ctx.h, rows, tok, capacity))
embed_grad_from_rows!(ctx, rows, tok, n_rows::Integer, grad; stream = C_NULL) =
    check(ctx, ccall((:lrcn_embed_grad_from_rows, lib), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Int32}, Cint, Ptr{Cfloat}, Ptr{Cvoid}), ctx.h, rows, tok, n_rows, pointer(grad), stream))

end # module
