# UNTESTED: no Julia in the image this repository is built and tested in -- nothing below has ever been run.
#
# export_jld_to_npy.jl -- run ON A MACHINE WITH JULIA + JLD (this repository's image has neither).
#
# The reference saves its trained model and vocabulary as JLD/HDF5 (`save(o[:savefile], "model", model, "vocab", vocab)`, lrcn.jl:183-186;
# best model per epoch :228-231) and its feature dictionaries as `Dict{Int,Array{Float32}}` (lrcn.jl:206-207, 220;
# feature_extractor.jl:50).  This image has no HDF5 reader, so a reference user exports those files ONCE with this script into plain
# NumPy `.npy` files (format 1.0, written by hand below: no Julia package beyond JLD is needed), which
# `lrcn_amd.formats.load_npy_dir` / `load_feature_npy_dir` and `tools/lrcn.py --loadfile <dir> / --features <dir>` read.
#
#   julia export_jld_to_npy.jl model  trained.jld   out_dir      # -> out_dir/param_<k>_<name>.npy (9 files), out_dir/vocab.tsv
#   julia export_jld_to_npy.jl feats  featsn.jld    out_dir      # -> out_dir/feature_ids.npy (Int64), out_dir/features.npy (4096 x N Float32)
#
# Arrays are written in Julia's own (column-major) memory order with `fortran_order: True`, so NumPy sees the reference's shapes
# unchanged: W1 (E+H1) x 4H1, b1 1 x 4H1, ..., Wembed V x E (initweights, lrcn.jl:489-510).  Vocabulary ids stay 1-based as in the
# reference (lrcn.jl:248-255: eos 1, bos 2, unk 3); the C ABI's 0-based shift happens in lrcn_amd.captions.
using JLD

# The reference does not save KnetArrays directly: `save` goes through its own wrapper type (lrcn.jl:776-781 -- `type KnetJLD; a::Array; end`
# with JLD.writeas / JLD.readas methods).  A standalone reader has to define a type of that NAME and field, or JLD reconstructs an opaque
# placeholder that `Array(p)` cannot convert (ADVICE r5).  Julia >= 0.7 spells `type` as `mutable struct`.
mutable struct KnetJLD
    a::Array
end

const PARAM_NAMES = ["W1", "b1", "W2", "b2", "Wproj", "Wcnn", "Wembed", "Wout", "bout"]

npy_descr(::Type{Float32}) = "<f4"
npy_descr(::Type{Int64}) = "<i8"

function write_npy(path, a::Array)
    T = eltype(a)
    shape = join(map(string, size(a)), ", ") * (ndims(a) == 1 ? "," : "")
    dict = "{'descr': '$(npy_descr(T))', 'fortran_order': True, 'shape': ($shape), }"
    # magic (6) + version (2) + header length (2) + dict + padding + '\n' must be a multiple of 64 bytes
    total = 10 + length(dict) + 1
    pad = (64 - total % 64) % 64
    header = dict * repeat(" ", pad) * "\n"
    open(path, "w") do io
        write(io, UInt8[0x93, 0x4e, 0x55, 0x4d, 0x50, 0x59, 0x01, 0x00])   # "\x93NUMPY" 1.0
        write(io, UInt16(length(header)))                                     # little-endian on every machine Julia runs on
        write(io, header)
        write(io, a)                                                          # column-major memory image
    end
end

# Array as saved on a CPU run; the KnetJLD wrapper of a GPU run (field `a`, above; also whatever placeholder JLD built if the type did not
# resolve, as long as it kept that field); anything else that converts
to_host(p) = convert(Array{Float32}, isa(p, Array) ? p : (isdefined(p, :a) ? getfield(p, :a) : Array(p)))

function export_model(jld, out)
    mkpath(out)
    model = load(jld, "model")
    length(model) == 9 || error("expected the 9-tensor model of initweights (lrcn.jl:489-510), got $(length(model)) tensors")
    for (k, (name, p)) in enumerate(zip(PARAM_NAMES, model))
        a = to_host(p)
        ndims(a) == 2 || (a = reshape(a, size(a, 1), :))
        write_npy(joinpath(out, "param_$(k - 1)_$(name).npy"), a)
    end
    vocab = load(jld, "vocab")
    open(joinpath(out, "vocab.tsv"), "w") do io
        for (w, i) in sort(collect(vocab), by = x -> x[2])
            println(io, w, "\t", i)
        end
    end
    println("wrote 9 tensors and $(length(vocab)) words to $out")
end

function export_feats(jld, out)
    mkpath(out)
    d = load(jld)
    feats = first(values(d))              # the file holds ONE Dict{Int,Array{Float32}} under whatever name it was saved with
    ids = sort(collect(keys(feats)))
    m = zeros(Float32, 4096, length(ids))
    for (j, i) in enumerate(ids)
        m[:, j] = vec(convert(Array{Float32}, feats[i]))   # 4096 or 1 x 4096 (SURVEY a15)
    end
    write_npy(joinpath(out, "feature_ids.npy"), convert(Array{Int64}, ids))
    write_npy(joinpath(out, "features.npy"), m)
    println("wrote $(length(ids)) feature vectors to $out")
end

length(ARGS) == 3 || error("usage: julia export_jld_to_npy.jl model|feats <file.jld> <out_dir>")
ARGS[1] == "model" ? export_model(ARGS[2], ARGS[3]) : ARGS[1] == "feats" ? export_feats(ARGS[2], ARGS[3]) : error("first argument: model or feats")
