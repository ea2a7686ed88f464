#!/usr/bin/env python3
"""Driver with the flag surface of `lrcn.jl`'s main (lrcn.jl:29-188): tokenise -> init/load -> one of
{train, generate, extfeatures}, running the hot path on an MI355X through liblrcn_hip (no CPU path).

    python tools/lrcn.py --coco --train --datafiles captions_train2014.json captions_val2014.json \
        --features train_feats.npz val_feats.npz --savefile m.npz
    python tools/lrcn.py --coco --loadfile m.npz --generate 30 --beam_width 5 --datafiles ... --features ...
    python tools/lrcn.py --cnn --model imagenet-vgg-verydeep-16.mat --loadfile m.npz --generate 30 photo.jpg
    python tools/lrcn.py --cnn --model vgg.mat --extfeatures --imagedir train2014 --prefix COCO_train2014_ --datafiles ...
    python tools/lrcn.py --coco --train --gpus 8 --batchsize 256 --datafiles ... --features ...           # data-parallel: 8 x 32 rows
    python tools/lrcn.py --coco --train --gpus 8 --cnn --model vgg.mat --imagedir train2014 --prefix COCO_train2014_ --datafiles ...   # from images

Training runs on dp.DataParallelTrainer (train.py = train! / train1 / average_loss, lrcn.jl:223-246, 330-397, 407-486): every bucketed
batch is split by rows over the ranks (same T everywhere, the global batch as normaliser, gradients summed over RCCL, identical Adam).
`--gpus N` starts the ranks itself from a parent that never touches the GPU; the torchrun form works too.  `--cnn --train --imagedir`
trains end to end from images: the VGG forward of the next batch runs beside the LSTM step of the current one.

Differences from the reference, all documented in SURVEY.md 5.6 / A.8: `--lr` (default 0.001 = what the reference's
Adam() actually uses) and `--gclip` (default 0 = off) are honoured instead of being parsed and ignored; `--bestfile` is
accepted; data locations are flags (`--features`, `--imagedir`, `--out`) instead of hard-coded paths; files are `.npz`
(formats.py) because JLD/HDF5 cannot be read here; `--dropout` exists (default 0.4 = the hard-coded pdrop of train!).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_parser():
    p = argparse.ArgumentParser(description="LRCN (MI355X): Long-term Recurrent Convolutional Networks for Visual "
                                            "Recognition and Description -- lrcn.jl's CLI on liblrcn_hip")
    p.add_argument("image", nargs="?", default=None, help="Image file (with --cnn --generate).")
    p.add_argument("--model", default="imagenet-vgg-verydeep-16.mat", help="Location of the MatConvNet VGG-16 file")
    p.add_argument("--datafiles", nargs="+", default=[], help="first file: training captions, second: dev, others: test")
    p.add_argument("--loadfile", help="Initialize model from file")
    p.add_argument("--savefile", help="Save final model to file")
    p.add_argument("--bestfile", help="accepted for compatibility (lrcn.jl:63 reads it)")
    p.add_argument("--generate", type=int, default=0, help="If non-zero generate captions of at most this many words.")
    p.add_argument("--hidden", nargs="+", type=int, default=[1000, 1000], help="Sizes of the two LSTM layers.")
    p.add_argument("--embed", type=int, default=1000, help="Size of the embedding vector.")
    p.add_argument("--epochs", type=int, default=10)
    p.add_argument("--capnumber", type=int, default=1000, help="Number of captions to generate.")
    p.add_argument("--batchsize", type=int, default=25, help="Number of sentences to train on in parallel.")
    p.add_argument("--lr", type=float, default=0.001, help="Adam learning rate (the reference always ran 0.001).")
    p.add_argument("--gclip", type=float, default=0.0, help="Gradient-norm clip (0 = off, as in the reference).")
    p.add_argument("--seed", type=int, default=-1)
    p.add_argument("--atype", default="bf16", choices=["bf16", "f32", "fp8", "KnetArray{Float32}", "Array{Float32}"],
                   help="arithmetic type of the LSTM/VGG kernels (the reference's array-type strings select f32; fp8 = e4m3 "
                        "conv2_2..conv5_3 for --generate / --extfeatures, calibrated on the first batch of crops, bf16 elsewhere)")
    p.add_argument("--train", action="store_true")
    p.add_argument("--cnn", action="store_true", help="load the VGG-16 weights")
    p.add_argument("--extfeatures", action="store_true", help="extract fc7 features for the first caption file's images")
    p.add_argument("--flickr", action="store_true")
    p.add_argument("--coco", action="store_true")
    p.add_argument("--beam_width", type=int, default=3)
    p.add_argument("--dropout", type=float, default=0.4, help="pdrop of train! (lrcn.jl:227)")
    p.add_argument("--features", nargs="+", default=[], help=".npz feature dictionaries: train [val] (formats.save_features)")
    p.add_argument("--imagedir", default=".", help="directory of the images for --extfeatures")
    p.add_argument("--prefix", default="", help="file-name prefix before the zero-padded id (COCO_train2014_)")
    p.add_argument("--out", default="eval", help="directory for candidates / ids files of --generate")
    p.add_argument("--workers", type=int, default=max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))),
                   help="threads that decode images for --cnn --train / --extfeatures")
    p.add_argument("--gpus", type=int, default=1, help="--train: ranks of the data-parallel job (one per GPU); batches are split by rows")
    p.add_argument("--dp_backend", default="torch", choices=["torch", "abi"], help="collectives through torch.distributed's RCCL group or the library's own")
    p.add_argument("--shard_adam", action="store_true", help="N > 1: reduce-scatter -> Adam on 1/N of the parameters -> all-gather (dp.py)")
    p.add_argument("--no_normalize", action="store_true", help="--cnn --train: do not divide fc7 features by their sum (lrcn.jl:595-597 does)")
    return p


def tokenize_all(o, cap):
    """Tokenizer.tokenize (tokenizer.jl:6-31): vocab over all files, caption lists per split."""
    lists, counted = [], []
    for path in o.datafiles:
        kind = path.split(".")[-1]
        with open(path) as f:
            if kind == "token":
                lines = f.readlines()
                counted.append(cap.tokenize_flickr(lines))  # vocab counts every caption incl. val/test (:13-15)
                lists.extend(cap.split_flickr(lines))
            elif kind == "json":
                d = cap.tokenize_coco(f.read())
                lists.append(d)
                counted.append(d)
            else:
                print("invalid caption file:", path)
    return cap.build_vocab(counted), lists


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    o = build_parser().parse_args(argv)
    world_env = os.environ.get("WORLD_SIZE")
    if o.gpus > 1 and world_env is None:   # before any torch / HIP import: this parent only starts the ranks and passes their output on
        if not o.train:
            raise SystemExit("--gpus N is the data-parallel TRAINING job (--train)")
        from lrcn_amd import launch
        rc, _ = launch.run_ranks(__file__, argv, o.gpus, {"LRCN_DP_BACKEND": o.dp_backend}, float(os.environ.get("LRCN_CLI_WATCHDOG_S", "86400")),
                                 capture=False)
        return rc
    world, rank, local_rank = int(world_env or "1"), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world != o.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (o.gpus, world))
    say = print if rank == 0 else (lambda *a, **k: None)
    say("opts=", sorted(vars(o).items()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import lrcn_amd
    from lrcn_amd import captions as cap
    from lrcn_amd import dp
    from lrcn_amd import formats as fmt
    from lrcn_amd import lrcn as L
    from lrcn_amd import train as trn

    # LRCN_CLI_FAKE_MULTI=1 (validation on a ONE-GPU box): every rank uses device 0 over gloo -- the N-rank control flow on the real kernels
    fake_multi = world > 1 and os.environ.get("LRCN_CLI_FAKE_MULTI", "0")[:1] == "1"
    torch.cuda.set_device(0 if fake_multi else local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if fake_multi:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if o.seed > 0:
        np.random.seed(o.seed)
    rng = np.random.default_rng(o.seed if o.seed > 0 else None)
    dt = lrcn_amd.LRCN_F32 if o.atype in ("f32", "KnetArray{Float32}", "Array{Float32}") else lrcn_amd.LRCN_BF16
    vdt = lrcn_amd.LRCN_FP8 if o.atype == "fp8" else dt
    if vdt == lrcn_amd.LRCN_FP8 and o.train and o.cnn:
        raise SystemExit("--atype fp8 is the caption-generation / feature-extraction precision; train with bf16 or f32")
    vocab, lists = None, []
    if o.datafiles:
        say("Tokenization starts")
        vocab, lists = tokenize_all(o, cap)
        say("Tokenization finished")
    adam_state = None
    host_model = None
    if o.loadfile:
        say("Loading model from", o.loadfile)
        host_model, vocab, adam_state, _ = fmt.load_checkpoint(o.loadfile)
    if vocab is None:
        raise SystemExit("need --datafiles or --loadfile (no vocabulary)")
    V = len(vocab)
    say("%d unique words" % V)
    if len(o.hidden) != 2 or o.hidden[1] % 2:
        raise SystemExit("--hidden takes two sizes, the second even (LRCN-2f, lrcn.jl:496-504)")
    H1, H2 = o.hidden
    gen_chunk = 256  # images decoded together by the batched beam search (1280 hypothesis rows at beam 5: the decode GEMMs fill the chip)
    ctx = L.Context(o.embed, H1, H2, V, max_B=max(o.batchsize, o.beam_width * (gen_chunk if o.generate > 0 else 1), 10), lstm_dtype=dt, vgg_dtype=vdt,
                    max_images=max(o.batchsize, 256 if o.train else 1) if o.cnn else 0)   # training: room for the crops of several batches per forward
    param = L.initweights(ctx, seed=o.seed if o.seed > 0 else 42) if host_model is None else L.model_from_arrays(host_model)
    say("LSTM is initialized")
    mean = L.VGG_MEAN
    if o.cnn:
        say("Reading", o.model)
        if o.model.startswith("synthetic"):  # "synthetic[:seed]": He-normal weights (no pretrained file offline) -- tests / benchmarks
            L.vgg_load(ctx, *L.synthetic_vgg_weights(seed=int(o.model.split(":")[1]) if ":" in o.model else 1, bias_std=0.05))
        else:
            cw, cb, fc6, fc7, m = fmt.load_vgg_mat(o.model)
            if m is not None:
                mean = tuple(float(v) for v in m)
            L.vgg_load(ctx, [L.to_jl(w) for w in cw], [torch.as_tensor(b).cuda() for b in cb],
                       (L.to_jl(fc6[0]), torch.as_tensor(fc6[1]).cuda()), (L.to_jl(fc7[0]), torch.as_tensor(fc7[1]).cuda()))
            if fmt.load_vgg_mat.average_image is not None:  # the reference subtracts the full array (lrcn.jl:113, 770)
                L.set_average_image(ctx, fmt.load_vgg_mat.average_image)
                mean = None
        say("Cnn is initialized")

    decode_pool = []

    def load_crops(paths):
        """read_image_data (lrcn.jl:750-765) for a batch: decode on the host -- in parallel over the CPUs this process may use (PIL's
        decoders release the GIL); the reference decodes image by image -- then resize / crop / grey->RGB on the GPU."""
        from PIL import Image

        def decode(pth):
            im = Image.open(pth)
            return np.asarray(im if im.mode in ("L", "RGB", "RGBA") else im.convert("RGB"))

        if len(paths) < 4 or o.workers <= 1:
            ims = [decode(p_) for p_ in paths]
        else:
            if not decode_pool:
                from concurrent.futures import ThreadPoolExecutor
                decode_pool.append(ThreadPoolExecutor(max_workers=o.workers))
            ims = list(decode_pool[0].map(decode, paths))
        return L.resize_crop_u8(ctx, ims)
    feats = [fmt.load_features(p) for p in o.features]
    idx2word = cap.index_to_word(vocab)

    def feature_rows(table, ids):
        return L.to_jl(np.stack([table[i].reshape(-1) for i in ids]).astype(np.float32))

    # ---------------------------------------------------------------- generate (lrcn.jl:127-160)
    if o.generate > 0:
        if o.cnn:
            crop = load_crops([o.image])
            if vdt == lrcn_amd.LRCN_FP8:
                L.vgg_calibrate(ctx, crop, mean=mean)
            f = L.convnet_u8(ctx, crop, mean=mean, normalize=True)  # input = input / sum(input)  (lrcn.jl:595-597)
            toks, _ = L.beam_search(ctx, param, f, o.beam_width, o.generate)
            print(cap.caption_text(toks, idx2word))
            return 0
        split = lists[2] if o.flickr and len(lists) > 2 else lists[min(1, len(lists) - 1)]
        table = feats[min(1, len(feats) - 1)] if o.coco else feats[0]
        ids = []
        for k in rng.permutation(len(split)):
            i = split[k][0][0]
            if i not in ids:
                ids.append(i)
            if len(ids) == o.capnumber:
                break
        os.makedirs(o.out, exist_ok=True)
        suffix = "_flickr" if o.flickr else ".txt"
        import gc  # tools/beam_bench.py: keep full cyclic-GC passes (tens of ms with torch's objects) out of the decode loop
        gc.collect()
        gc.freeze()
        with open(os.path.join(o.out, "candidates" + suffix), "w") as out, open(os.path.join(o.out, "candidate_ids" + suffix), "w") as ido:
            for s0 in range(0, len(ids), gen_chunk):  # the reference decodes image by image; here gen_chunk images x beam_width rows per step
                chunk = ids[s0:s0 + gen_chunk]
                for i, (toks, _) in zip(chunk, L.beam_search_batch(ctx, param, feature_rows(table, chunk), o.beam_width, o.generate)):
                    ido.write("%d\n" % i)
                    out.write(cap.caption_text(toks, idx2word) + "\n")
        return 0

    # ---------------------------------------------------------------- extract features (lrcn.jl:162-172, 190-221)
    if o.extfeatures:
        ids = sorted({c[0][0] for c in lists[0]})
        table, B = {}, max(o.batchsize, 1)
        for s in range(0, len(ids), B):  # the reference extracts image by image (lrcn.jl:190-221); here B images per VGG forward
            chunk = ids[s:s + B]
            crops = load_crops([os.path.join(o.imagedir, "%s%012d.jpg" % (o.prefix, i) if o.prefix else "%d.jpg" % i) for i in chunk])
            if vdt == lrcn_amd.LRCN_FP8 and s == 0:  # activation scales of the e4m3 layers from the first batch
                L.vgg_calibrate(ctx, crops, mean=mean)
            f = L.from_jl(L.convnet_u8(ctx, crops, mean=mean))
            for i, row in zip(chunk, f):
                table[i] = row.copy()
        fmt.save_features(o.savefile or "feats.npz", table)
        print("image features extracted")
        return 0

    # ---------------------------------------------------------------- train! (lrcn.jl:223-246) on dp.DataParallelTrainer
    if lists and o.train:
        say("Batching starts")
        seqs = [cap.minibatch(c, vocab, o.batchsize) for c in lists]
        say("Batching finished")
        B_global = seqs[0][3]   # splits with <= 30000 captions are forced to batch 10 (lrcn.jl:260-270)
        if B_global % world:
            raise SystemExit("the batch of %d captions does not split over %d ranks" % (B_global, world))
        optim = L.initparams(param)
        optim.lr = o.lr
        from_images = bool(o.cnn) and not o.features
        seed = o.seed if o.seed > 0 else 0
        # small per-GPU batches: the VGG forward of several upcoming batches as one forward (dp.DataParallelTrainer.step)
        lookahead = trn.batches_per_forward(B_global // world) if from_images else 1
        trainer = dp.DataParallelTrainer(ctx, param, optim, B_global, world, rank, pdrop=o.dropout, seed=seed, backend=o.dp_backend, ops=dp.HipOps(ctx, mean=mean),
                                         vgg_chunk=lookahead, rows=B_global // world,
                                         shard_adam=bool(o.shard_adam) and world > 1, normalize_features=from_images and not o.no_normalize,
                                         gclip=o.gclip)
        if adam_state is not None:   # resume: moments and step count (the reference never saved them)
            trainer.restore(adam=(adam_state["m"], adam_state["v"], adam_state["step"]))
        splits = [list(cap.batches(sq[0], sq[1], sq[2], sq[3])) for sq in seqs[:2]]

        def image_path(i):
            return os.path.join(o.imagedir, "%s%012d.jpg" % (o.prefix, i) if o.prefix else "%d.jpg" % i)

        def crops_of(ids):
            return load_crops([image_path(i) for i in ids])

        if from_images:   # end to end: ids -> crops -> VGG on the device; average_loss runs the same forward batch by batch
            def eval_f(ids):
                return L.convnet_u8(ctx, crops_of(ids), mean=mean, normalize=not o.no_normalize)
            kw = dict(crops_of=crops_of, eval_feats_of=[eval_f] * len(splits))
        else:
            if not feats:
                raise SystemExit("--train needs --features (precomputed fc7 features) or --cnn --imagedir (train from images)")
            tables = [feats[0], feats[min(1, len(feats) - 1)]]
            kw = dict(feats_of=lambda ids: feature_rows(tables[0], ids), eval_feats_of=[(lambda ids, t=t: feature_rows(t, ids)) for t in tables[:len(splits)]])

        def save(epoch):
            if o.savefile and rank == 0:
                fmt.save_checkpoint(o.savefile, [L.from_jl(p) for p in param], vocab,
                                    adam={"m": [L.from_jl(t) for t in optim.m], "v": [L.from_jl(t) for t in optim.v], "step": optim.t})

        trn.train(trainer, splits, o.epochs, seed, save=save, log=say, sync=ctx.sync, lookahead=lookahead, **kw)
        trainer.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0
    if o.savefile and not o.train:
        fmt.save_checkpoint(o.savefile, [L.from_jl(p) for p in param], vocab)
    return 0


if __name__ == "__main__":
    sys.exit(main())
