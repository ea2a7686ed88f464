#!/usr/bin/env python3
"""bench.py -- training images/sec of the LRCN step (VGG-16 -> fc7 forward + 2-layer LSTM fwd/bwd + Adam) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[3] / SURVEY.md 8(d) "C4": MS-COCO-shaped synthetic data, VGG-16 bf16 + LSTM
E=H1=H2=1000 bf16 (fp32 accumulate / master weights / Adam), V=10640, T=11, GLOBAL batch 256 split by rows over the
N ranks ("strong" scaling), dropout 0.4, one RCCL all-reduce(SUM) of the 39.8 M fp32 gradients per step.
A step = [VGG fwd on B/N images] + lossgradient + all-reduce + update!; inputs (uint8 crops, tokens) resident in HBM.
The VGG forward of step k+1 runs on a side HIP stream beside the LSTM work of step k (dp.py).  The timed region is the pipeline
in steady state: each of its K steps issues one VGG forward (of the next batch) and one LSTM step, K of each in total.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work, SURVEY.md 8(d)
VGG_CONV_GFLOP_PER_IMAGE = 30.693        # 13 conv layers
CONV11_GFLOP_PER_IMAGE = 2 * 224 * 224 * 64 * 27 / 1e9   # conv1_1 runs as a separate (im2col GEMM) launch
PEAK_BF16_TFLOPS = 2516.0                # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF)
PEAK_F32_TFLOPS = 157.3


def pmc_traffic(dtype, per_gpu_batch):
    """HBM-side bytes per launch of the dominant kernel family, from the committed rocprofv3 PMC passes of this same
    command (tools/pmc_traffic.py -> profiles/*.json; FETCH_SIZE/WRITE_SIZE cannot be read from inside the process).
    Only valid for the configuration it was measured on; otherwise null."""
    path = os.path.join(ROOT, "profiles", "r02d_pmc_traffic_bench_bf16_b256.json")
    if dtype != "bf16" or per_gpu_batch != 256 or not os.path.exists(path):
        return None
    try:
        return float(json.load(open(path))["traffic_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(vgg_w, E, H, V, T, rng, n_layers=2):
    """The CPU baseline (kind "port": the reference is Julia/GPU-only and cannot run): the SAME step through the SAME C ABI
    (include/lrcn.h) on this box's host cores -- oracle/liblrcn_cpu_f32.so, the oracle's baseline build behind lrcn_vgg_forward_u8 +
    lrcn_loss_grad -- on a bounded sample of the workload: 8 crops + 16 captions.  Threads = the container's CPU share.
    -> (cpu_baseline object, sample): the sample's inputs and outputs, which main() pushes through the HIP path afterwards
    (outside every timed region) for the `parity` spot-check of the same JSON line."""
    import ctypes as C
    import numpy as np
    from lrcn_amd import _lib
    from oracle import oracle as orc
    conv_w, conv_b, fc6, fc7 = vgg_w
    n_img, n_cap = 8, 16
    ncpu = orc.effective_cpus()       # the container's CPU share, not the host's core count
    A = orc.cpu_abi(_lib.SIGNATURES, fast=True)
    A.orc_set_num_threads(ncpu)
    cfg = _lib.Config(0, E, H, H, V, n_cap, T, _lib.LRCN_F32, _lib.LRCN_F32, n_img, n_layers)
    h = C.c_void_p()
    assert A.lrcn_create(C.byref(cfg), C.byref(h)) == 0

    def fp(a):
        return a.ctypes.data_as(C.c_void_p)

    keep = [orc.fa(a) for a in conv_w] + [orc.fa(a) for a in conv_b] + [orc.fa(fc6[0]), orc.fa(fc6[1]), orc.fa(fc7[0]), orc.fa(fc7[1])]
    assert A.lrcn_vgg_load(h, _lib.P13(*[fp(a).value for a in keep[:13]]), _lib.P13(*[fp(a).value for a in keep[13:26]]), fp(keep[26]),
                           fp(keep[27]), fp(keep[28]), fp(keep[29])) == 0
    img = rng.integers(0, 256, size=(n_img, 224, 224, 3), dtype=np.uint8)
    ref_feats = np.zeros((n_img, 4096), np.float32, order="F")
    mean = (C.c_float * 3)(123.68, 116.779, 103.939)
    t0 = time.time()
    assert A.lrcn_vgg_forward_u8(h, fp(img), n_img, mean, fp(ref_feats)) == 0
    t_vgg = (time.time() - t0) / n_img
    m = orc.init_weights(E, H, H, V, seed=42, n_layers=n_layers)
    g = m.zeros_like()
    feats = orc.fa((rng.standard_normal((n_cap, 4096)) * 0.01).astype(np.float32))
    tokens = rng.integers(3, V, size=(T, n_cap)).astype(np.int32)

    def p9(mm):
        return _lib.P9(*[fp(a).value if a.size else None for a in mm.arrays()])

    out = C.c_double()
    t0 = time.time()
    assert A.lrcn_loss_grad(h, p9(m), fp(feats), fp(tokens), T, n_cap, n_cap, None, p9(g), C.byref(out)) == 0
    t_lstm = (time.time() - t0) / n_cap
    A.lrcn_destroy(h)
    base = {"value": 1.0 / (t_vgg + t_lstm), "unit": "images/sec", "cores": ncpu, "kind": "port",
            "sample": "oracle/liblrcn_cpu_f32.so = include/lrcn.h on the host (oracle baseline build: float accumulate, OpenMP, "
                      "convolutions as im2col + register-blocked AVX2 SGEMM): lrcn_vgg_forward_u8 on %d crops (%.3f s/img = %.0f GFLOP/s) + "
                      "lrcn_loss_grad on %d captions of T=%d (%.3f s/caption); Adam excluded (<1%%); %d threads = the container's CPU "
                      "share" % (n_img, t_vgg, 30.93 / max(t_vgg, 1e-9), n_cap, T, t_lstm, ncpu)}
    return base, {"img": img, "ref_feats": ref_feats, "model": m, "feats": feats, "tokens": tokens, "ref_loss": out.value,
                  "ref_grads": g}


def parity_spot_check(ctx, L, sample, batch_imgs):
    """The cpu_baseline sample through the HIP path of THIS context: the 4 crops replace the first rows of one of the
    benchmark's own image batches, so the VGG forward runs at the benchmark's batch size with the benchmark's kernels and
    routes (reported); loss + gradients of the 16 captions against the oracle's.  Not timed."""
    import numpy as np
    import torch
    imgs = batch_imgs.clone()
    n = min(sample["img"].shape[0], imgs.shape[0])
    imgs[:n] = torch.as_tensor(sample["img"][:n]).cuda()
    got = L.from_jl(L.convnet_u8(ctx, imgs))[:n]
    ref = sample["ref_feats"][:n]
    vgg_err = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))
    m = sample["model"]
    grads, val = L.lossgradient(ctx, L.model_from_arrays(m.p), L.to_jl(sample["feats"]), sample["tokens"])
    cos = []
    for n, g in zip(L.PARAM_NAMES, grads):
        if g.numel() == 0:
            continue
        a, b = L.from_jl(g).ravel().astype(np.float64), sample["ref_grads"].p[n].ravel().astype(np.float64)
        cos.append(float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300)))
    return {"vgg_rel_max_err": vgg_err, "n_images": int(ref.shape[0]), "vgg_routes": L.debug_route(ctx, 1),
            "loss_rel_err": float(abs(val - sample["ref_loss"]) / abs(sample["ref_loss"])), "n_captions": int(sample["tokens"].shape[1]),
            "grad_cos_min": min(cos), "checker": "oracle (liblrcn_cpu_f32.so) on the cpu_baseline sample"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)    # >= 50 timed steps after >= 10 warm-up
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--global-batch", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--hidden", type=int, default=1000)
    ap.add_argument("--vocab", type=int, default=10640)
    ap.add_argument("--T", type=int, default=11)
    ap.add_argument("--pdrop", type=float, default=0.4)
    ap.add_argument("--layers", type=int, default=2, choices=[1, 2])  # 1 = LRCN-1f, BASELINE configs[1] (with --dtype f32 --global-batch 32 --hidden 512 --vocab 2540)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import lrcn_amd
    from lrcn_amd import dp
    from lrcn_amd import lrcn as L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (a.gpus, a.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    dt = lrcn_amd.LRCN_BF16 if a.dtype == "bf16" else lrcn_amd.LRCN_F32
    E = H = a.hidden
    V, T, Bg = a.vocab, a.T, a.global_batch
    rows = dp.shard_rows(Bg, world, rank)
    B = rows.stop - rows.start

    ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=dt, vgg_dtype=dt, max_images=B, n_layers=a.layers)
    vgg_w = L.synthetic_vgg_weights(seed=1)
    L.vgg_load(ctx, *vgg_w)
    param = L.initweights(ctx, seed=42)          # identical on every rank (same seed)
    optim = L.initparams(param)
    trainer = dp.DataParallelTrainer(ctx, param, optim, Bg, world, rank, pdrop=a.pdrop, seed=7)

    # synthetic inputs: uint8 crops uniform seed 1234; Zipf(1.0) word ids >= 3, seed 7; one T per batch
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    n_sets = 2
    imgs_all = [torch.randint(0, 256, (Bg, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)[rows].contiguous()
                for _ in range(n_sets)]
    rng = np.random.default_rng(7)
    pz = 1.0 / np.arange(1, V - 3 + 1)
    pz /= pz.sum()
    toks_all = [torch.as_tensor((rng.choice(V - 3, size=(T, Bg), p=pz) + 3).astype(np.int32)[:, rows.start:rows.stop]
                                .copy()).cuda() for _ in range(n_sets)]

    step_ev = []

    def run(nsteps, events=False):
        # steady-state pipeline: EVERY step (the last one too) issues the VGG forward of the batch after it, so a run of K steps
        # holds exactly K VGG forwards + K LSTM steps; the features a run's first step consumes were produced by the previous
        # run's last step (the warm-up's, for the timed region), or in order if there is none.
        for _ in range(nsteps):
            k = trainer.step_no  # global step index: batch k uses image/token set k mod n_sets, also across warm-up -> timed region
            trainer.step(imgs_all[k % n_sets], toks_all[k % n_sets], next_img_u8=imgs_all[(k + 1) % n_sets])
            if events:  # one event per step on the main stream (no host sync): per-step intervals -> the median SURVEY 8(d) asks for
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                step_ev.append(e)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import gc  # a full pass of Python's cyclic GC scans every object torch created (tens of ms): none inside the timed region
    if os.environ.get("LRCN_BENCH_GC_LATE"):
        run(a.warmup)
        gc.collect()
        gc.freeze()
    else:
        gc.collect()
        gc.freeze()
        run(a.warmup)
    barrier()
    _lib = lrcn_amd._lib
    _lib.check(ctx._h, _lib.lib().lrcn_profile(ctx._h, 1))
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    step_ev.append(e0)
    t0 = time.perf_counter()
    run(a.steps, events=True)
    barrier()
    dt_s = time.perf_counter() - t0
    step_ms_seq = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(len(step_ev) - 1)]
    step_ms = sorted(step_ms_seq)
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    import ctypes as C
    conv_ms, conv_n = C.c_double(), C.c_int64()
    _lib.check(ctx._h, _lib.lib().lrcn_profile_get(ctx._h, C.byref(conv_ms), C.byref(conv_n)))
    _lib.check(ctx._h, _lib.lib().lrcn_profile(ctx._h, 0))
    loss = trainer.loss_value()
    tt = torch.tensor([dt_s], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt_s = float(tt.item())

    if rank == 0:
        ms_step = 1e3 * dt_s / a.steps
        value = Bg * a.steps / dt_s
        # dominant kernel family: the 12 convolution launches conv1_2..conv5_3 (2 x conv64_kernel + 10 x gemm8p_kernel<CONV3>)
        # bf16: conv1_1 runs inside conv1_2's launch (conv64.hip FUSE), so its FLOPs belong to the 12 timed launches
        fused11 = a.dtype == "bf16" and os.environ.get("LRCN_FUSE11", "1")[:1] != "0" and os.environ.get("LRCN_CONV64", "1")[:1] != "0"
        flops_per_launch = (VGG_CONV_GFLOP_PER_IMAGE - (0.0 if fused11 else CONV11_GFLOP_PER_IMAGE)) * 1e9 * B / 12.0
        avg_launch_s = conv_ms.value * 1e-3 / max(conv_n.value, 1)
        achieved = flops_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        out = {
            "metric": "training images/sec (VGG16+LSTM, COCO, batch 256) at 1/2/4/8 MI355X",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "ms_per_step_median": median_ms,
            "ms_per_step_first8": [round(x, 3) for x in step_ms_seq[:8]], "ms_per_step_last": round(step_ms_seq[-1], 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "%s: VGG-16 -> fc7 fwd + %s LSTM "
                                   "E=H=%d V=%d T=%d fwd/bwd + Adam, global batch %d, dp%d, dropout %.1f; synthetic uint8 "
                                   "224x224 crops, He-normal VGG weights"
                                   % ("BASELINE.json configs[3] (C4), MS-COCO-shaped" if (a.layers == 2 and a.dtype == "bf16" and Bg == 256 and H == 1000)
                                      else ("BASELINE.json configs[1] (C2), Flickr8k-shaped" if (a.layers == 1 and a.dtype == "f32" and Bg == 32 and H == 512)
                                            else "custom"),
                                      "LRCN-2f (2-layer)" if a.layers == 2 else "LRCN-1f (1-layer)", H, V, T, Bg, world, a.pdrop),
                       "global_batch": Bg, "per_gpu_batch": B, "seq_len": T + 1, "parallelism": "dp%d" % world,
                       "last_loss": loss},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": pmc_traffic(a.dtype, B),
                         "kernel": "conv64_kernel (conv1_1+conv1_2 fused, conv2_1) + gemm8p_kernel<*,CONV3,*> (conv2_2..conv5_3): 12 launches/step"
                                   if a.dtype == "bf16" else "gemm_glds_kernel<float,*,CONV3,*> v_mfma_f32_32x32x2_f32 (conv1_2..conv5_3)",
                         "avg_launch_ms": 1e3 * avg_launch_s, "flops_per_launch": flops_per_launch},
        }
        if world == 1 and not a.no_cpu_baseline:
            host_w = ([L.from_jl(w) for w in vgg_w[0]], [b.cpu().numpy() for b in vgg_w[1]],
                      (L.from_jl(vgg_w[2][0]), vgg_w[2][1].cpu().numpy()), (L.from_jl(vgg_w[3][0]), vgg_w[3][1].cpu().numpy()))
            out["cpu_baseline"], sample = cpu_baseline(host_w, E, H, V, T, np.random.default_rng(3), a.layers)
            out["parity"] = parity_spot_check(ctx, L, sample, imgs_all[0])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
