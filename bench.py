#!/usr/bin/env python3
"""bench.py -- training images/sec of the LRCN step (VGG-16 -> fc7 forward + 2-layer LSTM fwd/bwd + Adam) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[3] / SURVEY.md 8(d) "C4": MS-COCO-shaped synthetic data, VGG-16 bf16 + LSTM
E=H1=H2=1000 bf16 (fp32 accumulate / master weights / Adam), V=10640, T=11, GLOBAL batch 256 split by rows over the
N ranks ("strong" scaling), dropout 0.4, one RCCL all-reduce(SUM) of the 39.8 M fp32 gradients per step.
A step = [VGG fwd on B/N images] + lossgradient + all-reduce + update!.  Crops: pinned HOST memory, uploaded per step on a copy stream
a step ahead of the forward that reads them (--inputs host, the default: BASELINE.md section 3 / lrcn.jl:369-376) or resident in HBM
(--inputs hbm); tokens resident.
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
SCRIPT = os.path.abspath(__file__)   # what launch() / supervise() start as ranks (tests/bench_dryrun.py wraps this file and puts itself here)

# algorithmic work, SURVEY.md 8(d)
VGG_CONV_GFLOP_PER_IMAGE = 30.693        # 13 conv layers
CONV11_GFLOP_PER_IMAGE = 2 * 224 * 224 * 64 * 27 / 1e9   # conv1_1 runs as a separate (im2col GEMM) launch
PEAK_BF16_TFLOPS = 2516.0                # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF)
PEAK_F32_TFLOPS = 157.3


class HwSampler:
    """Shader clock and socket power HELD DURING THE TIMED REGION ("a 7.19 ms lease is explained by a number").  A host
    thread reads the amdgpu hwmon nodes of this rank's device -- freq1_input (sclk, Hz) and power1_input (socket power, microwatts) under
    /sys/bus/pci/devices/<bdf>/hwmon/hwmon*; plain sysfs reads, no SMI library, no GPU call, ~0.1 ms per sample -- every `period_s` between
    start() and stop().  The step is power-bound (0.95-0.97 of the 1400 W cap), so the clock the package holds under
    this lease's silicon, cooling and neighbours IS the lease-to-lease spread of the headline.  Absent nodes (another driver, a
    container without sysfs) -> {"available": false}; never an error."""

    def __init__(self, device_index, period_s=0.010, hwmon_dir=None):   # hwmon_dir: tests
        self.period_s, self.samples, self._stop, self._rec, self._th, self.dir = period_s, [], False, False, None, None
        if hwmon_dir is not None:
            self.dir, self.bdf = hwmon_dir, "test"
        else:
            try:
                import glob
                import torch
                p = torch.cuda.get_device_properties(device_index)
                bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
                d = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
                if d and os.path.exists(os.path.join(d[0], "freq1_input")):
                    self.dir, self.bdf = d[0], bdf
            except Exception:
                self.dir = None
        if self.dir is not None:
            # the thread exists (asleep) from construction on; start() only raises a flag.  Creating it at the head of the timed region cost the
            # first timed step 0.5-1 ms (a new thread and its first sysfs reads contend with the host thread that is racing to refill the
            # launch queues after the barrier); at 4 ms per sample the whole run was 1 % slower than with the sampler off -- now 10 ms
            import threading
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read())
        except Exception:
            return None

    def _loop(self):
        while not self._stop:
            if self._rec:
                self.samples.append((self._read("freq1_input"), self._read("power1_input")))
            time.sleep(self.period_s)

    def start(self):
        self._rec = True

    def stop(self):
        self._rec = False
        self._stop = True
        if self._th is not None:
            self._th.join(timeout=1.0)
        if self.dir is None:
            return {"available": False}

        def stats(v, scale):
            v = sorted(x * scale for x in v if x is not None)
            return {"n": len(v), "min": round(v[0], 1), "median": round(v[len(v) // 2], 1), "max": round(v[-1], 1)} if v else None
        cap = self._read("power1_cap")
        return {"available": True, "source": "sysfs hwmon of %s, sampled every %.0f ms on a host thread inside the timed region" % (self.bdf, 1e3 * self.period_s),
                "sclk_mhz": stats([s[0] for s in self.samples], 1e-6), "socket_power_w": stats([s[1] for s in self.samples], 1e-6),
                "power_cap_w": cap * 1e-6 if cap else None}


PMC_TRAFFIC_FILE = "pmc_traffic_bench_bf16_b256.json"   # profiles/<latest tag>_pmc_traffic_bench_bf16_b256.json


def pmc_traffic(dtype, per_gpu_batch):
    """HBM-side bytes per launch of the dominant kernel family, from the committed rocprofv3 PMC passes of this same
    command (tools/collect_profiles.sh -> tools/pmc_traffic.py -> profiles/*.json; FETCH_SIZE/WRITE_SIZE cannot be read from
    inside the process, and a --pmc pass is its own run).  Only quoted for the configuration AND the kernel sources it was
    measured on: the file records the digest of csrc/ at measurement time; if the sources have changed since, traffic is null
    and the note says so.  -> (bytes per launch | None, note)"""
    if dtype != "bf16" or per_gpu_batch != 256:
        return None, "no PMC pass for this configuration"
    return pmc_traffic_file(PMC_TRAFFIC_FILE)


def pmc_traffic_file(suffix):
    """The newest profiles/*<suffix> whose csrc digest equals the current sources' -> (bytes per launch | None, note)."""
    pdir = os.path.join(ROOT, "profiles")
    files = sorted(f for f in os.listdir(pdir) if f.endswith(suffix)) if os.path.isdir(pdir) else []
    for f in reversed(files):
        try:
            d = json.load(open(os.path.join(pdir, f)))
        except Exception:
            continue
        if d.get("csrc_digest") == csrc_digest():
            return float(d["traffic_bytes_per_launch"]), "profiles/%s (csrc digest %s matches)" % (f, d["csrc_digest"])
    return None, "stale: no profiles/*_%s was taken on the current kernel sources (digest %s)" % (suffix, csrc_digest())


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


PEAK_HBM_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_PCIE_GBS = 64.0      # PCIe Gen5 x16, one direction


def sub_reports(segs, n_steps, rec=None):
    """lrcn_profile_segment's accumulators -> roofline.sub: per HBM-bound segment of SURVEY 8(d) the achieved GB/s on its ALGORITHMIC bytes
    (28 B/param for update!, one read of the recurrent weight block per timestep, (T+1) B E elements for the embedding gather, its dual's
    rows in + dense gradient out, 1 B + one element per pixel value for the preprocessing pass, the crops' bytes for the upload), the
    bytes and milliseconds per training step, and the fraction of the peak that bounds it (HBM 8 TB/s; the upload: PCIe Gen5 x16)."""
    names = {"update": "adam", "rec_fwd": "recurrence_weight_stream_fwd", "rec_bwd": "recurrence_weight_stream_bwd", "embed_gather": "embed_gather",
             "embed_grad": "embed_scatter", "preprocess": "preprocess", "upload": "upload"}
    out = {"measured_on": "%d extra steps after the timed region with lrcn_profile(ctx, 2): HIP-event pairs on the launching stream around "
                          "each segment; bytes are algorithmic (SURVEY 8d), not PMC traffic" % n_steps}
    for k, (ms, n, by) in segs.items():
        if n == 0 or ms <= 0:
            continue
        peak = PEAK_PCIE_GBS if k == "upload" else PEAK_HBM_GBS
        gbs = by / ms / 1e6
        out[names[k]] = {"GB/s": round(gbs, 1), "frac_of_peak": round(gbs / peak, 4), "peak_GB/s": peak, "ms_per_step": round(ms / n_steps, 4),
                         "algorithmic_MB_per_step": round(by / n_steps / 1e6, 3), "brackets": int(n)}
        if rec and k in ("rec_fwd", "rec_bwd"):
            # SURVEY 8(d) lists the recurrence as a weight stream, but from 256 rows per GPU it is a chain of 2 (B x 4H x H) contractions on the
            # CUs the capped convolution grids leave free: what bounds it is THEIR matrix-pipe share and the launch chain, so that is stated too
            tf = rec["gflop_per_step"] / (ms / n_steps)   # GFLOP per ms = TFLOP/s
            out[names[k]].update({"bound_in_fact": "mfma on %d free CUs + launch chain (DESIGN section 4), not hbm" % rec["free_cus"],
                                  "TFLOP/s": round(tf, 1), "frac_of_free_cu_mfma_peak": round(tf / (PEAK_BF16_TFLOPS * rec["free_cus"] / 256.0), 4)})
    return out


def parity_spot_check(ctx, L, sample, batch_imgs, dtype="bf16"):
    """The cpu_baseline sample through the HIP path of THIS context: the 4 crops replace the first rows of one of the
    benchmark's own image batches, so the VGG forward runs at the benchmark's batch size with the benchmark's kernels and
    routes (reported); loss + gradients of the 16 captions against the oracle's.  bf16: additionally against the bf16-EMULATING
    oracle (oracle/lrcn_oracle.h ORC_EMULATE_BF16, the checker build), elementwise -- the bound of tests/parity_util.py.  Not timed."""
    import numpy as np
    import torch
    from oracle import oracle as orc
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
    out = {"vgg_rel_max_err": vgg_err, "n_images": int(ref.shape[0]), "vgg_routes": L.debug_route(ctx, 1),
           "loss_rel_err": float(abs(val - sample["ref_loss"]) / abs(sample["ref_loss"])), "n_captions": int(sample["tokens"].shape[1]),
           "grad_cos_min": min(cos), "checker": "oracle (liblrcn_cpu_f32.so) on the cpu_baseline sample"}
    if dtype == "bf16":
        with orc.emulate_bf16():
            e_loss, e_g = orc.loss(m, sample["feats"], sample["tokens"], want_grad=True)
        worst, norm = 0.0, 0.0
        for n, g in zip(L.PARAM_NAMES, grads):
            if g.numel() == 0:
                continue
            a, b = L.from_jl(g).astype(np.float64), e_g.p[n].astype(np.float64)
            worst = max(worst, float((np.abs(a - b) / (5e-3 * np.abs(b) + 2.5e-3 * np.abs(b).max())).max()))
            norm = max(norm, float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300)))
        out["bf16_emulation"] = {"loss_rel_err": float(abs(val - e_loss) / abs(e_loss)), "grad_worst_over_tol": worst, "grad_rel_norm_max": norm,
                                 "tol": "per element |d| <= 5e-3 |ref| + 2.5e-3 max|ref| (worst_over_tol <= 1), per tensor ||d|| <= 3e-3 ||ref||",
                                 "checker": "bf16-emulating oracle (liblrcn_oracle.so, orc_set_emulate_bf16) on the same 16 captions"}
    return out


PRESETS = {  # BASELINE.json configs[k] -> flags (SURVEY.md 8d "Config -> shapes"); c4 is the headline and the default
    "c4": {},
    "c2": {"layers": 1, "dtype": "f32", "global_batch": 32, "hidden": 512, "vocab": 2540},
    "c3": {"global_batch": 128, "vocab": 7730, "T": 12},
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)    # >= 50 timed steps after >= 10 warm-up
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c4", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json configs[1..4]: c4 = the headline (default); c2 = fp32 VGG + LRCN-1f LSTM-512, batch 32; c3 = Flickr30k-shaped "
                         "bf16, batch 128; c5 = fp8 VGG + beam-5 caption generation (captions/s; runs tools/caption_bench.py's loop)")
    ap.add_argument("--global-batch", type=int, default=None)
    ap.add_argument("--dtype", default=None, choices=["bf16", "f32"])
    ap.add_argument("--hidden", type=int, default=None)
    ap.add_argument("--vocab", type=int, default=None)
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--pdrop", type=float, default=0.4)
    ap.add_argument("--layers", type=int, default=None, choices=[1, 2])  # 1 = LRCN-1f (this repo's definition of configs[1]'s "1-layer LSTM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vgg-chunk-images", type=int, default=int(os.environ.get("LRCN_VGG_CHUNK_IMAGES", "256")),
                    help="a rank whose own batch is smaller than this runs the VGG forward of its next m = chunk // batch steps as ONE forward (the "
                         "frozen extractor does not depend on the parameters) and feeds one feature block per step: 32 rows per GPU -> 4 steps per "
                         "forward.  Every image still passes through the VGG exactly once inside the timed region.  0 / 1 = one forward per step")
    ap.add_argument("--inputs", default="host", choices=["host", "hbm"],
                    help="where a step finds its crops: 'host' (default; BASELINE.md section 3) = pinned host memory, uploaded per step on the library's "
                         "copy stream one step ahead of the VGG forward that reads them (lrcn_upload_crops); 'hbm' = already resident in device memory")
    ap.add_argument("--emulate-world", type=int, default=1,
                    help="ONE process runs rank 0's shard of an N-rank job (global batch / N rows, the GLOBAL normaliser, no collective): the "
                         "per-rank step that bounds the N-GPU number, measurable on one GPU.  Labelled; NOT the headline metric.")
    ap.add_argument("--dp-backend", default=os.environ.get("LRCN_DP_BACKEND", "torch"), choices=["torch", "abi", "auto"],
                    help="N > 1: 'torch' = per-group all-reduces issued through torch.distributed's RCCL process group (default); 'abi' = RCCL "
                         "inside liblrcn_hip (lrcn_comm_init + lrcn_train_step_dp, one C call per step); 'auto' (self-launched jobs only) = "
                         "try 'abi' under a watchdog, rerun with 'torch' if it fails")
    ap.add_argument("--shard-adam", action="store_true", default=os.environ.get("LRCN_DP_SHARD_ADAM", "0")[:1] == "1",
                    help="N > 1, torch backend: reduce-scatter -> Adam on 1/N of the flat parameter buffer -> all-gather instead of all-reduce -> replicated "
                         "Adam (same wire bytes, 1/N of the update's HBM traffic per rank).  Opt-in for real N > 1 jobs: it has never met a second GPU.  "
                         "With --emulate-world N it is the DEFAULT since round 5 (the form an 8-GPU job should run once validated; 1.30 -> 1.25 ms at "
                         "32 rows); --replicated-update selects the replicated form there")
    ap.add_argument("--replicated-update", action="store_true", help="--emulate-world N: replicated update! (the whole Adam on this rank) instead of "
                    "the sharded one")
    ap.add_argument("--spinup-ms", type=float, default=150.0,
                    help="set-up, before the W warm-up steps: keep the GPU busy with VGG forwards of the benchmark's own crops for this long.  "
                         "The chip needs ~100 ms of load after idle before its clocks settle (round 3: steps 1..15 after idle run 7.5 -> 7.0 ms); "
                         "these are NOT training steps (no lossgradient, no update!), lie outside the timed region and are reported in the line "
                         "(config.setup_spinup).  Longer does not help (round 6, fresh single-command leases: 150 ms 6.95 6.96 6.98 7.20 7.29 ms, "
                         "2 s 6.75 6.99 7.00 7.02 7.09 ms -- what differs between leases is the box: profiles/r06b_*, r06d_*).  0 = off")
    ap.add_argument("--watchdog-s", type=float, default=float(os.environ.get("LRCN_BENCH_WATCHDOG_S", "900")),
                    help="self-launched jobs: kill the ranks and fail if they have not finished after this many seconds")
    a = ap.parse_args(argv)
    if a.config in PRESETS:
        for k, v in PRESETS[a.config].items():
            if getattr(a, k) is None:
                setattr(a, k, v)
    for k, v in {"global_batch": 256, "dtype": "bf16", "hidden": 1000, "vocab": 10640, "T": 11, "layers": 2}.items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    return a


def workload_name(a):
    H, Bg = a.hidden, a.global_batch
    if a.layers == 2 and a.dtype == "bf16" and Bg == 256 and H == 1000 and a.vocab == 10640:
        return "BASELINE.json configs[3] (C4), MS-COCO-shaped"
    if a.layers == 1 and a.dtype == "f32" and Bg == 32 and H == 512:
        return "BASELINE.json configs[1] (C2), Flickr8k-shaped"
    if a.layers == 2 and a.dtype == "bf16" and Bg == 128 and H == 1000 and a.vocab == 7730:
        return "BASELINE.json configs[2] (C3), Flickr30k-shaped"
    return "custom"


HEADLINE_METRIC = "training images/sec (VGG16+LSTM, COCO, batch 256) at 1/2/4/8 MI355X"


def metric_name(a, fake_multi=False):
    w = workload_name(a)
    if fake_multi:
        return "VALIDATION (%d ranks sharing ONE GPU over gloo) -- not a measurement; would be: %s" % (a.gpus, metric_name(a))
    if a.emulate_world > 1:
        return "EMULATED rank-0 step of a %d-rank job (%d of %d rows, global normaliser, no collective) -- not the headline" % (
            a.emulate_world, a.global_batch // a.emulate_world, a.global_batch)
    if w.startswith("BASELINE.json configs[3]"):
        return HEADLINE_METRIC
    return "training images/sec (VGG16+LSTM), %s" % w


def launch(a, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: this process NEVER touches the GPU (no torch import, no HIP call --
    a parent that had initialised the device could not be replaced or forked safely); it starts the N ranks as a CHILD job
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same flags>`),
    waits for it under a watchdog, relays rank 0's JSON line and exits with the job's code.  Every rank of that job is itself a GPU-free
    supervisor that walks the ladder of supervise() below -- the same thing happens when the driver starts the torchrun job itself."""
    from lrcn_amd import launch as lch   # imports neither torch nor the HIP library
    # The outer watchdog has to outlast the ladder it guards (a fixed 900 s around 2-3 rungs of up to rung_s each fired while the
    # fallback rung was still running, and the job ended with rc 124 and no number): every rung may take its whole budget plus the
    # supervisors' teardown and verdict exchange, so the default is derived from the rung list; an explicit --watchdog-s / LRCN_BENCH_WATCHDOG_S
    # larger than that is honoured, a smaller one is raised to it.
    n_rungs = len(lch.default_rungs(a.dp_backend))
    rung_s = float(os.environ.get("LRCN_BENCH_RUNG_S", "900"))
    stall_s = float(os.environ.get("LRCN_BENCH_STALL_S", "300"))
    need_s = n_rungs * (rung_s + 60.0) + stall_s + 120.0
    if a.watchdog_s < need_s and not os.environ.get("LRCN_BENCH_WATCHDOG_EXACT"):
        a.watchdog_s = need_s
    rc, out = lch.run_ranks(SCRIPT, list(argv), a.gpus, {"LRCN_BENCH_LAUNCHED": "1"}, a.watchdog_s)
    if rc == 124:
        print("bench.py: the %d-rank job did not finish within %.0f s and was stopped" % (a.gpus, a.watchdog_s), file=sys.stderr)
    lines = [ln for ln in (out or "").splitlines() if is_json_line(ln)]
    if rc == 0 and lines:
        line = json.loads(lines[-1])
        line.setdefault("rccl", {})["launcher"] = "self (bench.py spawned torch.distributed.run)"
        print(json.dumps(line), flush=True)
        return 0
    print("bench.py: the %d-rank job failed (rc %s)" % (a.gpus, rc), file=sys.stderr)
    return rc or 1


def is_json_line(ln):
    return ln.startswith("{") and '"metric"' in ln


def supervise(a, argv):
    """WORLD_SIZE > 1 and this is the process torchrun (or the driver) started for one rank: it stays GPU-free and runs the real rank as
    a fresh child per RUNG of the ladder (lrcn_amd/launch.py): "default" = the full N-rank pipeline (per-group [event -> all-reduce ->
    fused Adam] on a probed update stream, sparse exchange of the embedding gradient, several batches per VGG forward) -> "plain" = one
    all-reduce of the flat gradient buffer + one replicated Adam, no probes, one forward per step -> give up (rc != 0).  A rung is left
    when any rank's child exits non-zero, stops beating for LRCN_BENCH_STALL_S seconds (a hung collective), or fails its own step-1
    self-check (rank_main: sparse = dense embedding gradient, sum of the ranks' losses = rank 0's recomputation on all rows, parameter
    checksums equal on every rank).  The line names the rung that produced it: rccl.mode, rccl.rung, rccl.fallback_reason."""
    from lrcn_amd import launch as lch
    rungs = lch.default_rungs(a.dp_backend)
    only = os.environ.get("LRCN_BENCH_RUNGS_ONLY")   # e.g. "plain": development
    if only:
        rungs = [r for r in rungs if r[0] in only.split(",")] or rungs

    def annotate(line, name, k, n, reasons):
        d = json.loads(line)
        r = d.setdefault("rccl", {})
        r["mode"], r["rung"], r["fallback_reason"] = name, "%d of %d" % (k + 1, n), ("; ".join(reasons) or None)
        return json.dumps(d)

    return lch.supervise_rank(SCRIPT, list(argv), rungs, stall_s=float(os.environ.get("LRCN_BENCH_STALL_S", "300")),
                              rung_s=float(os.environ.get("LRCN_BENCH_RUNG_S", "900")), is_line=is_json_line, annotate=annotate)


PMC_MFMA_FILE = "pmc_mfma_busy_bench_bf16_b256.json"   # profiles/<tag>_pmc_mfma_busy_bench_bf16_b256.json (tools/pmc_mfma.py)


def pmc_mfma(dtype, per_gpu_batch):
    """Matrix-pipe utilisation of the convolution family and of the LSTM chain's contractions from the committed SQ-counter passes of this
    same command (tools/pmc_sq_passes.sh -> tools/pmc_mfma.py), quoted under the same rule as the traffic: only for the configuration and
    the kernel sources it was measured on.  -> (dict | None, note)"""
    if dtype != "bf16" or per_gpu_batch != 256:
        return None, "no SQ-counter pass for this configuration"
    pdir = os.path.join(ROOT, "profiles")
    files = sorted(f for f in os.listdir(pdir) if f.endswith(PMC_MFMA_FILE)) if os.path.isdir(pdir) else []
    for f in reversed(files):
        try:
            d = json.load(open(os.path.join(pdir, f)))
        except Exception:
            continue
        if d.get("csrc_digest") == csrc_digest():
            fam = d["families"]
            return ({"conv_family": fam["conv"]["mfma_busy"], "lstm_gemm_family": fam.get("lstm", {}).get("mfma_busy"),
                     "conv_waves_parked": fam["conv"]["waves_parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES)"],
                     "lstm_waves_parked": fam.get("lstm", {}).get("waves_parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES)"),
                     "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES), time-weighted over the family's launches; kernels serialised by --pmc"},
                    "profiles/%s (csrc digest %s matches)" % (f, d["csrc_digest"]))
    return None, "stale: no profiles/*_%s was taken on the current kernel sources (digest %s)" % (PMC_MFMA_FILE, csrc_digest())


def csrc_digest():
    """sha256 over the kernel sources: a committed PMC measurement is only quoted for the sources it was taken on."""
    import hashlib
    d = os.path.join(ROOT, "long-term-recurrent-convolutional-nn_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse_args(argv)
    if a.config == "c5":
        return caption_main(a, argv)
    world_env = os.environ.get("WORLD_SIZE")
    if a.gpus > 1 and world_env is None:
        return launch(a, argv)          # before any torch / HIP import
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.emulate_world > 1 and world > 1:
        raise SystemExit("--emulate-world is a one-process measurement")
    if world > 1 and not os.environ.get("LRCN_BENCH_CHILD") and os.environ.get("LRCN_BENCH_LADDER", "1")[:1] != "0":
        return supervise(a, argv)       # this process stays GPU-free; the rank runs as its child, rung by rung
    return rank_main(a, world, rank, local_rank)


def rank_main(a, world, rank, local_rank):
    """One rank of the job: everything that touches the GPU."""
    from lrcn_amd.launch import beat
    beat("started")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist
    import lrcn_amd
    from lrcn_amd import dp
    from lrcn_amd import lrcn as L

    # LRCN_BENCH_FAKE_MULTI=1 (validation on a ONE-GPU box; never a measurement): every rank uses device 0 and the process group is gloo
    # (RCCL refuses two ranks on one device) -- the N-rank control flow of this file and of dp.py (row shards, global normaliser, per-group
    # event -> all-reduce -> Adam pipeline, loss reduction, barrier timing) then runs on the REAL kernels, with only the transport swapped.
    fake_multi = world > 1 and os.environ.get("LRCN_BENCH_FAKE_MULTI", "0")[:1] == "1"
    if fake_multi:
        if world > 4:
            raise SystemExit("LRCN_BENCH_FAKE_MULTI: at most 4 ranks may share one GPU (the pool's limit is 6 processes per device)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if fake_multi:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    beat("process group up")

    if os.environ.get("LRCN_BENCH_MAIN_STREAM", "0")[:1] == "1":
        # development: run the whole job on a stream of its own instead of the device's null stream (whose work orders itself against the
        # first few other streams of a process: tools/stream_alias_probe.py)
        torch.cuda.set_stream(torch.cuda.Stream())
    dt = lrcn_amd.LRCN_BF16 if a.dtype == "bf16" else lrcn_amd.LRCN_F32
    E = H = a.hidden
    V, T, Bg = a.vocab, a.T, a.global_batch
    rows = dp.shard_rows(Bg, world * a.emulate_world, rank)
    B = rows.stop - rows.start

    plain_rung = os.environ.get("LRCN_BENCH_RUNG") == "plain"   # the ladder's last rung: command-line choices that would make it less plain do not apply
    # training batches per VGG forward: 8 at 32 rows per GPU, 4 at 64; one from 128 rows (measured: no gain there, dp.vgg_wg_cap_for)
    m_chunk = max(1, a.vgg_chunk_images // B) if (B <= 64 and not plain_rung) else 1
    m_chunk = 1 << (m_chunk.bit_length() - 1)   # a power of two of batches (tile counts stay round: 160 images measured slower than 128)
    while a.steps % m_chunk:   # ... that divides the timed step count: the timed region then holds exactly K batches of VGG forward and K LSTM steps
        m_chunk //= 2
    Bv = m_chunk * B
    ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=dt, vgg_dtype=dt, max_images=Bv, n_layers=a.layers)
    vgg_w = L.synthetic_vgg_weights(seed=1)
    L.vgg_load(ctx, *vgg_w)
    param = L.initweights(ctx, seed=42)          # identical on every rank (same seed)
    optim = L.initparams(param)
    backend = a.dp_backend if a.dp_backend != "auto" else "torch"   # 'auto' is resolved by the ladder (supervise): the rung sets LRCN_DP_BACKEND
    # one process: rank 0's side of the sharded update, collectives stubbed by copies (the default form of an emulated rank since round 5)
    emu_shard = a.emulate_world > 1 and not a.replicated_update and (bool(a.shard_adam) or os.environ.get("LRCN_DP_SHARD_ADAM", "1")[:1] != "0")
    trainer = dp.DataParallelTrainer(ctx, param, optim, Bg, world, rank, pdrop=a.pdrop, seed=7, backend=backend,
                                     shard_adam=(bool(a.shard_adam) and world > 1 and not plain_rung) or emu_shard, vgg_chunk=m_chunk, rows=B,
                                     emulate_shards=a.emulate_world if emu_shard else 0)

    # synthetic inputs: uint8 crops uniform seed 1234; Zipf(1.0) word ids >= 3, seed 7; one T per batch
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    n_sets = 2
    # one "chunk" = the crops of m_chunk consecutive steps of this rank (m_chunk = 1: one batch)
    imgs_dev = [torch.cat([torch.randint(0, 256, (Bg, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)[rows] for _ in range(m_chunk)]).contiguous()
                for _ in range(n_sets)]
    # --inputs host: the same crops in page-locked host memory; every step uploads the batch AFTER next on the copy stream
    imgs_all = [t.cpu().pin_memory() for t in imgs_dev] if a.inputs == "host" else imgs_dev
    rng = np.random.default_rng(7)
    pz = 1.0 / np.arange(1, V - 3 + 1)
    pz /= pz.sum()
    toks_glob = [(rng.choice(V - 3, size=(T, Bg), p=pz) + 3).astype(np.int32) for _ in range(n_sets)]   # every rank draws the GLOBAL batch
    toks_all = [torch.as_tensor(t[:, rows.start:rows.stop].copy()).cuda() for t in toks_glob]
    beat("trainer and inputs ready")

    # First contact (world > 1; LRCN_BENCH_SELFCHECK=1 forces it on one rank): the step-1 self-check, before anything is timed.  On a rung
    # that has a successor a violation ends this rank with an error -- the supervisors then move every rank to the next rung; on the
    # last rung it is reported in the line and the number stands with that caveat.
    selfcheck, strict = None, int(os.environ.get("LRCN_BENCH_RUNG_INDEX", "0")) + 1 < int(os.environ.get("LRCN_BENCH_RUNGS", "1"))

    def verdict(name, ok, detail):
        selfcheck.setdefault("violations", [])
        if not ok:
            selfcheck["violations"].append("%s: %s" % (name, detail))
            if strict:
                raise SystemExit("bench.py self-check failed on rung %r: %s: %s" % (os.environ.get("LRCN_BENCH_RUNG", "-"), name, detail))

    if (world > 1 or os.environ.get("LRCN_BENCH_SELFCHECK", "0")[:1] == "1") and a.emulate_world == 1:
        selfcheck = trainer.self_check(L.convnet_u8(ctx, imgs_dev[0][:B]), toks_glob[0])
        verdict("world", selfcheck["world_from_communicator"] == world == selfcheck["world_measured_by_allreduce"], selfcheck)
        verdict("loss over ranks vs rank 0 on all rows", selfcheck["loss_rel_diff"] <= 1e-5, selfcheck["loss_rel_diff"])
        sp = selfcheck["sparse_vs_dense_embed_grad_rel"]
        verdict("sparse vs dense embedding gradient", sp is None or sp <= 1e-4, sp)
        verdict("parameters identical before step 1", selfcheck["params_identical_before_step_1"], "checksums differ")
        beat("self-check passed")

    step_ev = []
    chunk_i = [0]   # index of the chunk whose features are being consumed
    host_t = [] if os.environ.get("LRCN_BENCH_HOST_TIMES") else None   # development: when the host ENTERED each step (issue-side timeline)

    def run(nsteps, events=False):
        # steady-state pipeline: EVERY step (the last one too) issues the VGG forward of the batch after it, so a run of K steps
        # holds exactly K VGG forwards + K LSTM steps; the features a run's first step consumes were produced by the previous
        # run's last step (the warm-up's, for the timed region), or in order if there is none.
        for _ in range(nsteps):
            k = trainer.step_no  # global step index: batch k uses image/token set k mod n_sets, also across warm-up -> timed region
            if host_t is not None:
                host_t.append(time.perf_counter())
            ci = chunk_i[0]
            if trainer.step(imgs_all[ci % n_sets][:B], toks_all[k % n_sets], next_img_u8=imgs_all[(ci + 1) % n_sets],
                            prefetch_img_u8=imgs_all[(ci + 2) % n_sets] if a.inputs == "host" else None):
                chunk_i[0] = ci + 1
            if events:  # one event per step on the main stream (no host sync): per-step intervals -> the median SURVEY 8(d) asks for
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                step_ev.append(e)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import gc  # a full pass of Python's cyclic GC scans every object torch created (tens of ms): none inside the timed region,
    gc.collect()  # and none between the warm-up and the timed region either (an idle gap there restarts the clock ramp)
    gc.freeze()
    # set-up: device spin-up (see --spinup-ms).  VGG forwards only: no training state changes, the W warm-up steps stay W.
    spun = 0
    if a.spinup_ms > 0:
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < a.spinup_ms:
            L.convnet_u8(ctx, imgs_dev[0])
            torch.cuda.synchronize()
            spun += 1
    # (constructed here, not between the warm-up's barrier and the timed region: finding the device's hwmon node takes milliseconds of host
    # time, and every millisecond the GPU idles there is paid by the first timed step -- it ran 7.4-7.5 ms against 6.8-6.9 for the rest)
    hw = HwSampler(local_rank) if rank == 0 and os.environ.get("LRCN_BENCH_HW_SAMPLER", "1")[:1] != "0" else None
    run(a.warmup)
    barrier()
    beat("warm-up done")
    if selfcheck is not None and a.warmup > 0:
        same, _ = trainer.check_replicas()
        selfcheck["params_identical_after_warmup"] = bool(same)
        verdict("parameters identical after %d warm-up step(s)" % a.warmup, same, "checksums differ")
        barrier()
    _lib = lrcn_amd._lib
    _lib.check(ctx._h, _lib.lib().lrcn_profile(ctx._h, 1))
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    step_ev.append(e0)
    if hw:
        hw.start()
    t0 = time.perf_counter()
    run(a.steps, events=True)
    barrier()
    dt_s = time.perf_counter() - t0
    hw_held = hw.stop() if hw else None
    beat("timed region done")
    if selfcheck is not None:
        same, _ = trainer.check_replicas()
        selfcheck["params_identical_after_last_step"] = bool(same)
        verdict("parameters identical after the last step", same, "checksums differ")
    step_ms_seq = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(len(step_ev) - 1)]
    step_ms = sorted(step_ms_seq)
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    import ctypes as C
    conv_ms, conv_n = C.c_double(), C.c_int64()
    _lib.check(ctx._h, _lib.lib().lrcn_profile_get(ctx._h, C.byref(conv_ms), C.byref(conv_n)))
    _lib.check(ctx._h, _lib.lib().lrcn_profile(ctx._h, 0))
    loss = trainer.loss_value()
    if getattr(trainer, "_tail_events", None):   # LRCN_DP_DEBUG_TAIL=1 (development)
        tails = [a_.elapsed_time(b_) for a_, b_ in trainer._tail_events[-a.steps:]]
        print("update chain past the end of the backward pass, ms per step: median %.3f min %.3f max %.3f  first16 %s" % (
            sorted(tails)[len(tails) // 2], min(tails), max(tails), [round(t, 3) for t in tails[:16]]), file=sys.stderr)
    # SURVEY 8(d)'s HBM-bound sub-reports, from a SEPARATE untimed pass of the same pipeline (the event pairs of lrcn_profile level 2 sit
    # between dependent launches and would cost the timed step a few microseconds each): rank 0 of a one-rank job only
    sub = None
    if world == 1 and os.environ.get("LRCN_BENCH_SUBREPORTS", "1")[:1] != "0":
        n_sub = 2 * max(m_chunk, 4)
        L.profile(ctx, 2)
        run(n_sub)
        torch.cuda.synchronize()
        free = 256 - (trainer.vgg_cap if getattr(trainer, "vgg_cap", 0) else 224)
        rec = {"gflop_per_step": a.layers * T * 2.0 * B * 4 * H * H / 1e9, "free_cus": free} if (a.dtype == "bf16" and B >= 256 and free > 0) else None
        sub = sub_reports(L.profile_segments(ctx), n_sub, rec)
        L.profile(ctx, 0)
        torch.cuda.synchronize()
    tt = torch.tensor([dt_s], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt_s = float(tt.item())

    if rank == 0:
        ms_step = 1e3 * dt_s / a.steps
        value = (B if a.emulate_world > 1 else Bg) * a.steps / dt_s
        # dominant kernel family: the 12 convolution launches conv1_2..conv5_3 (conv64f_kernel + conv64_kernel + 10 x gemm8p_kernel<CONV3>)
        # bf16: conv1_1 runs inside conv1_2's launch (conv64f.hip), so its FLOPs belong to the 12 timed launches
        fused11 = a.dtype == "bf16" and os.environ.get("LRCN_FUSE11", "1")[:1] != "0" and os.environ.get("LRCN_CONV64", "1")[:1] != "0"
        flops_per_launch = (VGG_CONV_GFLOP_PER_IMAGE - (0.0 if fused11 else CONV11_GFLOP_PER_IMAGE)) * 1e9 * Bv / 12.0
        avg_launch_s = conv_ms.value * 1e-3 / max(conv_n.value, 1)
        achieved = flops_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        traffic, traffic_note = pmc_traffic(a.dtype, B)
        mfma_busy, mfma_note = pmc_mfma(a.dtype, B)
        out = {
            "metric": metric_name(a, fake_multi),
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "ms_per_step_median": median_ms,
            "ms_per_step_first8": [round(x, 3) for x in step_ms_seq[:8]], "ms_per_step_last": round(step_ms_seq[-1], 3),
            **({"ms_per_step_all": [round(x, 3) for x in step_ms_seq]} if os.environ.get("LRCN_BENCH_ALL_STEPS") else {}),
            **({"host_issue_ms_all": [round(1e3 * (host_t[i + 1] - host_t[i]), 3) for i in range(len(host_t) - 1)]} if host_t else {}),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "%s: VGG-16 -> fc7 fwd + %s LSTM "
                                   "E=H=%d V=%d T=%d fwd/bwd + Adam, global batch %d, dp%d, dropout %.1f; synthetic uint8 "
                                   "224x224 crops, He-normal VGG weights"
                                   % (workload_name(a), "LRCN-2f (2-layer)" if a.layers == 2 else "LRCN-1f (1-layer)", H, V, T, Bg,
                                      world, a.pdrop),
                       "global_batch": Bg, "per_gpu_batch": B, "seq_len": T + 1, "parallelism": "dp%d" % world,
                       "vgg_forward": ("one forward per step (%d images)" % B) if m_chunk == 1 else
                                      ("one forward per %d steps (%d images: the crops of the next %d batches of this rank; one feature block per step)" % (m_chunk, Bv, m_chunk)),
                       "last_loss": loss,
                       "inputs": ("pinned host, H2D per step on a copy stream (%.1f MB per step, uploaded one step ahead of the forward that reads it)"
                                  % (B * 224 * 224 * 3 / 1e6)) if a.inputs == "host" else "resident in HBM",
                       "setup_spinup": "%d untimed VGG forwards (%.0f ms) before the warm-up steps; not training steps" % (spun, a.spinup_ms)},
            "rccl": {"world": (dist.get_world_size() if world > 1 else 1), "world_argv": a.gpus, "backend": ("none (one rank: no collective)" if world == 1 else
                                                 ("gloo on ONE shared GPU (LRCN_BENCH_FAKE_MULTI: validation, not a measurement)" if fake_multi else trainer.backend)),
                     "update": ("EMULATED sharded update (Adam on 1/%d of every gradient group; reduce-scatter / all-gather stubbed by device copies of the "
                                "bytes a rank receives; the other shards' parameters are not updated)" % a.emulate_world) if emu_shard else
                               ("sharded (reduce-scatter -> Adam on 1/N -> all-gather)" if trainer.shard else "replicated (all-reduce -> Adam)"),
                     "launched_by": "bench.py" if os.environ.get("LRCN_BENCH_LAUNCHED") else ("torch.distributed.run" if world > 1 else "direct"),
                     "selfcheck": selfcheck,
                     "pipeline": trainer.describe() if hasattr(trainer, "describe") else None},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_source": traffic_note, "mfma_busy": mfma_busy, "mfma_busy_source": mfma_note,
                         "kernel": "conv64f_kernel (conv1_1+conv1_2 fused) + conv64_kernel (conv2_1) + gemm8p_kernel<*,CONV3,*> (conv2_2..conv5_3): 12 launches/step"
                                   if a.dtype == "bf16" else "gemm_glds_kernel<float,*,CONV3,*> v_mfma_f32_32x32x2_f32 (conv1_2..conv5_3)",
                         "avg_launch_ms": 1e3 * avg_launch_s, "flops_per_launch": flops_per_launch, "sub": sub},
            "hw_held_in_timed_region": hw_held,
        }
        if a.emulate_world > 1:
            out["config"]["emulate_world"] = a.emulate_world
            out["config"]["note"] = ("value = this ONE rank's rows per second; an N-rank job adds the gradient exchange (DESIGN 5) and "
                                     "finishes when its slowest rank does")
        if world == 1 and not a.no_cpu_baseline:
            host_w = ([L.from_jl(w) for w in vgg_w[0]], [b.cpu().numpy() for b in vgg_w[1]],
                      (L.from_jl(vgg_w[2][0]), vgg_w[2][1].cpu().numpy()), (L.from_jl(vgg_w[3][0]), vgg_w[3][1].cpu().numpy()))
            out["cpu_baseline"], sample = cpu_baseline(host_w, E, H, V, T, np.random.default_rng(3), a.layers)
            out["parity"] = parity_spot_check(ctx, L, sample, imgs_dev[0], a.dtype)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    beat("finished")   # measured, reduced, printed: a teardown that hangs from here on no longer costs the rung its number (launch.supervise_rank)
    if world > 1:
        dist.destroy_process_group()
    return 0


def caption_main(a, argv):
    """--config c5: BASELINE.json configs[4], captions/sec of fp8 VGG + bf16 LSTM beam-5 generation.  The loop lives in
    tools/caption_bench.py (replicas only: no collective on the data path); N > 1 is self-launched the same way as training."""
    import runpy
    if a.gpus > 1 and os.environ.get("WORLD_SIZE") is None:
        return launch(a, argv)
    # 2048 images per pass and per batched beam search (10 240 hypotheses) since round 6: 35.9 k captions/s against 33.8 k at 1024 on the same
    # box -- the gate GEMMs' 1 280 tiles are five whole rounds on 256 CUs (640: two and a half), the per-step launches amortise over twice the rows
    n_img = os.environ.get("LRCN_C5_IMAGES", "2048")
    sys.argv = [os.path.join(ROOT, "tools", "caption_bench.py"), "--images", n_img, "--chunk", n_img, "--iters", str(max(2, a.steps // 10))]
    if a.no_cpu_baseline or os.environ.get("LRCN_C5_LIGHT"):   # LRCN_C5_LIGHT=1: counter passes under rocprofv3 (no CPU leg, no fixture)
        sys.argv += ["--no-cpu-baseline", "--no-fixture"]
    runpy.run_path(sys.argv[0], run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
