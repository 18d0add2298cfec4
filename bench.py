#!/usr/bin/env python3
"""Benchmark of the SELD training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): training clips/sec on synthetic 4-channel FOA, 60 s @ 24 kHz clips.
Workload (BASELINE.json configs[1]): CRNN (model_crnn.py), FOA 4-ch, batch 32 windows, bf16,
HIP feature kernels + PyTorch-ROCm forward/backward.

One STEP = one pass of the whole hot path over one batch of CLIPS_PER_STEP = 32 synthetic clips whose
PCM and metadata are already resident in HBM when the timed region starts:
  1. fused log-mel kernel over [32, 4, 1 440 000] PCM           (dataset.py:27-58)
  2. label rasteriser for the 32 clips                            (dataset.py:60-119)
  3. crop to 3000 aligned frames per clip, concatenate, cut 1920 windows of 250 frames / hop 50
     (dataset.py:243-317: 60 windows per clip, windows straddle clip boundaries as upstream)
  4. 60 optimiser iterations of 32 windows: window gather -> CRNN forward (bf16 autocast) ->
     fused softmax-MSE loss -> backward -> Adam; after three eager iterations the iteration is a replayed HIP graph
     (seld_graph.py; trainer.train_model's own path).  N > 1: the backward pass is cut into stages, one graph each, and
     the gradient buckets a stage completes are all-reduced over RCCL asynchronously while the next stages replay.
clips/s = N * 32 * K / (max-over-ranks wall time); weak scaling (every rank has its own 32 clips).

The JSON line also carries
  roofline      : the hand-written log-mel kernel against the HBM roofline -- algorithmic bytes
                  (26.11 MB per 60 s clip: 23.04 MB fp32 PCM in + 3.07 MB log-mel out) x 32
                  clips per launch / average launch duration measured with HIP events on the launch stream;
                  `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/);
  roofline_model: the optimiser iterations against the dense bf16 MFMA peak (39.4 GFLOP/window for the CRNN);
  kernels       : every other hand-written kernel (conv tail, loss, window gather, BiGRU recurrence) at the
                  workload's shapes against the roofline that bounds it, timed after the timed region;
  allreduce_overlap : N = 1: the data-parallel step's stage graphs replayed on this GPU without collectives, HIP events
                  between the replays (duration of every stage, bytes of every gradient bucket, the window that hides its
                  all-reduce); N > 1: true / false for the run, with `gradient_buckets`;
  other_workloads : 2 warm-up + 3 timed steps each of the other BASELINE.json workloads (configs[2], configs[4] on one GPU,
                  configs[3]'s per-GPU shard) in the same process, after the timed region -- never part of `value`;
  cpu_baseline  : the oracle (torch CPU restatement of the reference path) timed on this host's
                  cores on a bounded sample -- a reported, non-target number.

Launching N > 1 ranks.  Under torchrun (RANK / WORLD_SIZE set, the driver's form) this process IS one rank.  Started
plainly as ``python bench.py --gpus N`` it has not touched the GPU yet (``torch.cuda.device_count()`` does not initialise
it): it starts ``python -m torch.distributed.run --nproc-per-node N ... bench.py`` as a CHILD process, forwards the
child's output and exits with its code -- one rank per GPU over RCCL.  It refuses (exit code 2, nothing launched) when
fewer than N GPUs are visible, unless ``--rehearse-gloo`` (or SELD_DIST_BACKEND=gloo) asks for the rehearsal mode in
which the N ranks share the visible GPU(s) over gloo: that exercises DDP, the sharded device feed and the replica
agreement check on a one-GPU box and is never a measurement (the line says ``"rehearsal": true``).

Other workloads (parity / coverage cases of BASELINE.json, not the headline): --model conformer | resnet_conformer;
--features logmel_gcc --channels 8 (configs[3]'s per-GPU shard: 8-ch MIC array, 36 input channels).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
PKG = ROOT / "sound-event-localization-detection_amd"
for p in (str(ROOT), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CLIPS_PER_STEP = 32
CHANNELS = 4
CLIP_SAMPLES = 1_440_000          # 60 s @ 24 kHz
FRAMES_PER_CLIP = 3000            # min(3001 STFT frames, 3000 label frames), dataset.py:243-249
WINDOW, HOP = 250, 50
BATCH = 32
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0    # dense bf16
BYTES_PER_CLIP = CHANNELS * CLIP_SAMPLES * 4 + CHANNELS * 64 * (1 + CLIP_SAMPLES // 480) * 4   # 26.11 MB
# fwd+bwd GFLOP per 250-frame window, hook-counted on the reference modules
GFLOP_PER_WINDOW = {"crnn": 39.4, "conformer": 37.9, "resnet_conformer": 165.5}
# which BASELINE.json entry a --model run corresponds to (the headline metric is quoted on configs[1])
BASELINE_CONFIG = {"crnn": "BASELINE configs[1]", "conformer": "BASELINE configs[2]",
                   "resnet_conformer": "BASELINE configs[4]'s model, class-MSE loss, one GPU"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-rooflines", action="store_true",
                    help="skip the post-timing micro-benchmarks of the hand-written kernels (profiled runs: the kernel "
                         "trace then holds the steps only)")
    ap.add_argument("--model", default="crnn", choices=["crnn", "conformer", "resnet_conformer"])
    ap.add_argument("--fp32", action="store_true", help="disable bf16 autocast (parity runs)")
    ap.add_argument("--features", default="logmel", choices=["logmel", "logmel_iv", "logmel_gcc"],
                    help="feature set (logmel = the reference; logmel_gcc with --channels 8 = BASELINE configs[3]'s "
                         "per-GPU shard: 8-ch MIC, 8 log-mel + 28 GCC-PHAT input channels)")
    ap.add_argument("--channels", type=int, default=4, help="audio channels of the synthetic clips")
    ap.add_argument("--loss", default="class", choices=["class", "three_term"],
                    help="class = the modular loss.py (class MSE); three_term = MSE + AIUR + CL on probabilities "
                         "(smrl_seld_gaussian.py:946-1072, BASELINE configs[4])")
    ap.add_argument("--gaussian-augment", action="store_true",
                    help="labels by the Gaussian box rasteriser (smrl_seld_gaussian.py:397-534, BASELINE configs[4])")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short runs of the other BASELINE workloads after the timed region (`other_workloads`)")
    ap.add_argument("--rehearsal-clips", type=int, default=None,
                    help="clips per step of a --rehearse-gloo run (default 32 like a measurement; the GPU tests use 8: the "
                         "gradients of every iteration travel through the host there)")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N ranks share the visible GPU(s) over gloo (DDP rehearsal on a one-GPU box; not a measurement)")
    return ap.parse_args()


def launch_ranks(args) -> int:
    """``python bench.py --gpus N`` without torchrun: become the launcher.  Nothing here touches the GPU (a process
    that has initialised HIP must not be replaced or forked into ranks), the ranks are children of a child
    ``torch.distributed.run`` and this process only relays their output and exit code."""
    visible = torch.cuda.device_count()                 # does not initialise the GPU on this image
    rehearsal = args.rehearse_gloo or os.environ.get("SELD_DIST_BACKEND", "nccl") == "gloo"
    if visible < 1:
        print("bench.py: no ROCm GPU visible (the HIP extension has no CPU fallback)", file=sys.stderr)
        return 2
    if visible < args.gpus and not rehearsal:
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, {visible} visible.  One rank per GPU over RCCL is the "
              f"only measured mode; pass --rehearse-gloo to rehearse the {args.gpus}-rank path on {visible} GPU(s) "
              f"(DDP over gloo, ranks share the device; never a measurement).", file=sys.stderr)
        return 2
    if rehearsal and args.gpus > 6:
        print("bench.py: a rehearsal keeps at most 6 ranks on one GPU", file=sys.stderr)
        return 2
    with socket.socket() as sock:                       # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env["SELD_DIST_BACKEND"] = "gloo" if rehearsal else "nccl"
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    # stdout carries the ONE JSON line of rank 0 and nothing else: whatever the ranks' libraries print there (gloo's
    # connection banner in the rehearsal mode) is passed on to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        (sys.stdout if line.startswith("{") else sys.stderr).flush()
    return proc.wait()


def synth_metadata(clip_idx: int, meta_frames: int = 600) -> np.ndarray:
    """Seeded STARSS22-style rows (meta_frame, class, source, azimuth, elevation): 0-3 simultaneous
    sources per 100 ms frame, occasional second class in the same cell, two rows past the end."""
    rng = np.random.default_rng(1234 + clip_idx)
    rows = []
    for t in range(meta_frames):
        for s in range(int(rng.integers(0, 4))):
            az, el = int(rng.integers(-180, 181)), int(rng.integers(-90, 91))
            rows.append((t, int(rng.integers(0, 13)), s, az, el))
            if rng.random() < 0.05:
                rows.append((t, int(rng.integers(0, 13)), s + 1, az, el))
    rows += [(meta_frames, 1, 0, 10, 10), (meta_frames + 3, 1, 0, 10, 10)]
    return np.asarray(rows, dtype=np.int64).reshape(-1, 5)


def synth_clip_batch(seed: int, device, channels=CHANNELS):
    """Seeded synthetic inputs: PCM ~ N(0, 0.1^2) clipped to [-1, 1); STARSS22-style metadata."""
    g = torch.Generator(device=device).manual_seed(1234 + seed)
    pcm = (torch.randn(CLIPS_PER_STEP, channels, CLIP_SAMPLES, generator=g, device=device) * 0.1)
    pcm.clamp_(-1.0, 1.0 - 2.0 ** -15)
    events = []
    for i in range(CLIPS_PER_STEP):
        rows = synth_metadata(seed * CLIPS_PER_STEP + i, meta_frames=600)
        events.append(torch.from_numpy(rows).to(dtype=torch.int32, device=device))
    return pcm, events


class HotPath:
    """Holds the device-resident inputs and runs one step."""

    def __init__(self, args, device, rank, world, backend="nccl"):
        import seld_native
        import trainer
        self.native, self.trainer, self.device = seld_native, trainer, device
        cfg = trainer.config
        cfg.MODEL_TYPE = args.model
        cfg.BATCH_SIZE = BATCH
        cfg.AMP_DTYPE = "fp32" if args.fp32 else "bf16"
        # identical initial weights on every rank; the gloo rehearsal mode seeds the ranks DIFFERENTLY on purpose so
        # that it exercises what makes them agree (DDP's parameter broadcast + the master-weight broadcast)
        torch.manual_seed(rank if backend != "nccl" else 0)
        self.features, self.channels = args.features, args.channels
        extra = {"logmel": 0, "logmel_iv": 3, "logmel_gcc": args.channels * (args.channels - 1) // 2}[args.features]
        self.feature_channels = args.channels + extra
        model = trainer.prepare_model_for_device(trainer.build_model((18, 36), n_channels=self.feature_channels), device)
        trainer.enable_master_weights(model, device)
        # exactly what trainer.train_model sets up: captured iterations with their own flat gradient exchange
        # (seld_graph.py), or -- SELD_GRAPH_STEP=0 / Config.GRAPH_STEP off -- the eager loop under DistributedDataParallel
        graphed = trainer.graph_step_enabled(device, world)
        self.model = trainer.broadcast_replica_state(model, world) if graphed else trainer.wrap_ddp(model, device, world)
        weights = torch.ones(14, device=device)
        weights[13] = 0.05
        self.three_term = getattr(args, "loss", "class") == "three_term"
        self.criterion = trainer.SMRSELDLoss(loss_type="mse", w_class=1.0, w_aiur=1.0, w_cl=1.0, grid_size=(18, 36),
                                             class_weights=weights, three_term=self.three_term)
        self.optimizer = trainer.make_optimizer(self.model, cfg.LEARNING_RATE, device, capturable=graphed)
        self.stepper = trainer.make_stepper(self.model, self.criterion, self.optimizer, device, world)
        self.pcm, self.events = synth_clip_batch(rank, device, args.channels)
        self.centres = None
        if getattr(args, "gaussian_augment", False):
            # one normal draw per (class, source) of every clip, seeded (the reference draws them when it builds the
            # dataset, smrl_seld_gaussian.py:426-437); the box rasteriser itself runs in every step
            rng = np.random.default_rng(4321 + rank)
            self.centres = [torch.from_numpy(seld_native.gaussian_source_noise(ev.cpu().numpy(), rng=rng)).to(device)
                            for ev in self.events]
        total = CLIPS_PER_STEP * FRAMES_PER_CLIP
        self.starts = torch.arange(0, total, HOP, dtype=torch.int64, device=device)          # 480 windows
        self.spec_full = torch.empty(CLIPS_PER_STEP, 1 + CLIP_SAMPLES // 480, self.feature_channels, 64, device=device)
        self.mask_tm = torch.empty(total, 648, dtype=torch.uint16, device=device)
        self.feat_events = []          # (start, stop) HIP events around every log-mel launch
        self.model_events = []         # around the 15 optimiser iterations
        self.model.train()

    def step(self, timed: bool):
        nat, dev = self.native, self.device
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self.features == "logmel":
            nat.logmel(self.pcm, layout="tcf", out=self.spec_full)
        else:                               # log-mel + spectra in one pass, then intensity vectors / GCC-PHAT (csrc/spatial.hip)
            self.spec_full = nat.spatial_features(self.pcm, self.features)
        if timed:
            e1.record()
            self.feat_events.append((e0, e1))
        for i, ev in enumerate(self.events):
            if self.centres is not None:
                self.mask_tm[i * FRAMES_PER_CLIP:(i + 1) * FRAMES_PER_CLIP] = \
                    nat.rasterise_labels_gaussian(ev, self.centres[i], FRAMES_PER_CLIP, device=dev)
            else:
                nat.rasterise_labels(ev, FRAMES_PER_CLIP, device=dev,
                                     out=self.mask_tm[i * FRAMES_PER_CLIP:(i + 1) * FRAMES_PER_CLIP])
        spec_tm = self.spec_full[:, :FRAMES_PER_CLIP].reshape(-1, self.feature_channels, 64)   # crop + concatenate
        if timed:
            m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            m0.record()
        last = None
        for lo in range(0, self.starts.numel(), BATCH):
            starts = self.starts[lo:lo + BATCH]
            # straight into the captured step's input buffers once it has them (what trainer.DeviceFeed does)
            static = getattr(self.stepper, "static_inputs", None)
            bufs = static((len(starts), WINDOW, self.feature_channels, 64), torch.float32,
                          (len(starts), WINDOW, 648), torch.uint16) if static is not None else None
            spec = nat.gather_windows(spec_tm, starts, WINDOW, out=bufs[0] if bufs else None)
            mask = nat.gather_windows(self.mask_tm, starts, WINDOW, out=bufs[1] if bufs else None)
            last, _ = self.stepper(spec, mask)
            if os.environ.get("SELD_BENCH_TRACE"):          # developer diagnostic (synchronises: never a measurement)
                bad = [n for n, p in self.model.named_parameters() if not torch.isfinite(p).all()]
                self.loss_trace = getattr(self, "loss_trace", []) + [(round(float(last.item()), 5), len(bad), bad[:4])]
        if timed:
            m1.record()
            self.model_events.append((m0, m1))
        return last


def measured_traffic_per_clip():
    """HBM bytes per clip of the log-mel main kernel from the committed rocprofv3 --pmc passes
    (profiles/*_pmc_logmel_summary.json: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, per clip)."""
    best = None
    for path in sorted((ROOT / "profiles").glob("*_pmc_logmel_summary.json")):
        try:
            best = (path.name, float(json.loads(path.read_text())["traffic_bytes_per_clip"]))
        except (ValueError, KeyError):
            continue
    return best


def kernel_rooflines(device):
    """Every other hand-written kernel of the path against the roofline that bounds it, at the workload's shapes
    (batch 32 windows), timed with HIP events on the launch stream AFTER the timed region (not part of `value`).
    HBM kernels: algorithmic bytes per launch / duration vs the 8 TB/s peak.  The GRU recurrence runs on 8 CUs
    (one per direction and 4-sequence tile): its 2*B*T*2*H*3H flop against those CUs' dense bf16 MFMA peak."""
    import seld_native as nat
    out = []

    def timeit(fn, reps=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # hold the stream (3 x 1 ms, seld_stream_delay) while the host enqueues the repetitions: the events then
        # bracket back-to-back kernels, not the ~40 us of Python / ctypes time per call that exceeds the small ones
        for _ in range(3):
            nat.stream_delay(device, 1_000_000)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    def hbm(name, nbytes, seconds):
        gbs = nbytes / seconds / 1e9
        out.append({"kernel": name, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "bytes_per_launch": nbytes, "avg_launch_ms": seconds * 1e3})

    # conv-block tail (csrc/convtail.hip): 5 B (forward) / 8 B (backward) per pre-pool bf16 element
    for c, f in ((64, 64), (512, 8)):
        x = torch.randn(BATCH, c, WINDOW, f, device=device).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w, b = torch.ones(c, device=device), torch.zeros(c, device=device)
        rm, rv = torch.zeros(c, device=device), torch.ones(c, device=device)
        y, mi, ss = nat.conv_tail_forward(x, w, b, rm, rv, 0.1, 1e-5, True, 2)
        go = torch.randn_like(y)
        hbm(f"seld::tail_stats+apply (BN+ReLU+MaxPool forward, C={c})", 5 * x.numel(),
            timeit(lambda: nat.conv_tail_forward(x, w, b, rm, rv, 0.1, 1e-5, True, 2)))
        hbm(f"seld::tail_bwd_reduce+apply (BN+ReLU+MaxPool backward, C={c})", 8 * x.numel(),
            timeit(lambda: nat.conv_tail_backward(x, go, mi, ss, 2)))
    # fused softmax-MSE value + gradient (csrc/loss.hip): bf16 logits read + gradient written + mask read
    logits = torch.randn(BATCH, WINDOW, 648, 14, device=device).to(torch.bfloat16)
    mask = torch.zeros(BATCH, WINDOW, 648, dtype=torch.uint16, device=device)
    hbm("seld::softmax_mse_kernel (loss + gradient)", 2 * logits.numel() * 2 + mask.numel() * 2,
        timeit(lambda: nat.softmax_mse(logits, mask, grad_scale=1.0)))
    # head LayerNorm(512) -> ReLU (csrc/layernorm.hip): bf16 x read + y written; backward x, dy read + dx written
    xl = torch.randn(BATCH * WINDOW, 512, device=device).to(torch.bfloat16)
    lw, lb = torch.ones(512, device=device), torch.zeros(512, device=device)
    yl, st = nat.layernorm_forward(xl, lw, lb, 1e-5, True)
    gl = torch.randn_like(xl)
    hbm("seld::layernorm_forward_kernel (LayerNorm+ReLU, D=512)", 2 * xl.numel() * 2,
        timeit(lambda: nat.layernorm_forward(xl, lw, lb, 1e-5, True)))
    hbm("seld::layernorm_backward_kernel (+ column-sum reduction, D=512)", 3 * xl.numel() * 2,
        timeit(lambda: nat.layernorm_backward(xl, gl, lw, lb, st, True)))
    # label mask expansion and window gather (csrc/labels.hip)
    starts = torch.arange(0, BATCH * HOP, HOP, dtype=torch.int64, device=device)
    tm = torch.zeros(BATCH * HOP + WINDOW, 648, dtype=torch.uint16, device=device)
    hbm("seld::gather_rows_kernel (label mask windows)", 2 * BATCH * WINDOW * 648 * 2,
        timeit(lambda: nat.gather_windows(tm, starts, WINDOW)))
    # log-mel from int16 PCM (csrc/logmel.hip, the <int16_t> instantiation: 14.59 MB per 60 s clip) and the
    # 8-channel MIC-array feature pass (log-mel + spectra in one pass, then csrc/spatial.hip gcc_phat_kernel):
    # 8 x 5.76 MB PCM in + 36 x 64 x 3001 x 4 B out per clip
    clips = 8
    pcm16 = (torch.randn(clips, CHANNELS, CLIP_SAMPLES, device=device) * 3276.8).clamp(-32768, 32767).to(torch.int16)
    lm_out = torch.empty(clips, 1 + CLIP_SAMPLES // 480, CHANNELS, 64, device=device)
    hbm(f"seld::logmel_main_kernel<int16_t> + edge ({clips} clips, int16 PCM)",
        clips * (CHANNELS * CLIP_SAMPLES * 2 + lm_out[0].numel() * 4),
        timeit(lambda: nat.logmel(pcm16, layout="tcf", out=lm_out), reps=5))
    del pcm16, lm_out
    pcm8 = torch.randn(clips, 8, CLIP_SAMPLES, device=device) * 0.1
    hbm(f"spatial_features logmel_gcc ({clips} clips x 8 ch: logmel_main_kernel<float, 2> (Q15 phasors) + gcc_q15_kernel<2>)",
        clips * (8 * CLIP_SAMPLES * 4 + 36 * 64 * (1 + CLIP_SAMPLES // 480) * 4),
        timeit(lambda: nat.spatial_features(pcm8, "logmel_gcc"), reps=3))
    del pcm8
    # the FOA feature set: log-mel and the mel-projected intensity vectors in one kernel (seld::logmel_iv_kernel: the four
    # channels of a clip in one workgroup, spectra in LDS only): 4 x 5.76 MB PCM in + 7 x 64 x 3001 x 4 B out per clip
    pcm4 = torch.randn(clips, 4, CLIP_SAMPLES, device=device) * 0.1
    hbm(f"spatial_features logmel_iv ({clips} clips x 4 ch FOA: logmel_iv_kernel<float>, one pass)",
        clips * (4 * CLIP_SAMPLES * 4 + 7 * 64 * (1 + CLIP_SAMPLES // 480) * 4),
        timeit(lambda: nat.spatial_features(pcm4, "logmel_iv"), reps=3))
    del pcm4
    # GRU recurrence (csrc/gru.hip)
    h = 256
    gi = (torch.randn(BATCH, WINDOW, 2, 3 * h, device=device) * 0.5).to(torch.bfloat16)
    w_hh = (torch.rand(2, 3 * h, h, device=device) * 2 - 1) / 16
    b_hn = torch.zeros(2, h, device=device)
    dy = torch.randn(BATCH, WINDOW, 2 * h, device=device).to(torch.bfloat16)
    y, saved = nat.gru_forward(gi, w_hh, b_hn, True)
    flop = 2.0 * BATCH * WINDOW * 2 * h * 3 * h
    cus = 2 * ((BATCH + nat.GRU_TILE - 1) // nat.GRU_TILE)
    peak = MFMA_BF16_PEAK_TFLOPS * cus / 256.0
    for name, fn in (("seld::gru_forward_kernel (+ layout permute)", lambda: nat.gru_forward(gi, w_hh, b_hn, True)),
                     ("seld::gru_backward_kernel (+ layout permutes)", lambda: nat.gru_backward(dy, saved, y, w_hh))):
        sec = timeit(fn)
        out.append({"kernel": name, "bound": f"mfma on {cus} CUs (one per direction and 4-sequence tile)",
                    "achieved": flop / sec / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": flop / sec / 1e12 / peak,
                    # the MFMAs run on all 16 columns although only GRU_TILE carry sequences (csrc/gru.hip, header)
                    "mfma_issue_frac": flop / sec / 1e12 / peak * 16 / nat.GRU_TILE,
                    "us_per_step": sec * 1e6 / WINDOW, "avg_launch_ms": sec * 1e3})
    return out


def cpu_baseline():
    """The oracle path on this host's cores, on a bounded sample (SURVEY.md section 8d protocol): log-mel by the torch.stft
    restatement, 3 warm-up + 10 timed 60 s clips; reference-equivalent fp32 CRNN forward + loss + backward + Adam at
    batch 2, 1 warm-up + 5 timed iterations.  Threads: min(16, cores) -- the share of host cores one GPU of the box has;
    a batch-2 model on all 128 hardware threads runs slower, not faster (oversubscription).  The reference's pure-Python
    label loops are timed on a 2 s excerpt and reported in `sample`, not included."""
    from oracle import features as ofeat
    from oracle import labels as olab
    import model_crnn
    import loss as loss_mod
    before = torch.get_num_threads()
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    try:
        pcm = ofeat.synth_pcm(0, CHANNELS, CLIP_SAMPLES, "noise")
        for _ in range(3):
            ofeat.logmel_torch(pcm)                                     # warm-up
        feat = []
        for _ in range(10):
            t0 = time.perf_counter()
            ofeat.logmel_torch(pcm)
            feat.append(time.perf_counter() - t0)
        t_feat = float(np.median(feat))
        rows = olab.synth_metadata(0, meta_frames=20)
        t0 = time.perf_counter()
        olab.metadata_to_labels_loops(rows, 48000)
        t_label_loops_60s = (time.perf_counter() - t0) * 30.0          # 2 s excerpt -> 60 s, linear in frames
        torch.manual_seed(0)
        model = model_crnn.SELD_CRNN().train()
        crit = loss_mod.SMRSELDLoss(loss_type="mse", w_class=1.0, grid_size=(18, 36))
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        x = torch.randn(2, WINDOW, CHANNELS, 64) * 20 - 30
        y = torch.zeros(2, WINDOW, 648, 14)
        y[..., 13] = 1.0
        times = []
        for it in range(6):
            t0 = time.perf_counter()
            opt.zero_grad()
            out, _ = crit.loss_tensor(model(x), y)
            out.backward()
            opt.step()
            times.append(time.perf_counter() - t0)
        t_window = float(np.median(times[1:])) / 2.0
    finally:
        torch.set_num_threads(before)
    clip_seconds = t_feat + 60.0 * t_window
    return {
        "value": 1.0 / clip_seconds, "unit": "clips/s", "cores": threads, "kind": "port",
        "sample": (f"oracle on {threads} of {os.cpu_count()} host threads: log-mel 3 warm-up + 10 timed 60 s clips, median "
                   f"{t_feat * 1e3:.0f} ms/clip; CRNN fwd+bwd+Adam fp32 bs=2, 1 warm-up + 5 timed iterations, median "
                   f"{t_window * 1e3:.0f} ms/window x 60 windows/clip; reference-style Python label loops would add "
                   f"{t_label_loops_60s:.1f} s/clip (not included)"),
        "features_clips_per_s": 1.0 / t_feat, "model_windows_per_s": 1.0 / t_window,
    }


def overlap_windows(hot, device, iterations=20):
    """The data-parallel step's backward stages on ONE rank, without collectives (seld_graph.GraphedTrainStep(split=True):
    exactly the graphs the N-rank path replays): average duration of every replayed graph, measured with HIP events
    between the replays, after the timed region.  The window that hides the all-reduce of bucket k is everything replayed
    after stage k and before the update graph."""
    import seld_graph
    trainer = hot.trainer
    step = seld_graph.GraphedTrainStep(hot.model, hot.criterion, hot.optimizer, device, world=1,
                                       autocast=lambda: trainer.autocast_context(device), split=True)
    nat = hot.native
    spec_tm = hot.spec_full[:, :FRAMES_PER_CLIP].reshape(-1, hot.feature_channels, 64)
    starts = hot.starts[:BATCH]
    spec = nat.gather_windows(spec_tm, starts, WINDOW)
    mask = nat.gather_windows(hot.mask_tm, starts, WINDOW)
    for _ in range(seld_graph.WARMUP + 2):
        step(spec, mask)
    step.timing = True
    for _ in range(iterations):
        step(spec, mask)
    seg = step.segment_ms()
    stats = step.stats()
    step.close()
    if not seg or stats.get("capture_error"):
        return {"error": stats.get("capture_error") or "no staged graphs"}
    stages, update = seg[:-1], seg[-1]
    buckets = stats["gradient_buckets"]
    return {"measured": f"1 GPU, {iterations} replays of the staged graphs without collectives, HIP events between replays",
            "backward_stages": len(stages), "stage_ms": stages, "update_ms": update,
            "iteration_ms": sum(seg),
            "buckets": [{"stage": b["stage"], "bytes": b["bytes"],
                         "hidden_under_ms": sum(stages[b["stage"] + 1:])} for b in buckets],
            "reduce_dtype": stats["reduce_dtype"]}


OTHER_WORKLOADS = (
    ("conformer", dict(model="conformer")),
    ("resnet_conformer + three-term loss + Gaussian label augmentation",
     dict(model="resnet_conformer", loss="three_term", gaussian_augment=True)),
    ("crnn, 8-ch MIC array, log-mel + GCC-PHAT (36 input channels)", dict(model="crnn", features="logmel_gcc", channels=8)),
)


def other_workloads(args, device, steps=3, warmup=2):
    """Short runs (``warmup`` + ``steps`` steps of 32 clips each) of the other BASELINE.json workloads in this process,
    after the headline's timed region: configs[2] (Conformer), configs[4]'s model + loss + augmentation on one GPU,
    configs[3]'s per-GPU shard.  Reported beside the headline, never part of ``value``."""
    import copy
    import gc
    out = []
    for name, override in OTHER_WORKLOADS:
        a = copy.copy(args)
        for k, v in override.items():
            setattr(a, k, v)
        try:
            hot = HotPath(a, device, 0, 1)
            for _ in range(warmup):
                hot.step(timed=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = hot.step(timed=True)
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            feat_ms = float(np.mean([x.elapsed_time(y) for x, y in hot.feat_events]))
            model_ms = float(np.mean([x.elapsed_time(y) for x, y in hot.model_events]))
            windows = CLIPS_PER_STEP * FRAMES_PER_CLIP // HOP
            tflops = windows * GFLOP_PER_WINDOW[a.model] / (model_ms * 1e-3) / 1e3
            frames = 1 + CLIP_SAMPLES // 480
            feat_bytes = CLIPS_PER_STEP * (a.channels * CLIP_SAMPLES * 4 + hot.feature_channels * 64 * frames * 4)
            out.append({"workload": f"{name}, bs={BATCH} windows, bf16, 1 GPU", "baseline_config": WORKLOAD_CONFIG[name],
                        "clips_per_s": CLIPS_PER_STEP * steps / elapsed, "steps": steps, "warmup": warmup,
                        "ms_per_iteration": model_ms / (windows // BATCH),
                        "roofline_model_frac": tflops / MFMA_BF16_PEAK_TFLOPS, "model_tflops": tflops,
                        "feature_phase_ms": feat_ms, "feature_hbm_frac": feat_bytes / (feat_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "final_loss": float(loss.item()),
                        "captured_step": hot.stepper.stats() if hasattr(hot.stepper, "stats") else None})
            if hasattr(hot.stepper, "close"):
                hot.stepper.close()
        except Exception as exc:           # noqa: BLE001  -- a secondary workload must not take the headline line down
            out.append({"workload": name, "error": f"{type(exc).__name__}: {exc}"})
        hot = None
        gc.collect()
        torch.cuda.empty_cache()
    return out


WORKLOAD_CONFIG = {OTHER_WORKLOADS[0][0]: "BASELINE configs[2]",
                   OTHER_WORKLOADS[1][0]: "BASELINE configs[4] (model, loss and augmentation; one GPU of the 8)",
                   OTHER_WORKLOADS[2][0]: "BASELINE configs[3] (the per-GPU shard: 32 of the 256 windows)"}


def main():
    global CLIPS_PER_STEP
    args = parse_args()
    if args.rehearsal_clips is not None:
        if not (args.rehearse_gloo or os.environ.get("SELD_DIST_BACKEND") == "gloo"):
            raise SystemExit("--rehearsal-clips only applies to the gloo rehearsal (a measurement is 32 clips per step)")
        CLIPS_PER_STEP = int(args.rehearsal_clips)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))          # BEFORE anything initialises the GPU in this process
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP extension has no CPU fallback)")
    # one rank per GPU over RCCL; gloo = the rehearsal mode in which ranks may share a GPU (RCCL refuses that)
    backend = "gloo" if args.rehearse_gloo else os.environ.get("SELD_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"{world} ranks over RCCL need {world} GPUs ({torch.cuda.device_count()} visible); "
                         f"--rehearse-gloo rehearses on fewer")
    index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(index)
    device = torch.device("cuda", index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
    torch.backends.cudnn.benchmark = True

    hot = HotPath(args, device, rank, world, backend)
    for _ in range(args.warmup):
        hot.step(timed=False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = hot.step(timed=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if world > 1:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t.item()) for t in every]
        elapsed = max(per_rank)                               # the job is as slow as its slowest rank

    in_sync = None
    if world > 1:       # every replica must hold the same weights after the timed steps (sum of all parameters)
        with torch.no_grad():
            total = sum(p.double().sum() for p in hot.model.parameters()).reshape(1)
        if backend != "nccl":
            total = total.cpu()
        hi, lo = total.clone(), total.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        in_sync = bool((hi - lo).abs().item() <= 1e-9 * max(1.0, abs(hi.item())))
        out_of_sync = []
        if not in_sync:          # name the tensors that differ (a diagnostic: this never happens in a healthy run)
            with torch.no_grad():
                each = torch.stack([p.double().sum() for p in hot.model.parameters()])
            if backend != "nccl":             # RCCL moves device tensors only
                each = each.cpu()
            both = [torch.zeros_like(each) for _ in range(world)]
            dist.all_gather(both, each)
            names = [n for n, _ in hot.model.named_parameters()]
            for i, name in enumerate(names):
                vals = [b[i].item() for b in both]
                if any(v != vals[0] for v in vals) or any(v != v for v in vals):         # different, or NaN
                    out_of_sync.append(f"{name} {vals}")
    if rank == 0:
        feat_ms = float(np.mean([a.elapsed_time(b) for a, b in hot.feat_events]))
        model_ms = float(np.mean([a.elapsed_time(b) for a, b in hot.model_events]))
        clips = world * CLIPS_PER_STEP * args.steps
        frames = 1 + CLIP_SAMPLES // 480
        bytes_per_clip = args.channels * CLIP_SAMPLES * 4 + hot.feature_channels * 64 * frames * 4
        achieved_gbs = CLIPS_PER_STEP * bytes_per_clip / (feat_ms * 1e-3) / 1e9
        windows_per_step = CLIPS_PER_STEP * FRAMES_PER_CLIP // HOP
        model_tflops = windows_per_step * GFLOP_PER_WINDOW[args.model] / (model_ms * 1e-3) / 1e3
        traffic = measured_traffic_per_clip()
        line = {
            "metric": "training clips/sec (4ch FOA 60s@24kHz)",
            "value": clips / elapsed, "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp32" if args.fp32 else "bf16", "data": "synthetic",
            "config": {"workload": (f"{args.model} FOA 4-ch 60s clips, bs={BATCH} windows ({BASELINE_CONFIG[args.model]})"
                                    if args.features == "logmel" and args.channels == 4 else
                                    f"{args.model} {args.channels}-ch 60s clips, features {args.features} "
                                    f"({hot.feature_channels} input channels), bs={BATCH} windows per GPU"),
                       "clips_per_step": CLIPS_PER_STEP, "windows_per_step": windows_per_step,
                       "optimizer_iterations_per_step": windows_per_step // BATCH,
                       "parallelism": f"dp{world}", "windows_per_s": clips * 60 / elapsed,
                       "final_loss": float(loss.item())},
            "roofline": {"kernel": "seld::logmel_main_kernel<float, 0> (fused STFT+mel+dB; the edge iterations are its "
                                   "trailing workgroups)"
                         if args.features == "logmel" else f"feature phase: {'seld::logmel_main_kernel<float, 2> (log-mel + Q15 phasors) + seld::gcc_q15_kernel<2>' if args.features == 'logmel_gcc' else 'seld::logmel_iv_kernel<float> (log-mel + intensity vectors, one pass)'}",
                         "bound": "hbm",
                         "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": traffic[1] * CLIPS_PER_STEP if traffic and args.features == "logmel" and args.channels == 4 else None,
                         "traffic_source": f"committed PMC pass profiles/{traffic[0]}, not this run (rocprofv3 --pmc "
                                           f"FETCH_SIZE x2 + WRITE_SIZE per launch)"
                         if traffic and args.features == "logmel" and args.channels == 4 else None,
                         "bytes_per_launch": CLIPS_PER_STEP * bytes_per_clip, "avg_launch_ms": feat_ms,
                         "features_only_clips_per_s": CLIPS_PER_STEP / (feat_ms * 1e-3)},
            "roofline_model": {"phase": f"{args.model} fwd+bwd+Adam, {windows_per_step // BATCH} iterations of {BATCH} windows",
                               "bound": "mfma", "achieved": model_tflops, "peak": MFMA_BF16_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": model_tflops / MFMA_BF16_PEAK_TFLOPS, "avg_ms": model_ms},
        }
        if world > 1:
            line["config"]["backend"] = "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend
            line["config"]["replicas_in_sync"] = in_sync
            if out_of_sync:
                line["config"]["replicas_out_of_sync"] = out_of_sync
            line["config"]["per_rank_clips_per_s"] = [CLIPS_PER_STEP * args.steps / t for t in per_rank]
            if backend != "nccl":
                line["rehearsal"] = True          # ranks share a GPU over gloo: a functional check, not a measurement
        step_stats = hot.stepper.stats() if hasattr(hot.stepper, "stats") else None
        line["config"]["captured_step"] = step_stats
        if world > 1:
            # the gradient exchange of this run: buckets all-reduced asynchronously between the backward-stage graphs
            line["allreduce_overlap"] = bool(step_stats and step_stats.get("allreduce_overlap"))
            line["gradient_buckets"] = (step_stats or {}).get("gradient_buckets")
        if getattr(hot, "loss_trace", None):
            line["config"]["loss_trace"] = hot.loss_trace
        if hasattr(hot.optimizer, "fused_casts"):
            line["config"]["master_weights"] = {"one_launch_gradient_casts": hot.optimizer.fused_casts,
                                                "per_tensor_fallbacks": hot.optimizer.fallback_casts}
        if world == 1 and not args.no_kernel_rooflines:
            line["kernels"] = kernel_rooflines(device)
            if hasattr(hot.stepper, "stats"):
                line["allreduce_overlap"] = overlap_windows(hot, device)
        if world == 1 and not args.no_other_workloads and args.model == "crnn" and args.features == "logmel" \
                and args.loss == "class" and not args.fp32:
            if hasattr(hot.stepper, "close"):
                hot.stepper.close()
            hot.pcm = hot.spec_full = hot.mask_tm = None
            torch.cuda.empty_cache()
            line["other_workloads"] = other_workloads(args, device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
