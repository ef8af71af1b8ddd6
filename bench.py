"""Benchmark of the ViewNeTI textual-inversion train step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1 either way: plain `python bench.py --gpus N` re-launches itself as N ranks (one per GPU, RCCL) through
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`;
    started under torch.distributed.run already (WORLD_SIZE set) it just joins the rendezvous.

A "step" is one optimisation step of the reference's Coach.train loop body
(training/coach.py:154-231) on one micro-batch per GPU: VAE-encode -> sample/add-noise ->
16 x (NeTI mapper + CLIP text encoder) -> UNet forward with XTI per-layer contexts -> MSE ->
backward (dgrad through UNet and CLIP, wgrad of the mapper) -> AdamW — nothing cached or skipped.
Workload = BASELINE.json configs[1]: learnable_mode 0, SD-1.5 shapes, 512x512, fp16, bs=4,
gradient_accumulation 1, synthetic SD-shaped weights and inputs (no checkpoints / datasets exist
on the boxes).  Data-parallel weak scaling: every rank runs its own micro-batch, the flat
mapper-gradient bucket is all-reduced over RCCL; `value` = micro-steps completed by all ranks / s.

Rank 0 prints ONE JSON line with the metric plus
  "roofline":     measured-with-HIP-events average duration of the dominant kernel family
                  (the MFMA implicit-GEMM conv/linear kernel) vs the fp16 dense MFMA peak
  "cpu_baseline": the CPU oracle (oracle/sd_ref.py, a plain-torch fp32 restatement of the same
                  graph — diffusers itself is not installable here) timed on the host cores on a
                  bounded sample.  Reported, non-target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

from view_neti_amd.roofline import gemm_cost  # noqa: E402,F401  (cost model shared with tools/kernel_roofline.py)

MFMA_PEAK_TFLOPS = 2500.0  # fp16/bf16 dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_TBS = 8.0  # HBM3E spec peak, same guide
TILE_NAMES = {1: "gemm_kernel<128,128,64,64>", 2: "gemm_kernel<128,64,64,32>", 3: "gemm_kernel<64,64,32,32>",
              4: "gemm_kernel<256,128,64,64>", 5: "gemm_kernel<256,256,64,64>", 6: "gemm_kernel<256,128,64,64,ring3>",
              7: "gemm_kernel<256,128,64,32,ring3>", 8: "gemm_kernel<256,128,64,32>", 9: "gemm_kernel<128,128,64,32>",
              10: "gemm_kernel<128,128,64,32,ring3>", 11: "gemm_kernel<128,64,64,32,ring3>", 12: "gemm_kernel<64,64,32,32,ring3>",
              13: "gemm_kernel<128,128,64,32,ring4>", 14: "gemm_kernel<128,64,64,32,ring4>", 15: "gemm_kernel<64,64,32,32,ring4>",
              16: "gemm8_kernel<256,256,8-phase>", 17: "gemm8_kernel<256,128,8-phase>",
              18: "gemm8_kernel<256,128,8-phase,halo>"}
# algorithmic FLOPs per sample at 512^2, SD-1.5: VAE 1116.7 + CLIP 16x13.3 + UNet fwd 803.3
# + UNet dgrad 929.4 + CLIP dgrad 216 GF
ALGO_GFLOP_PER_SAMPLE_512 = 3278.0


def build_engine(args, rank, world, moment_cache: bool = False):
    from view_neti_amd import sd_config as sc, synth
    from view_neti_amd.engine.step import TrainStepEngine
    from view_neti_amd.mapper import fourier_frequencies, init_mapper_state
    cfg = sc.CONFIGS[args.model]()
    dev = "cuda"
    uw = synth.unet_weights(cfg.unet, device=dev)
    vw = synth.vae_weights(cfg.vae, device=dev)
    cw = synth.clip_weights(cfg.clip, device=dev)
    D = cfg.clip.hidden_size
    # one placeholder token appended to the vocabulary, initialised from a super-category row and
    # its norm used as the mapper's norm_scale (training/coach.py:367-395)
    tok = cw["text_model.embeddings.token_embedding.weight"]
    super_id = 1125 % cfg.clip.vocab_size
    cw["text_model.embeddings.token_embedding.weight"] = torch.cat([tok, tok[super_id:super_id + 1]], 0)
    norm_scale = float(tok[super_id].norm().item())
    placeholder_id = cfg.clip.vocab_size
    # reference quirk (App. C Q1): every mapper is initialised right after torch.manual_seed(0)
    w_enc = fourier_frequencies([0.03, 2.0], 64, 0)
    sd = init_mapper_state(64, 64, D)
    lr = 1e-3 * args.batch * world  # scale_lr rule of training/coach.py:728-733 (accum = 1)
    eng = TrainStepEngine(cfg, uw, vw, cw, args.batch, args.resolution, args.resolution, sd, w_enc, norm_scale, 0.2,
                          lr=lr, seed=1234 + rank, world_size=world, device_rng=True,
                          overlap=os.environ.get("VNETI_OVERLAP", "0") == "1",  # lab switch; the product default is no fork
                          moment_cache_images=args.batch if moment_cache else 0)
    del uw, vw, cw
    ids = synth.input_ids(args.batch, placeholder_id, cfg.clip.vocab_size)
    eng.set_batch(synth.pixel_values(args.batch, args.resolution, args.resolution, seed=1 + rank), ids,
                  torch.full((args.batch,), placeholder_id), image_idx=list(range(args.batch)) if eng.n_cache else None)
    return cfg, eng


def roofline_pass(eng, reps=3):
    """Average launch duration of each GEMM tile configuration, measured with HIP events on the
    launch stream by replaying exactly the step's GEMM launches back to back."""
    from view_neti_amd import ops
    groups = {}
    for f in eng.launches():
        if getattr(f, "func", None) is not ops.gemm:
            continue
        M, N, K, batch, flops, nbytes, t_mfma, t_hbm = gemm_cost(f)
        tile = f.keywords.get("tile_hint") or ops.gemm_select_tile(M, N, batch)
        g = groups.setdefault(tile, dict(launches=[], flops=0.0, bytes=0.0, floor_s=0.0, hbm_bound=0))
        g["launches"].append(f)
        g["flops"] += flops
        g["bytes"] += nbytes
        # the launch's TRUE bound: the slower of its MFMA time and its HBM time (the short-K linears sit below the ridge
        # of ~310 FLOP/B and are bandwidth kernels, whatever unit executes them)
        g["floor_s"] += max(t_mfma, t_hbm)
        g["hbm_bound"] += 1 if t_hbm > t_mfma else 0
    out = {}
    for tile, g in groups.items():
        for f in g["launches"]:
            f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(reps):
            for f in g["launches"]:
                f()
        e.record()
        torch.cuda.synchronize()
        total_ms = s.elapsed_time(e) / reps
        n = len(g["launches"])
        out[tile] = dict(n=n, total_ms=total_ms, avg_us=total_ms * 1e3 / n, flops_per_launch=g["flops"] / n,
                         bytes_per_launch=g["bytes"] / n,
                         tflops=g["flops"] / (total_ms * 1e-3) / 1e12,
                         frac_of_bound=g["floor_s"] / (total_ms * 1e-3), hbm_bound_launches=g["hbm_bound"])
    return out


def pmc_traffic(tile_name: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 counter passes
    (profiles/*_pmc.json, written by tools/profile_round.sh + tools/pmc_summary.py: separate --pmc
    FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    bench.py cannot run the counter passes itself, so `traffic` is null until a profile is committed."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc.json")))
    if not files:
        return {"traffic": None}
    dims = re.findall(r"\d+", tile_name.split(",ring")[0])
    stages = "3" if ",ring3" in tile_name else "4" if ",ring4" in tile_name else "2"

    def targs(name):
        """template arguments of a kernel name, demangled (`k<128, 1, true, true>`) or mangled (`kILi128ELi1ELb1ELb1EE`)"""
        m = re.search(r"gemm8?_kernel<([^>]*)>", name)
        if m:
            return [a.strip() for a in m.group(1).split(",")]
        m = re.search(r"gemm8?_kernelI((?:L[ib]\d+E)+)E", name)
        if m:
            return [("true" if v == "1" else "false") if t == "b" else v for t, v in re.findall(r"L([ib])(\d+)E", m.group(1))]
        return []

    # every epilogue instantiation (EPI level) of the tile counts as the same kernel; nothing else does
    if tile_name.startswith("gemm8_kernel"):
        # gemm8_kernel<BN, EPI, CONV, HALO>: BN is the tile WIDTH (the second number of the display name), HALO its own class
        bn, halo = dims[2], "true" if "halo" in tile_name else "false"

        def match(name):
            a = targs(name)
            return "gemm8_kernel" in name and len(a) >= 3 and a[0] == bn and (a[3] if len(a) > 3 else "false") == halo
    else:
        # gemm_kernel<BM, BN, WM, WN, F32OUT, DMA, STAGES, EPI, CONV>
        def match(name):
            a = targs(name)
            return "gemm_kernel" in name and "gemm8" not in name and a[:4] == dims[:4] and len(a) >= 7 and a[6] == stages
    try:
        doc = json.load(open(files[-1]))
        ks = doc["kernels"]
    except (OSError, ValueError, KeyError):
        return {"traffic": None}
    # a counter figure belongs to the kernels it was collected on: quote it only when the file is stamped with THIS tree's
    # kernel sources (tools/pmc_summary.py writes the stamp); otherwise the driver's record says null, not a stale number
    from view_neti_amd.roofline import kernel_tree_sha
    here = kernel_tree_sha()
    if doc.get("kernel_tree_sha") != here:
        return {"traffic": None, "traffic_source": os.path.basename(files[-1]),
                "traffic_note": f"counter passes in {os.path.basename(files[-1])} were collected on kernel tree "
                                f"{doc.get('kernel_tree_sha')}, this run's is {here}: not quoted"}
    hit = [e for name, e in ks.items() if match(name)]
    n = sum(e.get("launches", 1) for e in hit)
    if not hit or n == 0:
        return {"traffic": None}
    return {"traffic": sum(e["hbm_bytes"] * e.get("launches", 1) for e in hit) / n,
            "traffic_source": os.path.basename(files[-1]), "traffic_kernel_tree_sha": here,
            "traffic_note": "mean HBM bytes/launch over this tile's launches (all epilogue instantiations) in one eager "
                            "train step (FETCH_SIZE x2 + WRITE_SIZE), collected by the builder with tools/pmc_round.sh on "
                            "the kernel tree named in traffic_kernel_tree_sha (= this run's); bench.py cannot run the "
                            "separate --pmc passes itself"}


def usable_cores(cap: int = 32) -> int:
    """threads the CPU leg really gets: scheduler affinity and the cgroup CPU quota, capped (a 256-thread
    pool on the GPU box's host ran the oracle at <1 GFLOP/s — oversubscription, not a baseline)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def cpu_baseline_bounded(args, timeout_s: int = 240):
    """run cpu_baseline() in a child process under a hard wall-clock bound so the default bench run
    always finishes within minutes; a timeout is reported, never hidden.  First at the bench's own batch size
    (bs=4, 1 warm-up + 3 timed steps); if that does not fit the bound (or the host's memory) the leg
    falls back to bs=1 scaled by algorithmic FLOPs and says so in `sample`."""
    import subprocess
    notes = []
    for bs, bound in ((args.batch, timeout_s), (1, timeout_s // 2)):
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--model", args.model, "--batch",
               str(args.batch), "--resolution", str(args.resolution), "--cpu-resolution", str(args.cpu_resolution),
               "--cpu-steps", str(args.cpu_steps), "--cpu-batch", str(bs)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=bound)
            for line in reversed(r.stdout.strip().splitlines()):
                if line.startswith("{"):
                    out = json.loads(line)
                    if notes:
                        out["sample"] += "; " + "; ".join(notes)
                    return out
            notes.append(f"bs={bs} child failed (rc={r.returncode}): {r.stderr.strip()[-200:]}")
        except subprocess.TimeoutExpired:
            notes.append(f"bs={bs} at {args.cpu_resolution}x{args.cpu_resolution} did not finish within the {bound}s bound")
        if bs == 1:
            break
    return {"value": None, "unit": "steps/s", "cores": usable_cores(), "kind": "port",
            "sample": "oracle/sd_ref.py train step: " + "; ".join(notes)}


def cpu_baseline(args):
    """The oracle's train step on the host cores, on a bounded sample of the same workload
    (same SD-1.5 weights; bs=1 at a reduced resolution so the default run stays within minutes);
    converted to the bench unit by the algorithmic-FLOP ratio."""
    from oracle import sd_ref as R
    from view_neti_amd import sd_config as sc, synth
    from view_neti_amd.mapper import fourier_frequencies, init_mapper_state
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = sc.CONFIGS[args.model]()
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    uw = {k: v.cpu() for k, v in synth.unet_weights(cfg.unet, device=dev).items()}
    vw = {k: v.cpu() for k, v in synth.vae_weights(cfg.vae, device=dev).items()}
    cw = {k: v.cpu() for k, v in synth.clip_weights(cfg.clip, device=dev).items()}
    B, res = max(1, args.cpu_batch), args.cpu_resolution
    if B > 1:
        try:
            import psutil
            if psutil.virtual_memory().available < 20 * 2 ** 30 * B:
                raise SystemExit(f"host memory too small for the fp32 autograd graph at bs={B}")
        except ImportError:
            pass
    w_enc = fourier_frequencies([0.03, 2.0], 64, 0, preserve_rng=True)
    sd = {k: v.requires_grad_(True) for k, v in init_mapper_state(64, 64, cfg.clip.hidden_size).items()}
    ph = cfg.clip.vocab_size - 3
    ids = synth.input_ids(B, ph, cfg.clip.vocab_size)
    px, t = synth.pixel_values(B, res, res), synth.timesteps(B)
    eps, noise = synth.gaussian((B, 4, res // 8, res // 8), 3), synth.gaussian((B, 4, res // 8, res // 8), 4)
    def one_step():
        for v in sd.values():
            v.grad = None
        t0 = time.time()
        loss, _ = R.train_step_loss(cfg, uw, vw, cw, sd, w_enc, 0.4, px, ids, torch.full((B,), ph), t, eps, noise)
        loss.backward()
        return time.time() - t0, loss.item()

    # 1 warm-up + 3 timed steps (the warm-up pays oneDNN primitive creation and the allocator's first touch)
    warm, _ = one_step()
    times, loss_v = [], 0.0
    for _ in range(args.cpu_steps):
        dt, loss_v = one_step()
        times.append(dt)
    dt = sum(times) / len(times)
    # FLOP ratio between the sample and one bench step (VAE and UNet scale with pixels, CLIP with batch)
    pix = (res / 512.0) ** 2
    sample_gf = ((1116.7 + 803.3 + 929.4) * pix + 212.8 + 216.0) * B
    bench_gf = ALGO_GFLOP_PER_SAMPLE_512 * args.batch * (args.resolution / 512.0) ** 2
    est_steps_per_s = 1.0 / dt if (B, res) == (args.batch, args.resolution) else (sample_gf / dt) / bench_gf
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else -1
    return {"value": est_steps_per_s, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/sd_ref.py fp32 torch-CPU restatement (diffusers not installable): 1 warm-up ({warm:.1f}s) + "
                      f"{len(times)} timed train steps (fwd+bwd) at bs={B} {res}x{res}, mean {dt:.2f}s min {min(times):.2f}s "
                      f"({sample_gf / dt:.1f} GFLOP/s, loss {loss_v:.4f}); "
                      + ("the bench's own batch size and resolution, nothing scaled; " if (B, res) == (args.batch, args.resolution) else
                         f"bs={B} instead of the bench's bs={args.batch} to stay inside the wall-clock bound, scaled to "
                         f"bs={args.batch} {args.resolution}x{args.resolution} by algorithmic FLOPs (x{bench_gf / sample_gf:.2f}); ")
                      + f"torch.get_num_threads()={torch.get_num_threads()}, "
                      f"sched affinity {aff} cpus, os.cpu_count()={os.cpu_count()}, cpu='{model}'"}


def driver_timed_extras(cache_path):
    """Two numbers the headline does not carry, measured AFTER the timed region so that the driver's own run records them
    (VERDICT r4 item 5b): the end-to-end `Coach.train` rate at the headline size (dataloader-free device input pipeline,
    set_batch, graph replay, host bookkeeping: tools/bench_coach.py, training/coach.py:154-264) and BASELINE config 5's
    seconds per image (SD-2.1 768^2, DDIM-50, CFG: tools/bench_infer.py, sd_pipeline_call.py:73-98).  Each runs as a bounded
    subprocess on the now idle GPU and reuses this run's autotuner picks; a failure is recorded, never fatal."""
    import subprocess
    env = dict(os.environ, VNETI_ALLOW_SYNTHETIC_WEIGHTS="1")
    if cache_path:
        env["VNETI_AUTOTUNE_CACHE"] = cache_path
    out = {}
    for key, cmd, field in (
            ("coach_steps_per_s", [os.path.join(ROOT, "tools", "bench_coach.py"), "--steps", "60", "--variants", "device"], "steps_per_s"),
            ("infer_s_per_image_cfg5", [os.path.join(ROOT, "tools", "bench_infer.py"), "--steps", "50", "--reps", "1"], "s_per_image"),
            # the reference's own evaluation configuration: width 768 x height 576, DPM-Solver++ 30 steps, CFG 7.5
            # (training/validate.py:56-62,568-573; training/config.py eval.num_denoising_steps)
            ("infer_s_per_image_eval_768x576_dpmpp30",
             [os.path.join(ROOT, "tools", "bench_infer.py"), "--steps", "30", "--reps", "1", "--sampler", "dpm++2m",
              "--height", "576", "--width", "768"], "s_per_image")):
        try:
            r = subprocess.run([sys.executable, *cmd], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
            out[key] = json.loads(line)[field]
        except Exception as err:  # noqa: BLE001 (reported in the JSON line)
            out[key] = None
            out[key + "_error"] = f"{type(err).__name__}: {str(err)[:200]}"
    return out


def moment_cache_extra(args, steps: int):
    """NOT the headline: the same step with `data.cache_vae_moments` (mode 0, augmentation_key 0: the dataset is
    deterministic and the VAE posterior moments of an image are cached), on an engine of its OWN built after
    the headline's is gone — the headline engine carries no cache nodes, no extra graphs and no per-step host bookkeeping."""
    _, eng = build_engine(args, 0, 1, moment_cache=True)
    eng.capture()
    for _ in range(6):  # the first step runs the encoder and fills the cache; afterwards every batch image is cached
        eng.step()
    assert eng._use_cache(), "the batch should be served from the moment cache by now"
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        eng.step()
    torch.cuda.synchronize()
    v = steps / (time.perf_counter() - t1)
    del eng
    torch.cuda.empty_cache()
    return v


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks of one node, one per GPU,
    under torch.distributed.run (rendezvous on 127.0.0.1, a free port); rank 0's JSON line passes straight through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 20 warm-up + 200 timed steps (~6 s)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="sd15")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--resolution", type=int, default=512)
    ap.add_argument("--cpu-resolution", type=int, default=512)
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed oracle steps of the cpu_baseline leg (after 1 warm-up)")
    ap.add_argument("--cpu-batch", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-tile GEMM replay (counter-collection runs)")
    ap.add_argument("--no-extras", action="store_true", help="skip config.coach_steps_per_s / config.infer_s_per_image_cfg5")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start either plain `python bench.py --gpus N` or "
                         f"torch.distributed.run --nproc-per-node N bench.py --gpus N")
    # (debug aid: VNETI_DIST_BACKEND=gloo lets N ranks share one GPU to exercise the N > 1 control flow on a 1-GPU box)
    backend = os.environ.get("VNETI_DIST_BACKEND", "nccl")
    dev_index = local if backend == "nccl" else local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    # this run's autotuner picks, kept for the extras' subprocesses (same problems, no second tuning pass)
    tune_cache = None
    if world == 1 and "VNETI_AUTOTUNE_CACHE" not in os.environ:
        import tempfile
        tune_cache = os.path.join(tempfile.gettempdir(), f"vneti_bench_picks_{os.getpid()}.json")
        os.environ["VNETI_AUTOTUNE_CACHE"] = tune_cache
    if world > 1 and backend == "nccl" and not args.no_graph:
        # N GPUs = the one-GPU graph + ONE collective node; a rank that silently fell back to [graph A -> host all-reduce ->
        # graph B] would cost the scaling target and nobody would see why: make that an error (engine/step.py capture())
        os.environ.setdefault("VNETI_REQUIRE_ONE_GRAPH", "1")
    one_graph_error = None
    try:
        cfg, eng = build_engine(args, rank, world)
        if not args.no_graph:
            eng.capture()
    except RuntimeError as err:
        # the one-graph route was REQUIRED and is not available on this node (the requirement fails on every rank together:
        # parallel.all_agree).  A measurement with the reason attached beats no measurement: take the two-graph route and
        # say so in the line (`exchange_in_graph` false, `one_graph_error`).
        if not (world > 1 and os.environ.get("VNETI_REQUIRE_ONE_GRAPH") == "1"):
            raise
        one_graph_error = f"{type(err).__name__}: {str(err)[:300]}"
        print(f"[bench rank {rank}] one-graph route unavailable, falling back: {one_graph_error}", file=sys.stderr, flush=True)
        os.environ["VNETI_REQUIRE_ONE_GRAPH"] = "0"
        eng = None
        torch.cuda.empty_cache()
        cfg, eng = build_engine(args, rank, world)
        if not args.no_graph:
            eng.capture()
    for _ in range(args.warmup):
        eng.step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    barrier()
    dt = time.perf_counter() - t0
    dt_rank = dt
    rank_dts = [dt]
    if dist is not None:
        tall = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)]
        dist.all_gather(tall, torch.tensor([dt], dtype=torch.float64, device="cuda"))
        rank_dts = [float(t.item()) for t in tall]
        dt = max(rank_dts)  # the contract's MAX over ranks
    loss = eng.loss()
    # after the timed region: did every rank replay ONE schedule (rank 0's autotuner picks, parallel.shared_picks)?
    picks_equal = None
    if dist is not None:
        import hashlib
        from view_neti_amd import ops as _ops
        picks = [(f.keywords.get("tile_hint"), f.keywords.get("split_k"), (f.keywords.get("conv") or {}).get("korder"))
                 for f in eng.launches() if getattr(f, "func", None) is _ops.gemm]
        digest = hashlib.sha256(repr(picks).encode()).hexdigest()
        every = [None] * world
        dist.all_gather_object(every, digest)
        picks_equal = all(d_ == every[0] for d_ in every)
    cache_value = None  # measured after the headline engine is released (moment_cache_extra)

    if rank == 0 and args.no_roofline:
        print(json.dumps({"value": world * args.steps / dt, "unit": "steps/s", "note": "roofline pass skipped"}))
    elif rank == 0:
        from view_neti_amd import lib as _lib
        prec = _lib.precision()  # "fp16" (BASELINE.json's config) unless VNETI_PRECISION=bf16 selected the bf16 build
        prec_short = "bf16" if prec == "bf16" else "f16"
        ms = dt / args.steps * 1e3
        value = world * args.steps / dt
        rf = roofline_pass(eng)
        dom = max(rf, key=lambda k: rf[k]["total_ms"])
        d = rf[dom]
        out = {
            "metric": ("TI train steps/sec (SD-1.5 512^2 bs=4 per GPU)"
                       if (args.model, args.resolution, args.batch) == ("sd15", 512, 4) else
                       f"TI train steps/sec ({args.model} {args.resolution}^2 bs={args.batch} per GPU)"), "value": value, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": prec_short, "data": "synthetic",
            "config": {"workload": f"learnable_mode 0, {args.model} shapes, {args.resolution}x{args.resolution} {prec}, "
                                   f"bs={args.batch}/GPU, grad_accum 1, full train step (VAE+16xCLIP+UNet fwd/bwd+AdamW)",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "dist_backend": backend if world > 1 else None, "rccl_ranks": dist.get_world_size() if dist else 1,
                       "rank0_ms_per_step": dt_rank / args.steps * 1e3,
                       "rank_ms_per_step_min": min(rank_dts) / args.steps * 1e3,
                       "rank_ms_per_step_max": max(rank_dts) / args.steps * 1e3,
                       "hipgraph": not args.no_graph,
                       "exchange_in_graph": bool(eng.exchange_in_graph) if world > 1 else None,
                       "exchange": (None if world == 1 else "vneti_allreduce_flat (library RCCL communicator, captured node)"
                                    if eng.exchange_in_graph else "library RCCL communicator between two graphs"
                                    if eng.exchange is not None else f"torch.distributed.all_reduce ({backend}) between two graphs"),
                       "one_graph_error": one_graph_error,
                       "picks_identical_on_all_ranks": picks_equal,
                       "final_loss": loss,
                       "algorithmic_tflop_per_step": ALGO_GFLOP_PER_SAMPLE_512 * args.batch * (args.resolution / 512) ** 2 / 1e3,
                       "end_to_end_mfma_frac": ALGO_GFLOP_PER_SAMPLE_512 * args.batch * (args.resolution / 512) ** 2 / 1e3
                                               / (ms * 1e-3) / MFMA_PEAK_TFLOPS,
                       "engine_gib": eng.memory_bytes() / 2 ** 30,
                       "steps_per_s_with_vae_moment_cache": cache_value,
                       "vae_moment_cache_note": "labelled extra, NOT the metric: data.cache_vae_moments (deterministic dataset, "
                                                "augmentation_key 0) replays the step without the VAE encoder; the headline "
                                                "`value` runs the encoder in every timed step"},
            "roofline": {"kernel": TILE_NAMES[dom], "bound": "mfma", "achieved": d["tflops"], "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": d["tflops"] / MFMA_PEAK_TFLOPS,
                         **pmc_traffic(TILE_NAMES[dom]),
                         "launches_per_step": d["n"], "avg_launch_us": d["avg_us"],
                         "algorithmic_gflop_per_launch": d["flops_per_launch"] / 1e9,
                         "algorithmic_bytes_per_launch": d["bytes_per_launch"],
                         "algorithmic_bytes_note": "mean over this tile's launches of A (image for implicit convs) + weights "
                                                   "+ output + fused epilogue operands, each once; compare with `traffic`",
                         "frac_of_true_bound": d["frac_of_bound"],
                         "frac_of_true_bound_note": "sum over this tile's launches of max(FLOP / 2.5 PF, algorithmic bytes / 8 TB/s) "
                                                    "divided by the measured time: launches below the ridge are rated against HBM",
                         "all_gemm_tiles": {TILE_NAMES[k]: {"launches": v["n"], "ms_per_step": v["total_ms"],
                                                            "tflops": v["tflops"], "frac_of_bound": v["frac_of_bound"],
                                                            "hbm_bound_launches": v["hbm_bound_launches"]}
                                            for k, v in rf.items()}},
        }
        # every launch class of the step against ITS bound ("each kernel >= 60 %"), from this very run: one
        # pass over the eager schedule with an event pair around every launch (view_neti_amd/roofline.py)
        try:
            from view_neti_amd.roofline import time_classes
            cls = time_classes(eng.launches())
            out["roofline"]["classes"] = {
                k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()
                    if kk in ("launches", "ms_per_step", "bound", "achieved", "peak", "unit", "frac", "frac_of_cobound",
                              "cobound_frac_of_peak")}
                for k, v in sorted(cls.items(), key=lambda kv: -kv[1]["ms_per_step"])}
            out["roofline"]["classes_note"] = ("eager replay in schedule order, HIP events per launch, algorithmic FLOPs / bytes "
                                               "per class; attention also against the builder's MFMA+VALU co-bound at 2.4 GHz")
        except Exception as err:  # noqa: BLE001 (reported, never fatal for the metric)
            out["roofline"]["classes"] = None
            out["roofline"]["classes_error"] = f"{type(err).__name__}: {str(err)[:200]}"
        headline = (args.model, args.resolution, args.batch) == ("sd15", 512, 4)
        if world == 1 and (not args.no_cpu_baseline or (headline and not args.no_extras)):
            del eng
            torch.cuda.empty_cache()
        if world == 1 and headline and not args.no_extras and not args.no_cpu_baseline:
            try:
                out["config"]["steps_per_s_with_vae_moment_cache"] = moment_cache_extra(args, args.steps)
            except Exception as err:  # noqa: BLE001
                out["config"]["steps_per_s_with_vae_moment_cache_error"] = f"{type(err).__name__}: {str(err)[:200]}"
            out["config"].update(driver_timed_extras(tune_cache or os.environ.get("VNETI_AUTOTUNE_CACHE")))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_bounded(args)
        print(json.dumps(out))
    if tune_cache:
        try:
            os.unlink(tune_cache)
        except OSError:
            pass
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
