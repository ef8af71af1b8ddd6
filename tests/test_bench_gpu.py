"""bench.py's contract (one JSON line; N ranks launched by torch.distributed.run) exercised on ONE GPU: the N > 1 control
flow — rendezvous on 127.0.0.1, per-rank engine, captured graph A / all-reduce / graph B, barrier + max-over-ranks timing,
rank 0 prints — runs with the gloo backend standing in for RCCL (VNETI_DIST_BACKEND=gloo; every rank uses cuda:0)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly ONE JSON line expected, got {len(lines)}: {r.stdout[-500:]}"
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
def test_bench_single_gpu_line_tiny():
    d = _run([sys.executable, "bench.py", "--model", "tiny", "--resolution", "64", "--batch", "2", "--steps", "3",
              "--warmup", "1", "--no-cpu-baseline"], {})
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["unit"] == "steps/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f16" and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["algorithmic_bytes_per_launch"] > 0 and r["avg_launch_us"] > 0
    assert abs(d["ms_per_step"] * d["value"] - 1000.0) < 1.0


@pytest.mark.timeout(1200)
def test_bench_two_ranks_on_one_gpu():
    port = 29700 + os.getpid() % 200
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
              "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--model", "tiny", "--resolution", "64",
              "--batch", "2", "--steps", "3", "--warmup", "1"],
             {"VNETI_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 4
    assert d["value"] > 0 and abs(d["ms_per_step"] * d["value"] - 2000.0) < 2.0  # whole-job rate: 2 ranks' steps / max time
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only


@pytest.mark.timeout(1800)
def test_bench_eight_ranks_on_one_gpu():
    """the driver's SCALE run by construction (no 8-GPU node was ever available to the build): `bench.py --gpus 8` as eight
    ranks (gloo, sharing cuda:0) — weak scaling: global batch 8 x per-GPU batch, value = 8 ranks' steps / MAX rank time,
    every rank on rank 0's tile picks, per-rank timings populated, the exchange route named."""
    port = 29900 + os.getpid() % 90
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
              "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "8", "--model", "tiny", "--resolution", "64",
              "--batch", "2", "--steps", "3", "--warmup", "1"],
             {"VNETI_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "OMP_NUM_THREADS": "2"})
    c = d["config"]
    assert d["n_gpus"] == 8 and c["parallelism"] == "dp8" and c["global_batch"] == 16 and c["rccl_ranks"] == 8
    assert d["scaling"] == "weak" and abs(d["ms_per_step"] * d["value"] - 8000.0) < 8.0
    assert c["picks_identical_on_all_ranks"] is True
    assert 0 < c["rank_ms_per_step_min"] <= c["rank_ms_per_step_max"] and abs(c["rank_ms_per_step_max"] - d["ms_per_step"]) < 1e-6
    assert c["exchange_in_graph"] is False and "gloo" in c["exchange"]  # the one-graph route needs the nccl backend
    assert "cpu_baseline" not in d


@pytest.mark.timeout(1200)
def test_bench_self_launches_two_ranks():
    """the driver's own invocation: plain `python bench.py --gpus 2` (no launcher) must spawn the 2 ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VNETI_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--model", "tiny", "--resolution", "64", "--batch", "2",
                        "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["config"]["dist_backend"] == "gloo"
    assert d["config"]["rank0_ms_per_step"] <= d["ms_per_step"] * 1.0001
