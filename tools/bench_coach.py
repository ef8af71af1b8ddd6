"""Dev tool: END-TO-END throughput of `Coach.train` at the benchmark's size (SD-1.5 shapes, 512x512, bs 4) — dataloader,
H2D upload / device input pipeline, per-step host bookkeeping and the captured step — next to bench.py's resident-batch
number.  Prints one JSON line per variant.   python tools/bench_coach.py [--steps 60] [--aug 0|7] [--workers 4]"""
import os
os.environ.setdefault("VNETI_ALLOW_SYNTHETIC_WEIGHTS", "1")  # dev tool: synthetic SD-shaped weights on purpose
import argparse, json, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from view_neti_amd.compat import config as C
from view_neti_amd.compat.coach import Coach

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60); ap.add_argument("--aug", type=int, default=0)
ap.add_argument("--workers", type=int, default=4)
ap.add_argument("--variants", default="host,device", help="which input pipelines to time (bench.py asks for `device` only)")
a = ap.parse_args()
tmp = tempfile.mkdtemp()
root = os.path.join(tmp, "teapot"); os.makedirs(root)
rng = np.random.RandomState(0)
for i in range(5):
    Image.fromarray(rng.randint(0, 255, (600, 800, 3), dtype=np.uint8)).save(os.path.join(root, f"{i}.jpg"))
for name, device_pipe, workers in (("host pipeline", False, a.workers), ("device pipeline", True, 0)):
    if name.split()[0] not in a.variants.split(","):
        continue
    cfg = C.parse(C.RunConfig, ["--data.train_data_dir", root, "--data.placeholder_object_token", "<teapot>", "--learnable_mode", "0",
        "--model.word_embedding_dim", "768", "--model.arch_view_net", "15", "--model.arch_view_disable_tl", "False",
        "--model.arch_mlp_hidden_dims", "64", "--model.use_nested_dropout", "False", "--optim.max_train_steps", str(a.steps),
        "--optim.train_batch_size", "4", "--optim.gradient_accumulation_steps", "1", "--optim.mixed_precision", "fp16",
        "--data.augmentation_key", str(a.aug), "--data.dataloader_num_workers", str(workers),
        "--data.device_input_pipeline", str(device_pipe), "--log.save_steps", "100000", "--eval.validation_steps", "100000",
        "--log.exp_dir", os.path.join(tmp, "out"), "--log.exp_name", name.replace(" ", "_")])
    cfg.log.exp_dir = cfg.log.exp_dir / cfg.log.exp_name
    cfg.log.logging_dir = cfg.log.exp_dir / cfg.log.logging_dir
    torch.manual_seed(cfg.seed)
    coach = Coach(cfg)
    coach.logger.setLevel(100)
    warm = 10
    cfg.optim.max_train_steps = warm
    coach.train()                      # captures the graph, warms the dataloader workers and the image cache
    torch.cuda.synchronize()
    coach.engine.opt_step.zero_()
    cfg.optim.max_train_steps = a.steps
    t0 = time.perf_counter()
    coach.train()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"variant": name, "steps": a.steps, "steps_per_s": a.steps / dt, "ms_per_step": dt / a.steps * 1e3,
                      "augmentation_key": a.aug, "workers": workers, "includes": "dataloader + upload/pipeline + set_batch + "
                      "graph replay + final checkpoint save", "loss": coach.engine.loss()}), flush=True)
    del coach
    torch.cuda.empty_cache()
