import os
os.environ.setdefault("VNETI_ALLOW_SYNTHETIC_WEIGHTS", "1")  # dev tool: synthetic SD-shaped weights on purpose
import sys, os, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from PIL import Image
from view_neti_amd.compat import config as C
from view_neti_amd.compat.coach import Coach
tmp = tempfile.mkdtemp()
root = os.path.join(tmp, "teapot"); os.makedirs(root)
rng = np.random.RandomState(0)
for i in range(5):
    Image.fromarray(rng.randint(0, 255, (600, 800, 3), dtype=np.uint8)).save(os.path.join(root, f"{i}.jpg"))
cfg = C.parse(C.RunConfig, ["--data.train_data_dir", root, "--data.placeholder_object_token", "<teapot>", "--learnable_mode", "0",
    "--model.word_embedding_dim", "768", "--model.arch_view_net", "15", "--model.arch_view_disable_tl", "False",
    "--model.arch_mlp_hidden_dims", "64", "--model.use_nested_dropout", "True", "--optim.max_train_steps", "40",
    "--optim.train_batch_size", "4", "--optim.gradient_accumulation_steps", "1", "--optim.mixed_precision", "fp16",
    "--data.augmentation_key", "7", "--data.dataloader_num_workers", "4", "--log.save_steps", "20", "--eval.validation_steps", "20",
    "--eval.num_denoising_steps", "10", "--eval.num_validation_images", "1", "--eval.validation_seeds", "[0]",
    "--log.exp_dir", os.path.join(tmp, "out"), "--log.exp_name", "teapot"])
cfg.log.exp_dir = cfg.log.exp_dir / cfg.log.exp_name
cfg.log.logging_dir = cfg.log.exp_dir / cfg.log.logging_dir
torch.manual_seed(cfg.seed)
t0 = time.time(); coach = Coach(cfg); t1 = time.time()
coach.train(); torch.cuda.synchronize(); t2 = time.time()
print(f"build {t1-t0:.1f}s train(40 steps + 2 validations + saves) {t2-t1:.1f}s; files:", sorted(os.listdir(cfg.log.exp_dir))[:12])
print("loss", coach.engine.loss(), "opt_step", coach.engine.opt_step.item(), "mem GiB", torch.cuda.max_memory_allocated()/2**30)
