"""Entry point with the reference's CLI (scripts/train.py:19-35):

    python scripts/train.py --config_path input_configs/train.yaml --log.exp_name run0 --optim.train_batch_size 4

multi-GPU: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/train.py ...
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from view_neti_amd import parallel
from view_neti_amd.compat import config as cfgmod
from view_neti_amd.compat.coach import Coach


def fixseed(seed: int):
    """utils/fixseed.py:6-10"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def prepare_directories(cfg):
    cfg.log.exp_dir = cfg.log.exp_dir / cfg.log.exp_name
    if os.path.exists(cfg.log.exp_dir) and not cfg.log.overwrite_ok:
        raise ValueError(f"Experiment folder already exists and overwrite_ok=False: [{cfg.log.exp_dir}] "
                         f"to overwrite the old experiment, add --log.overwrite_ok")
    cfg.log.logging_dir = cfg.log.exp_dir / cfg.log.logging_dir


@cfgmod.wrap()
def main(cfg: cfgmod.RunConfig):
    rank, world, local = parallel.world_info()
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    fixseed(cfg.seed)
    prepare_directories(cfg)
    Coach(cfg).train()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
