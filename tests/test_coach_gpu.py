"""End-to-end on the GPU through the reference-shaped surface: config -> Coach -> train -> checkpoints that
load back (tiny SD shape family, synthetic image folder, gradient accumulation 2)."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def test_coach_mode0_trains_and_saves(tmp_path):
    from view_neti_amd.compat import config as C
    from view_neti_amd.compat.checkpoint_handler import CheckpointHandler
    from view_neti_amd.compat.coach import Coach
    root = tmp_path / "toys"
    root.mkdir()
    rng = np.random.RandomState(0)
    for i in range(3):
        Image.fromarray(rng.randint(0, 255, (90, 120, 3), dtype=np.uint8)).save(root / f"{i}.png")
    cfg = C.parse(C.RunConfig, [
        "--data.train_data_dir", str(root), "--data.placeholder_object_token", "<toy>", "--data.resolution", "64",
        "--data.dataloader_num_workers", "0", "--model.word_embedding_dim", "128", "--model.arch_view_net", "15",
        "--model.arch_view_disable_tl", "False", "--model.arch_mlp_hidden_dims", "64", "--model.use_nested_dropout",
        "False", "--optim.max_train_steps", "4", "--optim.train_batch_size", "2",
        "--optim.gradient_accumulation_steps", "2", "--optim.mixed_precision", "fp16", "--log.save_steps", "2",
        "--log.exp_dir", str(tmp_path / "out"), "--log.exp_name", "run"])
    cfg.log.exp_dir = cfg.log.exp_dir / cfg.log.exp_name
    cfg.log.logging_dir = cfg.log.exp_dir / cfg.log.logging_dir
    torch.manual_seed(cfg.seed)
    coach = Coach(cfg)
    p0 = coach.engine.params.clone()
    coach.train()
    out = cfg.log.exp_dir
    assert (out / "config.yaml").exists() and (out / "logs" / "log.txt").exists()
    for name in ("learned_embeds-steps-2.bin", "mapper-steps-2_object.pt", "learned_embeds-final.bin",
                 "mapper-final_object.pt"):
        assert (out / name).exists(), name
    assert coach.engine.opt_step.item() == 4 and not torch.equal(p0, coach.engine.params)
    # lr rule of coach.py:728-733: 1e-3 * accum(2) * bs(2) * world(1)
    assert abs(float(coach.engine.hyper[0]) - 4e-3) < 1e-9
    tok_id = coach.placeholder_object_token_ids[0]
    cfg2, lookup = CheckpointHandler.load_mapper(out / "mapper-final_object.pt", "object", ["<toy>"], [tok_id])
    from view_neti_amd.engine.text import flatten_mapper_state
    flat = flatten_mapper_state(lookup[tok_id].mapper_state())
    assert torch.allclose(flat, coach.engine.params.cpu(), atol=0, rtol=0)
    assert abs(cfg2.model.target_norm_object - cfg.model.target_norm_object) < 1e-6
    emb = torch.load(out / "learned_embeds-final.bin")
    assert list(emb) == ["<toy>"] and emb["<toy>"].shape == (128,)
