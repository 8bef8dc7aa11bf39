"""train.py-shaped loop: schedule on CPU; a short real run with eval + checkpoint save/restore on GPU."""
import json
import numpy as np
import os

import pytest

import train as train_cli
from oracle import vit_ref


def test_schedule_matches_oracle():
    for step in (0, 1, 49, 50, 51, 500, 999, 1000):
        assert abs(train_cli.warmup_cosine(step, 1e-3, 50, 1000) - vit_ref.warmup_cosine_lr(step, 1e-3, 50, 1000)) < 1e-12


def test_cli_flag_names_match_reference():
    """The reference's option names (train.py:130-190)."""
    src = open(train_cli.__file__).read()
    for flag in ("--data_dir", "--img_size", "--num_epochs", "--batch_size", "--label_smoothing", "--augmentation", "--model_name",
                 "--lr", "--weight_decay", "--clip_grad", "--checkpoint_dir", "--seed"):
        assert f'"{flag}"' in src, flag


@pytest.mark.gpu
def test_short_run_eval_checkpoint_restore(tmp_path, capsys):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ck = str(tmp_path / "ck")
    common = ["--model_name", "vit_ti_patch16", "--batch_size", "8", "--steps_per_epoch", "3", "--checkpoint_dir", ck, "--clip_grad", "1.0",
              "--eval_every_epochs", "1", "--save_every_epochs", "1", "--log_every", "1", "--lr", "1e-3"]
    end = train_cli.main(common + ["--num_epochs", "2"])
    assert end == 6
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    train = [l for l in lines if "train/loss" in l]
    evals = [l for l in lines if "eval/loss" in l]
    assert len(train) == 6 and len(evals) == 2
    assert abs(train[0]["train/loss"] - 6.9078) < 1e-2  # zero-init head: ln(1000) at step 1 (SURVEY 8c i)
    assert all(0.0 <= e["eval/top-5-acc"] <= 1.0 for e in evals)
    assert os.path.exists(os.path.join(ck, "checkpoint_6"))  # the reference's file name (flax.training.checkpoints)
    end2 = train_cli.main(common + ["--num_epochs", "3"])  # resumes from step 6
    assert end2 == 9
    lines2 = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{") and "train/loss" in l]
    assert [l["step"] for l in lines2] == [7, 8, 9]


@pytest.mark.gpu
def test_short_run_with_gpu_mix_augment(capsys):
    """Row f-2 in the loop: batch mixup / cutmix on the GPU feed the two-label loss (train.py:83-88); at the zero-init head the
    mixed loss is still ln(1000)."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    end = train_cli.main(["--model_name", "vit_ti_patch16", "--batch_size", "8", "--steps_per_epoch", "4", "--num_epochs", "1", "--log_every", "1",
                          "--lr", "1e-3", "--mixup_alpha", "0.8", "--cutmix_alpha", "1.0", "--eval_every_epochs", "100"])
    assert end == 4
    train = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{") and "train/loss" in l]
    assert len(train) == 4 and abs(train[0]["train/loss"] - 6.9078) < 1e-2
    assert all(np.isfinite(t["train/loss"]) for t in train)


@pytest.mark.gpu
def test_short_run_mlp_mixer(tmp_path, capsys):
    """The same loop drives the MLP-Mixer family: train, eval, Flax-format checkpoint, resume."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ck = str(tmp_path / "ck")
    common = ["--model_name", "mixer_s_patch32", "--batch_size", "8", "--steps_per_epoch", "3", "--checkpoint_dir", ck, "--clip_grad", "1.0",
              "--eval_every_epochs", "1", "--save_every_epochs", "1", "--log_every", "1", "--lr", "1e-3"]
    assert train_cli.main(common + ["--num_epochs", "1"]) == 3
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    train = [l for l in lines if "train/loss" in l]
    assert len(train) == 3 and all(np.isfinite(t["train/loss"]) for t in train)
    assert 5.5 < train[0]["train/loss"] < 8.5  # lecun-normal head (mlp_mixer.py:63): near, not at, ln(1000)
    assert os.path.exists(os.path.join(ck, "checkpoint_3"))
    assert train_cli.main(common + ["--num_epochs", "2"]) == 6  # resumes from step 3
    steps = [json.loads(l)["step"] for l in capsys.readouterr().out.splitlines() if l.startswith("{") and "train/loss" in l]
    assert steps == [4, 5, 6]


@pytest.mark.gpu
def test_short_run_tnt(capsys):
    """... and the TNT family: zero-init head => ln(1000) at step 1, finite afterwards."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    end = train_cli.main(["--model_name", "tnt_b_patch16", "--batch_size", "8", "--steps_per_epoch", "3", "--num_epochs", "1", "--log_every", "1",
                          "--lr", "1e-3", "--clip_grad", "1.0", "--eval_every_epochs", "100"])
    assert end == 3
    train = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{") and "train/loss" in l]
    assert len(train) == 3 and abs(train[0]["train/loss"] - 6.9078) < 1e-2
    assert all(np.isfinite(t["train/loss"]) for t in train)


def test_stochastic_depth_seed_depends_on_seed_rank_and_step():
    """every rank and every step gets its own masks, and a resumed run continues the sequence (the seed is a pure
    function of (--seed, rank, global step))."""
    import importlib

    sd = importlib.import_module("savit_amd.cait_engine").stochastic_depth_seed
    seen = {sd(s, r, k) for s in (0, 42) for r in range(8) for k in range(50)}
    assert len(seen) == 2 * 8 * 50
    assert sd(42, 3, 17) == sd(42, 3, 17) and 0 <= sd(42, 3, 17) < 2 ** 63


@pytest.mark.gpu
def test_cait_masks_differ_across_ranks_and_resume_consistently():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from savit_amd.cait_engine import CaiTEngine, stochastic_depth_seed
    from savit_amd.config import ModelConfig

    mc = ModelConfig(kind="cait", num_layers=2, num_heads=2, embed_dim=128, patch=8, num_classes=16, img_size=32, num_layers_token_only=2,
                     stoch_depth_rate=0.3, layerscale_eps=1e-5)
    eng = CaiTEngine(mc, 64)

    def masks(rank, step):
        eng.set_stochastic_depth(True, seed=stochastic_depth_seed(42, rank, step))
        return eng.sd.clone()

    a = masks(0, 5)
    assert torch.equal(a, masks(0, 5))            # resume: step 5 draws what step 5 drew
    assert not torch.equal(a, masks(1, 5))        # another rank
    assert not torch.equal(a, masks(0, 6))        # the next step
    keep = 1.0 - mc.stoch_depth_rate
    assert set(torch.unique(a).tolist()) <= {0.0, 1.0 / keep} or torch.allclose(torch.unique(a), torch.tensor([0.0, 1.0 / keep], device=a.device))
    assert abs(float((a > 0).float().mean()) - keep) < 0.08  # floor(keep + U) keeps with probability `keep` (stochastic_depth.py:21-23)


@pytest.mark.gpu
def test_short_run_float32_simple_train_mode(tmp_path, capsys):
    """--dtype float32: create_model's default dtype, the arithmetic simple_train.py:72-90 trains config 1 (ViT-Ti/16, batch 8) in -
    through the same loop: train, eval, Flax-format checkpoint, resume; other families / ranks / mixup say why not."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ck = str(tmp_path / "ck")
    common = ["--model_name", "vit_ti_patch16", "--dtype", "float32", "--batch_size", "8", "--steps_per_epoch", "3", "--checkpoint_dir", ck,
              "--clip_grad", "1.0", "--eval_every_epochs", "1", "--save_every_epochs", "1", "--log_every", "1", "--lr", "1e-3"]
    assert train_cli.main(common + ["--num_epochs", "1"]) == 3
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    train = [l for l in lines if "train/loss" in l]
    assert len(train) == 3 and abs(train[0]["train/loss"] - 6.9078) < 1e-4  # zero-init head: ln(1000), to fp32 accuracy
    assert all(np.isfinite(t["train/loss"]) for t in train) and any("eval/loss" in l for l in lines)
    assert os.path.exists(os.path.join(ck, "checkpoint_3"))
    assert train_cli.main(common + ["--num_epochs", "2"]) == 6  # resumes
    # round 6: CaiT trains in fp32 too - the arithmetic the reference always uses for it (cait.py:147-154); as do the MLP-Mixer and TNT; mixup says why not
    capsys.readouterr()
    assert train_cli.main(["--model_name", "cait_xxs_24", "--dtype", "float32", "--batch_size", "2", "--max_steps", "2", "--log_every", "1",
                           "--clip_grad", "1.0", "--lr", "1e-3"]) == 2
    tl = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{") and "train/loss" in l]
    assert len(tl) == 2 and abs(tl[0]["train/loss"] - 6.9078) < 1e-4 and np.isfinite(tl[1]["train/loss"])
    assert train_cli.main(["--model_name", "mixer_s_patch32", "--dtype", "float32", "--batch_size", "2", "--max_steps", "2", "--log_every", "1",
                           "--clip_grad", "1.0", "--lr", "1e-3"]) == 2
    tl = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{") and "train/loss" in l]
    assert len(tl) == 2 and 5.5 < tl[0]["train/loss"] < 8.5 and np.isfinite(tl[1]["train/loss"])
    capsys.readouterr()
    assert train_cli.main(["--model_name", "tnt_s_patch16", "--dtype", "float32", "--batch_size", "2", "--max_steps", "2", "--log_every", "1",
                           "--clip_grad", "1.0", "--lr", "1e-3"]) == 2
    tl = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{") and "train/loss" in l]
    assert len(tl) == 2 and all(np.isfinite(t["train/loss"]) for t in tl)
    with pytest.raises(SystemExit, match="bfloat16"):
        train_cli.main(["--model_name", "vit_ti_patch16", "--dtype", "float32", "--batch_size", "2", "--max_steps", "1", "--mixup_alpha", "0.8"])
