"""-m gpu: the fine-tuning entry point end to end on one GPU (tiny model, synthetic data): epochs, gradient
accumulation, checkpoint layout on disk, resume."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env=None):
    e = dict(os.environ)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "a3vlm_amd.main_finetune"] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("precision", ["tf32", "bf16"])
def test_main_finetune_synthetic(tmp_path, precision):
    gd = os.path.join(ROOT, "tests", "golden")
    extra = tmp_path / "vit.json"
    extra.write_text(json.dumps(dict(vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=1)))
    out = tmp_path / "out"
    base = ["--llama_type", "llama_ens5", "--llama_config", os.path.join(gd, "tiny_params.json"), str(extra),
            "--tokenizer_path", os.path.join(gd, "tokenizer.model"), "--batch_size", "2", "--accum_iter", "2",
            "--warmup_epochs", "0.5", "--lr", "1e-3", "--min_lr", "0", "--clip_grad", "8", "--weight_decay", "0",
            "--max_words", "120", "--precision", precision, "--output_dir", str(out), "--synthetic", "24", "--num_workers", "0",
            "--dialog", "--data_parallel", "sdp", "--model_parallel_size", "1", "--checkpointing"]
    log = run(base + ["--epochs", "2"])
    assert "closs" in log
    for ep in ("epoch0", "epoch1"):
        files = set(os.listdir(out / ep))
        assert {"consolidated.00-of-01.model.pth", "consolidated.00-of-01.optimizer.pth", "consolidated.00-of-01.other.pth",
                "config.json", "meta.json", "tokenizer.model", "rank-specific-00000-of-00001.pth"} <= files
    lines = [json.loads(x) for x in open(out / "log.txt")]
    assert [l["epoch"] for l in lines] == [0, 1] and all(0 < l["train_closs"] < 20 for l in lines)
    log2 = run(base + ["--epochs", "3", "--resume", str(out)])
    assert "resume:" in log2 and os.path.isdir(out / "epoch2")
    lines = [json.loads(x) for x in open(out / "log.txt")]
    assert [l["epoch"] for l in lines] == [0, 1, 2]


def test_main_finetune_dialog_dataset(tmp_path):
    """--data_config: the dialog dataset (image_text + text groups), PIL transform and FinetuneDistSampler feed the trainer."""
    from oracle.gen_golden import dialog_yaml
    gd = os.path.join(ROOT, "tests", "golden")
    extra = tmp_path / "vit.json"
    extra.write_text(json.dumps(dict(vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=1)))
    out = tmp_path / "out"
    log = run(["--llama_type", "llama_ens5", "--llama_config", os.path.join(gd, "tiny_params.json"), str(extra),
               "--tokenizer_path", os.path.join(gd, "tokenizer.model"), "--batch_size", "2", "--accum_iter", "1", "--epochs", "1",
               "--warmup_epochs", "0.5", "--lr", "1e-3", "--min_lr", "0", "--clip_grad", "8", "--weight_decay", "0.02",
               "--max_words", "220", "--precision", "bf16", "--output_dir", str(out), "--data_config", dialog_yaml(str(tmp_path)),
               "--image_transform", "padded_resize", "--num_workers", "0", "--dialog", "--model_parallel_size", "1"])
    assert "closs" in log and "total length: 13" in log
    line = json.loads(open(out / "log.txt").read().strip().splitlines()[-1])
    assert 0 < line["train_closs"] < 20


def test_main_finetune_lora_only_save_trainable(tmp_path):
    """--llama_type llama_ens5_peft: adapters + norms train, the base is frozen; --only_save_trainable writes just those keys
    (util/misc.py:340-350) and the run resumes / reloads on top of a base checkpoint by name."""
    import torch
    gd = os.path.join(ROOT, "tests", "golden")
    extra = tmp_path / "peft.json"
    extra.write_text(json.dumps(dict(vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=1, lora_rank=8)))
    out = tmp_path / "out"
    log = run(["--llama_type", "llama_ens5_peft", "--llama_config", os.path.join(gd, "tiny_params.json"), str(extra),
               "--tokenizer_path", os.path.join(gd, "tokenizer.model"), "--batch_size", "2", "--accum_iter", "1", "--epochs", "1",
               "--warmup_epochs", "0.5", "--lr", "1e-3", "--min_lr", "0", "--clip_grad", "8", "--weight_decay", "0.0",
               "--max_words", "120", "--precision", "bf16", "--output_dir", str(out), "--synthetic", "16", "--num_workers", "0",
               "--dialog", "--model_parallel_size", "1", "--only_save_trainable"])
    assert "closs" in log
    sd = torch.load(out / "epoch0" / "consolidated.00-of-01.model.pth", weights_only=False)["model"]
    keys = set(sd)
    assert any("lora_a" in k for k in keys) and any("attention_norm" in k for k in keys) and "llma.visual_proj.0.weight" in keys
    assert not any(k.endswith("attention.wq.weight") for k in keys) and "llma.tok_embeddings.weight" not in keys
    assert json.load(open(out / "epoch0" / "meta.json"))["llama_type"] == "llama_ens5_peft"
