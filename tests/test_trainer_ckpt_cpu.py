"""CPU (-m "not gpu"): checkpoint layout (vs the reference's own loader, fixture ckpt_tiny.json) and the
train_one_epoch control flow (accumulation boundaries, LR schedule, clipping) on a plain torch module."""
import json
import os
import types

import pytest
import torch

from a3vlm_amd import checkpoint as ck
from a3vlm_amd.engine_finetune import train_one_epoch
from a3vlm_amd.model.meta import MetaModel
from oracle import ref_cpu
from oracle.gen_golden import TINY


@pytest.fixture()
def mm(golden_dir):
    return MetaModel("llama_ens5", os.path.join(golden_dir, "tiny_params.json"), os.path.join(golden_dir, "tokenizer.model"),
                     with_visual=False, max_seq_len=64)


def _write_shards(d, full, fmt):
    names = ck.shard_file_names(fmt, 2)
    for r, fn in enumerate(names):
        shard = {}
        for k, v in full.items():
            dd = ck.merge_dim(k)
            t = torch.chunk(v, 2, dd)[r].clone() if dd >= 0 else v.clone()
            shard[k if fmt == "consolidated" else k[len("llma."):]] = t
        torch.save({"model": shard} if fmt == "consolidated" else shard, os.path.join(d, fn))


@pytest.mark.parametrize("fmt", ["consolidated", "meta_ori"])
def test_two_shard_checkpoint_merges_like_the_reference(golden_dir, tmp_path, mm, fmt):
    fix = json.load(open(os.path.join(golden_dir, "ckpt_tiny.json")))
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=mm.tokenizer.n_words, **TINY), seed=fix["seed"], std=0.05)
    full = {"llma." + k: v for k, v in sd.items()}
    _write_shards(str(tmp_path), full, fmt)
    assert ck.infer_checkpoint_format_and_mp_size(str(tmp_path)) == (fmt, 2)
    for p in mm.parameters():
        p.data.zero_()
    res = ck.load_tensor_parallel_model_list(mm, [str(tmp_path)])
    assert res == fix[fmt]["load_result"] == {"missing_keys": [], "unexpected_keys": []}
    got = mm.state_dict()
    for k, v in full.items():
        assert torch.equal(got[k], v), k
        assert abs(float(got[k].double().sum()) - fix[fmt]["sums"][k]) < 1e-9 * max(1.0, fix[fmt]["abs_sums"][k])


def test_unknown_and_missing_keys_are_reported(tmp_path, mm):
    sd = {k: v.clone() for k, v in mm.state_dict().items()}
    sd.pop("llma.norm.weight")
    sd["llma.qformer.some.weight"] = torch.zeros(3)          # out-of-scope encoder key: reported, ignored
    torch.save({"model": sd}, os.path.join(str(tmp_path), "consolidated.00-of-01.model.pth"))
    res = ck.load_tensor_parallel_model_list(mm, [str(tmp_path)])
    assert res == {"missing_keys": ["llma.norm.weight"], "unexpected_keys": ["llma.qformer.some.weight"]}
    with pytest.raises(NotImplementedError):
        ck.infer_checkpoint_format_and_mp_size(os.path.join(str(tmp_path), "nope"))


def test_save_checkpoint_layout_round_trip(tmp_path, mm):
    args = types.SimpleNamespace(precision="bf16", only_save_trainable=False, lr=1e-3)
    opt = torch.optim.AdamW(mm.parameters(), lr=1e-3)
    d = ck.save_checkpoint(str(tmp_path), args, mm, opt, None, None, epoch=1, iteration=39, rank=0, world_size=1)
    assert os.path.basename(d) == "epoch1-iter39"
    assert sorted(os.listdir(d)) == sorted(["consolidated.00-of-01.model.pth", "consolidated.00-of-01.optimizer.pth",
                                            "consolidated.00-of-01.other.pth", "config.json", "meta.json", "tokenizer.model",
                                            "rank-specific-00000-of-00001.pth"])
    assert json.load(open(os.path.join(d, "meta.json"))) == {"llama_type": "llama_ens5"}
    cfg = json.load(open(os.path.join(d, "config.json")))
    assert cfg["dim"] == 64 and cfg["n_kv_heads"] == 2
    blob = torch.load(os.path.join(d, "consolidated.00-of-01.model.pth"), weights_only=False)
    assert set(blob) == {"model"} and all(k.startswith("llma.") for k in blob["model"]) and all(v.dtype == torch.bfloat16 for v in blob["model"].values())
    ck.save_checkpoint(str(tmp_path), args, mm, None, None, None, epoch=1, iteration=None)
    ck.save_checkpoint(str(tmp_path), args, mm, None, None, None, epoch=0, iteration=99)
    assert os.path.basename(ck.latest_checkpoint_dir(str(tmp_path))) == "epoch1"
    before = {k: v.clone() for k, v in mm.state_dict().items()}
    for p in mm.parameters():
        p.data.zero_()
    ck.load_tensor_parallel_model_list(mm, [d])
    for k, v in mm.state_dict().items():
        assert torch.equal(v, before[k].to(torch.bfloat16).to(v.dtype)), k
    # a MetaModel can be rebuilt from the folder alone (config.json + meta.json + tokenizer.model): from_pretrained's probing
    m2 = MetaModel(json.load(open(os.path.join(d, "meta.json")))["llama_type"], [{k: v for k, v in cfg.items() if k not in ("max_seq_len", "max_batch_size", "vocab_size")}],
                   os.path.join(d, "tokenizer.model"), with_visual=False, max_seq_len=64)
    assert ck.load_tensor_parallel_model_list(m2, [d]) == {"missing_keys": [], "unexpected_keys": []}


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([1.0, -2.0, 0.5]))

    def forward(self, examples, labels, images=None, depth_imgs=None):
        x = examples.float()
        return ((x @ self.w - labels.float()) ** 2).mean(), {}


def test_train_one_epoch_accumulation_lr_and_clip():
    torch.manual_seed(0)
    data = [(torch.randn(4, 3), torch.randn(4), torch.ones(4)) for _ in range(8)]
    args = types.SimpleNamespace(accum_iter=2, lr=0.1, min_lr=0.0, warmup_epochs=0.5, epochs=2, clip_grad=0.05, print_freq=1,
                                 save_iteration_interval=0)
    m = _Toy()
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    logs = []
    train_one_epoch(m, data, opt, epoch=0, start_iter=0, args=args, log=logs.append)
    # manual replay of engine_finetune.py semantics
    w = torch.tensor([1.0, -2.0, 0.5], requires_grad=True)
    for b in range(0, 8, 2):
        lr = 0.1 * (b / 8) / 0.5 if (b / 8) < 0.5 else 0.1 * 0.5 * (1 + torch.cos(torch.tensor(torch.pi * ((b / 8) - 0.5) / 1.5))).item()
        g = torch.zeros(3)
        for j in (b, b + 1):
            x, y, _ = data[j]
            loss = ((x @ w - y) ** 2).mean() / 2
            g += torch.autograd.grad(loss, w)[0]
        norm = g.norm()
        g = g * min(1.0, 0.05 / (float(norm) + 1e-6))
        w = (w - lr * g).detach().requires_grad_(True)
    assert torch.allclose(m.w.detach(), w.detach(), atol=1e-6)
    assert len(logs) == 4 and "closs" in logs[0]
