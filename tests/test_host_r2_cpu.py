"""CPU (-m "not gpu"): host-side contracts added in round 2 -- ``consolidated_diff`` add-semantics and the MP split of the
checkpoint reader/writer (util/tensor_parallel.py:133-161,387-423), ``MetaModel.from_pretrained`` (model/meta.py:88-222),
shape-mismatch reporting, the trainer's non-finite guard and the accumulation-cycle behaviour of the DP reducer."""
import json
import os
import types

import pytest
import torch

from a3vlm_amd import checkpoint as ck
from a3vlm_amd.engine_finetune import train_one_epoch
from a3vlm_amd.model.meta import MetaModel


@pytest.fixture()
def mm(golden_dir):
    return MetaModel("llama_ens5", os.path.join(golden_dir, "tiny_params.json"), os.path.join(golden_dir, "tokenizer.model"),
                     with_visual=False, max_seq_len=64)


def _save(d, sd, fmt="consolidated", mp=1):
    os.makedirs(d, exist_ok=True)
    for fn, shard in zip(ck.shard_file_names(fmt, mp), ck.split_tensor_parallel_state_dict(sd, mp)):
        torch.save({"model": shard}, os.path.join(d, fn))
    return d


def test_diff_checkpoint_adds_to_loaded_values_and_sets_new_keys(tmp_path, mm):
    g = torch.Generator().manual_seed(0)
    base = {k: torch.randn(v.shape, generator=g) for k, v in mm.state_dict().items()}
    missing_in_base = "llma.norm.weight"
    first = {k: v for k, v in base.items() if k != missing_in_base}
    delta = {k: 0.25 * torch.randn(v.shape, generator=g) for k, v in base.items() if "layers.1." in k or k == missing_in_base}
    a = _save(str(tmp_path / "base"), first)
    b = _save(str(tmp_path / "diff"), delta, fmt="consolidated_diff", mp=2)       # a TP-sharded diff folder: merged, then added
    assert ck.infer_checkpoint_format_and_mp_size(b) == ("consolidated_diff", 2)
    res = ck.load_tensor_parallel_model_list(mm, [a, b])
    assert res == {"missing_keys": [], "unexpected_keys": []}
    got = mm.state_dict()
    for k, v in base.items():
        if k == missing_in_base:
            want = delta[k]                      # no earlier folder provided it: plainly set
        elif k in delta:
            want = first[k] + delta[k]           # value = old + diff
        else:
            want = first[k]
        assert torch.allclose(got[k], want, atol=1e-6), k
    with pytest.raises(AssertionError):
        ck.load_tensor_parallel_model_list(mm, [b])                                # a diff cannot come first


def test_split_is_the_inverse_of_the_shard_merge(tmp_path, mm):
    sd = {k: v.clone() for k, v in mm.state_dict().items()}
    shards = ck.split_tensor_parallel_state_dict(sd, 2)
    assert shards[0]["llma.layers.0.attention.wq.weight"].shape[0] * 2 == sd["llma.layers.0.attention.wq.weight"].shape[0]
    assert shards[1]["llma.layers.0.attention.wo.weight"].shape[1] * 2 == sd["llma.layers.0.attention.wo.weight"].shape[1]
    assert torch.equal(shards[0]["llma.norm.weight"], shards[1]["llma.norm.weight"])
    d = _save(str(tmp_path / "mp2"), sd, mp=2)
    merged = ck.load_merged_state_dict(d)
    assert set(merged) == set(sd) and all(torch.equal(merged[k], sd[k]) for k in sd)


def test_shape_mismatch_is_reported_not_raised(tmp_path, mm):
    sd = {k: v.clone() for k, v in mm.state_dict().items()}
    sd["llma.output.weight"] = torch.zeros(7, 3)
    d = _save(str(tmp_path / "odd"), sd)
    with pytest.warns(UserWarning, match="shapes differ"):
        res = ck.load_tensor_parallel_model_list(mm, [d])
    assert res["missing_keys"] == ["llma.output.weight"] and res["unexpected_keys"] == []
    assert len(res["mismatched_keys"]) == 1 and "llma.output.weight" in res["mismatched_keys"][0]


def test_from_pretrained_probes_the_folder(tmp_path, mm, capsys):
    args = types.SimpleNamespace(precision="tf32", only_save_trainable=False)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in mm.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    d = ck.save_checkpoint(str(tmp_path), args, mm, None, None, None, epoch=0, iteration=None)
    m2 = MetaModel.from_pretrained(d, max_seq_len=64, dtype=torch.float32, device="cpu")
    assert m2.llama_type == "llama_ens5" and not m2.training and m2.llma.args.dim == mm.llma.args.dim
    assert "all params match perfectly!" in capsys.readouterr().out
    for (k, a), (_, b) in zip(mm.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(ValueError):
        MetaModel.from_pretrained([], device="cpu")
    os.remove(os.path.join(d, "meta.json"))
    with pytest.raises(ValueError, match="llama_type"):
        MetaModel.from_pretrained(d, device="cpu")
    m3 = MetaModel.from_pretrained([d], llama_type="llama_ens5", max_seq_len=64, dtype=torch.float32, device="cpu")
    assert m3.llma.args.n_kv_heads == 2
    with pytest.raises(NotImplementedError):
        MetaModel.from_pretrained("hf://Alpha-VLLM/whatever", device="cpu")


class _Toy(torch.nn.Module):
    """loss = mean((x . w - y)^2); the batch with the marker value produces a NaN loss."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([1.0, -2.0, 0.5]))

    def forward(self, examples, labels, images=None, depth_imgs=None):
        x = examples.float()
        loss = ((x @ self.w - labels.float()) ** 2).mean()
        if bool((examples == 77).any()):
            loss = loss * float("nan")
        return loss, {}


def _args(**kw):
    base = dict(accum_iter=2, lr=1e-2, min_lr=0.0, warmup_epochs=0.0, epochs=1, clip_grad=1.0, print_freq=100, save_iteration_interval=2)
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_non_finite_loss_stops_before_step_and_before_checkpoint():
    g = torch.Generator().manual_seed(3)
    data = [(torch.randint(0, 5, (4, 3), generator=g), torch.randint(0, 5, (4,), generator=g), torch.ones(4, 3)) for _ in range(6)]
    data[2][0][0, 0] = 77                         # micro-step 2 (second accumulation cycle) is poisoned
    m = _Toy()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    saved, logs = [], []
    w_after_first_cycle = None

    def on_save(step):
        saved.append((step, m.w.detach().clone()))
    with pytest.raises(SystemExit) as e:
        train_one_epoch(m, data, opt, epoch=0, start_iter=0, args=_args(), log=logs.append, on_save=on_save)
    assert e.value.code == 1
    assert len(saved) == 1 and saved[0][0] == 1, "only the checkpoint of the clean first cycle may have been written"
    assert torch.isfinite(m.w).all() and torch.equal(m.w.detach(), saved[0][1]), "the poisoned cycle must not reach the weights"
    assert any("non-finite" in s for s in logs)


def test_resumed_epoch_logs_its_last_step():
    g = torch.Generator().manual_seed(4)
    data = [(torch.randint(0, 5, (4, 3), generator=g), torch.randint(0, 5, (4,), generator=g), torch.ones(4, 3)) for _ in range(8)]

    class Resumed(list):                          # the sampler contract: len() is the FULL epoch, iteration starts at start_iter
        def __len__(self):
            return 8
    logs = []
    train_one_epoch(_Toy(), Resumed(data[4:]), torch.optim.SGD(_Toy().parameters(), lr=0.0), epoch=0, start_iter=4,
                    args=_args(save_iteration_interval=0), log=logs.append)
    assert any("[8/8]" in s for s in logs), logs


def test_reducer_runs_on_the_last_micro_step_of_a_cycle_only():
    calls = []

    class Red:
        enabled = True

        def finish(self):
            calls.append(("finish", self.enabled))
    g = torch.Generator().manual_seed(5)
    data = [(torch.randint(0, 5, (4, 3), generator=g), torch.randint(0, 5, (4,), generator=g), torch.ones(4, 3)) for _ in range(6)]
    m = _Toy()
    red = Red()
    seen = []
    orig = m.forward

    def fwd(*a, **k):
        seen.append(red.enabled)
        return orig(*a, **k)
    m.forward = fwd
    train_one_epoch(m, data, torch.optim.SGD(m.parameters(), lr=0.01), epoch=0, start_iter=0, args=_args(accum_iter=3, save_iteration_interval=0),
                    reducer=red, log=lambda s: None)
    assert seen == [False, False, True, False, False, True], "all-reduce overlapped with the backward of the cycle's last micro-step"
    assert calls == [("finish", True), ("finish", True)]
