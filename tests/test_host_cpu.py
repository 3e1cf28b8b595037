"""CPU (-m "not gpu") checks of the boundary and the host logic: the C-ABI library loads and exports
every symbol include/a3vlm_hip.h declares, the ctypes table mirrors the header, the plugin module
tree / trim / tokenizer logic behaves like the reference, and ops refuse CPU tensors loudly."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from a3vlm_amd import lib
    return lib


def header_functions():
    src = open(os.path.join(ROOT, "include", "a3vlm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.findall(r"^\s*(?:int|int64_t)\s+(a3v_\w+)\s*\(([^;]*)\)\s*;", src, flags=re.M)


def test_library_exports_every_header_symbol(built):
    fns = header_functions()
    assert len(fns) >= 18
    lib = built.load()
    for name, _ in fns:
        assert hasattr(lib, name), name
    assert lib.a3v_version() >= 100
    assert {n for n, _ in fns} == set(built.SIGNATURES), "ctypes table and header must list the same entry points"


def test_ctypes_arity_matches_header(built):
    for name, params in header_functions():
        n = 0 if params.strip() in ("", "void") else params.count(",") + 1
        assert n == len(built.SIGNATURES[name][1]), name


def test_epilogue_flags_match_the_header(built):
    """The epilogue / tile flags of the Python host are the header's enum values (A3V_EPI_SWIGLU_BWD was added in round 3)."""
    src = open(os.path.join(ROOT, "include", "a3vlm_hip.h")).read()
    enum = {}
    for name, val in re.findall(r"\b(A3V_EPI_\w+)\s*=\s*([^,/\n]+)", src):
        enum[name] = int(eval(val.strip(), {"__builtins__": {}}))
    for name, val in enum.items():
        py = name[len("A3V_"):]
        assert hasattr(built, py), f"{name} has no counterpart in a3vlm_amd.lib"
        assert getattr(built, py) == val, (name, val, getattr(built, py))
    assert enum["A3V_EPI_SWIGLU_BWD"] == 128


def test_host_side_pure_functions(built):
    lib = built.load()
    assert lib.a3v_gemm_skinny_split(8, 4096, 4096) == 8
    assert lib.a3v_gemm_skinny_split(8, 32000, 4096) >= 1
    assert lib.a3v_gemm_skinny_split(8, 22016, 4096) == 4 and lib.a3v_gemm_skinny_split(8, 256, 64) == 1
    assert lib.a3v_gemm_skinny_ws_bytes(8, 4096, 4096) == 65536 + 64 * 8 * 4096
    assert lib.a3v_attention_scratch_floats(8, 32, 128, 1091) > 0


def test_ops_refuse_cpu_tensors(built):
    from a3vlm_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.rmsnorm(torch.zeros(2, 64), torch.ones(64), torch.zeros(2, 64), 1e-5)


def test_missing_library_fails_loudly(monkeypatch, built):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", "/nonexistent/liba3vlm_hip.so")
    with pytest.raises(RuntimeError, match="not built"):
        built.load()


def test_plugin_contract_and_trim(golden_dir):
    from a3vlm_amd.model.meta import MetaModel
    from a3vlm_amd.model.LLM import llama_ens5 as plugin
    from oracle import ref_cpu
    mm = MetaModel("llama_ens5", os.path.join(golden_dir, "tiny_params.json"), os.path.join(golden_dir, "tokenizer.model"),
                   with_visual=False, max_seq_len=64)
    assert hasattr(plugin, "ModelArgs") and hasattr(plugin, "Transformer")
    assert mm.get_basic_block_classes() == [plugin.TransformerBlock]
    assert all(p.requires_grad for p in mm.parameters())
    ex = torch.arange(30).view(3, 10) + 1
    lab = torch.zeros(3, 10, dtype=torch.long)
    lab[1, 6] = 5
    e1, l1 = mm._trim(ex, lab)
    e2, l2 = ref_cpu.trim_to_last_label(ex, lab)
    assert torch.equal(e1, e2) and torch.equal(l1, l2) and e1.shape[1] == 7
    e3, _ = mm._trim(ex, torch.zeros_like(lab))
    assert e3.shape[1] == 3
    import dataclasses
    cfg = dataclasses.asdict(mm.llma.args)      # what save_checkpoint writes into config.json (misc.py:371-377)
    assert cfg["dim"] == 64 and "rope_scaling" in cfg
    with pytest.raises(ValueError):
        mm.generate("not a list")


def test_cos_sin_table_is_the_reference_complex_table():
    from a3vlm_amd.model.LLM.llama_ens5 import precompute_cos_sin, _ffn_hidden
    from oracle import ref_cpu
    for scaling in (None, 0.5):
        cs = precompute_cos_sin(128, 4096, 10000.0, scaling)
        fc = ref_cpu.precompute_freqs_cis(128, 4096, 10000.0, scaling)
        assert torch.equal(cs[..., 0], fc.real) and torch.equal(cs[..., 1], fc.imag)
    assert _ffn_hidden(4096, 256, None) == 11008 and _ffn_hidden(5120, 256, None) == 13824
    assert ref_cpu.ffn_hidden_dim(4096, 256, None) == 11008


def test_tokenizer_matches_reference_semantics(golden_dir):
    import json
    from a3vlm_amd.model.tokenizer import Tokenizer
    j = json.load(open(os.path.join(golden_dir, "meta_tiny.json")))
    tk = Tokenizer(os.path.join(golden_dir, "tokenizer.model"))
    assert tk.n_words == j["vocab_size"] and tk.bos_id == j["bos"] and tk.eos_id == j["eos"]
    assert tk.need_space_before_segment == j["need_space_before_segment"]
    assert [tk.encode(p, bos=True, eos=False) for p in j["prompts"]] == j["prompt_ids"]
    assert tk.decode(j["gen12_ids"][2]) == j["gen12_text"][2]


def test_param_state_key_counts_optimizer_steps():
    """Weight-image caches are keyed on (Tensor._version, optimizer-step epoch): the epoch moves for every torch.optim optimizer
    step that saw a gradient (fused optimizers do not bump _version), and only for those parameters."""
    import torch
    from a3vlm_amd.util import install_param_epoch_hook, param_state_key
    install_param_epoch_hook()
    install_param_epoch_hook()          # idempotent
    a = torch.nn.Parameter(torch.ones(4))
    b = torch.nn.Parameter(torch.ones(4))
    opt = torch.optim.SGD([a, b], lr=0.1)
    k0a, k0b = param_state_key(a), param_state_key(b)
    a.grad = torch.ones(4)
    opt.step()
    assert param_state_key(a)[1] == k0a[1] + 1 and param_state_key(b) == k0b     # b had no gradient: untouched, key unchanged
    opt.step()
    assert param_state_key(a)[1] == k0a[1] + 2


def test_fused_adamw_refuses_cpu_parameters():
    """No CPU fallback in the optimizer either."""
    import pytest
    import torch
    from a3vlm_amd.optim import FusedAdamW
    p = torch.nn.Parameter(torch.ones(8))
    p.grad = torch.ones(8)
    with pytest.raises(Exception):
        FusedAdamW([p], lr=1e-3).step()


def test_adapter_product_slices_fill_one_block_per_cu():
    """train.skinny_slices: the 256-row form of a3v_gemm_nt_splitk wants S x ceil(M / 256) blocks between half and all of the CUs (its own
    dispatch rule, csrc/a3v_gemm.hip); never more slices than half the k-tiles; small M and non-GPU tensors keep the older rule."""
    from a3vlm_amd.train import skinny_slices
    for M in (2048, 4364, 8728, 8729, 17456, 40000):
        for K in (4096, 11008, 12288, 22016):
            S = skinny_slices(M, 64, K, 256)
            blocks = -(-M // 256) * S
            assert 1 <= S <= 32 and S <= K // 128
            assert blocks <= 256, (M, K, S)
            if S < min(32, K // 128):
                assert 2 * blocks > 256, (M, K, S)      # one slice more would not fit
    assert skinny_slices(8728, 64, 4096, 256) == 7
    assert skinny_slices(8728, 64, 4096, 304) == 8      # another CU count, same rule
    assert skinny_slices(1000, 64, 4096, 256) == 4      # below 2048 rows: 64-row tiles, four slices
    assert skinny_slices(8728, 64, 4096, 0) == 4        # not on a GPU
    assert skinny_slices(100000, 64, 4096, 256) == 1    # more row tiles than CUs: un-split
