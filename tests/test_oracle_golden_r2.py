"""CPU: the oracle and the host-side sampling code against the round-2 fixtures captured from the reference by
``oracle/gen_golden_r2.py`` -- head_dim-128 / GQA decoder (``decoder_mid.npz``), linear RoPE scaling pinned through the
reference's HF exporter + transformers (``rope_scaling.npz``) and ``MetaModel.sample_top_p`` (``sampling.json``)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from oracle.gen_golden import TINY, checksum
from oracle.gen_golden_r2 import MID


@pytest.fixture(scope="module")
def mid(golden_dir):
    j = json.load(open(os.path.join(golden_dir, "mid_meta.json")))
    return np.load(os.path.join(golden_dir, "decoder_mid.npz")), j


def test_mid_decoder_forward_and_cached_inference(mid):
    fx, j = mid
    args = ref_cpu.OracleArgs(vocab_size=j["vocab_size"], **MID)
    sd = ref_cpu.make_decoder_weights(args, seed=21, std=0.04)
    assert abs(checksum(sd) - float(fx["weight_checksum"])) < 1e-6 * float(fx["weight_checksum"])
    d = ref_cpu.OracleDecoder(args, sd)
    ex = torch.from_numpy(fx["examples"])
    np.testing.assert_allclose(d.forward(ex).numpy(), fx["logits"], atol=3e-5, rtol=1e-5)
    P = 41
    got = [d.forward_inference(ex[:, :P], 0)]
    for t in range(P, P + 6):
        got.append(d.forward_inference(ex[:, t:t + 1], t))
    np.testing.assert_allclose(torch.stack(got).numpy(), fx["inf_logits"], atol=3e-5, rtol=1e-5)
    # the cached path reproduces the teacher-forced forward (same positions) -- the property the GPU tests use at full size
    np.testing.assert_allclose(fx["inf_logits"][0], fx["logits"][:, P - 1], atol=3e-5)
    # the reference's own bf16 run deviates from its fp32 run by this much: the scale the bf16 GPU tolerances are stated against
    dev = np.abs(fx["logits_bf16"] - fx["logits"]).max() / np.abs(fx["logits"]).max()
    assert 1e-3 < dev < 5e-2, dev


def test_rope_scaling_matches_hf_linear_scaling(golden_dir):
    fx = np.load(os.path.join(golden_dir, "rope_scaling.npz"))
    s = float(fx["rope_scaling"])
    V = fx["hf_logits"].shape[-1]
    args = ref_cpu.OracleArgs(vocab_size=V, rope_scaling=s, **TINY)
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=V, **TINY), seed=0, std=0.08)
    assert abs(checksum(sd) - float(fx["weight_checksum"])) < 1e-6 * float(fx["weight_checksum"])
    got = ref_cpu.OracleDecoder(args, sd).forward(torch.from_numpy(fx["examples"])).numpy()
    np.testing.assert_allclose(got, fx["hf_logits"], atol=1e-4)       # transformers: positions / factor, factor = 1 / s
    np.testing.assert_allclose(got, fx["ref_logits"], atol=2e-5)      # the reference class on the restated table
    plain = ref_cpu.OracleDecoder(ref_cpu.OracleArgs(vocab_size=V, **TINY), sd).forward(torch.from_numpy(fx["examples"])).numpy()
    assert np.abs(plain - fx["hf_logits"]).max() > 1e-2               # the scaling is not a no-op on this input
    # the table the HIP kernels consume is built by the plugin's host code: same angles
    from a3vlm_amd.model.LLM.llama_ens5 import precompute_cos_sin
    fc = ref_cpu.precompute_freqs_cis(16, 128, 10000.0, s)
    cs = precompute_cos_sin(16, 128, 10000.0, s)
    np.testing.assert_allclose(cs[..., 0].numpy(), fc.real.numpy(), atol=1e-6)
    np.testing.assert_allclose(cs[..., 1].numpy(), fc.imag.numpy(), atol=1e-6)


def test_sample_top_p_matches_reference(golden_dir):
    """model/meta.py:568-583: nucleus = smallest prefix of the sorted probabilities whose mass BEFORE the token is <= p;
    renormalise; one multinomial draw per row (same torch RNG stream -> same ids as the reference)."""
    from a3vlm_amd.model.meta import MetaModel
    cases = json.load(open(os.path.join(golden_dir, "sampling.json")))["cases"]
    for c in cases:
        gg = torch.Generator().manual_seed(c["seed"])
        logits = torch.randn(c["rows"], c["vocab"], generator=gg) * 3
        probs = torch.softmax(logits / c["temperature"], dim=-1)
        torch.manual_seed(c["torch_seed"])
        nt = MetaModel.sample_top_p(None, probs.clone(), c["p"])
        assert nt.shape == (c["rows"], 1) and nt.reshape(-1).tolist() == c["sampled"]
        for r in range(c["rows"]):
            assert int(nt[r]) in c["nucleus"][r]
        # deterministic part, independently: every draw of many stays inside the reference's nucleus and covers it when it is small
        torch.manual_seed(1)
        seen = [set() for _ in range(c["rows"])]
        for _ in range(200):
            d = MetaModel.sample_top_p(None, probs.clone(), c["p"]).reshape(-1).tolist()
            for r, t in enumerate(d):
                seen[r].add(t)
        for r in range(c["rows"]):
            assert seen[r] <= set(c["nucleus"][r])
            if len(c["nucleus"][r]) <= 2:
                assert seen[r] == set(c["nucleus"][r]) or len(c["nucleus"][r]) == 2


def test_oracle_nucleus_is_the_references(golden_dir):
    """oracle.top_p_nucleus / sample_top_p_at (the checker of the device sampler a3v_sample_top_p) against what the reference's
    sample_top_p produced (sampling.json): the kept set is exactly the reference's nucleus, every inverse-CDF draw lies in it, and
    over a grid of uniforms the draws follow the renormalised probabilities."""
    cases = json.load(open(os.path.join(golden_dir, "sampling.json")))["cases"]
    for c in cases:
        gg = torch.Generator().manual_seed(c["seed"])
        logits = torch.randn(c["rows"], c["vocab"], generator=gg) * 3
        probs = torch.softmax(logits / c["temperature"], dim=-1)
        ps, idx = ref_cpu.top_p_nucleus(probs, c["p"])
        for r in range(c["rows"]):
            kept = set(idx[r][ps[r] > 0].tolist())
            assert kept == set(c["nucleus"][r]), (c["seed"], r)
        # the reference's own draws are inside it (same fixture), and so is every inverse-CDF draw
        grid = (torch.arange(400, dtype=torch.float32) + 0.5) / 400
        counts = torch.zeros(c["rows"], c["vocab"])
        for u in grid.tolist():
            d = ref_cpu.sample_top_p_at(probs, c["p"], torch.full((c["rows"],), u))
            for r, t in enumerate(d.tolist()):
                assert t in c["nucleus"][r]
                counts[r, t] += 1
        want = torch.zeros(c["rows"], c["vocab"]).scatter_(1, idx, ps)
        assert float((counts / 400 - want).abs().max()) < 1.0 / 400 + 1e-6          # a CDF on a 400-point grid


def test_deep_decoder_oracle_matches_the_reference_at_every_depth(golden_dir):
    """Round-3 fixture (oracle/gen_golden_r3.py, generated by running the reference): a 12-layer head_dim-128 GQA decoder and its
    4- and 8-layer truncations -- the oracle reproduces the reference's fp32 logits at each depth, the cached-inference logits and
    the reference's greedy ids; the reference's OWN bf16-vs-fp32 deviation grows with depth (the yardstick of the GPU test)."""
    from oracle.gen_golden_r3 import DEEP
    fx = np.load(os.path.join(golden_dir, "decoder_deep.npz"))
    j = json.load(open(os.path.join(golden_dir, "deep_meta.json")))
    V = j["vocab_size"]
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=V, **DEEP), seed=31, std=0.04)
    assert abs(checksum(sd) - float(fx["weight_checksum"])) < 1e-6 * float(fx["weight_checksum"])
    ex = torch.from_numpy(fx["examples"])
    pos = fx["positions"].tolist()
    devs = []
    for depth in fx["depths"].tolist():
        sub = {k: v for k, v in sd.items() if not k.startswith("layers.") or int(k.split(".")[1]) < depth}
        d = ref_cpu.OracleDecoder(ref_cpu.OracleArgs(vocab_size=V, **{**DEEP, "n_layers": depth}), sub)
        np.testing.assert_allclose(d.forward(ex)[:, pos].numpy(), fx[f"logits_L{depth}"], atol=1e-4, rtol=2e-5)
        devs.append(float(np.abs(fx[f"logits_bf16_L{depth}"] - fx[f"logits_L{depth}"]).max() / np.abs(fx[f"logits_L{depth}"]).max()))
    assert devs[0] < devs[-1] < 0.05, devs               # error growth with depth is real and bounded: 4 -> 12 layers
    d = ref_cpu.OracleDecoder(ref_cpu.OracleArgs(vocab_size=V, **DEEP), sd)
    P = j["P"]
    got = [d.forward_inference(ex[:, :P], 0)]
    for t in range(P, P + 4):
        got.append(d.forward_inference(ex[:, t:t + 1], t))
    np.testing.assert_allclose(torch.stack(got).numpy(), fx["inf_logits"], atol=1e-4, rtol=2e-5)
