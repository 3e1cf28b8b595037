"""-m gpu: the sampled branch of generation on the device (reference model/meta.py:456-460, 568-583; the eval
script's DEFAULT recipe is temperature 0.1 / top-p 0.75, eval_affordance_v2.py:46-49).

What can be pinned: the reference's nucleus (tests/golden/sampling.json, captured from the reference's sample_top_p) and, given the
uniform numbers of the draw, the token the inverse CDF of the renormalised nucleus reaches -- the oracle restates both
(oracle/ref_cpu.py: top_p_nucleus, sample_top_p_at; pinned on CPU in tests/test_oracle_golden_r2.py).  torch.multinomial's own
random stream differs between CPU and GPU builds of torch and is not part of the contract."""
import json
import os
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"

from a3vlm_amd import ops  # noqa: E402
from oracle import ref_cpu  # noqa: E402


def _boundary_ok(probs_row, p, u, got, want, eps=2e-6):
    """got == want, or u lies within eps (in probability mass) of the CDF interval of ``got`` in the oracle's renormalised nucleus:
    the device sums fp32 masses in histogram order, torch.cumsum in index order, so two CDFs of 32000 terms differ by ~1e-5 and a
    uniform that close to a step may land on either side (among tokens of probability 1e-5 that can be a few ranks apart)."""
    if got == want:
        return True
    ps, idx = ref_cpu.top_p_nucleus(probs_row[None], p)
    ids = idx[0].tolist()
    if got not in ids:
        return False
    a = ids.index(got)
    if float(ps[0][a]) <= 0:
        # top_p >= 1: the reference's fp32 cumsum drifts past 1.0 somewhere in the tail and its `cumsum - p > top_p` test then cuts
        # everything behind that point (an artefact of its summation order); a token from that tail is a legitimate draw of "keep all"
        return p >= 1.0 and float(probs_row[got]) > 0 and float(probs_row[idx[0][ps[0] <= 0]].sum()) < 1e-3
    cdf = torch.cumsum(ps[0].double(), -1)
    lo = float(cdf[a - 1]) if a > 0 else 0.0
    return lo - eps <= u <= float(cdf[a]) + eps


def test_device_sampler_stays_in_the_references_nucleus_and_matches_the_inverse_cdf(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "sampling.json")))["cases"]
    for c in cases:
        gg = torch.Generator().manual_seed(c["seed"])
        logits = torch.randn(c["rows"], c["vocab"], generator=gg) * 3
        probs = torch.softmax(logits / c["temperature"], dim=-1)
        ld = logits.to(DEV)
        out = torch.empty(c["rows"], dtype=torch.long, device=DEV)
        ug = torch.Generator().manual_seed(5)
        for _ in range(64):
            u = torch.rand(c["rows"], generator=ug)
            ops.sample_top_p(ld, c["temperature"], c["p"], u.to(DEV), out)
            got = out.cpu().tolist()
            want = ref_cpu.sample_top_p_at(probs, c["p"], u).tolist()
            for r in range(c["rows"]):
                assert got[r] in c["nucleus"][r], (c["seed"], r, got[r])
                assert _boundary_ok(probs[r], c["p"], float(u[r]), got[r], want[r]), (c["seed"], r, float(u[r]), got[r], want[r])


@pytest.mark.parametrize("T,p", [(0.1, 0.75), (1.0, 0.75), (1.0, 0.95), (0.7, 1.0), (2.0, 0.5)])
def test_device_sampler_full_vocabulary_with_ties(T, p):
    """V = 32000, logits rounded to bf16 (what the bf16 LM head hands over: thousands of exactly equal values, ties ranked by token
    index), 8 rows, a grid of uniforms incl. 0 and the largest float below 1; plus a row whose nucleus is one token and a row with
    a uniform distribution (every token tied)."""
    g = torch.Generator().manual_seed(11)
    V, B = 32000, 8
    logits = (torch.randn(B, V, generator=g) * 2.5).bfloat16().float()
    logits[1, 777] = 60.0                       # one-token nucleus
    logits[2] = 0.25                            # all tied
    probs = torch.softmax(logits / T, dim=-1)
    ld = logits.to(DEV)
    out = torch.empty(B, dtype=torch.long, device=DEV)
    us = [0.0, 0.99999994, 0.5] + torch.rand(29, generator=g).tolist()
    bad = 0
    for uval in us:
        u = torch.full((B,), uval)
        ops.sample_top_p(ld, T, p, u.to(DEV), out)
        got = out.cpu().tolist()
        want = ref_cpu.sample_top_p_at(probs, p, u).tolist()
        assert got[1] == 777
        for r in range(B):
            if r == 2:
                # uniform row: kept = the first ceil-ish(p V) tokens by index; the draw is u * kept, up to float rounding of 1/V sums
                n_keep = int((ref_cpu.top_p_nucleus(probs[2:3], p)[0] > 0).sum())
                assert abs(got[r] - uval * n_keep) <= 2 + 1e-4 * n_keep and got[r] < max(n_keep, 1) + 1, (uval, got[r], n_keep)
                continue
            if not _boundary_ok(probs[r], p, uval, got[r], want[r], eps=4e-5 if p < 1.0 else 1e-3):
                bad += 1
    assert bad == 0
    # reproducible: same inputs, same ids (per-wave histograms merged in a fixed order)
    u = torch.rand(B, generator=g).to(DEV)
    a = ops.sample_top_p(ld, T, p, u, out).clone()
    for _ in range(3):
        assert torch.equal(ops.sample_top_p(ld, T, p, u, out), a)


def test_device_sampler_lds_cached_and_recomputing_forms_agree_and_large_vocabularies_work():
    """Round 5: e_i is kept in LDS for V <= 32768 (one expf per element instead of one per pass); larger vocabularies (and
    A3V_SAMPLER_CACHE=0) recompute it per pass with the same expression, the same keys and the same summation order: identical ids.
    V = 50000 (no cache possible) against the oracle's inverse CDF."""
    from a3vlm_amd import lib
    g = torch.Generator().manual_seed(21)
    B = 8
    for V, T, p in ((32000, 0.1, 0.75), (32768, 1.0, 0.9), (1003, 0.7, 0.5)):
        logits = (torch.randn(B, V, generator=g) * 2.0).bfloat16().float().to(DEV)
        out = torch.empty(B, dtype=torch.long, device=DEV)
        for _ in range(6):
            u = torch.rand(B, generator=g).to(DEV)
            a = ops.sample_top_p(logits, T, p, u, out).clone()
            with lib.env(A3V_SAMPLER_CACHE="0"):
                b = ops.sample_top_p(logits, T, p, u, out).clone()
            assert torch.equal(a, b), (V, a.tolist(), b.tolist())
    V, T, p = 50000, 0.8, 0.9
    logits = torch.randn(B, V, generator=g) * 2.0
    probs = torch.softmax(logits / T, dim=-1)
    ld = logits.to(DEV)
    out = torch.empty(B, dtype=torch.long, device=DEV)
    for _ in range(8):
        u = torch.rand(B, generator=g)
        got = ops.sample_top_p(ld, T, p, u.to(DEV), out).cpu().tolist()
        want = ref_cpu.sample_top_p_at(probs, p, u).tolist()
        for r in range(B):
            assert _boundary_ok(probs[r], p, float(u[r]), got[r], want[r], eps=4e-5), (r, float(u[r]), got[r], want[r])


def test_device_sampler_follows_the_renormalised_distribution():
    """4000 draws of one row at T = 1, top-p 0.9 over a 50-token head: empirical frequencies match the renormalised nucleus."""
    g = torch.Generator().manual_seed(3)
    V = 1000
    logits = torch.randn(1, V, generator=g)
    logits[0, :50] += 4.0
    probs = torch.softmax(logits, -1)
    ps, idx = ref_cpu.top_p_nucleus(probs, 0.9)
    want = torch.zeros(V).scatter_(0, idx[0], ps[0])
    N = 4000
    rows = logits.expand(N, V).contiguous().to(DEV)
    out = torch.empty(N, dtype=torch.long, device=DEV)
    ops.sample_top_p(rows, 1.0, 0.9, torch.rand(N, generator=g).to(DEV), out)
    freq = torch.bincount(out.cpu(), minlength=V).float() / N
    assert float(freq[want == 0].sum()) == 0.0
    assert float((freq - want).abs().max()) < 0.02


def _tiny_meta(tmp_path, dtype):
    from a3vlm_amd.model.meta import MetaModel
    from oracle.gen_golden import TINY
    gd = os.path.join(ROOT, "tests", "golden")
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps({k: v for k, v in TINY.items() if k != "max_seq_len"}))
    mm = MetaModel("llama_ens5", str(cfgp), os.path.join(gd, "tokenizer.model"), with_visual=False, max_seq_len=256)
    V = mm.tokenizer.n_words
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=V, **TINY), seed=0, std=0.08)
    mm.llma.load_state_dict(sd)
    mm.to(dtype).to(DEV)
    dec = ref_cpu.OracleDecoder(ref_cpu.OracleArgs(vocab_size=V, **{**TINY, "max_seq_len": 256}), sd)
    return mm, dec


@pytest.mark.parametrize("T,p", [(0.1, 0.75), (1.0, 0.75), (1.3, 0.95)])
def test_generate_sampled_equals_the_oracle_replayed_with_the_same_uniforms(tmp_path, T, p):
    """MetaModel.generate(temperature > 0) on the device (fp32 parity path) == the oracle's generate with the sampled branch
    (meta.py:456-458) drawing from the reference's nucleus at the SAME uniforms: prompts of unequal length (teacher forcing),
    extra stop symbol, ids equal position by position."""
    mm, dec = _tiny_meta(tmp_path, torch.float32)
    prompts = ["Detect all manipulable object parts.", "Where is the handle of the drawer, and which way does it open?", "Lid?"]
    ids = [mm.tokenizer.encode(x, bos=True, eos=False) for x in prompts]
    gen_len = 24
    n_steps = gen_len + max(map(len, ids)) - min(map(len, ids)) + 1
    U = torch.rand(n_steps, len(prompts), generator=torch.Generator().manual_seed(17))
    _, got = mm.generate(prompts, None, max_gen_len=gen_len, temperature=T, top_p=p, additional_stop_symbols=["###"], return_ids=True,
                         sample_uniforms=U)
    stops = [mm.tokenizer.encode_segment("###"), mm.tokenizer.encode_wo_prefix_space("###")]
    flips = []

    def sampler(step, logits):
        probs = torch.softmax(logits / T, dim=-1)
        return ref_cpu.sample_top_p_at(probs, p, U[step])
    _, want = ref_cpu.generate_greedy(dec, ids, max_gen_len=gen_len, eos_id=mm.tokenizer.eos_id, extra_stop=stops, sampler=sampler)
    assert got == want, (got, want, flips)
    if T >= 1.0:       # the draw matters: not the greedy sequence
        _, greedy = ref_cpu.generate_greedy(dec, ids, max_gen_len=gen_len, eos_id=mm.tokenizer.eos_id, extra_stop=stops)
        assert greedy != want


def test_generate_sampled_default_generator_is_seedable(tmp_path):
    mm, _ = _tiny_meta(tmp_path, torch.bfloat16)
    prompts = ["Detect all manipulable object parts.", "Lid?"]
    outs = []
    for seed in (4, 4, 5):
        torch.manual_seed(seed)
        outs.append(mm.generate(prompts, None, max_gen_len=16, temperature=1.0, top_p=0.9, return_ids=True)[1])
    assert outs[0] == outs[1] and outs[0] != outs[2]


def test_eval_entry_default_recipe_matches_the_oracle_replay(tmp_path):
    """The batch-inference entry point with its DEFAULT sampling recipe (temperature 0.1, top-p 0.75: eval_affordance_v2.py:46-49;
    no --temperature flag), as a subprocess.  The entry seeds torch's device generator with --seed right before its generation loop
    and generate() draws its [steps, batch] uniforms from it in one call, so the same seed + shape here yields the numbers it used:
    the oracle replayed with them through the same image pipeline / prompt / post-processing yields the same records."""
    import subprocess
    import sys
    from PIL import Image
    from a3vlm_amd import checkpoint as ck
    from a3vlm_amd import eval_affordance_v2 as entry
    from a3vlm_amd.data.conversation import default_conversation
    from a3vlm_amd.data.transform import T_padded_resize
    from a3vlm_amd.model.meta import MetaModel
    from oracle.gen_golden import TINY
    gd = os.path.join(ROOT, "tests", "golden")
    vit = dict(vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=5)
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps({**{k: v for k, v in TINY.items() if k != "max_seq_len"}, **vit}))
    mm = MetaModel("llama_ens5", str(cfgp), os.path.join(gd, "tokenizer.model"), with_visual=True, max_seq_len=512)
    V = mm.tokenizer.n_words
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=V, **TINY), seed=0, std=0.3)      # wide logits: T = 0.1 still leaves choices
    vsd = ref_cpu.make_vision_weights(64, width=64, layers=2, patch=14, grid=8, seed=1, std=0.05)
    mm.llma.load_state_dict({**sd, **vsd})
    ckdir = ck.save_checkpoint(str(tmp_path / "ck"), types.SimpleNamespace(precision="tf32", only_save_trainable=False), mm, None, None, None, epoch=0)
    SEED, GEN = 7, 12
    e = dict(os.environ)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "a3vlm_amd.eval_affordance_v2", "--llama_type", "llama_ens5", "--llama_config", str(cfgp),
                        "--tokenizer_path", os.path.join(gd, "tokenizer.model"), "--pretrained_path", ckdir, "--batch_size", "3",
                        "--num_workers", "0", "--dataset", os.path.join(gd, "demo", "demo.json"), "--input_size", "224",
                        "--addition_flag", "s", "--max_gen_len", str(GEN), "--max_seq_len", "512", "--seed", str(SEED),
                        "--image_root", os.path.join(gd, "demo"), "--output_root", str(tmp_path / "logs"), "--precision", "tf32"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    recs = json.load(open(tmp_path / "logs" / "s" / "demo.json"))
    assert len(recs) == 3
    assert entry.get_args_parser().get_default("temperature") == 0.1 and entry.get_args_parser().get_default("top_p") == 0.75
    torch.manual_seed(SEED)
    U = torch.rand(GEN, 3, device=DEV, dtype=torch.float32).cpu()          # what generate() drew (all three prompts are equally long)
    img = T_padded_resize(224)(Image.open(os.path.join(gd, "demo", "render_336x300.png")).convert("RGB")).unsqueeze(0)
    conv = default_conversation()
    conv.load_qas([["Detect all manipulable object parts and provide their 3D bounding boxes.", None]])
    prompt = conv.get_prompt()
    views = ref_cpu.encode_image(img.expand(3, -1, -1, -1), vsd, vit_layers=2, vit_heads=4, n_views=5)
    itok = ref_cpu.assemble_image_tokens(views, vsd["start_img"], vsd["end_img"])
    dec = ref_cpu.OracleDecoder(ref_cpu.OracleArgs(vocab_size=V, **{**TINY, "max_seq_len": 512}), sd)
    ids = [mm.tokenizer.encode(prompt, bos=True, eos=False)] * 3

    def sampler(step, logits):
        return ref_cpu.sample_top_p_at(torch.softmax(logits / 0.1, dim=-1), 0.75, U[step])
    _, outs = ref_cpu.generate_greedy(dec, ids, image_tokens=itok, image_words=itok.shape[1], max_gen_len=GEN, eos_id=mm.tokenizer.eos_id,
                                      sampler=sampler)
    for r_ in range(3):
        want = entry.postprocess_answer(mm.tokenizer.decode(outs[r_]))
        assert recs[r_]["answer"] == want and recs[r_]["format_answer"] == entry.format_bounding_box(want), r_
