"""CPU: the pieces of bench.py that do not need a GPU -- argument defaults, the FLOP / byte accounting the reported fractions
are built on (SURVEY 8(d) conventions), and the committed bench lines under profiles/ carry every field of the contract."""
import glob
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_defaults_finish_quickly_and_name_the_workload():
    sys_argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = sys_argv
    assert a.gpus == 1 and a.steps <= 10 and a.warmup <= 5 and a.batch == 8 and a.prompt == 512 and a.model == "7b"


def test_flop_and_byte_accounting_7b():
    from a3vlm_amd.model.LLM.llama_ens5 import ModelArgs
    args = ModelArgs(vocab_size=32000, max_seq_len=2048, vit_patch=14, vit_crop=336, n_views=1, vit_width=1024, vit_layers=24, vit_heads=16,
                     **bench.GEOM["7b"])
    B, T, W = 8, 512, 579
    fl = bench.flops_forward(args, B, T, W)
    S = T + W
    lin = 2 * B * S * 32 * (4096 * 3 * 4096 + 4096 * 4096 + 3 * 4096 * 11008)
    assert abs(fl["gemm"] - lin) / lin < 0.05                          # decoder linears dominate the GEMM count
    assert 14.0e12 < fl["total"] / B < 15.5e12                          # 14.83 TFLOP per sample
    by = bench.bytes_decode_step(args, B, 1100)
    weights = 2 * (32 * (4096 * 3 * 4096 + 4096 * 4096 + 3 * 4096 * 11008) + 4096 * 32000)
    kv = B * 2 * 32 * 1100 * 32 * 128 * 2
    assert by == weights + kv
    assert bench.bytes_decoder_matrices(args) == 32 * (4096 * 3 * 4096 + 4096 * 4096 + 3 * 4096 * 11008)


def test_committed_bench_lines_follow_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_7b.json")))
    assert files, "no committed bench line under profiles/"
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "samples/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["peak"] == 2500.0
    assert r["traffic"] is None or r["traffic"]["bytes_per_launch"] > r["traffic"]["algorithmic_bytes_per_launch"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "samples/s" and c["sample"]
    assert abs(d["value"] - 8 * 1000.0 / d["ms_per_step"]) / d["value"] < 0.01
    if os.path.basename(files[-1]) >= "r03":
        # round 3 on: the headline is BASELINE configs[2] (LoRA fine-tune step), the full fine-tune is a leg; launch fields present
        assert "(train)" in d["metric"] and d["config"]["workload"].startswith("configs[2]")
        assert abs(d["train_lora"]["samples_s"] - d["value"]) / d["value"] < 1e-3 and d["train"]["samples_s"] > 0
        assert d["rccl_ranks"] == d["n_gpus"] and "exposed_allreduce_ms" in d
        assert d["roofline"]["families_of_the_headline_step"] in (["nt", "nn"], ["nt", "nt_dgrad"])
    elif os.path.basename(files[-1]) >= "r02":
        # round 2: the headline was the full fine-tune step of the configs[1] backbone
        assert "(train)" in d["metric"] and "FULL FINE-TUNE" in d["config"]["workload"]
        assert abs(d["train"]["samples_s"] - d["value"]) / d["value"] < 1e-3
    if os.path.basename(files[-1]) >= "r02":
        for leg in ("forward", "decode", "generate", "decode_fp8", "geometry_R", "config5", "train_lora", "m13b"):
            assert leg in d and "error" not in d[leg], leg
        assert d["geometry_R"]["image_words"] == 1455 and d["config5"]["seq_len"] == 1024 + 2 * 579
        assert d["m13b"]["train_replica"]["hbm_gib"] < 288 and d["m13b"]["train_replica"]["trainable_params"] > 13e9
        assert {"nt", "nn", "tn"} <= set(r["families"]) and r["traffic"]["file"][:3] in ("r02", "r03", "r04", "r05")
        if os.path.basename(files[-1]) >= "r05":
            # round 5: the reference's own recipes, the full-depth parity field and the stated DP-8 wire expectation ride the line
            rc = d["recipe"]
            assert "error" not in rc and rc["train_lora"]["seq_len"] == 2048 and 0 < rc["train_lora"]["attention_share"] < 0.5
            assert all(rc["eval"][f"ctx_{x}"]["tok_s_after_prefill"] > 0 for x in (1500, 2500, 3500)) and rc["eval"]["top_p"] == 0.75
            assert c["parity_full_depth"]["layers"] == 32 and c["parity_full_depth_rel_err"] < 5e-2 and c["parity_full_depth"]["argmax_equal"]
            w3 = d["wire_prediction_dp8"]["configs3_zero1"]
            assert w3["wire_ms_one_link_ring"] > w3["wire_ms_all_7_links"] > 0
            assert d["m13b"]["train_zero1_recipe"]["seq_len"] == 2048
        assert c["c1"]["ids_equal"] is True and c["decode_tok_s"] > 0
        if os.path.basename(files[-1]) >= "r06":
            # round 6: what the driver's record keeps -- scalars of the north-star legs inside `roofline`, the compact summary as the LAST key,
            # full-depth parity over every text position of two samples, the 13B recipe at its own memory regime
            assert list(d)[-1] == "summary" and d["summary"]["errors"] == []
            assert r["forward_ms"] == d["forward"]["ms_per_step"] and r["decode_tok_s"] == d["decode"]["tok_s"] and 0.3 < r["in_step_frac"] < 0.8
            assert d["summary"]["forward"][0] == d["forward"]["ms_per_step"] and d["summary"]["train_lora"][0] == d["train_lora"]["ms_per_step"]
            pf = c["parity_full_depth"]
            assert pf["samples"] == 2 and pf["positions"] == 1024 and pf["agreement_on_decided"] == 1.0 and pf["decided_frac"] > 0.2
            assert pf["agreement_on_decided_4sigma"] == 1.0 and pf["argmax_agreement_all_positions"] > 0.9 and "rel_err" in pf["fp8_vs_bf16"]
            assert c["parity_full_depth_decided_frac"] == pf["decided_frac"] and c["parity_full_depth_agreement"] == 1.0
            z = d["m13b"]["train_zero1_recipe_ckpt"]
            assert z["micro_batch"] == 2 and z["accum"] == 8 and z["recompute"] is True and z["hbm_gib"] < 288
            lo = d["train_lora_with_loader"]
            assert lo["persistent_workers"] and lo["prefetch_factor"] >= 4 and lo["vs_device_resident_synthetic_step"] > 0.98 and lo["loader_only_samples_s"]
            assert "note" not in d["generate"] and os.path.isfile(os.path.join(ROOT, "profiles", "bench_notes.md"))
        assert d["generate"]["tok_s_end_to_end"] > 0 and d["decode"]["roofline"]["bound"] == "hbm"


def test_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with NO torchrun around it and no rank environment must end in a 2-rank job (the driver's scaling
    run may call it either way).  A3V_BENCH_LAUNCH_ONLY=1 stops after the rendezvous (gloo, no GPU work): rank 0 reports how many
    ranks joined."""
    import subprocess
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE", "GROUP_RANK", "TORCHELASTIC_RUN_ID"):
        e.pop(k, None)
    e["A3V_BENCH_LAUNCH_ONLY"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "tiny"], cwd=ROOT, env=e, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["gpus_arg"] == 2
    # one rank: no launcher is started, the line says one
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "tiny"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1


def test_tail_of_the_line_carries_every_leg_and_roofline_holds_the_north_star_scalars():
    """the driver keeps ~8 KB of stdout tail and the scalars of `roofline` / `cpu_baseline`; the forward / decode
    legs must be inside both.  finalize_line() is what main() prints."""
    import copy
    sample = json.loads(open(os.path.join(ROOT, "profiles", "r05h_bench_7b.json")).read().strip().splitlines()[-1])
    before = copy.deepcopy(sample)
    out, notes = bench.finalize_line(sample)
    line = json.dumps(out)
    assert list(out)[-1] == "summary"
    tail = line[-6000:]
    assert '"summary"' in tail
    s = json.loads(tail[tail.index('"summary"') + len('"summary": '):-1])
    for leg in ("forward", "decode", "generate", "train", "train_lora", "decode_fp8", "m13b", "geometry_R"):
        assert leg in s and all(isinstance(x, (int, float)) for x in s[leg]), leg
    assert s["forward"] == [before["forward"]["ms_per_step"], before["forward"]["mfma_frac"]]
    assert s["decode"][1] == before["decode"]["tok_s"] and s["train"][0] == before["train"]["ms_per_step"] and s["errors"] == []
    r = out["roofline"]
    assert r["forward_ms"] == before["forward"]["ms_per_step"] and r["forward_mfma_frac"] == before["forward"]["mfma_frac"]
    assert r["decode_tok_s"] == before["decode"]["tok_s"] and r["decode_hbm_frac"] == before["decode"]["roofline"]["frac"]
    assert 0.4 < r["in_step_frac"] < 0.7 and r["in_step_file"].endswith(".txt") and 4 < r["traffic_ratio"] < 5.5
    assert 0.5 < r["l2_hit"] < 1 and 0.5 < r["mfma_busy"] < 1
    # scalars sit in front of the big tables, and the prose has left the line
    keys = list(r)
    assert keys.index("forward_ms") < keys.index("shapes") and keys.index("decode_hbm_frac") < keys.index("traffic")
    assert "note" not in out["decode"]["roofline"] and "note" not in out["generate"] and any(k.startswith("generate") for k in notes)
    assert len(line) < 14000
    # the contract fields are untouched
    for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "dtype", "config", "cpu_baseline"):
        assert out[k] == before[k] or k == "cpu_baseline"
    assert out["config"]["workload"] == before["config"]["workload"] and out["cpu_baseline"]["sample"] == before["cpu_baseline"]["sample"]


def test_stdout_is_restored_when_a_leg_outside_guarded_raises(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    monkeypatch.setattr(bench, "_main_legs", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("boom")))
    keep = sys.stdout
    try:
        bench.main()
    except RuntimeError:
        pass
    assert sys.stdout is keep


def test_gpus_8_starts_eight_ranks_on_eight_distinct_devices_and_duplicates_are_refused():
    """the first SCALE run must be self-checking.  `--gpus 8` (launch only: gloo rendezvous, no GPU work) ends in
    8 children with 8 distinct LOCAL_RANK / device ordinals on the line; two ranks on one device are refused."""
    import subprocess
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE", "GROUP_RANK", "TORCHELASTIC_RUN_ID"):
        e.pop(k, None)
    e["A3V_BENCH_LAUNCH_ONLY"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--model", "tiny"], cwd=ROOT, env=e, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["gpus_arg"] == 8
    devs = d["rank_devices"]
    assert sorted(x["rank"] for x in devs) == list(range(8)) and sorted(x["local_rank"] for x in devs) == list(range(8))
    assert sorted(x["device"] for x in devs) == list(range(8)) and len({x["pid"] for x in devs}) == 8
    # duplicates: by ordinal, by uuid, by LOCAL_RANK
    ok = [{"rank": i, "local_rank": i, "device": i, "uuid": f"GPU-{i}"} for i in range(8)]
    bench.check_rank_devices(ok)
    import pytest
    for bad in ([dict(ok[0]), dict(ok[1], device=0)] + ok[2:], [dict(ok[0]), dict(ok[1], uuid="GPU-0")] + ok[2:],
                [dict(ok[0]), dict(ok[1], local_rank=0)] + ok[2:]):
        with pytest.raises(RuntimeError):
            bench.check_rank_devices(bad)
    bench.check_rank_devices([dict(ok[0]), dict(ok[1], device=0, uuid="GPU-0", local_rank=0)], allow_shared=True)     # the tests' one-GPU emulation
