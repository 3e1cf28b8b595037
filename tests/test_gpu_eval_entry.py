"""-m gpu: the batch-inference entry point end to end (checkpoint in the reference layout -> transform -> conversation
prompt -> 5-view encode -> greedy generate -> post-processing -> JSON), greedy ids checked against the oracle."""
import json
import os
import subprocess
import sys
import types

import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eval_affordance_v2_demo(tmp_path):
    from a3vlm_amd import checkpoint as ck
    from a3vlm_amd.data.conversation import default_conversation
    from a3vlm_amd.data.transform import T_padded_resize
    from a3vlm_amd.eval_affordance_v2 import format_bounding_box, postprocess_answer
    from a3vlm_amd.model.meta import MetaModel
    from oracle import ref_cpu
    from oracle.gen_golden import TINY
    gd = os.path.join(ROOT, "tests", "golden")
    vit = dict(vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=5)
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps({**{k: v for k, v in TINY.items() if k != "max_seq_len"}, **vit}))
    # a checkpoint in the reference's on-disk layout with known weights
    mm = MetaModel("llama_ens5", str(cfgp), os.path.join(gd, "tokenizer.model"), with_visual=True, max_seq_len=512)
    V = mm.tokenizer.n_words
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=V, **TINY), seed=0, std=0.08)
    vsd = ref_cpu.make_vision_weights(64, width=64, layers=2, patch=14, grid=8, seed=1, std=0.05)
    mm.llma.load_state_dict({**sd, **vsd})
    args = types.SimpleNamespace(precision="tf32", only_save_trainable=False)
    ckdir = ck.save_checkpoint(str(tmp_path / "ck"), args, mm, None, None, None, epoch=0)
    e = dict(os.environ)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "a3vlm_amd.eval_affordance_v2", "--llama_type", "llama_ens5", "--llama_config", str(cfgp),
                        "--tokenizer_path", os.path.join(gd, "tokenizer.model"), "--pretrained_path", ckdir, "--batch_size", "2",
                        "--num_workers", "0", "--dataset", os.path.join(gd, "demo", "demo.json"), "--input_size", "224",
                        "--addition_flag", "t", "--max_gen_len", "10", "--max_seq_len", "512", "--temperature", "0",
                        "--image_root", os.path.join(gd, "demo"), "--output_root", str(tmp_path / "logs"), "--precision", "tf32"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "'missing_keys': [], 'unexpected_keys': []" in r.stdout
    # the default is the device-side preprocessing (--preprocess gpu: workers decode, a3v_preprocess_batch pads / resizes / normalises);
    # the PIL transform in the workers (--preprocess cpu) must give byte-identical records
    cmd = r.args
    r2 = subprocess.run([a if a != "t" else "tcpu" for a in cmd] + ["--preprocess", "cpu"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-3000:]
    assert open(tmp_path / "logs" / "t" / "demo.json", "rb").read() == open(tmp_path / "logs" / "tcpu" / "demo.json", "rb").read()
    recs = json.load(open(tmp_path / "logs" / "t" / "demo.json"))
    assert len(recs) == 3 and set(recs[0]) == {"answer", "format_answer", "annotation", "question", "image", "fail"}
    # oracle: same image pipeline + prompt + greedy decode on the CPU
    img = T_padded_resize(224)(Image.open(os.path.join(gd, "demo", "render_336x300.png")).convert("RGB")).unsqueeze(0)
    conv = default_conversation()
    conv.load_qas([["Detect all manipulable object parts and provide their 3D bounding boxes.", None]])
    prompt = conv.get_prompt()
    assert recs[0]["question"] == prompt
    views = ref_cpu.encode_image(img, vsd, vit_layers=2, vit_heads=4, n_views=5)
    itok = ref_cpu.assemble_image_tokens(views, vsd["start_img"], vsd["end_img"])
    dec = ref_cpu.OracleDecoder(ref_cpu.OracleArgs(vocab_size=V, **{**TINY, "max_seq_len": 512}), sd)
    ids = [mm.tokenizer.encode(prompt, bos=True, eos=False)]
    _, outs = ref_cpu.generate_greedy(dec, ids, image_tokens=itok, image_words=itok.shape[1], max_gen_len=10, eos_id=mm.tokenizer.eos_id)
    want = postprocess_answer(mm.tokenizer.decode(outs[0]))
    assert recs[0]["answer"] == want and recs[0]["format_answer"] == format_bounding_box(want)
    assert recs[1]["answer"] == want      # the three demo items share image and question
