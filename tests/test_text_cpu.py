"""CPU: conversation template, answer post-processing and image transform (host side of the inference entry point)
against fixtures captured from the reference (tests/golden/text_tiny.json)."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from a3vlm_amd.data.conversation import conv_v1_2, default_conversation
from a3vlm_amd.data.transform import CLIP_MEAN, CLIP_STD, PadToSquare, T_padded_resize, get_transform
from a3vlm_amd.eval_affordance_v2 import format_bounding_box, normalize_number, postprocess_answer, shard_range


@pytest.fixture(scope="module")
def fx(golden_dir):
    return json.load(open(os.path.join(golden_dir, "text_tiny.json")))


def test_demo_prompts_and_label_spans(fx):
    q = "Detect all manipulable object parts and provide their 3D bounding boxes."
    for want in fx["demo_prompts"]:
        c = default_conversation()
        c.load_qas([[q, None]])
        assert c.get_prompt() == want
    c = conv_v1_2()
    c.load_qas([["Detect all manipulable object parts.", "<box>lid</box>[[0.12,0.34,0.56]]"], ["And the joint?", "<axis>revolute</axis>[0.10,0.20]"], ["Again?", None]])
    assert c.process() == fx["multi_turn"] and c.response_end_signal == fx["response_end_signal"]
    c2 = c.copy()
    c2.append_message("Human", "x")
    assert len(c.messages) + 1 == len(c2.messages)


def test_format_bounding_box_matches_reference(fx):
    for case in fx["bbox_cases"]:
        assert format_bounding_box(case["in"]) == case["out"], case["in"]
    assert [normalize_number(x) for x in (1234, 56, 7, 0.3)] == [1.234, 0.56, 0.7, 0.3]
    assert postprocess_answer("The answer is 42. ### Human: more") == "42"
    assert postprocess_answer(" [0.1, 0.2] ###") == "[01, 02]"     # '.' removed first (eval_affordance_v2.py:344-345)


def test_shard_range_covers_everything():
    for total, world in [(10, 4), (3, 8), (200, 8)]:
        seen = [i for r in range(world) for i in shard_range(total, world, r)]
        assert seen == list(range(total))


def test_padded_resize_transform(golden_dir):
    img = Image.open(os.path.join(golden_dir, "demo", "render_336x300.png")).convert("RGB")
    assert img.size == (336, 300)
    sq = PadToSquare(CLIP_MEAN)(img)
    assert sq.size == (336, 336)
    fill = tuple(int(x * 255) for x in CLIP_MEAN)
    assert sq.getpixel((0, 0)) == fill and sq.getpixel((5, 18)) == img.getpixel((5, 0))   # 18 rows of padding on top
    t = T_padded_resize(336)(img)
    assert t.shape == (3, 336, 336) and t.dtype == torch.float32
    # size == padded side: the resize is the identity -> exactly (x/255 - mean)/std of the padded image
    a = torch.from_numpy(np.asarray(sq).copy()).permute(2, 0, 1).float() / 255
    want = (a - torch.tensor(CLIP_MEAN).view(3, 1, 1)) / torch.tensor(CLIP_STD).view(3, 1, 1)
    assert torch.equal(t, want)
    t2 = get_transform("padded_resize", 448)(img)
    assert t2.shape == (3, 448, 448) and torch.isfinite(t2).all()
    with pytest.raises(ValueError):
        get_transform("no_such_transform")
    # the other two reference transforms: output geometry, determinism under torch's RNG, crop parameters in range
    c = get_transform("resized_center_crop", 224)(img)
    assert c.shape == (3, 224, 224)
    rrc = get_transform("random_resized_crop", 224)
    torch.manual_seed(3)
    r1 = rrc(img)
    torch.manual_seed(3)
    r2 = rrc(img)
    assert r1.shape == (3, 224, 224) and torch.equal(r1, r2)
    for _ in range(20):
        top, left, ch, cw = rrc.get_params(336, 300)
        assert 0 <= top and top + ch <= 300 and 0 <= left and left + cw <= 336
        assert 0.9 * 336 * 300 * 0.98 <= ch * cw <= 336 * 300 and 0.74 <= cw / ch <= 1.35


def test_dialog_dataset_matches_reference_items(tmp_path, capsys):
    """G9: FinetuneDialogDataset items (tokens, assistant-span labels, pad mask, image pass-through, group layout incl.
    ratio sub-sampling, length sort, str() of non-string turns, nothing-to-predict fallback) equal the reference's."""
    import json
    import os
    from a3vlm_amd.data.conversation.dataset import FinetuneDialogDataset, find_sublist
    from oracle.gen_golden import GOLD, dialog_transform, dialog_yaml
    want = json.load(open(os.path.join(GOLD, "dialog", "items.json")))
    ds = FinetuneDialogDataset(dialog_yaml(str(tmp_path)), dialog_transform, max_words=150, image_words=30,
                               tokenizer=os.path.join(GOLD, "tokenizer.model"))
    assert len(ds) == want["len"] and ds.groups() == want["groups"]
    n_img = 0
    for i, w in enumerate(want["items"]):
        it = ds[i]
        assert it[0].tolist() == w["tokens"] and it[1].tolist() == w["labels"], i
        assert float(it[2].sum()) == w["mask_sum"]
        assert (len(it) == 4) == ("image_sum" in w)
        if len(it) == 4:
            n_img += 1
            assert it[0].numel() == 150 - 30 and abs(float(it[3].double().sum()) - w["image_sum"]) < 1e-6
        else:
            assert it[0].numel() == 150
        assert it[1].count_nonzero() > 0 and set(it[1][it[1] != 0].tolist()) <= set(it[0].tolist())
    assert n_img == 6
    assert find_sublist([1, 2, 3, 4], [3, 4]) == 2 and find_sublist([1, 2], [3]) == -1
