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
        get_transform("random_resized_crop")
