"""Two-image (RGB + depth) variant of the ``llama_ens5`` plugin -- interface of the reference's
``accessory.model.LLM.llama_ens5_2images`` (model/LLM/llama_ens5_2images.py).

Differences from ``llama_ens5`` (reference lines):
  * ``image_words`` counts BOTH images, ``visual_image_words`` one (:335-336);
  * extra learned tags ``start_depth_img`` / ``end_depth_img`` (:343-344);
  * ``forward(examples, image, depth_imgs)``: the depth image's words (own tags, same encoders and projector)
    follow the RGB block: h = [BOS | RGB words | depth words | text] (:487-500);
  * ``forward_inference(tokens, start_pos, image, depth_images)``: image words are only inserted when BOTH
    images are given (:517-547), and ``cache_image_words`` is their total.
Here both images go through ONE ViT + projector pass (2x the crop batch) and the projector LayerNorm scatters
each image's rows to its word range of the sequence buffer, so the extra image costs no extra passes over ``h``.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .llama_ens5 import ModelArgs, Transformer as _Base  # noqa: F401  (ModelArgs is part of the plugin contract)


class Transformer(_Base):
    def __init__(self, args: ModelArgs, with_visual: bool = False):
        super().__init__(args, with_visual=with_visual)
        if with_visual:
            self.visual_image_words = self.image_words
            self.image_words = self.visual_image_words * 2
            self.start_depth_img = nn.Parameter(torch.rand(1, 1, args.dim))
            self.end_depth_img = nn.Parameter(torch.rand(1, 1, args.dim))

    def _image_slots(self):
        return [(self.start_img, self.end_img), (self.start_depth_img, self.end_depth_img)]

    def forward(self, examples: torch.Tensor, image: Optional[torch.Tensor] = None, depth_imgs: Optional[torch.Tensor] = None, *,
                qformer_feats=None, extra_feats=None) -> torch.Tensor:
        images, slots = [], []
        if image is not None:
            images.append(image)
            slots.append(0)
        if depth_imgs is not None:
            images.append(depth_imgs)
            slots.append(1)
        # reference quirk kept (:505): the LM head runs from word ``visual_image_words`` on, so with a depth image the
        # returned rows are [depth words | text] (MetaModel.forward never passes depth_imgs, meta.py:251)
        out_from = self.visual_image_words if image is not None else 0
        return self._forward_images(examples, images, tuple(slots), qformer_feats, extra_feats, out_from=out_from)

    @torch.no_grad()
    def forward_inference(self, tokens: torch.Tensor, start_pos: int, image: Optional[torch.Tensor] = None,
                          depth_images: Optional[torch.Tensor] = None, *, qformer_feats=None, extra_feats=None) -> torch.Tensor:
        if image is not None and depth_images is not None:
            return self._forward_inference_images(tokens, start_pos, [image, depth_images], (0, 1), qformer_feats, extra_feats)
        return self._forward_inference_images(tokens, start_pos, [], (), qformer_feats, extra_feats)
