"""Providers for the three frozen third-party encoders of the reference's ensemble, through stock
PyTorch-ROCm modules behind the plugin's feature hooks -- exactly the packages the reference itself instantiates
(LLM/llama_ens5.py:283-322) and the pre/post-processing it applies around them (:399-440):

  * BLIP-2 Q-Former + EVA ViT-g  (``transformers.Blip2Model``, :285-293; 32 query tokens x 768 per view, :399)
  * ConvNeXt-XXL trunk            (``open_clip`` 'convnext_xxlarge', :304-315; 256x256 input, 8x8x3072 map repeated to
                                   16x16, global-average "cls" token prepended, :402-418)
  * DINOv2 ViT-g/14               (``torch.hub`` facebookresearch/dinov2, :317-322; CLIP->ImageNet renormalisation, :420-434)

The hand-written HIP path covers the CLIP ViT branch, the projectors and everything downstream; these nets are out-of-scope
frozen feature extractors.  Each ``attach_*`` registers the module under the reference's attribute name (so the
``qformer.*`` / ``openclip_convnext_xxl.*`` / ``dinov2_vitg14.*`` keys of a published checkpoint load by name), keeps it frozen,
and installs the feature callback.  A missing package raises ImportError with the package name: nothing is stubbed.
The model must have been built with the matching geometry (``ModelArgs.qformer_tokens = 32``, ``extra_feat_dim = 3072 + 1536``).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
DINO_MEAN = (0.485, 0.456, 0.406)
DINO_STD = (0.229, 0.224, 0.225)


def _freeze(mod: torch.nn.Module, like: torch.Tensor) -> torch.nn.Module:
    mod.to(device=like.device, dtype=like.dtype)
    mod.eval()
    for p in mod.parameters():
        p.requires_grad = False
    return mod


def attach_qformer(model, pretrained: bool = False, config=None):
    """``model.qformer`` = transformers.Blip2Model without its language model (llama_ens5.py:285-293); features =
    ``get_qformer_features(pixel_values=views).last_hidden_state`` [N, 32, 768] under no_grad (:399)."""
    try:
        from transformers import Blip2Config, Blip2Model
    except ImportError as e:      # pragma: no cover
        raise ImportError("attach_qformer needs the `transformers` package (Blip2Model)") from e
    if not getattr(model.args, "qformer_tokens", 0):
        raise ValueError("build the plugin with ModelArgs.qformer_tokens = 32 to use a Q-Former")
    if pretrained:
        q = Blip2Model.from_pretrained("Salesforce/blip2-opt-2.7b")
    else:
        q = Blip2Model(config if config is not None else Blip2Config())
    q.language_projection = None
    q.language_model = None
    if q.config.num_query_tokens != model.args.qformer_tokens or q.config.qformer_config.hidden_size != 768:
        raise ValueError("Q-Former geometry does not match the plugin (query tokens / 768-wide output)")
    model.qformer = _freeze(q, model.norm.weight)

    @torch.no_grad()
    def fn(views: torch.Tensor) -> torch.Tensor:
        out = model.qformer.get_qformer_features(pixel_values=views.to(model.norm.weight.dtype))
        return getattr(out, "last_hidden_state", out)        # newer transformers return the tensor itself
    model.qformer_fn = fn
    return model


def convnext_tokens(feat_map: torch.Tensor) -> torch.Tensor:
    """[N, 3072, 8, 8] trunk output -> [N, 257, 3072] tokens (llama_ens5.py:406-418): 2x nearest upsampling to 16x16, flatten,
    mean token prepended as "cls"."""
    assert feat_map.dim() == 4 and feat_map.shape[2:] == (8, 8), feat_map.shape      # (3072, 8, 8) for the real ConvNeXt-XXL trunk
    x = feat_map.repeat_interleave(2, dim=-1).repeat_interleave(2, dim=-2).flatten(-2).permute(0, 2, 1)
    return torch.cat([x.mean(dim=1, keepdim=True), x], dim=1)


def attach_convnext_xxl(model, pretrained: bool = False, trunk: Optional[torch.nn.Module] = None):
    """``model.openclip_convnext_xxl`` = open_clip 'convnext_xxlarge' visual trunk with identity pooling (llama_ens5.py:304-315).
    ``trunk``: an already built timm-style ConvNeXt trunk (``trunk.head.global_pool`` / ``.flatten`` present, stride 32) to use
    instead of building the open_clip model -- an environment without ``open_clip`` can hand in its own copy; its channel count must
    equal the plugin's ``extra_feat_dim`` share."""
    if trunk is None:
        try:
            import open_clip
        except ImportError as e:
            raise ImportError("attach_convnext_xxl needs the `open_clip` package (open_clip_torch) and `timm`") from e
        net, _, _ = open_clip.create_model_and_transforms("convnext_xxlarge", pretrained="laion2b_s34b_b82k_augreg_soup" if pretrained else None)
        trunk = net.visual.trunk
    trunk.head.global_pool = torch.nn.Identity()
    trunk.head.flatten = torch.nn.Identity()
    model.openclip_convnext_xxl = _freeze(trunk, model.norm.weight)

    @torch.no_grad()
    def fn(views: torch.Tensor) -> torch.Tensor:
        x = F.interpolate(views.half(), size=(256, 256)).to(views)
        return convnext_tokens(model.openclip_convnext_xxl(x))
    _set_extra(model, 0, fn)
    return model


def dinov2_input(views: torch.Tensor) -> torch.Tensor:
    """CLIP-normalised pixels -> ImageNet-normalised pixels (llama_ens5.py:420-428)."""
    t = lambda v: torch.tensor(v, device=views.device, dtype=views.dtype).view(3, 1, 1)
    return (views * t(CLIP_STD) + t(CLIP_MEAN) - t(DINO_MEAN)) / t(DINO_STD)


def attach_dinov2(model, pretrained: bool = False, repo: str = "facebookresearch/dinov2", source: Optional[str] = None,
                  net: Optional[torch.nn.Module] = None):
    """``model.dinov2_vitg14`` via torch.hub (llama_ens5.py:317-322); tokens = [x_norm_clstoken | x_norm_patchtokens] (:429-434).
    ``source='local'`` with ``repo`` = a checkout of the hub repository works without network access; ``net``: an already built
    module with DINOv2's ``forward_features`` contract (dict with ``x_norm_clstoken`` [N, D] and ``x_norm_patchtokens`` [N, 256, D])."""
    if net is None:
        kw = dict(source=source) if source else {}
        net = torch.hub.load(repo, "dinov2_vitg14", pretrained=pretrained, **kw)
    model.dinov2_vitg14 = _freeze(net, model.norm.weight)

    @torch.no_grad()
    def fn(views: torch.Tensor) -> torch.Tensor:
        f = model.dinov2_vitg14.forward_features(dinov2_input(views))
        return torch.cat([f["x_norm_clstoken"].unsqueeze(1), f["x_norm_patchtokens"]], dim=1)
    _set_extra(model, 1, fn)
    return model


def _set_extra(model, slot: int, fn) -> None:
    """extra features are concatenated after the CLIP features in the order (ConvNeXt, DINOv2) (llama_ens5.py:436-440)."""
    slots = getattr(model, "_extra_slots", None) or [None, None]
    slots[slot] = fn
    model._extra_slots = slots
    model.extra_feat_fns = [f for f in slots if f is not None]


def attach_reference_encoders(model, pretrained: bool = False, qformer_config=None, convnext_trunk=None, dinov2_net=None):
    """All three, as ``Transformer.__init__`` of the reference does with ``with_visual=True`` (llama_ens5.py:283-322).  The keyword
    arguments hand in pre-built modules / a Q-Former config (reduced-width runs, environments without open_clip / torch.hub access)."""
    attach_qformer(model, pretrained, config=qformer_config)
    attach_convnext_xxl(model, pretrained, trunk=convnext_trunk)
    attach_dinov2(model, pretrained, net=dinov2_net)
    return model
