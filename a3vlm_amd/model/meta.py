"""``MetaModel`` with the interface of the reference's ``accessory.model.meta.MetaModel``
(reference: model/accessory/model/meta.py:20-597), driving the MI355X-native plugin.

Kept: constructor signature, plugin discovery by ``llama_type`` (meta.py:30-32), the params-json
merge (:34-47), ``forward -> (loss, {})`` with the trailing-pad trim and the all-zero-label
guard (:234-263), batched ``generate`` with shortest-prompt start, teacher forcing of longer
prompts, multi-token stop strings and left truncation (:379-485), ``stream_generate``,
``compute_logits``, ``evaluate_examples``, ``sample_top_p``, ``get_trainable_params`` etc.

Changed on purpose: no fairscale / torch.distributed requirement for a single process, tensors
are created on the model's device instead of ``.cuda()``, the loss and the
argmax run in HIP kernels (a3v_cross_entropy / a3v_argmax) instead of torch ops.
"""
from __future__ import annotations

import dataclasses
import importlib
import inspect
import json
from typing import Dict, Iterable, List, Optional

import torch
import torch.nn as nn

from .. import ops
from .tokenizer import Tokenizer, probe_tokenizer_path_from_pretrained  # noqa: F401

# llama_type -> module; extend to plug further model files (same contract as accessory.model.LLM.*)
_PLUGIN_PACKAGE = "a3vlm_amd.model.LLM"


class MetaModel(nn.Module):
    def __init__(self, llama_type: str, llama_config, tokenizer_path: str, with_visual: bool = False,
                 max_seq_len: int = 4096, pretrain_stage: bool = False) -> None:
        super().__init__()
        self.llama_type = llama_type
        self.with_visual = with_visual
        model_module = importlib.import_module(f"{_PLUGIN_PACKAGE}.{llama_type}")
        ModelArgs, Transformer = model_module.ModelArgs, model_module.Transformer

        llama_args = {}
        if isinstance(llama_config, str):
            llama_config = [llama_config]
        for cfg in llama_config or []:
            if isinstance(cfg, dict):
                llama_args.update(cfg)
            else:
                with open(cfg, "r") as f:
                    llama_args.update(json.loads(f.read()))
        llama_args["max_seq_len"] = max_seq_len
        llama_args["max_batch_size"] = 32
        tokenizer = Tokenizer(model_path=tokenizer_path)
        llama_args["vocab_size"] = tokenizer.n_words
        known = {f.name for f in dataclasses.fields(ModelArgs)}
        unknown = set(llama_args) - known
        if unknown:
            raise TypeError(f"unknown ModelArgs fields in config: {sorted(unknown)}")
        args = ModelArgs(**llama_args)

        if "tokenizer" in inspect.signature(Transformer.__init__).parameters:
            model = Transformer(args, tokenizer, with_visual=with_visual)
            self.tokenizer = model.tokenizer
        else:
            model = Transformer(args, with_visual=with_visual)
            self.tokenizer = tokenizer
        self.llma = model
        self.criterion = torch.nn.CrossEntropyLoss(ignore_index=0)  # kept as an attribute (meta.py:67)
        self._set_default_trainability(pretrain_stage)
        self.is_peft = getattr(model, "is_peft", False)

    # ------------------------------------------------------------------ construction from a checkpoint folder
    @classmethod
    def from_pretrained(cls, pretrained_path, llama_type: Optional[str] = None, llama_config=None,
                        tokenizer_path: Optional[str] = None, with_visual: bool = False, max_seq_len: int = 4096,
                        mp_group=None, dtype=torch.bfloat16, device="cuda", quant=False) -> "MetaModel":
        """Build the model a checkpoint folder describes and load it (reference: model/meta.py:88-222).

        What is not given is looked up in the LAST folder of ``pretrained_path``: ``llama_type`` in ``meta.json``,
        ``llama_config`` in ``config.json`` (absent: ModelArgs defaults), the tokenizer as ``tokenizer.model`` or an HF pair.
        Folders are loaded in order (``consolidated`` / ``meta_ori`` override, ``consolidated_diff`` adds).  Differences to the
        reference, all forced by the DP-replica design: no process group is created (``mp_group`` must be None or of size 1;
        TP-sharded folders are merged on load), ``hf://`` ids need a network and are refused, and ``quant=True`` selects this
        package's weight-only fp8 decoder images (``Transformer.quantize_decode_weights("fp8")``) instead of bitsandbytes NF4."""
        import os
        import warnings
        from ..checkpoint import load_tensor_parallel_model_list
        paths = [pretrained_path] if isinstance(pretrained_path, str) else list(pretrained_path or [])
        if not paths:
            raise ValueError("pretrained_path should be specified")
        for path in paths:
            if path.startswith("hf://"):
                raise NotImplementedError(f"{path}: downloading from the hub is not available here; pass a local folder")
        if mp_group is not None:
            import torch.distributed as dist
            if dist.get_world_size(mp_group) != 1:
                raise NotImplementedError("model parallel size > 1 is replaced by DP replicas; TP-sharded folders are merged on load")
        last = paths[-1]
        if llama_type is None:
            meta_file = os.path.join(last, "meta.json")
            if not os.path.exists(meta_file):
                raise ValueError("Cannot determine llama_type")
            with open(meta_file) as f:
                llama_type = json.load(f)["llama_type"]
        if llama_config is None:
            cfg_file = os.path.join(last, "config.json")
            llama_config = [cfg_file] if os.path.exists(cfg_file) else []
        if tokenizer_path is None:
            tokenizer_path = probe_tokenizer_path_from_pretrained(last)
            if tokenizer_path is None:
                raise FileNotFoundError("No tokenizer available")
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            with torch.device(device):
                model = cls(llama_type, llama_config, tokenizer_path, with_visual, max_seq_len)
        finally:
            torch.set_default_dtype(old)
        print(f"Loading pretrained weights from {paths} ...")
        load_result = load_tensor_parallel_model_list(model, paths)
        if load_result != {"missing_keys": [], "unexpected_keys": []}:
            warnings.warn(f"checkpoint and model mismatch: \n{load_result}")
        else:
            print("all params match perfectly!")
        if quant:
            model.llma.quantize_decode_weights("fp8")
        model.eval()
        return model

    # ------------------------------------------------------------------ trainability
    def get_trainable_params(self, pretrain_stage: bool = False):
        return {"llma." + n: p for n, p in self.llma.get_trainable_params(pretrain_stage).items()}

    def _set_default_trainability(self, pretrain_stage: bool = False) -> None:
        for _, v in self.named_parameters():
            v.requires_grad = False
        for _, v in self.get_trainable_params(pretrain_stage=pretrain_stage).items():
            v.requires_grad = True

    @property
    def _device(self) -> torch.device:
        return next(self.parameters()).device

    # ------------------------------------------------------------------ loss
    @staticmethod
    def _trim(examples: torch.Tensor, labels: torch.Tensor):
        """meta.py:235-249: cut the batch after the last column that holds any label."""
        nz = torch.count_nonzero(labels, dim=0).tolist()
        pos = len(nz) - 1
        while pos >= 0 and nz[pos] == 0:
            pos -= 1
        if pos == -1:
            pos = 2
        return examples[:, :pos + 1], labels[:, :pos + 1]

    def loss_from_logits(self, output: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """meta.py:256-262 on logits [B,T,V] and UNshifted labels [B,T]: mean CE over labels != 0."""
        B, T, V = output.shape
        lab = labels[:, 1:].contiguous().view(-1)
        if int(lab.sum()) == 0:
            return torch.zeros((), dtype=torch.float32, device=output.device)
        dev = output.device
        row_loss = torch.empty(B, T - 1, dtype=torch.float32, device=dev)
        for b in range(B):     # logits rows [b, 0:T-1] are contiguous per sample
            ops.cross_entropy(output[b, :T - 1], lab[b * (T - 1):(b + 1) * (T - 1)], row_loss[b])
        n_valid = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.count_valid(lab, n_valid)
        # final scalar reduction of B*(T-1) fp32 values: plumbing, not the hot path
        return row_loss.sum() / n_valid.to(torch.float32)[0]

    def train_engine(self, compute_dtype: torch.dtype = None, zero1_world: int = 0):
        """The HIP forward/backward engine of the plugin (a3vlm_amd/train.py).  compute dtype: bf16
        ("autocast") unless the model is all-fp32 (parity path)."""
        from ..train import TrainEngine
        if getattr(self, "_engine", None) is None:
            if compute_dtype is None:
                frozen = [p for p in self.llma.parameters() if not p.requires_grad]
                compute_dtype = frozen[0].dtype if frozen else torch.bfloat16
                if getattr(self, "train_compute_dtype", None) is not None:
                    compute_dtype = self.train_compute_dtype
            self._engine = TrainEngine(self.llma, compute_dtype, zero1_world=zero1_world)
            self._anchor = torch.zeros((), dtype=torch.float32, device=self._device, requires_grad=True)
        return self._engine

    def _sync_engine(self):
        eng = getattr(self, "_engine", None)
        if eng is not None:
            eng.sync_optimizer()

    def zero_grad(self, set_to_none: bool = True):
        """``set_to_none=False`` WRITES the gradients: an overlapped optimizer step (``FusedAdamW.step(overlap=True)``) may still
        be reading them on its own stream, so it is joined first (``set_to_none=True`` only drops references)."""
        if not set_to_none:
            self._sync_engine()
        return super().zero_grad(set_to_none=set_to_none)

    def state_dict(self, *args, **kwargs):
        self._sync_engine()            # checkpoints read the parameters on the current stream
        return super().state_dict(*args, **kwargs)

    def forward(self, examples, labels, images=None, depth_imgs=None, trimmed: bool = False):
        """``trimmed=True`` (not in the reference's signature): the caller already cut the batch after its last labelled column
        (``MetaModel._trim`` on the CPU batch, as ``engine_finetune.train_one_epoch`` does) -- skips the host read of the
        per-column label counts that the trim needs on device tensors."""
        if not trimmed:
            with torch.no_grad():
                examples, labels = self._trim(examples, labels)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.llma.parameters()):
            # training: loss and gradients through the HIP backward (loss.backward() fills param.grad)
            from ..train import step_loss
            eng = self.train_engine()
            return step_loss(eng, self._anchor, examples, labels, images), {}
        output = self.llma(examples, images)
        additional_loss = {}
        if isinstance(output, tuple):
            output, additional_loss = output
        return self.loss_from_logits(output, labels), additional_loss

    # ------------------------------------------------------------------ logits / eval
    @torch.no_grad()
    def compute_logits(self, examples, images=None, bos=True, eos=False) -> List[torch.Tensor]:
        if isinstance(examples, str):
            raise ValueError(f"{self.__class__}.generate expects a batched LIST of prompts, but str is given")
        if isinstance(examples[0], str):
            examples = [self.tokenizer.encode(x, bos, eos) for x in examples]
        dev = self._device
        if images is not None:
            images = images.to(dev)
        lens = [len(x) for x in examples]
        tok = torch.zeros((len(examples), max(lens)), dtype=torch.long, device=dev)
        for i, t in enumerate(examples):
            tok[i, :len(t)] = torch.tensor(t, dtype=torch.long)
        out = self.llma(tok, images)
        logits = out[0] if isinstance(out, tuple) else out
        return [lg[:n].float() for lg, n in zip(logits, lens)]

    @torch.no_grad()
    def evaluate_examples(self, examples, contexts=None, images=None, bos=True, eos=False) -> Dict[str, List]:
        """meta.py:306-377."""
        if isinstance(examples, str):
            raise ValueError(f"{self.__class__}.generate expects a batched LIST of prompts, but str is given")
        if isinstance(examples[0], str):
            examples = [self.tokenizer.encode(x, bos, eos) for x in examples]
            if contexts is not None:
                contexts = [self.tokenizer.encode(x, bos, False) for x in contexts]
        if contexts is not None:
            assert all(e[:len(c)] == c for e, c in zip(examples, contexts))
        logits = self.compute_logits(examples, images)
        res = {"log_likelihood": [], "ppl": [], "max_equal": [], "non_context_logits": []}
        for i, lg in enumerate(logits):
            start = 0 if contexts is None else len(contexts[i]) - 1
            assert start >= 0
            lg = lg[start:-1].contiguous()
            lab = torch.tensor(examples[i][start + 1:], dtype=torch.long, device=lg.device)
            row = torch.empty(lg.shape[0], dtype=torch.float32, device=lg.device)
            ops.cross_entropy(lg, lab, row)
            res["log_likelihood"].append(-row.sum().item())
            res["ppl"].append(row.mean().item())
            am = torch.empty(lg.shape[0], dtype=torch.long, device=lg.device)
            ops.argmax(lg, am)
            res["max_equal"].append(bool((am == lab).all().item()))
            res["non_context_logits"].append(lg)
        return res

    # ------------------------------------------------------------------ generation
    @torch.no_grad()
    def generate(self, prompts: List[str], images: Optional[torch.Tensor] = None,
                 depth_images: Optional[torch.Tensor] = None, max_gen_len: int = 512,
                 temperature: float = 0.0, top_p: float = 0.95,
                 additional_stop_symbols: Iterable[str] = (), return_ids: bool = False, poll_every: int = 4,
                 sample_uniforms: Optional[torch.Tensor] = None) -> List[str]:
        """meta.py:379-485.  ``return_ids`` additionally returns the generated id lists (the
        arguments of tokenizer.decode at :482-484) for bit-exact parity checks.
        ``temperature > 0`` (the eval script's default recipe, T 0.1 / top-p 0.75): the nucleus draw of :456-459 / :568-583 runs on
        the device (``a3v_sample_top_p``: no full-vocabulary sort, no host-visible op per token); its one random ingredient is a
        uniform number per row and step, drawn up front from torch's device generator (``torch.manual_seed`` reproduces a run) or
        handed in as ``sample_uniforms`` [steps, bsz] (parity tests replay the oracle with the same numbers).
        ``poll_every``: the all-rows-stopped flag lives on the device and is read back (a host sync) only every
        ``poll_every`` steps instead of every step (:478-479); steps taken after every row has stopped write past
        ``stop_pos`` and are discarded, so the outputs are identical for any value (1 = the reference's cadence)."""
        if isinstance(prompts, str):
            raise ValueError(f"{self.__class__}.generate expects a batched LIST of prompts, but str is given")
        dev = self._device
        if images is not None:
            images = images.to(dev)
        if depth_images is not None:
            depth_images = depth_images.to(dev)
        bsz = len(prompts)
        args = self.llma.args
        assert bsz <= args.max_batch_size, (bsz, args.max_batch_size)
        prompt_tokens = [self.tokenizer.encode(x, bos=True, eos=False) for x in prompts]
        min_prompt = min(len(t) for t in prompt_tokens)
        max_prompt = max(len(t) for t in prompt_tokens)
        max_seq_len = args.max_seq_len
        if images is not None:
            max_seq_len -= self.llma.image_words
        total_len = min(max_seq_len, max_gen_len + max_prompt)
        prompt_tokens = [t[-(max_seq_len - max_gen_len):] for t in prompt_tokens]

        tokens_cpu = torch.zeros((bsz, total_len), dtype=torch.long)
        mask_cpu = torch.zeros((bsz, total_len), dtype=torch.bool)
        for k, t in enumerate(prompt_tokens):
            tokens_cpu[k, :len(t)] = torch.tensor(t, dtype=torch.long)
            mask_cpu[k, :len(t)] = True
        tokens = tokens_cpu.to(dev)
        text_mask = mask_cpu.to(dev)
        start_pos, prev_pos = min_prompt, 0

        # stop sequences: EOS, then each extra stop string tokenised as a word-initial and as a word-internal piece (:438-444)
        l_stop = [[self.tokenizer.eos_id]]
        l_stop += [self.tokenizer.encode_segment(s) for s in additional_stop_symbols]
        l_stop += [self.tokenizer.encode_wo_prefix_space(s) for s in additional_stop_symbols]
        offs = [0]
        for st in l_stop:
            offs.append(offs[-1] + len(st))
        stop_seq = torch.tensor([t for st in l_stop for t in st], dtype=torch.long, device=dev)
        stop_off = torch.tensor(offs, dtype=torch.int32, device=dev)
        stopped = torch.zeros(bsz, dtype=torch.bool, device=dev)
        stop_pos = torch.full((bsz,), start_pos + 1, dtype=torch.long, device=dev)
        live = torch.full((1,), bsz, dtype=torch.int32, device=dev)      # rows still generating (device counter)
        uni = sampled = None
        if temperature > 0:
            n_steps = max(total_len - start_pos, 1)
            uni = (torch.rand(n_steps, bsz, device=dev, dtype=torch.float32) if sample_uniforms is None
                   else sample_uniforms.to(device=dev, dtype=torch.float32).contiguous())
            assert uni.shape[0] >= total_len - start_pos and uni.shape[1] == bsz, tuple(uni.shape)
            sampled = torch.empty(bsz, dtype=torch.long, device=dev)

        # One iteration = the model step (one C call for a decode step) + ONE launch of a3v_generate_step, which does everything
        # meta.py:456-477 does with a dozen small torch ops: argmax, teacher forcing of longer prompts, the tokens[:, cur_pos]
        # write, stop_pos / stopped bookkeeping and the multi-token stop match.  The host only reads `live` every poll_every steps.
        for cur_pos in range(start_pos, total_len):
            if depth_images is not None:                     # two-image plugin (meta.py:447-451)
                if prev_pos == 0:
                    logits = self.llma.forward_inference(tokens[:, prev_pos:cur_pos], prev_pos, images, depth_images)
                else:
                    logits = self.llma.forward_inference(tokens[:, prev_pos:cur_pos], prev_pos)
            else:
                logits = self.llma.forward_inference(tokens[:, prev_pos:cur_pos], prev_pos,
                                                     images if prev_pos == 0 else None)
            if temperature > 0:
                self._sample_into(logits, temperature, top_p, uni[cur_pos - start_pos], sampled)
            ops.generate_step(logits, sampled, tokens, text_mask, cur_pos, stop_seq, stop_off, len(l_stop), stopped, stop_pos, live)
            if ((cur_pos - start_pos) % poll_every == poll_every - 1 or cur_pos == total_len - 1) and int(live.item()) == 0:
                break
            prev_pos = cur_pos

        decoded, ids = [], []
        sp = stop_pos.tolist()
        for i, t in enumerate(tokens.tolist()):
            t = t[len(prompt_tokens[i]):sp[i]]
            ids.append(t)
            decoded.append(self.tokenizer.decode(t))
        return (decoded, ids) if return_ids else decoded

    @torch.no_grad()
    def stream_generate(self, prompt: str, image: Optional[torch.Tensor] = None, max_gen_len: int = 512,
                        temperature: float = 0.0, top_p: float = 0.95,
                        additional_stop_symbols: Iterable[str] = ()):
        """meta.py:487-566."""
        args = self.llma.args
        dev = self._device
        prompt_tokens = self.tokenizer.encode(prompt, bos=True, eos=False)
        max_seq_len = args.max_seq_len
        if image is not None:
            max_seq_len -= self.llma.image_words
            if image.dim() == 4:
                assert image.shape[0] == 1
            else:
                assert image.dim() == 3
                image = image.unsqueeze(0)
            image = image.to(dev)
        prompt_tokens = prompt_tokens[-(max_seq_len - max_gen_len):]
        prompt_size = len(prompt_tokens)
        total_len = min(max_seq_len, max_gen_len + prompt_size)
        tokens = torch.zeros(total_len, dtype=torch.long, device=dev)
        tokens[:prompt_size] = torch.tensor(prompt_tokens, dtype=torch.long)
        start_pos, prev_pos, generate_until = prompt_size, 0, prompt_size
        nt_buf = torch.empty(1, dtype=torch.long, device=dev)
        for cur_pos in range(start_pos, total_len):
            logits = self.llma.forward_inference(tokens[None, prev_pos:cur_pos], prev_pos,
                                                 image if prev_pos == 0 else None)
            if temperature > 0:
                nt = self._sample_into(logits, temperature, top_p, torch.rand(1, device=dev), nt_buf)
            else:
                nt = ops.argmax(logits, nt_buf)
            nt = int(nt.reshape(-1)[0].item())
            if nt == self.tokenizer.eos_id:
                break
            tokens[cur_pos] = nt
            prev_pos = cur_pos
            generate_until = cur_pos + 1
            generated = self.tokenizer.decode(tokens[start_pos:generate_until].tolist())
            for s in additional_stop_symbols:
                sp = generated.find(s)
                if sp != -1:
                    yield {"text": generated[:sp], "end_of_content": True}
                    return
            yield {"text": generated, "end_of_content": False}
        generated = self.tokenizer.decode(tokens[start_pos:generate_until].tolist())
        yield {"text": generated, "end_of_content": True}

    def _sample_into(self, logits: torch.Tensor, temperature: float, top_p: float, uniforms: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """One top-p draw per row into ``out``: the device kernel where it applies (V <= 65536, 0 < top_p), otherwise the reference's
        sort / cumsum / multinomial recipe (meta.py:456-458, 568-583) -- larger vocabularies must keep working as they do there."""
        if logits.shape[-1] <= 65536 and top_p > 0:
            return ops.sample_top_p(logits, temperature, top_p, uniforms, out)
        probs = torch.softmax(logits.float() / temperature, dim=-1)
        out.copy_(self.sample_top_p(probs, top_p).reshape(-1))
        return out

    def sample_top_p(self, probs: torch.Tensor, p: float) -> torch.Tensor:
        """meta.py:568-583, kept for callers of the reference's method (torch ops; ``generate`` itself draws on the device with
        ``a3v_sample_top_p``, same nucleus, explicit uniforms)."""
        ranked, order = probs.sort(dim=-1, descending=True)
        before = ranked.cumsum(dim=-1) - ranked                 # mass ranked ahead of each token
        kept = ranked.masked_fill(before > p, 0.0)
        draw = torch.multinomial(kept / kept.sum(dim=-1, keepdim=True), num_samples=1)
        return order.gather(-1, draw)

    def get_image_words(self) -> int:
        return self.llma.image_words

    def get_quant_blocklist(self) -> List[str]:
        if hasattr(self.llma, "get_quant_blocklist"):
            return ["llma." + x for x in self.llma.get_quant_blocklist()]
        return []

    def get_basic_block_classes(self):
        if hasattr(self.llma, "get_basic_block_classes"):
            return self.llma.get_basic_block_classes()
        return [type(self.llma.layers[0])]
