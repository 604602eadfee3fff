"""Task-prompt tokens: the learned `P_ctxt` / `P_shape` / `P_obj` placeholders of PowerPaint.

Host-side mirror of /root/reference/powerpaint/utils/utils.py (same class / function names, argument meaning and
error behaviour) for the inference path:

  TokenizerWrapper          utils.py:15-254   "P_obj" -> "P_obj_0 ... P_obj_9" before the wrapped CLIPTokenizer runs
  EmbeddingLayerWithFixes   utils.py:257-483  ids >= num_embeddings -> row 0, then the learned block is spliced in
  add_tokens                utils.py:486-530  registers the placeholders on both objects
  add_task                  app.py:38-66      task name -> (promptA, promptB, negative_promptA, negative_promptB)

The integer part (which row of which table every output position takes) is decided on the host by `splice_plan`,
a restatement of the reference's left-to-right scan *including* its corner cases (see that function).  The gather
itself is one launch of `pp_embed_splice` (include/pp_hip.h) on the device the embedding table lives on; there is no
CPU execution path -- a table that is not on the GPU raises.
"""
import copy
import os
import random
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L


# ------------------------------------------------------------------------------------------------------ tokenizer
class TokenizerWrapper:
    """CLIPTokenizer plus multi-vector placeholder tokens (utils.py:15-254).

    Unknown attributes are forwarded to the wrapped tokenizer, so the object can be registered as `pipe.tokenizer`.
    `tokenizer=` accepts an already constructed tokenizer (any object with `add_tokens`, `__call__`, `decode`);
    otherwise `from_pretrained` is handed to `transformers.CLIPTokenizer.from_pretrained` like the reference does.
    """

    def __init__(self, from_pretrained: Optional[Union[str, os.PathLike]] = None,
                 from_config: Optional[Union[str, os.PathLike]] = None, *args, tokenizer=None, **kwargs):
        if from_pretrained and from_config:
            raise AssertionError("'from_pretrained' and 'from_config' should not be passed at the same time.")
        if from_config:                                   # utils.py:49-58: HF tokenizers have no from_config
            from_pretrained = from_config
        if tokenizer is not None:
            wrapped = tokenizer
        else:
            import transformers
            if from_pretrained:
                wrapped = transformers.CLIPTokenizer.from_pretrained(from_pretrained, *args, **kwargs)
            else:
                wrapped = transformers.CLIPTokenizer(*args, **kwargs)
        self.__dict__["wrapped"] = wrapped
        self._from_pretrained = from_pretrained
        self.token_map: Dict[str, List[str]] = {}

    def __getattr__(self, name: str) -> Any:
        # only reached when normal lookup fails
        wrapped = self.__dict__.get("wrapped")
        if name == "wrapped" or wrapped is None:
            raise AttributeError(name)
        try:
            return getattr(wrapped, name)
        except AttributeError:
            raise AttributeError(f"'{name}' cannot be found in both '{type(self).__name__}' and "
                                 f"'{type(self).__name__}.tokenizer'.") from None

    def try_adding_tokens(self, tokens: Union[str, List[str]], *args, **kwargs):
        """utils.py:83-94: adding a token the vocabulary already holds is an error."""
        added = self.wrapped.add_tokens(tokens, *args, **kwargs)
        assert added != 0, (f"The tokenizer already contains the token {tokens}. Please pass a different "
                            "`placeholder_token` that is not already in the tokenizer.")

    def get_token_info(self, token: str) -> dict:
        """Id range [start, end) a placeholder occupies: first id after BOS .. last id before EOS (utils.py:96-109)."""
        ids = self(token).input_ids
        return {"name": token, "start": ids[1], "end": ids[-2] + 1}

    def add_placeholder_token(self, placeholder_token: str, *args, num_vec_per_token: int = 1, **kwargs):
        """Register `placeholder_token` as `num_vec_per_token` fresh vocabulary entries (utils.py:111-138)."""
        if num_vec_per_token == 1:
            pieces = [placeholder_token]
        else:
            pieces = [f"{placeholder_token}_{i}" for i in range(num_vec_per_token)]
        for p in pieces:
            self.try_adding_tokens(p, *args, **kwargs)
        for known in self.token_map:
            if known in placeholder_token:                # substring clash, checked after the vocabulary grew (as :131)
                raise ValueError(f"The tokenizer already has placeholder token {known} that can get confused with "
                                 f"{placeholder_token} keep placeholder tokens independent")
        self.token_map[placeholder_token] = pieces

    def replace_placeholder_tokens_in_text(self, text: Union[str, List[str]], vector_shuffle: bool = False,
                                           prop_tokens_to_load: float = 1.0) -> Union[str, List[str]]:
        """"a P_obj" -> "a P_obj_0 P_obj_1 ..." (utils.py:140-170).  For a list, `prop_tokens_to_load` is NOT
        forwarded to the items (the reference's recursion drops it, :159-161); kept."""
        if isinstance(text, list):
            return [self.replace_placeholder_tokens_in_text(t, vector_shuffle=vector_shuffle) for t in text]
        for placeholder, pieces in self.token_map.items():
            if placeholder not in text:
                continue
            pieces = pieces[: 1 + int(len(pieces) * prop_tokens_to_load)]
            if vector_shuffle:
                pieces = copy.copy(pieces)
                random.shuffle(pieces)
            text = text.replace(placeholder, " ".join(pieces))
        return text

    def replace_text_with_placeholder_tokens(self, text: Union[str, List[str]]) -> Union[str, List[str]]:
        """Inverse mapping used by `decode` (utils.py:172-192)."""
        if isinstance(text, list):
            return [self.replace_text_with_placeholder_tokens(t) for t in text]
        for placeholder, pieces in self.token_map.items():
            joined = " ".join(pieces)
            if joined in text:
                text = text.replace(joined, placeholder)
        return text

    def __call__(self, text: Union[str, List[str]], *args, vector_shuffle: bool = False,
                 prop_tokens_to_load: float = 1.0, **kwargs):
        text = self.replace_placeholder_tokens_in_text(text, vector_shuffle=vector_shuffle,
                                                       prop_tokens_to_load=prop_tokens_to_load)
        return self.wrapped(text, *args, **kwargs)

    def encode(self, text: Union[str, List[str]], *args, **kwargs):
        """utils.py:218-226 -- note: returns the wrapped tokenizer's __call__ result, as the reference does."""
        return self.wrapped(self.replace_placeholder_tokens_in_text(text), *args, **kwargs)

    def decode(self, token_ids, return_raw: bool = False, *args, **kwargs) -> Union[str, List[str]]:
        text = self.wrapped.decode(token_ids, *args, **kwargs)
        return text if return_raw else self.replace_text_with_placeholder_tokens(text)

    def __repr__(self):
        src = f", from_pretrained={self._from_pretrained!r}" if self._from_pretrained else ""
        return f"TokenizerWrapper({type(self.wrapped).__name__}{src}, placeholders={list(self.token_map)})"


# ------------------------------------------------------------------------------------------------------ splice plan
def splice_plan(input_ids: np.ndarray, num_embeddings: int, spans: Sequence[dict]) -> np.ndarray:
    """Source row of every output position of `EmbeddingLayerWithFixes.forward` (utils.py:378-483), as int32.

    input_ids : [batch, length] integer array.
    spans     : dicts with 'name', 'start', 'end', 'row0' -- id range [start, end) of an external embedding whose rows
                sit at row0.. of the concatenated external table.
    returns   : [batch, length]; v >= 0 -> base-table row v, v < 0 -> external row (-v - 1).

    What the reference's scan does, position by position, and is reproduced here:
      * every id >= num_embeddings is looked up as row 0 of the base table (:378-389);
      * for each external embedding, in registration order, a cursor walks the ids; where it meets `start` it requires
        the next (end - start) ids to be exactly start..end-1 (AssertionError otherwise, also when the run is cut short
        by the end of the sequence) and takes the external block there (:419-436);
      * the position right after a block is never tested against `start` (:438-439 `s_idx = e_idx + n;
        e_idx = s_idx + 1`), so a second run directly adjacent to a first one is not spliced and keeps row 0;
      * ids of a range that do not belong to a spliced run (orphans) keep row 0.
    """
    ids = np.asarray(input_ids)
    if ids.ndim != 2:
        raise ValueError("input_ids must be [batch, length]")
    B, n = ids.shape
    plan = np.where(ids >= num_embeddings, 0, ids).astype(np.int64)
    for sp in spans:
        start, end, row0, name = int(sp["start"]), int(sp["end"]), int(sp["row0"]), sp["name"]
        width = end - start
        want = list(range(start, end))
        for b in range(B):
            row = ids[b]
            if not (row == start).any():
                continue
            i = 0
            while i < n:
                if row[i] != start:
                    i += 1
                    continue
                got = [int(v) for v in row[i:i + width]]
                assert got == want, (f"Invalid 'input_ids' in position: {i} to {i + width}. Expect '{want}' for "
                                     f"embedding '{name}' but found '{got}'.")
                plan[b, i:i + width] = -(row0 + np.arange(width)) - 1
                i += width + 1                             # the slot right after a block is skipped, see docstring
    return plan.astype(np.int32)


# ------------------------------------------------------------------------------------------------------ embedding
class EmbeddingLayerWithFixes(nn.Module):
    """nn.Embedding with external (placeholder) embeddings spliced in (utils.py:257-483); inference only.

    `external_embeddings` entries are dicts {name, embedding [n_vec, dim], start, end[, trainable]}.  A `trainable`
    entry is registered in `self.trainable_embeddings` under the same key the reference uses, so the reference's
    `text_encoder` state dicts (…token_embedding.trainable_embeddings.P_obj, …wrapped.weight) load unchanged.
    """

    def __init__(self, wrapped: nn.Embedding, external_embeddings: Optional[Union[dict, List[dict]]] = None):
        super().__init__()
        self.wrapped = wrapped
        self.num_embeddings = wrapped.weight.shape[0]
        self.external_embeddings: List[dict] = []
        self.trainable_embeddings = nn.ParameterDict()
        self._packed = None                               # (key, table, spans)
        if external_embeddings:
            self.add_embeddings(external_embeddings)

    @property
    def weight(self):
        return self.wrapped.weight

    def check_duplicate_names(self, embeddings: List[dict]):
        names = [e["name"] for e in embeddings]
        assert len(names) == len(set(names)), f"Found duplicated names in 'external_embeddings'. Name list: '{names}'"

    def check_ids_overlap(self, embeddings: List[dict]):
        ranges = sorted([e["start"], e["end"], e["name"]] for e in embeddings)
        for (s0, e0, n0), (s1, e1, n1) in zip(ranges, ranges[1:]):
            assert e0 <= s1, f"Found ids overlapping between embeddings '{n0}' and '{n1}'."

    def add_embeddings(self, embeddings: Optional[Union[dict, List[dict]]]):
        """utils.py:312-376."""
        if isinstance(embeddings, dict):
            embeddings = [embeddings]
        self.external_embeddings += embeddings
        self.check_duplicate_names(self.external_embeddings)
        self.check_ids_overlap(self.external_embeddings)
        for e in embeddings:
            if e.get("trainable", False):
                e["embedding"] = nn.Parameter(e["embedding"], requires_grad=e["embedding"].is_floating_point())
                self.trainable_embeddings[e["name"]] = e["embedding"]
        self._packed = None

    def replace_input_ids(self, input_ids: torch.Tensor) -> torch.Tensor:
        """ids the base table does not hold -> 0 (utils.py:378-389)."""
        out = input_ids.clone()
        out[out >= self.num_embeddings] = 0
        return out

    # -- device side
    def _pack(self, extra: List[dict]):
        """All external blocks concatenated along rows in the base table's promoted dtype, cached until a block's
        storage or version changes (load_state_dict / .to() rewrite the Parameters in place)."""
        embs = self.external_embeddings + extra
        w = self.wrapped.weight
        key = tuple((e["name"], e["start"], e["end"], e["embedding"].data_ptr(), e["embedding"]._version,
                     e["embedding"].dtype, str(e["embedding"].device)) for e in embs) + (w.dtype, str(w.device))
        if self._packed is not None and self._packed[0] == key and not extra:
            return self._packed[1], self._packed[2]
        dt = w.dtype
        for e in embs:
            dt = torch.promote_types(dt, e["embedding"].dtype)   # what torch.cat does in utils.py:446
        spans, blocks, row0 = [], [], 0
        for e in embs:
            blk = e["embedding"].detach()
            width = int(e["end"]) - int(e["start"])
            if blk.dim() != 2 or blk.shape[0] != width or blk.shape[1] != w.shape[1]:
                raise L.PPError(f"external embedding '{e['name']}' has shape {tuple(blk.shape)}, expected "
                                f"({width}, {w.shape[1]}) for ids [{e['start']}, {e['end']})")
            spans.append(dict(name=e["name"], start=int(e["start"]), end=int(e["end"]), row0=row0))
            blocks.append(blk.to(device=w.device, dtype=dt))
            row0 += width
        table = torch.cat(blocks).contiguous() if blocks else None
        if not extra:
            self._packed = (key, table, spans)
        return table, spans

    def _gather(self, table: torch.Tensor, ext: Optional[torch.Tensor], plan: np.ndarray) -> torch.Tensor:
        if not table.is_cuda:
            raise L.PPError("EmbeddingLayerWithFixes: the embedding table must live on the GPU (no CPU path)")
        table = table if table.is_contiguous() else table.contiguous()
        n = plan.size
        src = torch.from_numpy(np.ascontiguousarray(plan.reshape(-1))).to(table.device, non_blocking=False)
        out = torch.empty((n, table.shape[1]), dtype=table.dtype, device=table.device)
        with torch.cuda.device(table.device):
            L.check(L.lib().pp_embed_splice(table.data_ptr(), ext.data_ptr() if ext is not None else None,
                                            src.data_ptr(), out.data_ptr(), n, table.shape[1] * table.element_size(),
                                            torch.cuda.current_stream().cuda_stream), "pp_embed_splice")
        return out

    def forward(self, input_ids: torch.Tensor, external_embeddings: Optional[Union[dict, List[dict]]] = None):
        """input_ids [batch, length] or [length] -> [batch, length, dim] (utils.py:448-483)."""
        assert input_ids.ndim in [1, 2]
        if input_ids.ndim == 1:
            input_ids = input_ids.unsqueeze(0)
        if external_embeddings is None:
            extra = []
        elif isinstance(external_embeddings, dict):
            extra = [external_embeddings]
        else:
            extra = list(external_embeddings)
        w = self.wrapped.weight.detach()
        ext, spans = self._pack(extra)
        ids_host = input_ids.detach().cpu().numpy()
        if ids_host.size and (ids_host.min() < 0 or (ext is None and ids_host.max() >= self.num_embeddings)):
            raise IndexError("index out of range in self")          # what nn.Embedding raises
        plan = splice_plan(ids_host, self.num_embeddings, spans)
        B, n = ids_host.shape
        if ext is None or ext.dtype == w.dtype:
            out = self._gather(w, ext, plan)
        else:
            # mixed dtypes: torch.cat in the reference promotes; gather the base rows first, promote them, then splice
            base = self._gather(w, None, np.maximum(plan, 0)).to(ext.dtype)
            pos = np.arange(plan.size, dtype=np.int32).reshape(plan.shape)
            out = self._gather(base, ext, np.where(plan >= 0, pos, plan).astype(np.int32))
        return out.view(B, n, -1)


def add_tokens(tokenizer, text_encoder, placeholder_tokens: list, initialize_tokens: list = None,
               num_vectors_per_token: int = 1):
    """Register placeholder tokens on a TokenizerWrapper and a CLIP text encoder (utils.py:486-530).

    Each placeholder gets `num_vectors_per_token` vectors, initialised from the embedding of `initialize_tokens[i]`
    (its first token) or, without it, uniformly in [-0.25, 0.25).  The learned values arrive afterwards with the
    checkpoint's text-encoder state dict (app.py:110-117).
    """
    if initialize_tokens is not None:
        assert len(initialize_tokens) == len(placeholder_tokens), \
            "placeholder_token should be the same length as initialize_token"
    for p in placeholder_tokens:
        tokenizer.add_placeholder_token(p, num_vec_per_token=num_vectors_per_token)
    # transformers 4.x (the reference's pin) keeps the embeddings under `.text_model`; 5.x flattened CLIPTextModel
    holder = getattr(text_encoder, "text_model", text_encoder).embeddings
    layer = holder.token_embedding
    if not isinstance(layer, EmbeddingLayerWithFixes):     # powerpaint_amd.models.CLIPTextModel is born wrapped
        layer = EmbeddingLayerWithFixes(layer)
        holder.token_embedding = layer
    init = []
    for i, _ in enumerate(placeholder_tokens):
        if initialize_tokens is not None:
            row = layer.weight[tokenizer(initialize_tokens[i]).input_ids[1]]
            init.append(row.detach()[None, ...].repeat(num_vectors_per_token, 1))
        else:
            dim = layer.weight.shape[1]
            init.append((torch.rand(num_vectors_per_token, dim) - 0.5) / 2.0)
    infos = []
    for p, e in zip(placeholder_tokens, init):
        info = tokenizer.get_token_info(p)
        info["embedding"] = e
        info["trainable"] = True
        infos.append(info)
    layer.add_embeddings(infos)


# ------------------------------------------------------------------------------------------------------ task prompts
def add_task(prompt: str, negative_prompt: str, control_type: str, version: str = "ppt-v1"):
    """Task name -> the four prompts the pipelines blend (app.py:38-66).

    'object-removal' / 'image-outpainting' use P_ctxt against P_obj, 'shape-guided' blends P_shape with P_ctxt by the
    fitting degree, everything else ('text-guided', …) uses P_obj.  For versions other than 'ppt-v1' the user text is
    dropped from these prompts (the BrushNet pipeline takes it through `prompt=` instead, app.py:397-405).
    """
    quality = ", worst quality, low quality, normal quality, bad quality, blurry "
    v1 = version == "ppt-v1"
    if control_type in ("object-removal", "image-outpainting"):
        pos = ("empty scene blur " + prompt) if v1 else ""
        neg = negative_prompt if v1 else ""
        return pos + " P_ctxt", pos + " P_ctxt", neg + " P_obj", neg + " P_obj"
    pos = prompt if v1 else ""
    neg = (negative_prompt + quality) if v1 else ""
    if control_type == "shape-guided":
        return pos + " P_shape", pos + " P_ctxt", neg + "P_shape", neg + "P_ctxt"
    return pos + " P_obj", pos + " P_obj", neg + "P_obj", neg + "P_obj"
