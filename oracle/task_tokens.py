"""Oracle (TEST INFRASTRUCTURE, never imported by the product): CPU restatement of PowerPaint's task-prompt token path.

Follows /root/reference/powerpaint/utils/utils.py literally -- the output is *assembled by concatenation*, as the
reference does it, rather than through the per-position source map the product computes (powerpaint_amd.utils
.splice_plan), so the two are independent derivations of the same rule.

Pinned: tests/golden/ref_task_tokens.json holds outputs of the reference's own TokenizerWrapper / add_tokens /
EmbeddingLayerWithFixes (imported unmodified in the build container by tests/golden/make_ref_task_tokens.py);
tests/test_oracle.py checks every function below against it.
"""
from typing import List, Sequence

import numpy as np


def expand_placeholders(text, token_map: dict, prop_tokens_to_load: float = 1.0):
    """utils.py:140-170 without the shuffle: every placeholder keyword -> its sub-tokens joined by blanks; the
    replacement runs in registration order on the progressively rewritten string."""
    if isinstance(text, list):
        return [expand_placeholders(t, token_map) for t in text]              # :159-161 drops prop_tokens_to_load
    for key in token_map:
        if key in text:
            parts = token_map[key]
            parts = parts[: 1 + int(len(parts) * prop_tokens_to_load)]
            text = " ".join(parts).join(text.split(key))
    return text


def placeholder_names(placeholder: str, num_vec_per_token: int) -> List[str]:
    """utils.py:120-129."""
    return [placeholder] if num_vec_per_token == 1 else [placeholder + "_%d" % i for i in range(num_vec_per_token)]


def zero_external_ids(ids: np.ndarray, num_embeddings: int) -> np.ndarray:
    """utils.py:378-389."""
    out = np.array(ids, copy=True)
    out[out >= num_embeddings] = 0
    return out


def splice_one(ids_1d: np.ndarray, emb: np.ndarray, ext: dict) -> np.ndarray:
    """utils.py:391-446 for one sequence and one external embedding: walk (s, e) over the ids, cut the running
    embedding into pieces and glue the external block in."""
    start, end = int(ext["start"]), int(ext["end"])
    block = np.asarray(ext["embedding"])
    if not np.any(ids_1d == start):
        return emb
    n = len(ids_1d)
    pieces = []
    s = e = 0
    while e < n:
        if ids_1d[e] == start:
            if e != 0:
                pieces.append(emb[s:e])
            found = [int(v) for v in ids_1d[e:e + end - start]]
            if found != list(range(start, end)):
                raise AssertionError("Invalid 'input_ids' for embedding '%s': %r" % (ext["name"], found))
            pieces.append(block)
            s = e + end - start
            e = s + 1
        else:
            e += 1
    if e == n:
        pieces.append(emb[s:e])
    return np.concatenate(pieces, axis=0)


def embedding_with_fixes(ids: np.ndarray, weight: np.ndarray, externals: Sequence[dict]) -> np.ndarray:
    """utils.py:448-483: [batch, length] ids -> [batch, length, dim]."""
    ids = np.asarray(ids)
    if ids.ndim == 1:
        ids = ids[None]
    if not externals:
        return weight[ids]
    base = weight[zero_external_ids(ids, weight.shape[0])]
    rows = []
    for seq, emb in zip(ids, base):
        for ext in externals:
            emb = splice_one(seq, emb, ext)
        rows.append(emb)
    return np.stack(rows)


def task_prompts(prompt: str, negative_prompt: str, task: str, version: str = "ppt-v1"):
    """/root/reference/app.py:38-66."""
    pos = neg = ""
    tail = ", worst quality, low quality, normal quality, bad quality, blurry "
    if task in ("object-removal", "image-outpainting"):
        if version == "ppt-v1":
            pos, neg = "empty scene blur " + prompt, negative_prompt
        return pos + " P_ctxt", pos + " P_ctxt", neg + " P_obj", neg + " P_obj"
    if version == "ppt-v1":
        pos, neg = prompt, negative_prompt + tail
    if task == "shape-guided":
        return pos + " P_shape", pos + " P_ctxt", neg + "P_shape", neg + "P_ctxt"
    return pos + " P_obj", pos + " P_obj", neg + "P_obj", neg + "P_obj"
