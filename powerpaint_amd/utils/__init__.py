"""Host-side helpers of the hot path's callers (mirrors /root/reference/powerpaint/utils/__init__.py)."""
from .utils import EmbeddingLayerWithFixes, TokenizerWrapper, add_task, add_tokens, splice_plan

__all__ = ["TokenizerWrapper", "EmbeddingLayerWithFixes", "add_tokens", "add_task", "splice_plan"]
