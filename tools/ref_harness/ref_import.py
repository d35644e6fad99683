"""Import the read-only reference (/root/reference) inside THIS container so that golden
vectors can be generated from the reference's own code.

Harness-side measures only; nothing under /root/reference is edited and nothing from it is
copied.  This module must never be imported by the product, by tests, or on the GPU box."""
import argparse
import collections
import collections.abc
import os
import sys

import numpy as np
import torch

REF = os.environ.get("CST_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    stubs = os.path.join(_HERE, "stubs")
    for p in (REF, stubs):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REF)
    sys.path.insert(0, stubs)
    # numpy>=1.24 / py>=3.10 aliases the 2020-era code still uses
    for name, typ in (("float", float), ("int", int), ("bool", bool), ("object", object)):
        if not hasattr(np, name):
            setattr(np, name, typ)
    for name in ("Collection", "Iterable", "Mapping", "Sequence", "MutableMapping", "Callable"):
        if not hasattr(collections, name):
            setattr(collections, name, getattr(collections.abc, name))
    torch.serialization.add_safe_globals([argparse.Namespace])
    import fairseq  # noqa: F401
    from fairseq.modules import transformer_layer as tl

    # torch>=2.x F.multi_head_attention_forward rejects the float64 memory mask
    # (w2v2_transformer_interlingua.py:284-288); cast to x.dtype, values unchanged (0 / -1e8).
    if not getattr(tl.TransformerEncoderLayer, "_cst_patched", False):
        orig = tl.TransformerEncoderLayer.forward

        def fwd(self, x, encoder_padding_mask, attn_mask=None):
            if attn_mask is not None:
                attn_mask = attn_mask.to(x.dtype)
            return orig(self, x, encoder_padding_mask, attn_mask)

        tl.TransformerEncoderLayer.forward = fwd
        tl.TransformerEncoderLayer._cst_patched = True
    return fairseq
