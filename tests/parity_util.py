"""Shared helpers of the parity tests: run the oracle (CPU) on a parameter dict, with or without storage-rounding emulation,
and compare gradient sets in norm.  The oracle is the checker only."""
import numpy as np
import torch

from oracle import chimera_oracle as O


def oracle_leaves(state_dict, storage=None):
    """Oracle parameter dict from a (CPU, any dtype) state dict: fp32 leaves (rounded to `storage` first when emulating
    reduced-precision storage), tied output projection shared with the embedding as the reference ties it."""
    p = {}
    for k, v in state_dict.items():
        t = v.detach().to("cpu")
        if t.is_floating_point():
            t = t.float()
            if storage is not None:
                t = t.to(storage).float()
            t = t.clone().requires_grad_("_float_tensor" not in k and k != "decoder.version")
        p[k] = t
    if "decoder.embed_tokens.weight" in p:
        p["decoder.output_projection.weight"] = p["decoder.embed_tokens.weight"]
    return p


def run_oracle(fn, state_dict, sample, cfg, storage=None, **kw):
    """fn = O.triplet_criterion / O.lsce_criterion; returns (outputs, {name: grad}) of one forward + backward."""
    prev = O.STORAGE
    O.STORAGE = storage
    try:
        p = oracle_leaves(state_dict, storage)
        out = fn(p, sample, cfg, **kw)
        out["loss"].backward()
    finally:
        O.STORAGE = prev
    grads = {k: v.grad for k, v in p.items() if k != "decoder.output_projection.weight" and v.requires_grad}
    return out, grads


def cpu_sample(sample):
    def mv(x):
        if torch.is_tensor(x):
            return x.cpu()
        if isinstance(x, dict):
            return {k: mv(v) for k, v in x.items()}
        return x
    return mv(sample)


def grad_errors(got, ref):
    """got / ref: {name: tensor or None}.  Returns (global relative L2 error, {name: (rel L2 error, share of the total norm)})."""
    num = den = 0.0
    per = {}
    for k, r in ref.items():
        r = (r if r is not None else torch.zeros(1)).double().cpu()
        g = got.get(k)
        g = torch.zeros_like(r) if g is None else g.detach().double().cpu()
        assert torch.isfinite(g).all(), k
        e, n = float(((g - r) ** 2).sum()), float((r ** 2).sum())
        num += e
        den += n
        per[k] = (e, n)
    return (num / den) ** 0.5, {k: ((e / n) ** 0.5 if n > 0 else 0.0, (n / den) ** 0.5) for k, (e, n) in per.items()}


def max_abs_rel(got, ref):
    """max |got - ref| / max(1, |ref|max): the north_star's '1e-3' measure."""
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu() if torch.is_tensor(ref) else torch.from_numpy(np.asarray(ref, dtype=np.float32))
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return float((got - ref).abs().max()) / max(1.0, float(ref.abs().max()))
