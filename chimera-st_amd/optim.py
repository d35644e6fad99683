"""Optimizer path: the reference chains FP16Optimizer (fp16 model + flat fp32 master,
optim/fp16_optimizer.py:16-300) -> multiply_grads -> utils.clip_grad_norm_ (utils.py:323-364) -> Adam.step
(optim/adam.py:146-226) -> copy back, i.e. 5-6 passes over parameter-sized buffers.  Here:
  * model parameters AND gradients live in two flat buffers (params/grads are views) — the gradient buffer is what
    the data-parallel wrapper all-reduces bucket by bucket;
  * one cst_sumsq pass gives the global grad norm on device;
  * one cst_adam_step pass applies  g * (world/sample_size) * clip_coef,  Adam with fairseq's semantics on the fp32
    master, and writes the model-dtype parameter.  The combined scale is a DEVICE scalar: no host sync in the step.
Flags mirror fairseq: --adam-betas --adam-eps --weight-decay --lr --clip-norm --warmup-updates --warmup-init-lr."""
import math
import weakref

import torch

from . import kernels as K
from .profiling import scope

ALIGN = 8  # elements: keeps every parameter view 16-byte aligned (bf16) for the GEMM loaders

# Bumped by every write of the flat parameter buffer that the parameters' own autograd version counters do not see: cst_adam_step
# (raw pointers) and copies into `flat_param` itself (each parameter was re-homed with `p.data = view` and counts on its own).
# Anything that caches tensors DERIVED from parameters (decode_engine.BeamDecodeEngine._pack, functional.WEIGHT_TRANSPOSES) keys on it.
PARAM_EPOCH = [0]


class FlatParamBuffers:
    """Re-home a model's parameters (and their .grad) as views of two flat tensors."""

    def __init__(self, params, adjacent=None, align=ALIGN):
        """`align` (elements, a multiple of ALIGN): every parameter's slot starts at a multiple of it — the sharded optimizer asks for
        ALIGN x world so that the 1 / world shards of every bucket keep the kernels' 16-byte alignment.
        `adjacent`: groups of parameters to be laid out back to back, in the given order (the q | k | v projection weights and
        biases of a self-attention module: the packed [3C, C] projection is then a VIEW of the flat buffer instead of a torch.cat
        per layer and update).  Only the storage layout changes; `self.params` keeps model.parameters() order, which is the index
        space of the reference's optimizer state (optim/fairseq_optimizer.py)."""
        self.params = [p for p in params if p.requires_grad]
        assert len(self.params) > 0
        self.dtype, self.device = self.params[0].dtype, self.params[0].device
        index = {id(p): i for i, p in enumerate(self.params)}
        lead, follow = {}, set()
        for grp in adjacent or []:
            ids = [index[id(p)] for p in grp if id(p) in index]
            if len(ids) == len(grp) and all(i not in follow and i not in lead for i in ids):
                lead[min(ids)] = ids  # the group takes the slot of its first member in parameter order
                follow.update(i for i in ids if i != min(ids))
        order = []
        for i in range(len(self.params)):
            if i in lead:
                order.extend(lead[i])
            elif i not in follow:
                order.append(i)
        assert align >= ALIGN and align % ALIGN == 0
        self.align = int(align)
        self.offsets, total = [0] * len(self.params), 0
        for i in order:
            p = self.params[i]
            assert p.dtype == self.dtype and p.device == self.device, "all trainable parameters must share dtype/device"
            self.offsets[i] = total
            total += (p.numel() + self.align - 1) // self.align * self.align
        self.total = total
        self.flat_param = torch.zeros(total, dtype=self.dtype, device=self.device)
        self.flat_grad = torch.zeros(total, dtype=self.dtype, device=self.device)
        self.grad_views = []
        self.epoch = 1  # bumped by zero_grad: a gradient slot is handed out at most once per epoch (grad_slot)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.flat_param[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o:o + n].view(p.shape)
            self.grad_views.append(self.flat_grad[o:o + n].view(p.shape))
            p.grad = None
            # functional._linear_backward writes a weight gradient straight into its slot of the flat buffer (grad_slot below)
            p._cst_grad_slot = (weakref.ref(self), len(self.grad_views) - 1)
            p._cst_grad_claim = 0
        if self.flat_param.is_cuda:
            from .functional import WEIGHT_TRANSPOSES
            WEIGHT_TRANSPOSES.register_storage(self, self.flat_param)  # views of this buffer may be cached as W^T copies

    def zero_grad(self):
        """Gradients are left to autograd as free-standing tensors (p.grad = None lets AccumulateGrad STEAL the tensor our
        backward kernels wrote, instead of launching one `grad += new` kernel per parameter); they are gathered into the flat
        buffer bucket-by-bucket (gather_grads) right before they are all-reduced / consumed by the fused optimizer."""
        self.flat_grad.zero_()
        self.epoch += 1
        for p in self.params:
            p.grad = None

    def gather_grads(self, indices=None):
        """Copy the autograd-owned gradients of the given parameters into their flat-buffer slots with ONE multi-tensor copy;
        parameters without a gradient keep zeros (legacy_distributed_data_parallel.py:155-156)."""
        idxs = range(len(self.params)) if indices is None else indices
        dsts, srcs = [], []
        for i in idxs:
            g = self.params[i].grad
            if g is not None and g.data_ptr() != self.grad_views[i].data_ptr():
                dsts.append(self.grad_views[i])
                srcs.append(g if g.shape == self.grad_views[i].shape else g.view(self.grad_views[i].shape))
        if dsts:
            torch._foreach_copy_(dsts, srcs)
            for i in idxs:
                if self.params[i].grad is not None:
                    self.params[i].grad = self.grad_views[i]


def grad_slot(w):
    """Where a backward kernel may write the gradient of weight `w` directly: its view of the flat gradient buffer — instead of a
    fresh tensor that gather_grads copies there later (339 MB per update on the bench model) — or None.  Handed out at most once
    per zero_grad epoch and only while w.grad is None: a second use of a shared weight in the same backward pass, or a later
    micro-batch of an accumulated update, gets None and takes the ordinary route (autograd then ADDS into the slot; writing there
    again would overwrite what it already holds).  `w` is a Parameter re-homed by FlatParamBuffers, or the stacked q | k | v view of
    three adjacent ones (functional.stacked_rows: their slots are adjacent too)."""
    parts = getattr(w, "_cst_parts_params", None)
    ps = parts if parts is not None else (w,)
    slots = []
    for p in ps:
        s = getattr(p, "_cst_grad_slot", None)
        buf = s[0]() if s is not None else None
        if (buf is None or p.grad is not None or p._cst_grad_claim == buf.epoch or not p.requires_grad
                or getattr(p, "_cst_slot_frozen", False)):  # (frozen: the slot is travelling in a bucket all-reduce, distributed.py)
            return None
        slots.append((buf, s[1], p))
    buf = slots[0][0]
    if any(b is not buf for b, _, _ in slots):
        return None
    views = [buf.grad_views[i] for _, i, _ in slots]
    if len(views) == 1:
        out = views[0].detach()  # a fresh alias: AccumulateGrad only adopts a gradient tensor nobody else holds
    else:
        n = views[0].numel() * views[0].element_size()
        if not all(v.shape == views[0].shape and views[k].data_ptr() + n == v.data_ptr() for k, v in enumerate(views[1:])):
            return None
        out = torch.as_strided(views[0], (len(views) * views[0].shape[0],) + tuple(views[0].shape[1:]), views[0].stride(), views[0].storage_offset())
    if tuple(out.shape) != tuple(w.shape):
        return None
    for _, _, p in slots:
        p._cst_grad_claim = buf.epoch
    return out


def qkv_groups(model):
    """Adjacency groups for FlatParamBuffers: the q, k, v projection weights (and biases) of every self-attention module, the k, v
    projections of every encoder-decoder attention module (they read the same rows: one [2C, C] GEMM, functional.attention_kv)."""
    groups = []
    for m in model.modules():
        if not all(hasattr(m, n) for n in ("q_proj", "k_proj", "v_proj")):
            continue
        if getattr(m, "self_attention", False):
            names = ("q_proj", "k_proj", "v_proj")
        elif getattr(m, "encoder_decoder_attention", False) and m.k_proj.weight.shape == m.v_proj.weight.shape:
            names = ("k_proj", "v_proj")
        else:
            continue
        groups.append([getattr(m, n).weight for n in names])
        if m.q_proj.bias is not None:
            groups.append([getattr(m, n).bias for n in names])
    return groups


def inverse_sqrt_lr(num_updates, lr, warmup_updates, warmup_init_lr):
    """optim/lr_scheduler/inverse_square_root_schedule.py:52-94."""
    if warmup_init_lr < 0:
        warmup_init_lr = 0 if warmup_updates > 0 else lr
    if warmup_updates > 0 and num_updates < warmup_updates:
        return warmup_init_lr + num_updates * (lr - warmup_init_lr) / warmup_updates
    decay_factor = lr * max(warmup_updates, 1) ** 0.5
    return decay_factor * max(num_updates, 1) ** -0.5 if warmup_updates > 0 else lr


class FusedAdam:
    """FairseqAdam + FP16Optimizer(flat fp32 master) + clip_grad_norm fused over flat buffers."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_norm=0.0,
                 warmup_updates=0, warmup_init_lr=-1.0, buffers=None):
        self.buf = buffers if buffers is not None else FlatParamBuffers(params)
        self.base_lr, self.betas, self.eps, self.weight_decay, self.clip_norm = lr, betas, eps, weight_decay, clip_norm
        self.warmup_updates, self.warmup_init_lr = warmup_updates, warmup_init_lr
        n, dev = self.buf.total, self.buf.device
        self.master = self.buf.flat_param.float().clone()
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.num_updates = 0
        self.lr = inverse_sqrt_lr(0, lr, warmup_updates, warmup_init_lr)
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._scale = torch.ones(1, dtype=torch.float32, device=dev)
        self.segments = None   # sharded optimizer state (shard()): the [lo, hi) spans of the flat buffers this rank owns
        self._gather = None    # ... and the collective that assembles a flat tensor from every rank's spans

    # ---- ZeRO-1 ("optimizer state sharding": --zero-sharding os, fairseq/trainer.py:241-252 + optim/shard.py, where the reference
    #      hands its optimizer to fairscale's OSS).  Behind a reduce-scatter of every gradient bucket (distributed.py, collective
    #      "rs") rank r holds the reduced gradient of ITS 1 / world span of each bucket only; it keeps fp32 master weights and both
    #      Adam moments for those spans alone (1 / world of the optimizer memory and of the update's 28 bytes per parameter), runs
    #      cst_sumsq + cst_adam_step on them, and the bf16 parameters are all-gathered bucket by bucket afterwards — the same bytes
    #      on the wire as the gradient all-gather they replace.  The update itself is element-wise: with the same gradient scale
    #      every parameter gets the bits of the unsharded route. ----
    def shard(self, segments, gather):
        segs = [(int(lo), int(hi)) for lo, hi in segments if hi > lo]
        assert all(lo % ALIGN == 0 and hi % ALIGN == 0 for lo, hi in segs), "shard spans must keep the 16-byte alignment of the kernels"
        full = (self.master, self.exp_avg, self.exp_avg_sq) if self.segments is None else self._full_state()
        self.segments, self._gather = segs, gather
        self._seg_loc, at = [], 0
        for lo, hi in segs:
            self._seg_loc.append(at)
            at += hi - lo
        self.master, self.exp_avg, self.exp_avg_sq = (self._take(t) for t in full)

    def _take(self, flat):
        """This rank's spans of a flat [total] tensor, back to back."""
        dev = self.buf.device
        if not self.segments:
            return torch.zeros(0, dtype=flat.dtype, device=dev)
        return torch.cat([flat[lo:hi] for lo, hi in self.segments]).contiguous()

    def _full_state(self):
        """(master, exp_avg, exp_avg_sq) as flat [total] fp32 tensors on EVERY rank (a collective when the state is sharded: all
        ranks must call it — checkpoints, re-sharding)."""
        if self.segments is None:
            return self.master, self.exp_avg, self.exp_avg_sq
        out = []
        for local in (self.master, self.exp_avg, self.exp_avg_sq):
            flat = torch.zeros(self.buf.total, dtype=torch.float32, device=self.buf.device)
            for (lo, hi), a in zip(self.segments, self._seg_loc):
                flat[lo:hi].copy_(local[a:a + hi - lo])
            self._gather(flat)
            out.append(flat)
        return tuple(out)

    def _spans(self):
        """(lo, hi, offset into the local state) of the spans this rank updates: the whole buffer when nothing is sharded."""
        if self.segments is None:
            return [(0, self.buf.total, 0)]
        return [(lo, hi, a) for (lo, hi), a in zip(self.segments, self._seg_loc)]

    # the two kernels behind the update (the world-2 CPU tests of the sharding logic replace them with torch arithmetic)
    def _sumsq_span(self, grad, out):
        K.sumsq(grad, out)  # out[0] += sum(grad^2): fixed-order two-stage sum, spans added in span order

    def _adam_span(self, master, m, v, grad, param):
        K.adam_step(master, m, v, grad, param, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.num_updates, self._scale)

    @classmethod
    def from_args(cls, args, params, buffers=None):
        betas = eval(args.adam_betas) if isinstance(args.adam_betas, str) else args.adam_betas
        return cls(params, lr=args.lr[0], betas=betas, eps=args.adam_eps, weight_decay=args.weight_decay,
                   clip_norm=getattr(args, "clip_norm", 0.0), warmup_updates=getattr(args, "warmup_updates", 0),
                   warmup_init_lr=getattr(args, "warmup_init_lr", -1.0), buffers=buffers)

    def zero_grad(self):
        self.buf.zero_grad()

    def backward(self, loss):
        # second stages of the small fixed-order reductions of this backward pass in one launch (kernels.DEFER): only when the
        # trainer established that every parameter receives ONE gradient per backward pass (Trainer.__init__)
        with K.deferred_reductions(getattr(self, "defer_reductions", False)):
            loss.backward()

    def get_lr(self):
        return self.lr

    def grad_sumsq(self):
        """sum(grad^2) over the flat gradient buffer as a 1-element fp32 device tensor (fixed-order two-stage sum: replicas that hold
        the same all-reduced gradients get the same bits)."""
        self._sumsq.zero_()
        for lo, hi, _ in self._spans():  # (sharded state: the sum over THIS rank's spans; the trainer adds the ranks' sums)
            self._sumsq_span(self.buf.flat_grad[lo:hi], self._sumsq)
        return self._sumsq

    def grad_norm(self, multiply=1.0):
        """|| multiply * grad ||_2 as a device tensor (what clip_grad_norm_ returns after multiply_grads)."""
        return self.grad_sumsq().sqrt() * multiply

    def step(self, multiply=1.0, gnorm=None):
        """One update.  `multiply` = world_size / sample_size (trainer.py:606).  Returns the pre-clip grad norm.
        gnorm: the norm of multiply * grad if the caller already has it on the HOST (the trainer reads the raw sum of squares back
        together with the logging vector); then the clip coefficient is computed there and no device scalar math is queued."""
        if gnorm is not None:
            with scope("multiply-grads"):  # (trainer.py:601-606 + :613: here both are ONE scalar the Adam kernel applies on its way)
                multiply = float(multiply)
                coef = min(1.0, self.clip_norm / (gnorm + 1e-6)) if self.clip_norm > 0 else 1.0
                self._scale.fill_(coef * multiply)
        else:
            gnorm = self.grad_norm(multiply)  # `multiply` may be a python float or a 1-element device tensor
            if self.clip_norm > 0:
                coef = (self.clip_norm / (gnorm + 1e-6)).clamp(max=1.0)
                self._scale.copy_((coef * multiply).reshape(1))
            elif torch.is_tensor(multiply):
                self._scale.copy_(multiply.reshape(1))
            else:
                self._scale.fill_(multiply)
        self.num_updates += 1
        PARAM_EPOCH[0] += 1
        for lo, hi, a in self._spans():
            n = hi - lo
            self._adam_span(self.master[a:a + n], self.exp_avg[a:a + n], self.exp_avg_sq[a:a + n], self.buf.flat_grad[lo:hi], self.buf.flat_param[lo:hi])
        if self.segments is not None:
            self._gather(self.buf.flat_param)  # every rank's freshly updated spans -> the full parameter buffer on every rank
        self.lr = inverse_sqrt_lr(self.num_updates, self.base_lr, self.warmup_updates, self.warmup_init_lr)
        return gnorm

    # ---- the reference's on-disk optimizer state (torch.optim state dict of optim/adam.py, or FP16Optimizer's single flat fp32
    #      parameter, optim/fp16_optimizer.py:33-60, :71-76) <-> the flat master / moment buffers ----
    def fairseq_state_dict(self):
        """(Sharded state: a collective — every rank calls it, every rank gets the full dict.)"""
        _, exp_avg, exp_avg_sq = self._full_state()
        state = {}
        for i, (p, o) in enumerate(zip(self.buf.params, self.buf.offsets)):
            n = p.numel()
            state[i] = {"step": self.num_updates, "exp_avg": exp_avg[o:o + n].view(p.shape).cpu().clone(),
                        "exp_avg_sq": exp_avg_sq[o:o + n].view(p.shape).cpu().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "params": list(range(len(self.buf.params)))}
        return {"state": state, "param_groups": [group]}

    def load_fairseq_state_dict(self, sd, num_updates=None):
        """Moments from a reference checkpoint; the fp32 master is re-derived from the (just loaded) model parameters, as
        FP16Optimizer does on construction.  `num_updates` (optimizer_history[-1]) drives the lr schedule."""
        st = {int(k): v for k, v in sd["state"].items()}  # keys = parameter indices; parameters that never had a gradient have none
        entries = [st[k] for k in sorted(st)]
        params, offs = self.buf.params, self.buf.offsets
        total = sum(p.numel() for p in params)
        sharded = self.segments is not None
        ea = torch.zeros(self.buf.total, dtype=torch.float32, device=self.buf.device) if sharded else self.exp_avg.zero_()
        es = torch.zeros(self.buf.total, dtype=torch.float32, device=self.buf.device) if sharded else self.exp_avg_sq.zero_()
        if len(st) == 1 and len(params) > 1 and entries[0]["exp_avg"].numel() == total:  # FP16Optimizer: one flat fp32 parameter
            m, v, at = entries[0]["exp_avg"].reshape(-1).float(), entries[0]["exp_avg_sq"].reshape(-1).float(), 0
            for p, o in zip(params, offs):
                ea[o:o + p.numel()].copy_(m[at:at + p.numel()])
                es[o:o + p.numel()].copy_(v[at:at + p.numel()])
                at += p.numel()
        else:
            for i, e in st.items():
                if not (0 <= i < len(params)) or e["exp_avg"].numel() != params[i].numel():
                    raise ValueError("optimizer state entry %d does not match the model's parameter list (the index space of a "
                                     "fairseq optimizer state is model.parameters() order)" % i)
                n, o = params[i].numel(), offs[i]
                ea[o:o + n].copy_(e["exp_avg"].reshape(-1).float())
                es[o:o + n].copy_(e["exp_avg_sq"].reshape(-1).float())
        step = int(entries[0]["step"]) if entries else 0
        self.num_updates = int(num_updates) if num_updates is not None else step
        if sharded:
            self.exp_avg, self.exp_avg_sq, self.master = self._take(ea), self._take(es), self._take(self.buf.flat_param.float())
        else:
            self.master.copy_(self.buf.flat_param.float())
        self.lr = inverse_sqrt_lr(self.num_updates, self.base_lr, self.warmup_updates, self.warmup_init_lr)

    def state_dict(self):
        """Flat fp32 master / moments of the whole buffer (sharded state: a collective, as fairseq_state_dict)."""
        master, exp_avg, exp_avg_sq = self._full_state()
        return {"master": master, "exp_avg": exp_avg, "exp_avg_sq": exp_avg_sq, "num_updates": self.num_updates, "lr": self.lr}

    def load_state_dict(self, sd):
        if self.segments is not None:
            self.master, self.exp_avg, self.exp_avg_sq = (self._take(sd[k].to(self.buf.device)) for k in ("master", "exp_avg", "exp_avg_sq"))
        else:
            self.master.copy_(sd["master"]); self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.num_updates, self.lr = sd["num_updates"], sd["lr"]
        self.buf.flat_param.copy_(sd["master"])
        PARAM_EPOCH[0] += 1  # the parameters changed under their views
