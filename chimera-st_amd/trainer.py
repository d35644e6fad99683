"""The trainer seam of the hot path: fairseq/trainer.py train_step (:455-700) — micro-batch loop with no_sync
(:479-492), task.train_step, summed logging stats (:1005-1043), model.all_reduce() (:588-589), multiply_grads
(world/sample_size, :601-606), clip (:615), optimizer step (:627), lr schedule — with the MI355X pieces swapped in:
flat bf16 parameter/gradient buffers, bucketed RCCL all-reduce overlapped with backward, one fused Adam pass."""
import contextlib
import math
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from .distributed import DistributedFairseqModel, all_reduce_stats
from . import rng
from .profiling import scope
from .hostcfg import limit_host_threads
from .optim import ALIGN, FlatParamBuffers, FusedAdam, qkv_groups


class _StateView:
    """The optimizer as checkpoint_utils.save_state sees it, with a state dict that was assembled beforehand (sharded state)."""

    def __init__(self, opt, full):
        self._opt, self._full = opt, full

    def fairseq_state_dict(self):
        return self._full

    def __getattr__(self, name):
        return getattr(self._opt, name)


class Trainer:
    def __init__(self, args, task, model, criterion, device="cuda"):
        self.args, self.task, self.criterion = args, task, criterion
        self.device = torch.device(device)
        limit_host_threads(int(os.environ.get("LOCAL_WORLD_SIZE", dist.get_world_size() if dist.is_initialized() else 1)))  # (hostcfg.py)
        dtype = torch.bfloat16 if getattr(args, "bf16", False) else torch.float32
        if getattr(args, "fp16", False):
            raise NotImplementedError("--fp16: this build computes in bf16 (--bf16) or fp32; gfx950 MFMA rates are equal and "
                                      "bf16 needs no loss scaling (DESIGN.md)")
        model = model.to(device=self.device, dtype=dtype)  # trainer.py:70-78
        self.criterion = criterion.to(self.device)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        # CST_DDP_FORCE=1 keeps the collective path on for a 1-rank group (a 1-GPU box can only form that RCCL communicator)
        self.ddp = self.world > 1 or (dist.is_initialized() and os.environ.get("CST_DDP_FORCE") == "1")
        # --zero-sharding os (fairseq/trainer.py:241-252, optim/shard.py: the reference wraps its optimizer in fairscale's OSS):
        # optimizer state sharded over the data-parallel ranks behind reduce-scattered gradient buckets (optim.FusedAdam.shard)
        self.zero = self.ddp and (getattr(args, "zero_sharding", "none") == "os" or os.environ.get("CST_ZERO1") == "1")
        self.buffers = FlatParamBuffers(model.parameters(), adjacent=qkv_groups(model), align=ALIGN * self.world if self.zero else ALIGN)
        self.optimizer = FusedAdam.from_args(args, None, buffers=self.buffers)
        self.model = DistributedFairseqModel(args, model, self.buffers) if self.ddp else model
        self._model = model
        if self.zero:
            self.optimizer.shard(self.model.reducer.shard_state(), self.model.reducer.all_gather_shards)
        # Deferred reductions (kernels.DEFER): allowed when every parameter receives exactly one gradient per backward pass — the
        # criterion runs ONE pass over the model (label_smoothed_cross_entropy; the triplet criterion runs two) and the model
        # declares that a pass uses each parameter once.  Parameters reachable under two names (tied embeddings) are marked and
        # excluded at the call sites: their two gradients are added by autograd as soon as the second one exists.
        seen = {}
        for _, p in model.named_parameters(remove_duplicate=False):
            seen[id(p)] = seen.get(id(p), 0) + 1
        for p in model.parameters():
            if seen.get(id(p), 1) > 1:
                p._cst_shared = True  # (modules mark parameters they apply twice in one pass themselves: the memory layers' LayerNorm)
        self._defer_cache = None  # (criterion id, decision, the parameters THIS trainer marked for it)

        def _defer_ok():
            # decided once per (criterion, environment switches the criterion's twice_used reads): the walk over named_parameters is
            # not an every-update cost, and marks set for one criterion are taken back before another one is evaluated
            key = (id(self.criterion),) + tuple(os.environ.get(k) for k in ("CST_NO_PACK", "CST_NO_PAIR_DECODER", "CST_NO_PAIR_ENCODER", "CST_NO_PAIR_MEMORY"))
            if self._defer_cache is not None and self._defer_cache[0] == key:
                return self._defer_cache[1]
            if self._defer_cache is not None:
                for p in self._defer_cache[2]:
                    p._cst_shared = False
            marked, ok = [], False
            if getattr(self._model, "single_use_parameters", False):
                if getattr(self.criterion, "single_pass", False):
                    ok = True
                else:
                    # two passes that share the decoder call (TripletSTMTContrastiveCriterion.one_decoder_pass): the parameters both
                    # passes run through are named by the criterion and marked; all the others receive one gradient per backward pass
                    twice = getattr(self.criterion, "twice_used", lambda m: None)(self._model)
                    if twice is not None:
                        ok = True
                        for p in twice:
                            if not getattr(p, "_cst_shared", False):
                                p._cst_shared = True
                                marked.append(p)
            self._defer_cache = (key, ok, marked)
            return ok

        self._defer_ok = _defer_ok
        self.optimizer.defer_reductions = self._defer_ok()
        self.num_updates = 0
        self.dtype = dtype
        self._dummy_batch = None
        # layout of the all-reduced statistics vector: fixed by the criterion, not by the first batch — a rank that runs out of
        # memory before it ever completed an update must still be able to build the vector the other ranks are reducing
        keys = getattr(self.criterion, "logging_keys", None)
        self._log_keys = sorted(keys()) if keys is not None else None
        assert self._log_keys is not None or not self.ddp, "data-parallel training needs criterion.logging_keys()"
        # non-finite gradients: skip the update on every rank; give up after this many in a row (the reference's DynamicLossScaler
        # halves the scale from 2^7 down to min_loss_scale = 1e-4 before raising FloatingPointError: 20 consecutive overflows,
        # optim/dynamic_loss_scaler.py:42-70)
        self.nonfinite_tolerance = int(getattr(args, "nonfinite_tolerance", 20))
        self._nonfinite_run = 0

    def get_model(self):
        return self._model

    def _set_seed(self):
        """trainer.py:934-938 — seed + num_updates so dropout/layerdrop RNG is resumable."""
        seed = getattr(self.args, "seed", 1) + self.num_updates
        torch.manual_seed(seed)
        if self.device.type == "cuda":
            torch.cuda.manual_seed(seed)
        np.random.seed(seed)
        random.seed(seed)
        rng.reseed(seed)  # dropout sites of the HIP path: keys = f(seed, site ordinal)

    def _prepare_sample(self, sample):
        """trainer.py:896-932: H2D; the waveform stays fp32 (conv0 reads it directly), token tensors stay int64."""
        def mv(x):
            if torch.is_tensor(x):
                return x.to(self.device, non_blocking=True)
            if isinstance(x, dict):
                return {k: mv(v) for k, v in x.items()}
            return x

        return mv(sample)

    def train_step(self, samples, raise_oom=False):
        """`_train_step` inside the "train_step-N" range of the reference (fairseq_cli/train.py:225-227; N = updates done so far).
        Phase ranges are roctx markers, off unless CST_ROCTX=1 (profiling.py)."""
        with scope("train_step-%d" % self.num_updates):
            return self._train_step(samples, raise_oom)

    def _train_step(self, samples, raise_oom=False):
        """One update over a list of micro-batches.  Returns the summed logging output (device scalars converted to floats in the
        step's single host read), or None when the update was skipped because a rank ran out of memory (trainer.py:524-544, 564-570).

        The cross-rank guards of the reference ride in the ONE fp64 vector that already carries the logging scalars (C3-C5):
          * `distributed_cuda_oom` (trainer.py:564-570): a rank that ran out of memory in forward/backward still issues every
            bucket all-reduce (finish()) and the stat all-reduce, with the flag set; every rank then drops the update;
          * `_check_grad_norms` (trainer.py:1045-1077): each rank writes the sum of squares of ITS copy of the reduced gradient
            into its own slot; after the all-reduce every rank sees all of them and raises if they disagree beyond 1e-6."""
        self._set_seed()
        self.optimizer.zero_grad()
        self.optimizer.defer_reductions = self._defer_ok()  # (read per update: the criterion is an attribute a driver may replace)
        logs, sample_size, ooms = [], 0, 0
        for i, sample in enumerate(samples):
            # an empty batch (this rank's shard ran out: ShardedIterator pads with []) runs the dummy batch with its loss zeroed
            # so that every rank issues the same collectives (trainer.py:469-477, 552-556)
            ignore = sample is None or len(sample) == 0
            if ignore:
                assert self._dummy_batch is not None, "an empty batch before any real batch was seen"
                sample = self._dummy_batch
            elif self._dummy_batch is None:
                self._dummy_batch = sample
            last = i == len(samples) - 1
            ctx = self.model.no_sync() if (self.ddp and not last) else contextlib.nullcontext()
            try:
                sample = self._prepare_sample(sample)
                with ctx:
                    loss, ss, log = self.task.train_step(sample, self.model, self.criterion, self.optimizer, self.num_updates,
                                                         ignore_grad=ignore)
                del loss
            except RuntimeError as e:
                if "out of memory" not in str(e) or raise_oom:
                    raise
                ooms += 1
                self.optimizer.zero_grad()
                torch.cuda.empty_cache()
                if not self.ddp:
                    return None
                continue
            if ignore:
                log, ss = {k: v * 0 for k, v in log.items()}, 0
            logs.append(log)
            sample_size += ss
        with scope("reduce-grads"):
            if self.ddp:
                self.model.all_reduce()  # waits for the overlapped bucket reductions; launches the stragglers
            else:
                self.buffers.gather_grads()  # one multi-tensor copy of every autograd-owned gradient into the flat buffer
        # logging scalars + sample_size + OOM flag + late-gradient count + one gradient-norm slot per rank: ONE fp64 device vector,
        # one small all-reduce, one host read
        if self._log_keys is None:
            assert logs, "out of memory before any update completed and the criterion does not declare logging_keys()"
            self._log_keys = sorted(logs[0].keys())
        keys = self._log_keys
        late = list(self.model.reducer.late_params) if self.ddp else []
        # layout: [one slot per logging key | OOM flag | late-gradient count | one gradient-norm slot per rank].  Host-side numbers
        # (token / sentence counts, the flags) travel in ONE pinned buffer, copied asynchronously — a Python scalar turned into a
        # device tensor is a pageable copy, i.e. a stream synchronisation each (five per update until round 4: the gradient-norm
        # kernel and this vector's assembly were then enqueued on an idle GPU); device-side numbers (losses, the gradient's sum of
        # squares) are scattered into their slots by one kernel.
        nk = len(keys)
        n = nk + 2 + self.world
        host = getattr(self, "_stat_host", None)
        if host is None or host.numel() != n:
            host = torch.zeros(n, dtype=torch.float64)
            self._stat_host = host = host.pin_memory() if self.device.type == "cuda" else host
        host.zero_()
        dev_idx, dev_val = [], []
        for i, k in enumerate(keys):
            parts = [l[k] for l in logs]
            if any(torch.is_tensor(v) for v in parts):
                dev_idx.append(i)
                dev_val.append(sum((v.to(torch.float64) if torch.is_tensor(v) else v for v in parts)))
            else:
                host[i] = float(sum(parts))
        host[nk], host[nk + 1] = float(ooms), float(len(late))

        def assemble():
            vec = host.to(self.device, non_blocking=True) if self.device.type == "cuda" else host.clone()
            idx = dev_idx + [nk + 2 + self.rank]
            val = dev_val + [self.optimizer.grad_sumsq()[0].double()]
            cache = getattr(self, "_stat_idx", None)
            if cache is None or cache[0] != idx:
                cache = self._stat_idx = (idx, torch.tensor(idx, dtype=torch.int64).to(self.device))
            vec[cache[1]] = torch.stack([v.reshape(()).to(self.device) for v in val])
            return vec

        def norm_slots():
            slots = torch.zeros(self.world, dtype=torch.float64, device=self.device)
            slots[self.rank] = self.optimizer.grad_sumsq()[0].double()
            return slots

        # "clip-grads" (trainer.py:613-614): the gradient's sum of squares is the device half of the clip; its host half (the
        # coefficient) and "multiply-grads" (trainer.py:601-606) are one scalar folded into the Adam kernel (optim.step)
        with scope("clip-grads"):
            vec = assemble()
            if self.ddp:
                dist.all_reduce(vec)
            vals = vec.tolist()  # the step's only host sync
        out = dict(zip(keys, vals[:len(keys)]))
        if vals[len(keys)] != 0:  # some rank lost this update's gradients: nobody steps (trainer.py:564-570)
            self.optimizer.zero_grad()
            return None
        sumsq = vals[len(keys) + 2:]
        self.last_late_count = int(vals[len(keys) + 1])  # diagnostics: late gradients seen by all ranks in this update
        if vals[len(keys) + 1] != 0:
            # some rank saw a gradient arrive after its bucket had been reduced (a parameter reported unused that took part after
            # all).  Rare; every rank takes this branch together: all-gather the 0/1 masks (WHO was late matters: a rank where the
            # gradient was on time already holds a partial mean, see late_reduce), reduce those parameters again and exchange the
            # norms of the corrected gradients.
            mask = torch.zeros(len(self.buffers.params), dtype=torch.int32, device=self.device)
            if late:
                mask[torch.tensor(late, device=self.device)] = 1
            masks = [torch.zeros_like(mask) for _ in range(self.world)]
            dist.all_gather(masks, mask)
            self.model.reducer.late_reduce([{i for i, m in enumerate(mk.tolist()) if m} for mk in masks])
            slots = norm_slots()
            dist.all_reduce(slots)
            sumsq = slots.tolist()
        if self.zero:
            # sharded optimizer: a rank's slot is the sum over ITS spans of the reduced gradient — the slots are the parts of one sum,
            # not copies of one number, so there is nothing for _check_grad_norms to compare; every rank adds them in slot order
            sumsq = [math.fsum(sumsq)] * self.world if all(math.isfinite(v) for v in sumsq) else [float("nan")] * self.world
        else:
            self._check_grad_norms(sumsq)  # raises when the replicas disagree (a mix of finite and non-finite norms included)
        if not all(math.isfinite(v) for v in sumsq):
            # NaN / Inf in the reduced gradient (every rank holds the same vector, so every rank is here): drop the update, as the
            # reference does on an fp16 overflow (trainer.py:629-646, optim/fp16_optimizer.py:182); master weights and both Adam
            # moments stay untouched.  A persistent condition is an error (optim/dynamic_loss_scaler.py:58-66).
            self.optimizer.zero_grad()
            self._nonfinite_run += 1
            if self._nonfinite_run > self.nonfinite_tolerance:
                raise FloatingPointError("gradients are NaN/Inf in %d consecutive updates (update %d): giving up"
                                         % (self._nonfinite_run, self.num_updates))
            out.update(overflow=1.0, gnorm=float("nan"), lr=self.optimizer.get_lr())
            return out
        self._nonfinite_run = 0
        multiply = self.world / out["sample_size"] if out["sample_size"] > 0 else 0.0
        with scope("optimizer"):
            out["gnorm"] = self.optimizer.step(multiply=multiply, gnorm=math.sqrt(sumsq[self.rank]) * multiply)
        self.num_updates += 1
        out["lr"] = self.optimizer.get_lr()
        return out

    def _check_grad_norms(self, sumsq):
        """trainer.py:1045-1077 — replicas all-reduce the same buckets in the same order, so their gradient norms must agree to
        rounding; anything else (a rank that skipped a bucket, mixed hardware, a corrupted buffer) is fatal."""
        norms = [math.sqrt(v) if v >= 0 and math.isfinite(v) else float("nan") for v in sumsq]
        if len(norms) < 2 or not any(math.isfinite(n) for n in norms):
            return  # the reference's predicate: all-non-finite is the overflow path's business, a mix of finite and not is fatal
        if all(math.isfinite(n) for n in norms) and max(abs(n - norms[0]) for n in norms) / (norms[0] + 1e-6) < 1e-6:
            return
        detail = "\n".join("rank {:3d} = {:.8f}".format(r, n) for r, n in enumerate(norms))
        raise RuntimeError("Fatal error: gradients are inconsistent between workers.\n" + "-" * 80 +
                           "\ngrad_norm across the workers:\n{}\n".format(detail) + "-" * 80)

    @torch.no_grad()
    def valid_step(self, sample):
        """trainer.py:702-760: criterion outputs of one validation batch (device scalars)."""
        if sample is None or len(sample) == 0:
            return None
        loss, ss, log = self.task.valid_step(self._prepare_sample(sample), self._model, self.criterion)
        return log

    # ---- checkpoints in the reference's format (fairseq/trainer.py:270-395; checkpoint_utils.py) -------------------------
    def save_checkpoint(self, filename, extra_state=None):
        from . import checkpoint_utils
        if self.zero:  # the moments live in every rank's shards: assembling them is a collective, so every rank builds the state
            full = self.optimizer.fairseq_state_dict()
            if self.rank != 0:
                return None
            opt = _StateView(self.optimizer, full)
        else:
            opt = self.optimizer
        if self.rank != 0:
            return None
        extra = {"train_iterator": {"epoch": 1, "iterations_in_epoch": 0}, "val_loss": None}
        extra.update(extra_state or {})
        return checkpoint_utils.save_state(filename, self.args, self._model.state_dict(), self.criterion, opt,
                                           self.num_updates, extra_state=extra)

    def load_checkpoint(self, filename, reset_optimizer=False):
        """Model + optimizer moments + update counter (lr schedule, per-update seeds) from a checkpoint written by this build or
        by the reference.  Returns extra_state (train_iterator position, val_loss) or None if the file does not exist."""
        from . import checkpoint_utils
        import os
        if not os.path.exists(filename):
            return None
        state = checkpoint_utils.load_checkpoint_to_cpu(filename)
        sd = state["model"]
        self._model.upgrade_state_dict(sd)
        self._model.load_state_dict(sd, strict=True)  # copies into the flat parameter buffer (parameters are views of it)
        last = state["optimizer_history"][-1]
        if not reset_optimizer and state.get("last_optimizer_state") is not None:
            self.optimizer.load_fairseq_state_dict(state["last_optimizer_state"], num_updates=last["num_updates"])
            self.num_updates = int(last["num_updates"])
        else:
            self.optimizer.load_fairseq_state_dict({"state": {}}, num_updates=0)
            self.num_updates = 0
        return state["extra_state"]
