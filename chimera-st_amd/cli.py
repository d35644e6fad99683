"""Command-line drivers — the part of fairseq_cli/train.py (:44-330) and fairseq_cli/generate.py (:57-390) that sits
around the hot path, so that the flag sets of chimera/scripts/train-en2any-ST.sh and chimera/generate/generate-mustc-final.sh
drive this build unchanged:

  python fairseq_train.py <data> --task triplet --arch s2t_transformer_w2v2_interlingua_base --config-yaml config_wave.yaml ...
  python fairseq_generate.py <data> --task triplet --path ckpt.pt --gen-subset tst-COMMON_wave --beam 10 --lenpen 1.5 ...

`--fp16` selects the reduced-precision path of this hardware: bf16 storage with fp32 accumulation (no loss scaling).  One
process per GPU (torchrun-style env: RANK / WORLD_SIZE / LOCAL_RANK); each rank reads its own shard of the epoch's batches."""
import argparse
import json
import math
import os
import sys
import time

import torch

from . import checkpoint_utils, criterions, registry, s2t_transformer, tasks, w2v2_transformer, w2v2_transformer_interlingua  # noqa: F401
from .distributed import distributed_init, launch_ranks, needs_self_launch
from .hostcfg import limit_host_threads
from .trainer import Trainer


def _log(rank, **kw):
    if rank == 0:
        print(json.dumps(kw), flush=True)


def _extra_train_flags(argv):
    """Flags of the reference scripts that configure subsystems outside this build (or have a fixed meaning here)."""
    p = argparse.ArgumentParser(add_help=False, allow_abbrev=False)
    p.add_argument("--best-checkpoint-metric", default="loss")
    p.add_argument("--maximize-best-checkpoint-metric", action="store_true")
    p.add_argument("--validate-interval", type=int, default=1)
    p.add_argument("--no-epoch-checkpoints", action="store_true")
    p.add_argument("--no-save", action="store_true")
    p.add_argument("--restore-file", default="checkpoint_last.pt")
    p.add_argument("--disable-validation", action="store_true")
    p.add_argument("--data-buffer-size", type=int, default=8)
    p.add_argument("--label-smoothing-placeholder", default=None, help=argparse.SUPPRESS)
    return p.parse_known_args(argv)


def validate(trainer, task, args, subset, rank, world):
    ds = task.load_dataset(subset) if subset not in task.datasets else task.dataset(subset)
    itr = task.get_batch_iterator(ds, max_tokens=args.max_tokens, max_sentences=args.batch_size, max_positions=task.max_positions(),
                                  ignore_invalid_inputs=True, seed=args.seed, num_shards=world, shard_id=rank)
    totals = {}
    for sample in itr.next_epoch_itr(shuffle=False):
        log = trainer.valid_step(sample)
        for k, v in (log or {}).items():
            totals[k] = totals.get(k, 0.0) + torch.as_tensor(v, dtype=torch.float64, device=trainer.device)
    # every rank takes part in the SAME collectives with the SAME vector layout, whatever its shard held: a rank whose shard had
    # no batch (fewer validation batches than ranks) contributes zeros instead of returning before the all-reduce
    keys = sorted(totals)
    if world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, keys)
        keys = sorted(set(k for ks in gathered for k in ks))
    if not keys:
        return {}
    zero = torch.zeros((), dtype=torch.float64, device=trainer.device)
    vec = torch.stack([totals.get(k, zero) for k in keys])
    if world > 1:
        torch.distributed.all_reduce(vec)
    out = dict(zip(keys, vec.tolist()))
    ss = max(out.get("sample_size", 1.0), 1.0)
    return {k: (v / ss / math.log(2) if k.endswith("loss") else v) for k, v in out.items()}  # per-token, base 2 (criterions reduce_metrics)


def train_main(argv=None):
    # Self-launch (below) is the SCRIPT entry's behaviour (fairseq_train.py's __main__ calls train_main() with no argument list); a
    # programmatic caller that passes argv wants the Trainer back from THIS process and gets ranks only by asking for them
    # (--distributed-world-size N > 1).
    from_script = argv is None
    argv = list(sys.argv[1:] if argv is None else argv)
    extra, rest = _extra_train_flags(argv)
    args = registry.parse_args_and_arch(rest)
    for k, v in vars(extra).items():
        setattr(args, k, v)
    if args.fp16 or getattr(args, "memory_efficient_fp16", False):
        args.fp16, args.memory_efficient_fp16, args.bf16 = False, False, True
    # Started plainly on a multi-GPU node, fairseq-train spawns one process per GPU itself (fairseq/distributed_utils.py:286-303,
    # nprocs = min(device_count, distributed_world_size); the flag's default is the device count, options.py:310).  Same here: this
    # process has not touched the GPU yet (device_count() does not initialise it) and becomes the launcher of the rank processes.
    if "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        ndev = torch.cuda.device_count()
        want = args.distributed_world_size if args.distributed_world_size is not None else (max(ndev, 1) if from_script else 1)
        nproc = want if os.environ.get("CST_DIST_BACKEND") else min(max(ndev, 1), want)  # (gloo: ranks may share a GPU — tests)
        if needs_self_launch(nproc):
            entry = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fairseq_train.py")
            code = launch_ranks(nproc, [sys.executable, entry] + argv)
            if code != 0:
                raise SystemExit(code)
            return None
    rank, world = distributed_init()
    limit_host_threads(int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    device = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(args.seed)
    task = registry.setup_task(args)
    train_ds = task.load_dataset(args.train_subset)
    model = task.build_model(args)
    criterion = task.build_criterion(args)
    trainer = Trainer(args, task, model, criterion, device=device)
    _log(rank, event="start", arch=args.arch, task=args.task, criterion=args.criterion, world_size=world,
         params=sum(p.numel() for p in model.parameters()), train_examples=len(train_ds), dtype=str(trainer.dtype))
    os.makedirs(args.save_dir, exist_ok=True)
    start_epoch, skip = 1, 0
    restore = args.restore_file if os.path.isabs(args.restore_file) else os.path.join(args.save_dir, args.restore_file)
    extra_state = trainer.load_checkpoint(restore, reset_optimizer=args.reset_optimizer)
    if extra_state is not None:
        if not args.reset_dataloader and extra_state.get("train_iterator"):
            start_epoch, skip = extra_state["train_iterator"]["epoch"], extra_state["train_iterator"]["iterations_in_epoch"]
        _log(rank, event="loaded_checkpoint", path=restore, num_updates=trainer.num_updates, epoch=start_epoch, iterations_in_epoch=skip)
    itr = task.get_batch_iterator(train_ds, max_tokens=args.max_tokens, max_sentences=args.batch_size, max_positions=task.max_positions(),
                                  ignore_invalid_inputs=True, required_batch_size_multiple=8 if args.batch_size is None else 1,
                                  seed=args.seed, num_shards=world, shard_id=rank, epoch=start_epoch)
    itr.pin_memory = True
    best = extra_state.get("best") if extra_state is not None else None  # checkpoint_utils.py:47-55: save_checkpoint.best survives a restart
    max_update = args.max_update or math.inf
    max_epoch = args.max_epoch or math.inf
    epoch = start_epoch

    def save(tag_files, epoch_, it_in_epoch, val):
        # `args.no_save` is the same on every rank, so this return is rank-agreed.  With --zero-sharding os the optimizer's moments
        # live in every rank's shards and Trainer.save_checkpoint assembles them with an all-gather: EVERY rank must enter it (only
        # rank 0 writes, trainer.py save_checkpoint); the file copies stay rank 0's.
        if args.no_save or (rank != 0 and not trainer.zero):
            return
        extra_ = {"train_iterator": {"epoch": epoch_, "iterations_in_epoch": it_in_epoch}, "val_loss": val, "best": best}
        first = os.path.join(args.save_dir, tag_files[0])
        trainer.save_checkpoint(first, extra_)
        if rank != 0:
            return
        for f in tag_files[1:]:
            import shutil
            shutil.copyfile(first, os.path.join(args.save_dir, f))

    while epoch <= max_epoch and trainer.num_updates < max_update:
        uf = args.update_freq[min(epoch - 1, len(args.update_freq) - 1)]
        t0, group, n_it, agg = time.time(), [], 0, {}
        for i, sample in enumerate(itr.next_epoch_itr(shuffle=True, buffer_size=max(getattr(args, "data_buffer_size", 8), 0))):
            if epoch == start_epoch and i < skip:
                continue
            group.append(sample)
            n_it = i + 1
            if len(group) < uf:
                continue
            out = trainer.train_step(group)
            group = []
            if out is None:  # a rank ran out of memory: every rank dropped this update (trainer.py:564-570)
                _log(rank, event="oom_skipped_update", epoch=epoch, num_updates=trainer.num_updates)
                continue
            if out.get("overflow"):  # NaN / Inf gradients: every rank dropped this update (trainer.py:629-646)
                _log(rank, event="nonfinite_gradient_skipped_update", epoch=epoch, num_updates=trainer.num_updates)
                continue
            for k, v in out.items():
                agg[k] = agg.get(k, 0.0) + (v if k not in ("lr", "gnorm") else 0.0)
            if trainer.num_updates % max(args.log_interval, 1) == 0:
                ss = max(out.get("sample_size", 1.0), 1.0)
                _log(rank, event="train_inner", epoch=epoch, num_updates=trainer.num_updates, loss=out["loss"] / ss / math.log(2),
                     gnorm=out["gnorm"], lr=out["lr"], ntokens=out.get("ntokens"))
            if args.save_interval_updates > 0 and trainer.num_updates % args.save_interval_updates == 0:
                save(["checkpoint_%d_%d.pt" % (epoch, trainer.num_updates), "checkpoint_last.pt"], epoch, n_it, None)
            if trainer.num_updates >= max_update:
                break
        if group and trainer.num_updates < max_update:  # the epoch's tail group is a (smaller) update of its own (iterators.py GroupedIterator)
            out = trainer.train_step(group)
            if out is not None and out.get("overflow"):
                out = None
            for k, v in (out or {}).items():
                agg[k] = agg.get(k, 0.0) + (v if k not in ("lr", "gnorm") else 0.0)
        ss = max(agg.get("sample_size", 1.0), 1.0)
        _log(rank, event="train", epoch=epoch, num_updates=trainer.num_updates, loss=agg.get("loss", 0.0) / ss / math.log(2),
             wall=time.time() - t0, nsentences=agg.get("nsentences"))
        val = None
        if not args.disable_validation and epoch % args.validate_interval == 0:
            stats = validate(trainer, task, args, args.valid_subset.split(",")[0], rank, world)
            val = stats.get(args.best_checkpoint_metric, stats.get("loss"))
            _log(rank, event="valid", epoch=epoch, num_updates=trainer.num_updates, **{k: v for k, v in stats.items()})
        files = [] if args.no_epoch_checkpoints else ["checkpoint%d.pt" % epoch]
        better = val is not None and (best is None or (val > best if args.maximize_best_checkpoint_metric else val < best))
        if better:
            best = val
            files.append("checkpoint_best.pt")
        files.append("checkpoint_last.pt")
        save(files, epoch + 1, 0, val)
        epoch += 1
        skip = 0
    _log(rank, event="done", num_updates=trainer.num_updates, best=best)
    if world > 1:
        torch.distributed.destroy_process_group()
    return trainer


# --------------------------------------------------------------------------------------------------------------------
def corpus_bleu(hyps, refs, max_n=4):
    """Corpus BLEU-4 with brevity penalty over whitespace tokens (the reference scores with sacrebleu's 13a tokeniser,
    fairseq_cli/generate.py:352-376 — sacrebleu is not in this image; numbers are comparable between runs of THIS tool only)."""
    from collections import Counter
    match, total, hl, rl = [0] * max_n, [0] * max_n, 0, 0
    for h, r in zip(hyps, refs):
        h, r = h.split(), r.split()
        hl, rl = hl + len(h), rl + len(r)
        for n in range(1, max_n + 1):
            hc = Counter(tuple(h[i:i + n]) for i in range(len(h) - n + 1))
            rc = Counter(tuple(r[i:i + n]) for i in range(len(r) - n + 1))
            match[n - 1] += sum(min(c, rc[g]) for g, c in hc.items())
            total[n - 1] += max(len(h) - n + 1, 0)
    if min(total) == 0 or min(match) == 0:
        return 0.0
    logp = sum(math.log(m / t) for m, t in zip(match, total)) / max_n
    bp = 1.0 if hl > rl else math.exp(1 - rl / max(hl, 1))
    return 100.0 * bp * math.exp(logp)


def generate_main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    p = argparse.ArgumentParser(allow_abbrev=False)
    p.add_argument("data")
    p.add_argument("--path", required=True, help="checkpoint(s), colon separated (the first is decoded: single-model search)")
    p.add_argument("--task", default="triplet")
    p.add_argument("--config-yaml", default="config.yaml")
    p.add_argument("--gen-subset", default="test")
    p.add_argument("--max-tokens", type=int, default=None)
    p.add_argument("--max-sentences", "--batch-size", type=int, default=None, dest="batch_size")
    p.add_argument("--max-source-positions", type=int, default=2000000)
    p.add_argument("--max-target-positions", type=int, default=1024)
    p.add_argument("--beam", type=int, default=5)
    p.add_argument("--max-len-a", type=float, default=0)
    p.add_argument("--max-len-b", type=int, default=200)
    p.add_argument("--min-len", type=int, default=1)
    p.add_argument("--lenpen", type=float, default=1.0)
    p.add_argument("--unkpen", type=float, default=0.0)
    p.add_argument("--temperature", type=float, default=1.0)
    p.add_argument("--unnormalized", action="store_true")
    p.add_argument("--remove-bpe", "--post-process", nargs="?", const="@@ ", default=None, dest="post_process")
    p.add_argument("--scoring", default="bleu")
    p.add_argument("--results-path", default=None)
    p.add_argument("--fp16", action="store_true")
    p.add_argument("--bf16", action="store_true")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--quiet", action="store_true")
    args, ignored = p.parse_known_args(argv)
    limit_host_threads()
    overrides = {"data": args.data, "config_yaml": args.config_yaml, "max_source_positions": args.max_source_positions,
                 "max_target_positions": args.max_target_positions}
    models, margs, task = checkpoint_utils.load_model_ensemble_and_task(args.path.split(":")[:1], arg_overrides=overrides)
    dtype = torch.bfloat16 if (args.fp16 or args.bf16) else torch.float32
    model = models[0].to("cuda", dtype).eval()
    ds = task.load_dataset(args.gen_subset)
    itr = task.get_batch_iterator(ds, max_tokens=args.max_tokens, max_sentences=args.batch_size,
                                  max_positions=(args.max_source_positions, args.max_target_positions), ignore_invalid_inputs=True)
    gen = task.build_generator([model], args)
    tgt_dict = task.target_dictionary
    if args.results_path:
        os.makedirs(args.results_path, exist_ok=True)
    out = open(os.path.join(args.results_path, "generate-%s.txt" % args.gen_subset), "w") if args.results_path else sys.stdout

    def detok(s):
        if args.post_process == "sentencepiece":
            return s.replace(" ", "").replace("▁", " ").strip()
        if args.post_process:
            return (s + " ").replace(args.post_process, "").rstrip()
        return s

    hyps, refs, nsent, ntok, t0 = [], [], 0, 0, time.time()
    for sample in itr.next_epoch_itr(shuffle=False):
        if not sample:
            continue
        src = sample["net_input"]["src_tokens"]
        # raw waveforms [B, S] stay fp32 as in training (Trainer._prepare_sample: conv0 reads fp32 samples); only feature
        # inputs [B, T, F] take the model's storage dtype
        src = src.to("cuda", dtype) if (src.dim() == 3 and src.is_floating_point()) else src.cuda()
        s = {"net_input": {"src_tokens": src, "src_lengths": sample["net_input"]["src_lengths"].cuda()}}
        results = task.inference_step(gen, [model], s)
        for i, sid in enumerate(sample["id"].tolist()):
            ref = tgt_dict.string(sample["target"][i]) if sample.get("target") is not None else None
            h = results[i][0]
            hyp = tgt_dict.string(h["tokens"].cpu())
            if not args.quiet:
                if ref is not None:
                    print("T-%d\t%s" % (sid, ref), file=out)
                print("H-%d\t%.6f\t%s" % (sid, float(h["score"]) / math.log(2), hyp), file=out)
                print("D-%d\t%.6f\t%s" % (sid, float(h["score"]) / math.log(2), detok(hyp)), file=out)
                print("P-%d\t%s" % (sid, " ".join("%.4f" % (x / math.log(2)) for x in h["positional_scores"].tolist())), file=out)
            hyps.append(detok(hyp))
            refs.append(detok(ref) if ref is not None else "")
            ntok += len(h["tokens"])
        nsent += len(results)
    torch.cuda.synchronize()
    dt = time.time() - t0
    summary = {"event": "generate", "subset": args.gen_subset, "sentences": nsent, "tokens": ntok, "seconds": dt,
               "sentences_per_s": nsent / max(dt, 1e-9), "tokens_per_s": ntok / max(dt, 1e-9), "beam": args.beam,
               "bleu4_whitespace": corpus_bleu(hyps, refs) if any(refs) else None, "ignored_flags": ignored}
    print(json.dumps(summary), file=out, flush=True)
    if out is not sys.stdout:
        out.close()
        print(json.dumps(summary))
    return summary
