#!/usr/bin/env python3
"""bench.py — train utterances/sec of the Chimera-ST hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...; started
   plainly, `python bench.py --gpus N` launches its N rank processes itself)

Workload (config.workload): BASELINE configs[1] — `s2t_transformer_w2v2` with the s2t_transformer_m dimensions
(d 512, ffn 2048, 8 heads, 12 encoder + 6 decoder layers, tied 10 000-way vocabulary) behind the full wav2vec2-small
front end (7-layer CNN over raw 16 kHz samples + 12 x 768 Transformer), label-smoothed CE, Adam — one full update
(fwd + bwd + gradient all-reduce + optimizer) per step on a synthetic batch of 32 utterances x <= 30 s per GPU
(weak scaling), inputs resident in HBM when the clock starts.  Random-init weights, seeded synthetic data.
Prints ONE JSON line on rank 0."""
import argparse
import importlib
import json
import os
import sys
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {"bf16": 2500.0, "f32": 157.3}  # dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md


def build(args, device):
    importlib.import_module("chimera-st_amd")
    w2t = importlib.import_module("chimera-st_amd.w2v2_transformer")
    importlib.import_module("chimera-st_amd.w2v2_transformer_interlingua")
    importlib.import_module("chimera-st_amd.criterions")
    tasks = importlib.import_module("chimera-st_amd.tasks")
    reg = importlib.import_module("chimera-st_amd.registry")
    w2v = importlib.import_module("chimera-st_amd.wav2vec2")
    Trainer = importlib.import_module("chimera-st_amd.trainer").Trainer
    # wav2vec_small hyper-parameters.  --dropout 0.1 (default) is the training recipe: dropout = attention_dropout
    # = 0.1, dropout_input 0.1 in wav2vec2 (activation_dropout 0.0 there), dropout = attention = activation 0.1 in the
    # s2t transformer (w2v2_transformer.py:458-460); --dropout 0 switches every site off (the parity configuration).
    # layerdrop (wav2vec_small: 0.05) is off by default so that every step executes the same, full amount of work.
    dp = float(getattr(args, "dropout", 0.1))
    w2t.SYNTHETIC_W2V["wav2vec_small_bench"] = w2v.wav2vec_small_args(
        dropout=dp, attention_dropout=dp, activation_dropout=0.0, encoder_layerdrop=float(getattr(args, "layerdrop", 0.0)),
        dropout_input=dp, dropout_features=dp)
    chimera = args.model == "chimera"
    ns = Namespace(
        arch="s2t_transformer_w2v2_interlingua_base" if chimera else "s2t_transformer_w2v2",
        task="triplet", criterion="triplet_st_mt_contrastive" if chimera else "label_smoothed_cross_entropy",
        w2v2_model_path="synthetic:wav2vec_small_bench", data=None, synthetic_vocab_size=10000,
        dropout=dp, attention_dropout=dp, activation_dropout=dp, share_decoder_input_output_embed=True,
        max_source_positions=2000000, max_target_positions=1024, label_smoothing=0.1,
        bf16=(args.dtype == "bf16"), lr=[2e-4], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.0, clip_norm=10.0,
        warmup_updates=4000, warmup_init_lr=1e-7, seed=1, bucket_cap_mb=64,
        loss_ratio=[1.0, 1.0, 1.0], contrastive_temp=0.1, contrastive_increase_until=None, kd_ratio=[None, None])
    if chimera:  # chimera/scripts/train-en2any-ST.sh: 6 shared encoder layers + 3 memory layers, M = 64
        ns.encoder_layers, ns.interlingua_layers, ns.interlingua_length = 6, 3, 64
    torch.manual_seed(1)
    task = reg.setup_task(ns)
    reg.ARCH_CONFIG_REGISTRY[ns.arch](ns)
    model = task.build_model(ns)
    criterion = task.build_criterion(ns)
    trainer = Trainer(ns, task, model, criterion, device=device)
    return trainer, task, tasks, ns


def make_batch(tasks, task, args, rank, device):
    # Every rank draws the SAME utterance / target lengths (the contents differ, seed = 1 + rank): weak scaling means a fixed amount
    # of work per GPU, which is also what the reference's --max-tokens batching aims at (batches of about equal frame counts on every
    # worker).  Independent draws would put a 5 % spread on the ranks' frame totals and the slowest rank's luck into the curve.
    g = torch.Generator().manual_seed(1)
    smax = int(args.seconds * 16000)
    if args.lengths == "max":
        audio = [smax] * args.batch
    else:  # uniform in [10 s, 30 s], multiples of 320, collater sorts descending
        lo = min(160000, smax)
        audio = [int(torch.randint(lo // 320, smax // 320 + 1, (1,), generator=g)) * 320 for _ in range(args.batch)]
        audio[0] = smax
    tgt = [int(torch.randint(16, 129, (1,), generator=g)) for _ in range(args.batch)]
    src = [int(torch.randint(8, 97, (1,), generator=g)) for _ in range(args.batch)]
    return tasks.synthetic_sample(task.target_dictionary, args.batch, audio, tgt, src, seed=1 + rank, device=device)


def algorithmic_tflop(sample, ns, chimera, conv_spec):
    """ALGORITHMIC FLOPs of one update on `sample`, by SURVEY section 8(d)'s formulas (FLOP = 2 MAC, training = 3 x forward), from the
    batch's own lengths: padding, skipped tiles and tile rounding earn nothing.  Split by the launch class that executes them:
    `gemm` (conv layers 1-6, every Linear, pos-conv, subsampler, vocabulary projection), `attention` (QK^T + PV), `conv0`.
    Chimera's memory layers are priced at the MINIMAL variant of 8(d) (M query rows), its text pass included."""
    C, F, H = ns.encoder_embed_dim, ns.encoder_ffn_embed_dim, 768
    V = 10000
    N = ns.encoder_layers
    ND = ns.decoder_layers
    M = int(getattr(ns, "interlingua_length", 0) or 0)
    NM = int(getattr(ns, "interlingua_layers", 0) or 0) if chimera else 0
    lin = lambda rows, d, f: rows * (4 * d * d + 2 * d * f)
    gemm = attn = conv0 = 0

    def tail(T2, U):  # everything behind the (audio or text) front end: encoder layers, memory, decoder, vocabulary
        g = lin(T2, C, F) * N
        a = N * 2 * T2 * T2 * C
        K = T2
        if NM:
            g += NM * (M * (2 * C * C + 2 * C * F) + (T2 + M) * 2 * C * C)
            a += NM * 2 * M * (T2 + M) * C
            K = M
        g += ND * (U * (6 * C * C + 2 * C * F) + K * 2 * C * C) + U * C * V
        a += ND * (2 * U * U * C + 2 * U * K * C)
        return g, a

    S_all = sample["net_input"]["src_lengths"].tolist()
    U_all = sample["target_lengths"].tolist()
    for S, U in zip(S_all, U_all):
        L, cin = S, 1
        for i, (c, k, st) in enumerate(conv_spec):
            L = (L - k) // st + 1
            if i == 0:
                conv0 += cin * c * k * L
            else:
                gemm += cin * c * k * L
            cin = c
        T1 = L
        gemm += T1 * cin * H + T1 * H * (H // 16) * 128 + 12 * lin(T1, H, 4 * H)
        attn += 12 * 2 * T1 * T1 * H
        Ts = (T1 - 1) // 2 + 1
        T2 = (Ts - 1) // 2 + 1
        gemm += H * 1024 * 5 * Ts + C * 1024 * 5 * T2
        g, a = tail(T2, U)
        gemm, attn = gemm + g, attn + a
    if chimera and "src_text_lengths" in sample:
        for Ls, U in zip(sample["src_text_lengths"].tolist(), U_all):
            g, a = tail(Ls, U)
            gemm, attn = gemm + g, attn + a
    k = 6.0 / 1e12
    return {"gemm": gemm * k, "attention": attn * k, "conv0": conv0 * k, "total": (gemm + attn + conv0) * k}


def dominant_gemm_launch(args, device):
    """The single most expensive GEMM launch of the step — wav2vec2 fc1 forward, [B*T1, 768] x [3072, 768]^T with the
    bias + GELU + pre-activation epilogue (12 launches per update) — timed alone with HIP events on the launch stream, with
    the HBM-side traffic of the same launch from the committed PMC pass (profiles/r01_pmc_gemm.json: TCC_EA0_RDREQ/WRREQ,
    64 B per request, reads doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950)."""
    K = importlib.import_module("chimera-st_amd.kernels")
    Lb = importlib.import_module("chimera-st_amd.lib")
    t1 = int(args.seconds * 16000)
    for (k, s_) in [(10, 5), (3, 2), (3, 2), (3, 2), (3, 2), (2, 2), (2, 2)]:
        t1 = (t1 - k) // s_ + 1
    M, D, F = args.batch * t1, 768, 3072
    x = torch.randn(M, D, device=device).to(torch.bfloat16)
    w = (torch.randn(F, D, device=device) * 0.02).to(torch.bfloat16)
    b = torch.zeros(F, device=device, dtype=torch.bfloat16)
    h, z = torch.empty(M, F, device=device, dtype=torch.bfloat16), torch.empty(M, F, device=device, dtype=torch.bfloat16)

    def launch():
        K.gemm(x, w, h, M, F, D, a_kmajor=1, b_kmajor=1, lda=D, ldb=D, ldc=F, bias=b, act=Lb.ACT_GELU, aux_out=z, ld_aux_out=F, split_k=1)

    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flops = 2.0 * M * F * D
    out = {"kernel": "gemm8p_kernel<k-major, k-major>", "shape": [M, F, D], "epilogue": "bias+gelu+aux_out", "avg_launch_ms": ms,
           "achieved": flops / ms / 1e9, "unit": "TFLOP/s", "frac": flops / ms / 1e9 / PEAK["bf16"],
           "algorithmic_bytes": 2.0 * (M * D + F * D + 2 * M * F), "traffic": None}
    pmc = next((f for f in (os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", n) for n in ("r06_pmc_gemm.json", "r05_pmc_gemm.json", "r04_pmc_gemm.json", "r03_pmc_gemm.json", "r02_pmc_gemm.json", "r01_pmc_gemm.json"))
                if os.path.exists(f)), None)
    if pmc is not None:
        rec = json.load(open(pmc))
        if rec.get("shape") == [47968, F, D] and M == 47968:
            out["traffic"] = rec["traffic_bytes_per_launch"]
            out["traffic_note"] = rec["note"]
            out["traffic_source"] = {"file": os.path.relpath(pmc, ROOT), "build": rec.get("build", "unrecorded")}
    return out


def oracle_cfg(ns):
    """The oracle's config dict for the model `build` made (checker side only: cpu_baseline and tests/test_fullsize_gpu.py)."""
    wa = importlib.import_module("chimera-st_amd.w2v2_transformer").SYNTHETIC_W2V["wav2vec_small_bench"]
    return dict(conv_layers=eval(wa.conv_feature_layers), conv_pos=wa.conv_pos, conv_pos_groups=wa.conv_pos_groups,
                w2v_layers=wa.encoder_layers, w2v_heads=wa.encoder_attention_heads, feature_grad_mult=wa.feature_grad_mult,
                d=ns.encoder_embed_dim, heads=ns.encoder_attention_heads, dec_heads=ns.decoder_attention_heads,
                enc_layers=ns.encoder_layers, dec_layers=ns.decoder_layers, mem_layers=getattr(ns, "interlingua_layers", 0))


def cpu_baseline(trainer, task, tasks, ns, args):
    """The oracle (CPU fp32 restatement pinned to the reference, oracle/chimera_oracle.py) timed on this box's host
    cores on a bounded sample of the same workload: ONE full update (fwd + bwd + Adam) on one 30 s utterance."""
    from oracle import chimera_oracle as O
    cfg = oracle_cfg(ns)
    p = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point() and "_float_tensor" not in k and k != "decoder.version")
         for k, v in trainer.get_model().state_dict().items()}
    p["decoder.output_projection.weight"] = p["decoder.embed_tokens.weight"]
    cores = torch.get_num_threads()
    secs = min(args.seconds, args.cpu_seconds)
    fn = O.triplet_criterion if args.model == "chimera" else O.lsce_criterion

    def one(nsamp, audio_s, u):
        s = tasks.synthetic_sample(task.target_dictionary, nsamp, [int(audio_s * 16000)] * nsamp, [u] * nsamp, [u] * nsamp, seed=5)
        t0 = time.time()
        out = fn(p, s, cfg)
        out["loss"].backward()
        leaves = list({id(t): t for t in p.values() if t.requires_grad and t.grad is not None}.values())
        with torch.no_grad():
            for t in leaves:
                O.adam_step(t, t.grad, torch.zeros_like(t), torch.zeros_like(t), 1, 2e-4)
                t.grad = None
        return time.time() - t0

    one(1, 1.0, 8)  # warm the thread pool / allocator
    # thread count: the fastest of a few candidates on a 4 s utterance (on a 128-thread host the intra-op pools stop scaling — and
    # then lose — well before all threads are used: the baseline should be the CPU path at its best, not at its widest)
    best, all_threads, probe = None, cores, {}
    for n in sorted({c for c in (8, 16, 32, 64, all_threads) if c <= all_threads}):
        torch.set_num_threads(n)
        one(1, 1.0, 8)
        t = one(1, min(4.0, secs), 32)
        probe[str(n)] = round(t, 3)
        if best is None or t < best[1]:
            best = (n, t)
    cores = best[0]
    torch.set_num_threads(cores)
    # 1 warm-up + 3 timed updates at the chosen thread count; the reported value is the mean of the three
    times = [one(1, secs, 64) for _ in range(1 + 3)][1:]
    dt = sum(times) / len(times)
    torch.set_num_threads(all_threads)
    return {"value": 1.0 / dt, "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": "CPU fp32 oracle, 1 warm-up + 3 timed updates (fwd+bwd+Adam) of 1 utterance x %.0f s + 64 target tokens, same model "
                      "dims; %d threads = the fastest of the probe" % (secs, cores),
            "seconds_per_update": [round(t, 3) for t in times], "host_threads_available": all_threads,
            "thread_probe_seconds_4s_utterance": probe}


def measure_train(args, device, rank, lib, traffic=True):
    """Build the model of `args`, run warm-up + timed updates on one resident batch (barrier + synchronize on both sides, MAX over
    ranks) and — unless --no-roofline — one more update with every C-ABI launch bracketed by hipEvents on its launch stream."""
    import torch.distributed as dist
    trainer, task, tasks, ns = build(args, device)
    sample = make_batch(tasks, task, args, rank, device)  # resident in HBM before the clock starts

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_step([sample])
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.train_step([sample])
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)

    roof = None
    if not args.no_roofline:
        # one extra (untimed) update on EVERY rank (its gradient all-reduce is a collective); rank 0 wraps each of its launches
        # in a hipEvent pair on the launch stream
        if rank == 0:
            lib.prof_enable(True)
        trainer.train_step([sample])
        barrier()
    if not args.no_roofline and rank == 0:
        table = lib.prof_query()
        lib.prof_enable(False)
        dom = max(table.items(), key=lambda kv: kv[1]["ms"])
        name, r = dom
        if r["flops"] > 0:
            # `achieved` / `frac` are by ALGORITHMIC FLOPs (the batch's own lengths through the survey's formulas; tile
            # padding and whatever the kernels execute beyond that earn nothing); the executed-FLOP figure (what the launches' descriptors
            # add up to, tile rounding included) stays beside it as `*_executed`.
            wa = importlib.import_module("chimera-st_amd.w2v2_transformer").SYNTHETIC_W2V["wav2vec_small_bench"]
            alg = algorithmic_tflop(sample, ns, args.model == "chimera", eval(wa.conv_feature_layers))
            ach_x = r["flops"] / (r["ms"] * 1e-3) / 1e12
            ach = alg[name] / (r["ms"] * 1e-3) if name in alg else ach_x
            roof = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": PEAK[args.dtype], "unit": "TFLOP/s",
                    "frac": ach / PEAK[args.dtype], "frac_algorithmic": ach / PEAK[args.dtype], "achieved_executed": ach_x,
                    "frac_executed": ach_x / PEAK[args.dtype], "traffic": None, "launches": r["launches"],
                    "avg_launch_ms": r["ms"] / max(r["launches"], 1),
                    "algorithmic_tflop_per_update": {k: round(v, 4) for k, v in alg.items()},
                    "step_mfu_algorithmic": alg["total"] / (dt / args.steps) / PEAK[args.dtype]}
        else:
            ach = r["bytes"] / (r["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                    "traffic": None, "launches": r["launches"], "avg_launch_ms": r["ms"] / max(r["launches"], 1)}
        roof["per_class_ms"] = {k: round(v["ms"], 3) for k, v in table.items() if v["launches"]}
        # the HBM-bound launch classes against 8 TB/s: ALGORITHMIC bytes of every launch of the class (each operand once, as its entry
        # point states them to the profiling table) over the class's time in this update
        roof["hbm_bound_classes"] = {k: {"ms": round(v["ms"], 3), "achieved_GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                         "frac": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / 8000.0, 4), "launches": v["launches"]}
                                     for k, v in table.items() if k in ("layernorm", "elementwise", "optim", "loss") and v["launches"] and v["ms"] > 0}
        if r["flops"] > 0:
            a_ms = table["attn_fwd"]["ms"] + table["attn_bwd"]["ms"]
            # QK^T + PV of every attention of the update (forward + 2x backward, algorithmic) over the attention kernels' time
            roof["attention"] = {"ms": round(a_ms, 3), "algorithmic_tflop": round(alg["attention"], 4),
                                 "achieved": alg["attention"] / (a_ms * 1e-3), "frac": alg["attention"] / (a_ms * 1e-3) / PEAK[args.dtype]}
        roof["algorithmic_bytes"] = r["bytes"] / max(r["launches"], 1)
        # HBM-side traffic of the same launches from the committed PMC pass of this command (counters need their own rocprofv3
        # run and cannot be read inside the timed process); only quoted for the workload it was collected on
        pmc = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r06_pmc_gemm_class.json", "r05_pmc_gemm_class.json", "r04_pmc_gemm_class.json", "r03_pmc_gemm_class.json", "r02_pmc_gemm_class.json", "r01f_pmc_gemm_class.json")) if os.path.exists(f)), None)
        if traffic and name == "gemm" and pmc and args.model == "s2t_w2v2" and args.batch == 32 and args.seconds == 30.0 and args.dtype == "bf16":
            rec = json.load(open(pmc))
            if abs(rec["gemm_class_launches_per_update"] - r["launches"]) <= 16:
                roof["traffic"] = rec["traffic_bytes_per_launch"]
                roof["traffic_note"] = rec["note"]
                # PMC counters need their own rocprofv3 --pmc passes and cannot be read inside the timed process: the figure is the
                # committed result of tools/pmc_gemm_class.py over this same command; source file and the build it was taken on:
                roof["traffic_source"] = {"file": os.path.relpath(pmc, ROOT), "build": rec.get("build", "unrecorded")}
        if traffic and name == "gemm" and args.dtype == "bf16":
            roof["dominant_launch"] = dominant_gemm_launch(args, device)
    return trainer, task, tasks, ns, sample, dt, out, roof


def h2d_overlapped(trainer, sample, device, steps=12):
    """Updates fed from PINNED HOST batches: batch i + 1 crosses PCIe on a copy stream while update i runs.  Two RESIDENT device
    batches are allocated once and refilled in place (`copy_`, non-blocking): nothing is allocated on the copy stream inside the
    loop, the compute stream waits only for the event of its own batch, and the copy into a buffer waits for the update that last
    read it.  Returns utterances/s with the transfer inside the clock (SURVEY section 8d defines the metric with H2D; the bench
    contract's `value` starts with the batch resident, so this is reported beside it)."""
    def walk(x, fn):
        if torch.is_tensor(x):
            return fn(x)
        if isinstance(x, dict):
            return {k: walk(v, fn) for k, v in x.items()}
        return x

    def refill(dst, src):
        if torch.is_tensor(dst):
            dst.copy_(src, non_blocking=True)
        elif isinstance(dst, dict):
            for k in dst:
                refill(dst[k], src[k])

    host = [walk(sample, lambda t: t.detach().cpu().pin_memory()) for _ in range(2)]
    dev = [walk(sample, lambda t: torch.empty_like(t)) for _ in range(2)]  # on the current (main) stream, before the loop
    copy, main = torch.cuda.Stream(), torch.cuda.current_stream()
    filled = [torch.cuda.Event(), torch.cuda.Event()]   # copy stream: batch landed in dev[j]
    freed = [torch.cuda.Event(), torch.cuda.Event()]    # main stream: the update that read dev[j] is enqueued behind this point

    def stage(i):
        j = i % 2
        with torch.cuda.stream(copy):
            if i >= 2:
                copy.wait_event(freed[j])
            refill(dev[j], host[j])
            filled[j].record(copy)

    nutt = sample["target"].size(0)
    torch.cuda.synchronize()
    stage(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if i + 1 < steps:
            stage(i + 1)
        main.wait_event(filled[i % 2])
        trainer.train_step([dev[i % 2]])
        freed[i % 2].record(main)
    torch.cuda.synchronize()
    return nutt * steps / (time.perf_counter() - t0)


def h2d_ms(sample, device, reps=3):
    """PCIe cost of handing one batch over (excluded from `value`: the contract times with inputs resident in HBM): pinned host
    copy of every tensor of the batch -> device, non-blocking, timed with events on the copy stream."""
    host = {}

    def pin(x):
        if torch.is_tensor(x):
            return x.detach().cpu().pin_memory()
        if isinstance(x, dict):
            return {k: pin(v) for k, v in x.items()}
        return x

    host = pin(sample)

    def put(x):
        if torch.is_tensor(x):
            return x.to(device, non_blocking=True)
        if isinstance(x, dict):
            return {k: put(v) for k, v in x.items()}
        return x

    put(host)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        put(host)
    e1.record()
    torch.cuda.synchronize()

    def nbytes(x):
        if torch.is_tensor(x):
            return x.numel() * x.element_size()
        if isinstance(x, dict):
            return sum(nbytes(v) for v in x.values())
        return 0

    return e0.elapsed_time(e1) / reps, nbytes(host)


def decode_line(args, device):
    """--mode decode: BASELINE configs[4] — s2t_transformer_l (12 + 6 layers, d 1024, 16 heads, ffn 4096, V = 10 000), filter-bank
    input, beam 5, incremental-state decode on 1 MI355X.  A "step" is one full beam search over one resident batch.  The dominant
    loop is HBM-bound: per decode step every decoder weight, the self-attention K/V caches written so far and the per-sentence
    encoder K/V are read once; `roofline` prices those algorithmic bytes against 8 TB/s."""
    s2t = importlib.import_module("chimera-st_amd.s2t_transformer")
    tasks = importlib.import_module("chimera-st_amd.tasks")
    reg = importlib.import_module("chimera-st_amd.registry")
    SG = importlib.import_module("chimera-st_amd.sequence_generator").SequenceGenerator
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    es = 2 if args.dtype == "bf16" else 4
    torch.manual_seed(1)
    task = tasks.SpeechToTextTask(Namespace(data=None, synthetic_vocab_size=10000))
    ns = Namespace(share_decoder_input_output_embed=True, dropout=0.0)
    reg.ARCH_CONFIG_REGISTRY["s2t_transformer_l"](ns)
    model = s2t.S2TTransformerModel.build_model(ns, task).to(device, dt).eval()
    frames, beam, max_len = int(args.seconds * 100), args.beam, args.max_len
    g = torch.Generator().manual_seed(1)
    lens = torch.randint(frames // 3, frames + 1, (args.batch,), generator=g).sort(descending=True)[0]
    lens[0] = frames
    src = torch.randn(args.batch, frames, 80, generator=g).to(dt).to(device)
    sample = {"net_input": {"src_tokens": src, "src_lengths": lens.to(device)}}
    gen = SG([model], task.target_dictionary, beam_size=beam, max_len_a=0, max_len_b=max_len)
    for _ in range(max(args.warmup, 1)):
        hyps = gen.generate([model], sample)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hyps = gen.generate([model], sample)
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / args.steps
    with torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            model.encoder(src, sample["net_input"]["src_lengths"])
        torch.cuda.synchronize()
        enc_s = (time.perf_counter() - t0) / 3
    nsteps = max_len + 1  # random-init weights emit eos only when forced: every sentence runs all max_len + 1 steps
    ms_step = (dt_s - enc_s) / nsteps * 1e3
    dec = model.decoder
    C, nl = dec.embed_dim, len(dec.layers)
    wbytes = sum(p.numel() for p in dec.parameters()) * es                  # every decoder weight (tied table read once as the projection)
    S = (frames - 1) // 2 + 1
    S = (S - 1) // 2 + 1
    cross = 2 * nl * args.batch * S * C * es                                  # per-sentence encoder K and V, every layer
    selfkv = 2 * nl * args.batch * beam * C * es * (nsteps + 1) / 2.0        # average over the steps of the caches read so far
    step_bytes = wbytes + cross + selfkv
    ach = step_bytes / (ms_step * 1e-3) / 1e9
    ntok = sum(len(h[0]["tokens"]) for h in hyps)
    eng = getattr(gen, "_engine", None)
    nodes = eng.nodes_per_step(dt, args.batch * beam) if eng is not None else 0
    line = {"metric": "decode utterances/sec, s2t_transformer_l beam 5 incremental decode, 1 MI355X", "value": args.batch / dt_s,
            "unit": "utterances/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_s * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "s2t_transformer_l (12 enc + 6 dec, d1024/ffn4096/16h, V=10000), filter-bank input, beam search with "
                                   "incremental state (device-resident loop, one HIP graph per decode step and lane; lanes = groups of sentences decoded side by side on their own streams)", "batch": args.batch, "beam": beam,
                       "max_frames": frames, "max_len": max_len, "tokens_per_s": ntok / dt_s, "encoder_ms": enc_s * 1e3,
                       "ms_per_decode_step": ms_step, "hypothesis_rows_per_step": args.batch * beam,
                       "lanes": eng.lanes if eng is not None else 0},
            "roofline": {"bound": "hbm", "kernel": "one decode step (%d graph nodes: LayerNorm-folded / skinny GEMMs, cache attention, beam step)" % nodes,
                         "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None,
                         "algorithmic_bytes": step_bytes, "avg_launch_ms": ms_step,
                         "bytes_breakdown": {"decoder_weights": wbytes, "encoder_kv": cross, "self_kv_avg": selfkv}},
            "cpu_baseline": None}
    return line


def decode_main(args, device):
    print(json.dumps(decode_line(args, device)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU per step")
    ap.add_argument("--seconds", type=float, default=30.0, help="max audio length")
    ap.add_argument("--lengths", default="uniform", choices=["uniform", "max"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--model", default="s2t_w2v2", choices=["s2t_w2v2", "chimera"])
    ap.add_argument("--dropout", type=float, default=0.1, help="training-recipe dropout (0 = every dropout site off)")
    ap.add_argument("--layerdrop", type=float, default=0.0, help="wav2vec2 encoder_layerdrop (wav2vec_small: 0.05)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=30.0)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the Chimera / decode / H2D-overlap legs of the default 1-GPU run")
    ap.add_argument("--mode", default="train", choices=["train", "decode"], help="decode = BASELINE configs[4] (s2t_transformer_l beam search)")
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--max-len", type=int, default=200)
    args = ap.parse_args()

    dist_mod = importlib.import_module("chimera-st_amd.distributed")
    if dist_mod.needs_self_launch(args.gpus):
        # started plainly (`python bench.py --gpus N`, as the driver starts the N = 1 run): this process has not touched the GPU and
        # becomes the launcher — N rank processes with the torchrun environment, rank 0's JSON line on our stdout
        # (fairseq/distributed_utils.py:286-303).  Under torchrun (WORLD_SIZE set) nothing changes.
        sys.exit(dist_mod.launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    rank, world = dist_mod.distributed_init()
    # intra-op CPU threads: this rank's share of what the container may run (cgroup quota), not the machine's thread count
    importlib.import_module("chimera-st_amd.hostcfg").limit_host_threads(int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if os.environ.get("CST_BENCH_FAIL_RANK") == str(rank):  # (tests: a rank that dies must take the self-launched job down with it)
        sys.exit(7)
    assert world == args.gpus or (world == 1 and args.gpus == 1), "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    lib = importlib.import_module("chimera-st_amd.lib")
    lib.load()
    if args.mode == "decode":
        assert world == 1, "--mode decode is a 1-GPU measurement (BASELINE configs[4])"
        return decode_main(args, device)

    import torch.distributed as dist
    trainer, task, tasks, ns, sample, dt, out, roof = measure_train(args, device, rank, lib)
    h2d = h2d_ms(sample, device) if rank == 0 else (None, None)
    h2d_rate = h2d_overlapped(trainer, sample, device) if (world == 1 and not args.no_extra) else None
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(trainer, task, tasks, ns, args)
        # the oracle restates the reference with plain ATen calls and is slower than the reference itself on the same cores; the
        # ratio was measured in the build container, where /root/reference can be imported (profiles/r01d_cpu_reference_vs_oracle.txt)
        cpu["reference_equivalent"] = {"value": cpu["value"] * 1.82, "unit": cpu["unit"], "oracle_over_reference_time": 1.82,
                                       "approximate": True,
                                       "source": "APPROXIMATE, provenance only: the ratio was measured once, at 8 threads, in the build container "
                                                 "(profiles/r01d_cpu_reference_vs_oracle.txt: 30 s utterance, fp32, reference 4.65 s, oracle 8.43 s per "
                                                 "update, identical loss) and is applied here to whatever thread count this box's probe picked"}
    extra = None
    if rank == 0 and world == 1 and not args.no_extra and args.model == "s2t_w2v2" and args.mode == "train":
        # BASELINE configs[3] (Chimera M = 64, joint MT + ST batches) and configs[4] (s2t_transformer_l beam-5 decode) on the same box,
        # so that the driver's record carries them too (short runs: ~15 s together)
        CF = importlib.import_module("chimera-st_amd.functional")
        extra = {}
        if args.lengths != "max":
            # the all-30 s variant of the headline workload (no padding, nothing to skip: the configuration BASELINE.md section 3 prices),
            # same trainer, 2 warm-ups + 4 timed updates
            ma = argparse.Namespace(**vars(args))
            ma.lengths = "max"
            m_sample = make_batch(tasks, task, ma, rank, device)
            for _ in range(2):  # (new shapes: the caching allocator grows during the first update)
                trainer.train_step([m_sample])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                m_out = trainer.train_step([m_sample])
            torch.cuda.synchronize()
            m_dt = (time.perf_counter() - t0) / 4
            lib.prof_enable(True)
            trainer.train_step([m_sample])
            torch.cuda.synchronize()
            m_tab = lib.prof_query()
            lib.prof_enable(False)
            m_flops = sum(v["flops"] for v in m_tab.values())
            extra["maxlen"] = {"metric": "train utterances/sec, same model, every utterance 30 s long (--lengths max)", "value": args.batch / m_dt,
                               "unit": "utterances/s", "ms_per_step": m_dt * 1e3, "steps": 4, "loss": float(m_out["loss"]),
                               "executed_tflop_per_update": m_flops / 1e12, "mfu": m_flops / m_dt / 1e12 / PEAK[args.dtype],
                               "per_class_ms": {k: round(v["ms"], 3) for k, v in m_tab.items() if v["launches"]}}
            del m_sample, m_out
        del trainer, sample
        CF.WEIGHT_TRANSPOSES.invalidate()  # the next model must not refresh (or keep alive) this one's W^T copies
        torch.cuda.empty_cache()
        ca = argparse.Namespace(**vars(args))
        ca.model, ca.steps, ca.warmup = "chimera", 4, 2
        c_tr, _, _, _, c_sample, c_dt, c_out, c_roof = measure_train(ca, device, rank, lib, traffic=False)
        extra.update({"chimera": {"metric": "train utterances/sec, Chimera s2t_transformer_w2v2_interlingua_base M=64 (triplet_st_mt_contrastive), 1 MI355X",
                             "value": args.batch * ca.steps / c_dt, "unit": "utterances/s", "ms_per_step": c_dt / ca.steps * 1e3, "steps": ca.steps,
                             "loss": float(c_out["loss"]), "roofline": c_roof}})
        del c_tr, c_sample
        CF.WEIGHT_TRANSPOSES.invalidate()
        torch.cuda.empty_cache()
        da = argparse.Namespace(**vars(args))
        da.steps, da.warmup = 1, 1
        dl = decode_line(da, device)
        extra["decode"] = {"metric": dl["metric"], "value": dl["value"], "unit": dl["unit"], "ms_per_step": dl["ms_per_step"],
                           "ms_per_decode_step": dl["config"]["ms_per_decode_step"], "tokens_per_s": dl["config"]["tokens_per_s"],
                           "roofline": dl["roofline"]}

    # RCCL prints a version banner through C stdio, which is block-buffered when stdout is a pipe and would otherwise land AFTER
    # the JSON line at process exit: flush every rank's C stdio, then rank 0 prints the line last.
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        utt = world * args.batch * args.steps
        line = {
            "metric": "train utterances/sec, MuST-C EN-DE s2t_transformer_m, 1/2/4/8 MI355X",
            "value": utt / dt, "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("s2t_transformer_w2v2 (s2t_transformer_m dims: d512/ffn2048/8h, 12 enc + 6 dec) + wav2vec2-small "
                                    "frontend (7-layer CNN on raw 16 kHz + 12x768 encoder), label_smoothed_cross_entropy, Adam"
                                    if args.model == "s2t_w2v2" else
                                    "Chimera s2t_transformer_w2v2_interlingua_base (6 enc + 3 memory layers, M=64) + wav2vec2-small, "
                                    "triplet_st_mt_contrastive, Adam"),
                       "batch_per_gpu": args.batch, "global_batch": world * args.batch, "max_audio_s": args.seconds,
                       "audio_lengths": args.lengths, "lengths_per_rank": "identical on every rank (fixed work per GPU); contents differ", "target_tokens": "16-128", "vocab": 10000, "dropout": args.dropout, "w2v_layerdrop": args.layerdrop,
                       "parallelism": "dp%d" % world, "loss": float(out["loss"]),
                       "parity": {"fp32": "1e-3 vs oracle (logits, loss terms, gradients; tests/test_fullsize_gpu.py, test_model_gpu.py)",
                                  "bf16": "<= 1.5x storage-rounding emulation of the oracle (~1.8e-2 gradient rel-L2 on both sides); "
                                          "the 1e-3 bar of north_star is met in fp32 storage only"},
                       "h2d": {"included_in_value": False, "note": "inputs resident in HBM when the clock starts (bench contract); one "
                               "batch pinned host -> device measured separately; value_with_h2d = 12 updates fed from pinned host batches, "
                               "batch i + 1 copied on a side stream under update i into one of two resident device batches", "ms_per_batch": h2d[0], "bytes_per_batch": h2d[1],
                               "value_with_h2d": h2d_rate}},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if args.dropout > 0:
            line["config"]["parity_note"] = ("dropout masks are counter-based (not torch's Philox stream): this configuration is checked by "
                                             "mask-exact kernel tests and a finite-difference test, oracle parity runs at dropout 0")
        if extra is not None:
            line["extra"] = extra
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
