#!/usr/bin/env python3
"""Decode throughput of BASELINE config 5 (SURVEY §8d "Decode"): s2t_transformer_l (d 1024, ffn 4096, 16 heads, 12 encoder +
6 decoder layers, 10 000-way tied vocabulary), filter-bank input, beam 5, incremental-state decode, 1 x MI355X, bf16.

  python tools/bench_decode.py [--batch 32] [--frames 3000] [--beam 5] [--max-len 200] [--reps 3] [--mirror]

Prints one JSON line: utterances/s and generated tokens/s of the device-resident loop (decode_engine.py: one captured HIP
graph per step), the per-step time, the encoder time, and — with --mirror — the same numbers for the host-driven
module-by-module loop (fused=False).  Random-init weights emit eos only when forced, so every sentence runs the full
max_len + 1 steps: the reported rate is the worst case for the configured max_len."""
import argparse
import importlib
import json
import os
import sys
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=3000, help="max filter-bank frames (10 ms each)")
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--max-len", type=int, default=200)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--arch", default="s2t_transformer_l")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--mirror", action="store_true", help="also time the host-driven loop (fused=False)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cross-kernel", default="flash", choices=["flash", "flash_hm", "shared"])
    ap.add_argument("--profile", action="store_true", help="per-class GPU time of one eager decode loop (hipEvent pairs)")
    args = ap.parse_args()

    importlib.import_module("chimera-st_amd")
    s2t = importlib.import_module("chimera-st_amd.s2t_transformer")
    tasks = importlib.import_module("chimera-st_amd.tasks")
    reg = importlib.import_module("chimera-st_amd.registry")
    lib = importlib.import_module("chimera-st_amd.lib")
    SG = importlib.import_module("chimera-st_amd.sequence_generator").SequenceGenerator
    lib.load()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(1)
    task = tasks.SpeechToTextTask(Namespace(data=None, synthetic_vocab_size=10000))
    ns = Namespace(share_decoder_input_output_embed=True, dropout=0.0)
    reg.ARCH_CONFIG_REGISTRY[args.arch](ns)
    model = s2t.S2TTransformerModel.build_model(ns, task).to("cuda", dt).eval()

    g = torch.Generator().manual_seed(1)
    lens = torch.randint(args.frames // 3, args.frames + 1, (args.batch,), generator=g).sort(descending=True)[0]
    lens[0] = args.frames
    src = torch.randn(args.batch, args.frames, 80, generator=g).to(dt).cuda()
    sample = {"net_input": {"src_tokens": src, "src_lengths": lens.cuda()}}

    def timed(gen):
        hyps = gen.generate([model], sample)  # warm-up (graph capture, code objects)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            hyps = gen.generate([model], sample)
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - t0) / args.reps
        ntok = sum(len(h[0]["tokens"]) for h in hyps)
        return dt_, ntok

    with torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            model.encoder(src, sample["net_input"]["src_lengths"])
        torch.cuda.synchronize()
        enc_s = (time.perf_counter() - t0) / args.reps

    fused = SG([model], task.target_dictionary, beam_size=args.beam, max_len_a=0, max_len_b=args.max_len, use_graph=not args.no_graph,
               cross_kernel=args.cross_kernel)
    t_f, ntok = timed(fused)
    steps = args.max_len + 1
    out = {"metric": "decode utterances/sec, s2t_transformer_l beam 5, 1 MI355X", "config": {"arch": args.arch, "batch": args.batch,
           "beam": args.beam, "max_frames": args.frames, "max_len": args.max_len, "dtype": args.dtype, "graph": not args.no_graph},
           "utterances_per_s": args.batch / t_f, "tokens_per_s": ntok / t_f, "best_hyp_tokens": ntok, "s_per_batch": t_f,
           "encoder_s": enc_s, "ms_per_step": (t_f - enc_s) / steps * 1e3, "hyp_rows_per_step": args.batch * args.beam}
    if args.mirror:
        mirror = SG([model], task.target_dictionary, beam_size=args.beam, max_len_a=0, max_len_b=args.max_len, fused=False)
        t_m, ntok_m = timed(mirror)
        out["mirror_host_loop"] = {"utterances_per_s": args.batch / t_m, "tokens_per_s": ntok_m / t_m, "s_per_batch": t_m,
                                   "ms_per_step": (t_m - enc_s) / steps * 1e3}
        out["speedup_vs_host_loop"] = t_m / t_f
    if args.profile:
        eager = SG([model], task.target_dictionary, beam_size=args.beam, max_len_a=0, max_len_b=args.max_len, use_graph=False)
        eager.generate([model], sample)
        lib.prof_enable(True)
        eager.generate([model], sample)
        torch.cuda.synchronize()
        table = lib.prof_query()
        lib.prof_enable(False)
        out["per_class_ms_per_step"] = {k: round(v["ms"] / steps, 4) for k, v in table.items() if v["launches"]}
        out["launches_per_step"] = sum(v["launches"] for v in table.values()) / steps
    print(json.dumps(out))


if __name__ == "__main__":
    main()
