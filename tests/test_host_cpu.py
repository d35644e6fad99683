"""Host-logic tests that need no GPU: registry surface, arch presets (incl. quirk Q4), dictionary, collater-shaped
synthetic batches, schedules, positional tables vs the oracle, flat parameter buffers, state-dict key compatibility
with the reference fixtures, and the world_size-2 gloo path of the bucketed gradient all-reduce."""
import ast
import math
import os
import socket
from argparse import Namespace
from importlib import import_module

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, load_pkg
from oracle import chimera_oracle as O

load_pkg()
reg = import_module("chimera-st_amd.registry")
tasks = import_module("chimera-st_amd.tasks")
w2t = import_module("chimera-st_amd.w2v2_transformer")
inter = import_module("chimera-st_amd.w2v2_transformer_interlingua")
s2t = import_module("chimera-st_amd.s2t_transformer")
crit = import_module("chimera-st_amd.criterions")
optim = import_module("chimera-st_amd.optim")
modules = import_module("chimera-st_amd.modules")
Dictionary = import_module("chimera-st_amd.dictionary").Dictionary
distributed = import_module("chimera-st_amd.distributed")


def test_registry_names_match_reference():
    for m in ("s2t_transformer_w2v2_interlingua", "s2t_transformer_w2v2", "s2t_transformer", "wav2vec2"):
        assert m in reg.MODEL_REGISTRY
    for a in ("s2t_transformer_w2v2_interlingua_base", "s2t_transformer_w2v2", "s2t_transformer_w2v2_s", "s2t_transformer_w2v2yr_s",
              "s2t_transformer_w2v2_sp", "s2t_transformer_w2v2asr_s", "s2t_transformer_s", "s2t_transformer_sp", "s2t_transformer_m",
              "s2t_transformer_mp", "s2t_transformer_l", "s2t_transformer_lp"):
        assert a in reg.ARCH_MODEL_REGISTRY and a in reg.ARCH_CONFIG_REGISTRY
    assert set(("triplet", "speech_to_text")) <= set(reg.TASK_REGISTRY)
    assert set(("triplet_st_mt_contrastive", "label_smoothed_cross_entropy")) <= set(reg.CRITERION_REGISTRY)


def test_arch_quirk_q4_and_script_flags():
    """chimera/scripts/train-en2any-ST.sh flag set parses unchanged; Q4: the interlingua preset's 256/4 defaults never apply."""
    argv = ["/data", "--task", "triplet", "--arch", "s2t_transformer_w2v2_interlingua_base", "--criterion", "triplet_st_mt_contrastive",
            "--w2v2-model-path", "synthetic:wav2vec_small", "--max-tokens", "2000000", "--update-freq", "1", "--label-smoothing", "0.1",
            "--loss-ratio", "1", "1", "1", "--encoder-layers", "6", "--interlingua-length", "64", "--interlingua-layers", "3",
            "--optimizer", "adam", "--adam-betas", "(0.9, 0.98)", "--lr", "1e-4", "--lr-scheduler", "inverse_sqrt", "--warmup-updates", "25000",
            "--clip-norm", "10.0", "--ddp-backend", "no_c10d", "--share-decoder-input-output-embed", "--seed", "1", "--max-update", "150000",
            "--max-source-positions", "2000000"]
    args = reg.parse_args_and_arch(argv)
    assert args.encoder_embed_dim == 512 and args.encoder_attention_heads == 8 and args.encoder_ffn_embed_dim == 2048
    assert args.encoder_layers == 6 and args.decoder_layers == 6 and args.interlingua_length == 64
    assert args.encoder_normalize_before and args.decoder_normalize_before
    assert args.dropout == 0.1 and args.attention_dropout == 0.1 and args.activation_dropout == 0.1
    assert args.loss_ratio == [1.0, 1.0, 1.0] and args.ddp_backend == "no_c10d"
    # flags the scripts do not pass keep the reference's defaults: world size = visible GPUs (None until the launcher counts them,
    # options.py:310), no optimizer-state sharding (dataclass/configs.py:327); `--zero-sharding os` / `--distributed-world-size N` parse
    assert args.zero_sharding == "none" and args.distributed_world_size is None
    a2 = reg.parse_args_and_arch(argv + ["--zero-sharding", "os", "--distributed-world-size", "8"])
    assert a2.zero_sharding == "os" and a2.distributed_world_size == 8
    ns = Namespace()
    reg.ARCH_CONFIG_REGISTRY["s2t_transformer_m"](ns)
    assert (ns.encoder_embed_dim, ns.encoder_attention_heads, ns.dropout, ns.encoder_layers) == (512, 8, 0.15, 12)
    ns = Namespace()
    reg.ARCH_CONFIG_REGISTRY["s2t_transformer_l"](ns)
    assert (ns.encoder_embed_dim, ns.encoder_ffn_embed_dim, ns.decoder_attention_heads) == (1024, 4096, 16)


def test_dictionary_and_collater_layout():
    d = Dictionary.synthetic(10000)
    assert len(d) == 10000 and (d.bos(), d.pad(), d.eos(), d.unk()) == (0, 1, 2, 3)
    s = tasks.synthetic_sample(d, 3, [3200, 8000, 4800], [4, 9, 2], [3, 5, 7], seed=3)
    assert s["net_input"]["src_lengths"].tolist() == [8000, 4800, 3200]  # sorted descending like the collater
    assert s["net_input"]["src_tokens"].shape == (3, 8000) and s["net_input"]["mask"] is False
    assert (s["net_input"]["prev_output_tokens"][:, 0] == d.eos()).all()
    tl = s["target_lengths"].tolist()
    for i, n in enumerate(tl):
        assert s["target"][i, n - 1] == d.eos() and (s["target"][i, n:] == d.pad()).all()
        assert (s["net_input"]["prev_output_tokens"][i, 1:n] == s["target"][i, :n - 1]).all()
    assert s["ntokens"] == sum(tl) and set(s) >= {"id", "net_input", "target", "target_lengths", "src_text", "src_text_lengths", "ntokens", "nsentences"}


def test_lr_schedule_and_positions_match_oracle():
    for n in (0, 1, 3, 4, 5, 100, 25000, 30000):
        assert optim.inverse_sqrt_lr(n, 1e-3, 4, 1e-7) == pytest.approx(O.inverse_sqrt_lr(n, 1e-3, 4, 1e-7), rel=1e-12)
    toks = torch.tensor([[5, 6, 7, 1, 1], [8, 9, 4, 6, 2]])
    pe = modules.PositionalEmbedding(1024, 64, 1)
    np.testing.assert_allclose(pe(toks).numpy(), O.positional_embedding(toks, 64, 1).numpy(), rtol=0, atol=0)
    mask = torch.tensor([[False, False, True], [False, False, False]])
    np.testing.assert_allclose(pe(mask).numpy(), O.positional_embedding(mask, 64, 1).numpy(), rtol=0, atol=0)


@pytest.mark.parametrize("fixture,kind", [("chimera_tiny.npz", "chimera"), ("s2t_w2v2_tiny.npz", "s2t")])
def test_state_dict_keys_match_reference(fixture, kind):
    """Checkpoint compatibility: our modules expose exactly the reference's state-dict keys and shapes."""
    g = load_golden(fixture)
    w = ast.literal_eval(str(g["meta/w2v_args"]))
    m = ast.literal_eval(str(g["meta/model_args"]))
    w2t.SYNTHETIC_W2V["golden_tiny_cpu"] = Namespace(**w)
    args = Namespace(**m)
    args.w2v2_model_path = "synthetic:golden_tiny_cpu"
    V = g["param/decoder.embed_tokens.weight"].shape[0]
    task = tasks.TripletTask(Namespace(data=None, synthetic_vocab_size=V))
    cls = inter.S2TTransformerInterlinguaModelW2V2 if kind == "chimera" else w2t.S2TTransformerModelW2V2
    model = cls.build_model(args, task)
    own = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    ref = {k[len("param/"):]: tuple(v.shape) for k, v in g.items() if k.startswith("param/")}
    assert own == ref
    names = [n for n, _ in model.named_parameters()]
    assert "decoder.output_projection.weight" not in names  # tied to decoder.embed_tokens.weight (Q5)
    assert model.decoder.output_projection.weight is model.decoder.embed_tokens.weight
    if kind == "chimera":
        assert model.encoder.text_embed_tokens.weight is not model.decoder.embed_tokens.weight  # Q5: separate tables


def test_flat_param_buffers_alias_and_zero():
    lin = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 7))
    ref = [p.detach().clone() for p in lin.parameters()]
    buf = optim.FlatParamBuffers(lin.parameters())
    for p, r, o in zip(lin.parameters(), ref, buf.offsets):
        assert torch.equal(p, r) and o % optim.ALIGN == 0
        assert p.data_ptr() == buf.flat_param.data_ptr() + o * 4
    lin(torch.randn(2, 5)).sum().backward()
    want = [p.grad.clone() for p in lin.parameters()]
    assert float(buf.flat_grad.abs().sum()) == 0  # autograd owns the fresh gradients until they are gathered
    buf.gather_grads()
    for p, w, v in zip(lin.parameters(), want, buf.grad_views):
        assert torch.equal(v, w) and p.grad.data_ptr() == v.data_ptr()
    buf.zero_grad()
    assert float(buf.flat_grad.abs().sum()) == 0 and all(p.grad is None for p in lin.parameters())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, q, collective="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), CST_DDP_COLLECTIVE=collective)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 30), torch.nn.ReLU(), torch.nn.Linear(30, 4),
                              torch.nn.Linear(4, 4))  # last layer: only used on some steps
    buf = optim.FlatParamBuffers(net.parameters())
    model = distributed.DistributedFairseqModel(Namespace(bucket_cap_mb=0.0005), net, buf)  # ~128 floats/bucket -> several buckets
    assert len(model.reducer.buckets) >= 3 and model.reducer.collective == collective
    g = torch.Generator().manual_seed(100 + rank)
    xs = [torch.randn(5, 6, generator=g) for _ in range(3)]
    # step 1: plain update; every rank uses its own data
    buf.zero_grad()
    distributed.notify_unused_parameters(net[5].parameters())  # what wav2vec2 layerdrop reports for a skipped layer
    net[:5](xs[0]).pow(2).sum().backward()  # net[5] gets NO gradient -> must be all-reduced as zeros
    # net[5]'s bucket is the first in launch order: without the notification nothing could start before finish()
    assert model.reducer._next == len(model.reducer.buckets), "buckets were not launched during backward"
    model.all_reduce()
    step1 = buf.flat_grad.clone()
    # step 2: gradient accumulation over two micro-batches, collective only on the last one
    buf.zero_grad()
    with model.no_sync():
        net(xs[1]).pow(2).sum().backward()
    net(xs[2]).pow(2).sum().backward()
    model.all_reduce()
    q.put((rank, step1.numpy(), buf.flat_grad.clone().numpy(), [x.numpy() for x in xs]))
    dist.destroy_process_group()


def test_qkv_parameters_laid_out_back_to_back_are_one_view():
    """FlatParamBuffers(adjacent=qkv_groups(model)) puts the q | k | v projections of a self-attention module next to each other;
    functional.stacked_rows then returns the packed [3C, C] operand as a VIEW (no copy) and its backward hands each parameter its
    block of the packed gradient.  The optimizer-state index space (model.parameters() order) is unchanged."""
    optim = import_module("chimera-st_amd.optim")
    CF = import_module("chimera-st_amd.functional")
    torch.manual_seed(0)

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.self_attention = True
            # fairseq registration order: k, v, q, out (modules/multihead_attention.py:70-82)
            self.k_proj, self.v_proj, self.q_proj = torch.nn.Linear(16, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 16)
            self.out_proj = torch.nn.Linear(16, 16)

    net = torch.nn.Sequential(torch.nn.Linear(16, 16), Attn(), torch.nn.Linear(16, 8))
    before = [p.detach().clone() for p in net.parameters()]
    buf = optim.FlatParamBuffers(net.parameters(), adjacent=optim.qkv_groups(net))
    assert [id(p) for p in buf.params] == [id(p) for p in net.parameters()]  # index space untouched
    for p, b in zip(net.parameters(), before):
        assert torch.equal(p, b)
    spans = sorted((o, o + p.numel()) for p, o in zip(buf.params, buf.offsets))
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= buf.total  # disjoint slots
    a = net[1]
    w = CF.stacked_rows(a.q_proj.weight, a.k_proj.weight, a.v_proj.weight)
    bq = CF.stacked_rows(a.q_proj.bias, a.k_proj.bias, a.v_proj.bias)
    assert w.data_ptr() == a.q_proj.weight.data_ptr() and bq.data_ptr() == a.q_proj.bias.data_ptr()
    assert torch.equal(w, torch.cat((a.q_proj.weight, a.k_proj.weight, a.v_proj.weight), 0))
    x = torch.randn(5, 16)
    (torch.nn.functional.linear(x, w, bq) ** 2).sum().backward()
    ref = torch.cat((a.q_proj.weight, a.k_proj.weight, a.v_proj.weight), 0).detach().requires_grad_(True)
    refb = torch.cat((a.q_proj.bias, a.k_proj.bias, a.v_proj.bias), 0).detach().requires_grad_(True)
    (torch.nn.functional.linear(x, ref, refb) ** 2).sum().backward()
    got = torch.cat((a.q_proj.weight.grad, a.k_proj.weight.grad, a.v_proj.weight.grad), 0)
    assert torch.allclose(got, ref.grad) and torch.allclose(torch.cat((a.q_proj.bias.grad, a.k_proj.bias.grad, a.v_proj.bias.grad)), refb.grad)
    # not adjacent (no buffers): falls back to a copy
    free = Attn()
    w2 = CF.stacked_rows(free.q_proj.weight, free.k_proj.weight, free.v_proj.weight)
    assert w2.data_ptr() != free.q_proj.weight.data_ptr() and w2.shape == (48, 16)
    # the reducer cuts its buckets by storage position, whatever the parameter order
    D = import_module("chimera-st_amd.distributed")
    red = D.BucketedGradAllReduce(buf.params, buf.offsets, buf.flat_grad, None, bucket_cap_mb=1e-3)
    seen = sorted(i for b in red.buckets for i in b["members"])
    assert seen == list(range(len(buf.params)))
    for b in red.buckets:
        for i in b["members"]:
            assert b["lo"] <= buf.offsets[i] and buf.offsets[i] + buf.params[i].numel() <= b["hi"]


@pytest.mark.parametrize("collective", ["allreduce", "rs_ag"])
def test_bucketed_all_reduce_gloo_world2(collective):
    """rs_ag: every bucket as reduce-scatter + all-gather in place (CST_DDP_COLLECTIVE) — the same means, bit for bit on every rank."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q, collective)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # every rank ends with identical (averaged) gradients
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=0, atol=0)
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=0, atol=0)
    # ... equal to the mean of the per-rank gradients computed serially
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 30), torch.nn.ReLU(), torch.nn.Linear(30, 4),
                              torch.nn.Linear(4, 4))
    buf = optim.FlatParamBuffers(net.parameters())
    want1 = torch.zeros_like(buf.flat_grad)
    want2 = torch.zeros_like(buf.flat_grad)
    for r in range(world):
        xs = [torch.from_numpy(x) for x in res[r][3]]
        buf.zero_grad()
        net[:5](xs[0]).pow(2).sum().backward()
        buf.gather_grads()
        want1 += buf.flat_grad / world
        buf.zero_grad()
        net(xs[1]).pow(2).sum().backward()
        net(xs[2]).pow(2).sum().backward()
        buf.gather_grads()
        want2 += buf.flat_grad / world
    np.testing.assert_allclose(res[0][1], want1.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(res[0][2], want2.numpy(), rtol=1e-5, atol=1e-6)
    tail = buf.offsets[-2]
    assert np.abs(res[0][1][tail:]).sum() == 0  # the unused layer stayed exactly zero in step 1


# ---- Trainer-level cross-rank guards (fairseq/trainer.py:564-570 OOM flag, :1045-1077 _check_grad_norms), world 2 over gloo ----
class _ToyCriterion(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fail_next = False

    def forward(self, model, sample):
        if self.fail_next:
            self.fail_next = False
            raise RuntimeError("HIP out of memory. Tried to allocate 20.00 GiB (injected by the test)")
        if getattr(self, "report_unused", None) is not None:  # what wav2vec2 layerdrop does for a skipped layer — wrongly, here
            import_module("chimera-st_amd.distributed").notify_unused_parameters(self.report_unused)
        loss = model(sample["x"]).pow(2).sum()
        n = sample["x"].shape[0]
        return loss, n, {"loss": loss.detach(), "sample_size": n, "ntokens": n, "nsentences": n}

    @staticmethod
    def logging_keys():
        return ("loss", "sample_size", "ntokens", "nsentences")


def _cpu_trainer(rank):
    """A Trainer on CPU tensors: the two optimizer KERNELS (cst_sumsq, cst_adam_step) are replaced by torch arithmetic — the
    guards under test live in the host logic around them."""
    Trainer = import_module("chimera-st_amd.trainer").Trainer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 4))
    args = Namespace(bf16=False, lr=[1e-2], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.0, clip_norm=0.0, seed=1,
                     bucket_cap_mb=0.0005)
    tr = Trainer(args, tasks.FairseqTask(args), net, _ToyCriterion(), device="cpu")
    opt = tr.optimizer
    opt.grad_sumsq = lambda: opt.buf.flat_grad.double().pow(2).sum().float().reshape(1)

    def step(multiply=1.0, gnorm=None):
        opt.num_updates += 1
        opt.buf.flat_param.sub_(opt.lr * float(multiply) * opt.buf.flat_grad)  # plain SGD stands in for the Adam kernel
        return gnorm

    opt.step = step
    return tr


def _guard_worker(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = _cpu_trainer(rank)
    g = torch.Generator().manual_seed(100 + rank)
    batch = lambda: {"x": torch.randn(5, 6, generator=g)}
    out = {"rank": rank}
    if mode == "oom_first":  # out of memory in the very FIRST update of rank 1: it has never seen a logging output
        if rank == 1:
            tr.criterion.fail_next = True
        out["first_none"] = tr.train_step([batch()]) is None and tr.num_updates == 0
    o1 = tr.train_step([batch()])
    out["step1"] = (o1["sample_size"], o1["gnorm"], tr.buffers.flat_param.clone().numpy())
    if mode == "nonfinite":
        tr.nonfinite_tolerance = 2
        before = tr.buffers.flat_param.clone()
        bad = batch()
        if rank == 1:
            bad["x"][0, 0] = float("inf")  # one rank's batch blows up; the all-reduced gradient is non-finite everywhere
        o2 = tr.train_step([bad])
        out["skipped"] = bool(o2 is not None and o2.get("overflow") == 1.0 and math.isnan(o2["gnorm"]))
        out["unchanged"] = bool(torch.equal(before, tr.buffers.flat_param)) and tr.num_updates == 1
        o3 = tr.train_step([batch()])  # a clean update goes through and resets the run counter
        out["step3"] = (o3["sample_size"], tr.buffers.flat_param.clone().numpy(), tr.num_updates)
        raised = None
        try:
            for _ in range(4):
                b = batch()
                b["x"][0, 0] = float("nan")
                tr.train_step([b])
        except FloatingPointError as e:
            raised = str(e)
        out["raised"] = raised
        out["updates_at_raise"] = tr.num_updates
    elif mode in ("late", "late_one", "late_oom"):
        # a parameter reported unused that takes part after all: its bucket may leave without its gradient; the ranks agree on the
        # late set through the statistics vector and reduce it again.  late_oom: rank 1 loses its pass (no late set of its own).
        w0 = tr.get_model()[0].weight
        if mode != "late_one" or rank == 1:  # late_one: the gradient is late on rank 1 only; rank 0's bucket carried its gradient
            tr.criterion.report_unused = [w0]
        if mode == "late_oom" and rank == 1:
            tr.criterion.fail_next = True
        xs = batch()
        o2 = tr.train_step([xs])
        tr.criterion.report_unused = None
        out["step2_none"] = o2 is None
        out["x"] = xs["x"].numpy()
        out["param"] = tr.buffers.flat_param.clone().numpy()
        out["updates"] = tr.num_updates
        out["late_count"] = getattr(tr, "last_late_count", None)
        out["offsets"] = list(tr.buffers.offsets)
    elif mode == "oom":
        before = tr.buffers.flat_param.clone()
        if rank == 1:
            tr.criterion.fail_next = True  # rank 1 loses its forward pass; rank 0 completes a normal backward
        o2 = tr.train_step([batch()])
        out["step2_none"] = o2 is None
        out["unchanged"] = bool(torch.equal(before, tr.buffers.flat_param)) and tr.num_updates == 1
        o3 = tr.train_step([batch()])  # the job carries on, replicas still identical
        out["step3"] = (o3["sample_size"], tr.buffers.flat_param.clone().numpy(), tr.num_updates)
    else:  # a replica whose reduced gradient differs (a skipped bucket, a corrupted buffer)
        orig = tr.model.all_reduce

        def bad():
            orig()
            if rank == 1:
                tr.buffers.flat_grad[3] += 0.5

        tr.model.all_reduce = bad
        try:
            tr.train_step([batch()])
            out["raised"] = None
        except RuntimeError as e:
            out["raised"] = str(e)
    q.put(out)
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["oom", "oom_first", "diverge", "nonfinite", "late", "late_one", "late_oom"])
def test_trainer_oom_flag_and_grad_norm_check_gloo_world2(mode):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_guard_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0]["step1"][0] == res[1]["step1"][0] == 10.0  # sample sizes were summed over the ranks
    assert res[0]["step1"][1] == res[1]["step1"][1]
    np.testing.assert_array_equal(res[0]["step1"][2], res[1]["step1"][2])
    if mode == "oom":
        for r in res:
            assert r["step2_none"] and r["unchanged"], r  # BOTH ranks dropped the update although only rank 1 failed
            assert r["step3"][0] == 10.0 and r["step3"][2] == 2
        np.testing.assert_array_equal(res[0]["step3"][1], res[1]["step3"][1])
    elif mode == "oom_first":
        assert all(r["first_none"] for r in res)  # nobody hung in a collective, nobody stepped
    elif mode == "nonfinite":
        for r in res:
            assert r["skipped"] and r["unchanged"], r  # inf on ONE rank: every rank skipped, parameters bit-unchanged
            assert r["step3"][0] == 10.0 and r["step3"][2] == 2
            assert r["raised"] is not None and "consecutive" in r["raised"] and r["updates_at_raise"] == 2  # tolerance 2 -> third one raises
        np.testing.assert_array_equal(res[0]["step3"][1], res[1]["step3"][1])
    elif mode in ("late", "late_one"):
        # the update equals plain SGD on the mean gradient of the two batches, the late parameter included
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 4))
        flat0, offs = res[0]["step1"][2], res[0]["offsets"]
        params = list(net.parameters())
        with torch.no_grad():
            for p_, o in zip(params, offs):
                p_.copy_(torch.from_numpy(flat0[o:o + p_.numel()]).view(p_.shape))
        sum(net(torch.from_numpy(r["x"])).pow(2).sum() for r in res).backward()
        for p_, o in zip(params, offs):  # SGD stand-in: param -= lr * (world / sample_size) * mean gradient
            want = (p_ - 1e-2 * (2 / 10.0) * p_.grad / 2).detach().reshape(-1).numpy()
            np.testing.assert_allclose(res[0]["param"][o:o + p_.numel()], want, rtol=1e-5, atol=1e-6)
        for r in res:
            assert not r["step2_none"] and r["updates"] == 2
            assert r["late_count"] == (2 if mode == "late" else 1), r["late_count"]  # ranks that saw the gradient arrive after its bucket had left
        np.testing.assert_array_equal(res[0]["param"], res[1]["param"])
        assert not np.array_equal(res[0]["param"], res[0]["step1"][2])
    elif mode == "late_oom":
        for r in res:
            assert r["step2_none"] and r["updates"] == 1  # dropped together, no deadlock on mismatched late sets
        np.testing.assert_array_equal(res[0]["param"], res[1]["param"])
    else:
        for r in res:  # every rank sees every rank's norm in the all-reduced vector: all of them stop
            assert r["raised"] is not None and "gradients are inconsistent between workers" in r["raised"] and "rank   1" in r["raised"]


def test_grad_slot_hands_a_flat_buffer_slot_out_once_per_epoch():
    """optim.grad_slot (backward kernels write a weight gradient straight into the flat gradient buffer): the slot is adopted by
    autograd as p.grad (no copy in gather_grads), handed out once per zero_grad epoch and only while p.grad is None — a later
    micro-batch accumulates into it, a second use of a shared weight in the same pass is added to it — and never while it is
    travelling in a bucket all-reduce (`_cst_slot_frozen`, distributed.py)."""
    lin = torch.nn.Linear(8, 4)
    buf = optim.FlatParamBuffers(lin.parameters())
    w = lin.weight
    handed = []

    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, dy):
            x, w_ = ctx.saved_tensors
            assert w_ is lin.weight  # the Parameter itself comes back from saved_tensors: its attributes are there
            out = optim.grad_slot(w_)
            handed.append(out is not None)
            g = dy.t() @ x
            if out is not None:
                out.copy_(g)
                g = out
            return dy @ w_, g

    x = torch.randn(3, 8)
    want = torch.ones(3, 4).t() @ x
    buf.zero_grad()
    F.apply(x, w).sum().backward()
    assert handed == [True] and w.grad.data_ptr() == buf.grad_views[0].data_ptr() and torch.allclose(w.grad, want)
    F.apply(x, w).sum().backward()  # accumulation into an existing gradient: not handed out, autograd adds into the slot
    assert handed == [True, False] and torch.allclose(w.grad, 2 * want) and w.grad.data_ptr() == buf.grad_views[0].data_ptr()
    buf.zero_grad()
    handed.clear()
    (F.apply(x, w).sum() + F.apply(2 * x, w).sum()).backward()  # a shared weight: two gradients in one pass, one slot
    assert sorted(handed) == [False, True] and torch.allclose(w.grad, 3 * want)
    buf.gather_grads()
    assert torch.allclose(buf.grad_views[0], 3 * want)
    buf.zero_grad()
    handed.clear()
    w._cst_slot_frozen = True  # what the reducer sets on members of a bucket that left without their gradient
    F.apply(x, w).sum().backward()
    w._cst_slot_frozen = False
    assert handed == [False] and w.grad.data_ptr() != buf.grad_views[0].data_ptr() and torch.allclose(w.grad, want)


def _odd_order_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", CST_DDP_FORCE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    torch.manual_seed(0)
    layers = [torch.nn.Linear(6, 6) for _ in range(6)]
    # the flat buffer (and with it the buckets and their hooked parameters) is laid out in an order that has nothing to do with
    # the order in which the gradients arrive
    order = [3, 0, 5, 1, 4, 2]
    params = [p for i in order for p in layers[i].parameters()]
    buf = optim.FlatParamBuffers(params)
    red = distributed.BucketedGradAllReduce(buf.params, buf.offsets, buf.flat_grad, None, bucket_cap_mb=0.0002, gather=buf.gather_grads)
    assert red.active and len(red.buckets) >= 3 and len(red._hooks) == len(buf.params)
    x = torch.randn(4, 6)
    outs = []
    for step in range(2):
        buf.zero_grad()
        h = x
        for l in layers:
            h = torch.tanh(l(h))
        h.pow(2).sum().backward()
        early = red._next
        red.finish()
        outs.append((early, buf.flat_grad.clone()))
    h = x
    for l in layers:
        h = torch.tanh(l(h))
    ref = torch.autograd.grad(h.pow(2).sum(), buf.params)
    err = max(float((buf.grad_views[i] - ref[i]).abs().max()) for i in range(len(buf.params)))
    q.put((outs[0][0], outs[1][0], len(red.buckets), bool(torch.equal(outs[0][1], outs[1][1])), err))
    dist.destroy_process_group()


def test_reducer_with_an_unexpected_arrival_order():
    """Buckets leave strictly in storage order; when the gradients do not arrive in reverse storage order a bucket waits for its
    last member's hook (or for finish()).  Whatever the order: every bucket is reduced exactly once and the gradients are complete."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_odd_order_worker, args=(port, q))
    p.start()
    early1, early2, nb, same, err = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0
    assert 0 <= early1 <= nb and early1 == early2 and same and err < 1e-6


# ---- accumulated updates (update_freq > 1) with weight gradients written straight into the flat buffer (optim.grad_slot) --------------
class _SlotLinear(torch.autograd.Function):
    """What functional._linear_backward does on the GPU: the weight gradient goes into the parameter's slot of the flat gradient
    buffer when optim.grad_slot hands it out (first micro-batch), otherwise it is an ordinary tensor autograd adds in place."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        g = dy.t() @ x
        out = optim.grad_slot(w)
        if out is not None:
            out.copy_(g)
            g = out
        return dy @ w, g


class _SlotNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(12, 12) * 0.3) for _ in range(6)])  # bias-free layers

    def forward(self, x):
        for w in self.w:
            x = torch.tanh(_SlotLinear.apply(x, w))
        return x


def _accum_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = _SlotNet()
    buf = optim.FlatParamBuffers(net.parameters())
    model = distributed.DistributedFairseqModel(Namespace(bucket_cap_mb=0.0006), net, buf)  # one 144-element weight per bucket
    red = model.reducer
    assert len(red.buckets) == 6
    g = torch.Generator().manual_seed(100 + rank)
    xs = [torch.randn(5, 12, generator=g) for _ in range(2)]
    launched_at = []  # (bucket, had every member's hook fired in the reducing pass?)
    orig = red._launch

    def spy(b):
        launched_at.append((b, all(i in red._fired for i in red.buckets[b]["members"])))
        orig(b)

    red._launch = spy
    buf.zero_grad()
    with model.no_sync():
        net(xs[0]).pow(2).sum().backward()
    assert all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(buf.params, buf.grad_views))  # slot-written
    net(xs[1]).pow(2).sum().backward()  # the reducing pass: in-place adds into the slots, which share ONE version counter
    early = red._next
    model.all_reduce()
    q.put((rank, buf.flat_grad.clone().numpy(), [x.numpy() for x in xs], launched_at, early, list(red.late_params)))
    dist.destroy_process_group()


def test_accumulated_update_with_slot_written_gradients():
    """Advisor, round 4 (high): with update_freq > 1 the slot-written gradients are views of one flat tensor and share its version
    counter; a reducer that reads arrival off `p.grad._version` sees every weight arrive with the first in-place add of the last
    micro-batch and lets buckets go before their members' gradients exist.  Arrival is a per-parameter hook record now: every bucket
    leaves only after all of its members fired in the reducing pass, nothing is flagged late, and the result is the mean over the
    ranks of the sum over both micro-batches."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])
    torch.manual_seed(0)
    net = _SlotNet()
    want = [torch.zeros_like(p) for p in net.parameters()]
    for r in range(world):
        for x in res[r][2]:
            gs = torch.autograd.grad(net(torch.from_numpy(x)).pow(2).sum(), list(net.parameters()))
            want = [w + g / world for w, g in zip(want, gs)]
    np.testing.assert_allclose(res[0][1], torch.cat([w.reshape(-1) for w in want]).numpy(), rtol=1e-5, atol=1e-6)
    for r in res:
        assert len(r[3]) == 6 and all(ok for _, ok in r[3]), r[3]   # no bucket left before its member's gradient of the last pass
        assert r[4] == 6 and r[5] == []                              # all of them under backward; nobody "late"


def test_launch_ranks_starts_one_process_per_rank_and_propagates_failure(tmp_path, capfd):
    """distributed.launch_ranks (what `python bench.py --gpus N` / `fairseq_train.py` do when started without a launcher;
    fairseq/distributed_utils.py:286-303): N children with the torchrun environment that can form a process group; rank 0 owns stdout;
    a failing rank stops the job with its exit code."""
    import subprocess
    import sys
    prog = tmp_path / "rank.py"
    prog.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "if len(sys.argv) > 1 and sys.argv[1] == str(r): sys.exit(5)\n"
        "dist.init_process_group('gloo', rank=r, world_size=w)\n"
        "t = torch.tensor([float(r + 1)]); dist.all_reduce(t)\n"
        "print('rank %d of %d sum %d' % (r, w, int(t.item())), flush=True)\n"
        "dist.destroy_process_group()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    assert distributed.needs_self_launch(3) and not distributed.needs_self_launch(1)
    r = subprocess.run([sys.executable, "-c", "import sys, importlib; sys.path.insert(0, %r); d = importlib.import_module('chimera-st_amd.distributed'); "
                        "sys.exit(d.launch_ranks(3, [sys.executable, %r]))" % (ROOT, str(prog))], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    assert [l for l in r.stdout.splitlines() if l.startswith("rank")] == ["rank 0 of 3 sum 6"]  # rank 0 alone writes to stdout
    assert "rank 1 of 3 sum 6" in r.stderr and "rank 2 of 3 sum 6" in r.stderr
    r = subprocess.run([sys.executable, "-c", "import sys, importlib; sys.path.insert(0, %r); d = importlib.import_module('chimera-st_amd.distributed'); "
                        "sys.exit(d.launch_ranks(3, [sys.executable, %r, '2']))" % (ROOT, str(prog))], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 5 and "rank 2 exited with code 5" in r.stderr


# ---- ZeRO-1: optimizer state sharded behind reduce-scattered gradient buckets (--zero-sharding os) ---------------------------------
def _torch_adam_trainer(zero):
    """A CPU Trainer whose two optimizer KERNELS are torch arithmetic (real Adam with both moments, so that a mis-sliced shard of the
    state shows); everything around them — buckets, reduce-scatter, spans, the all-gather of the parameters — is the product code."""
    Trainer = import_module("chimera-st_amd.trainer").Trainer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 24), torch.nn.ReLU(), torch.nn.Linear(24, 4))
    net.upgrade_state_dict = lambda sd: None  # (the fairseq model surface Trainer.load_checkpoint calls)
    args = Namespace(bf16=False, lr=[1e-2], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.0, clip_norm=0.0, seed=1,
                     bucket_cap_mb=0.0008, zero_sharding="os" if zero else "none")
    tr = Trainer(args, tasks.FairseqTask(args), net, _ToyCriterion(), device="cpu")
    opt = tr.optimizer

    def sumsq_span(grad, out):
        out.add_(grad.double().pow(2).sum().float())

    def adam_span(master, m, v, grad, param):
        b1, b2, t = opt.betas[0], opt.betas[1], opt.num_updates
        g = grad.float() * opt._scale
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(opt.eps)
        master.addcdiv_(m, denom, value=-opt.lr / (1 - b1 ** t))
        param.copy_(master)

    opt._sumsq_span, opt._adam_span = sumsq_span, adam_span
    return tr


def _zero_worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {"rank": rank}
    runs = {}
    for zero in (True, False):
        tr = _torch_adam_trainer(zero)
        g = torch.Generator().manual_seed(100 + rank)
        gn = []
        for _ in range(3):
            o = tr.train_step([{"x": torch.randn(5, 6, generator=g)}])
            gn.append(o["gnorm"])
        runs[zero] = (tr, gn)
    tz, tu = runs[True][0], runs[False][0]
    out["zero_on"] = tz.zero and not tu.zero and tz.model.reducer.collective == "rs"
    out["spans"] = list(tz.optimizer.segments)
    out["state_numel"] = (tz.optimizer.exp_avg.numel(), tz.buffers.total, tu.optimizer.exp_avg.numel(), tu.buffers.total)
    # the sharded buffers are laid out with align = 8 * world: compare parameter by parameter
    out["params_equal"] = all(torch.equal(a, b) for a, b in zip(tz.buffers.params, tu.buffers.params))
    out["gnorm"] = (runs[True][1], runs[False][1])
    sz, su = tz.optimizer.fairseq_state_dict(), tu.optimizer.fairseq_state_dict()   # (a collective for the sharded one: both ranks call it)
    out["state_equal"] = all(torch.equal(sz["state"][i][k], su["state"][i][k]) for i in su["state"] for k in ("exp_avg", "exp_avg_sq"))
    # checkpoint round trip through the reference's format: rank 0 writes, every rank loads into a fresh sharded trainer and continues
    path = os.path.join(tmp, "ckpt.pt")
    tz.save_checkpoint(path)
    dist.barrier()
    t2 = _torch_adam_trainer(True)
    t2.load_checkpoint(path)
    g2 = torch.Generator().manual_seed(500 + rank)
    batch = {"x": torch.randn(5, 6, generator=g2)}
    t2.train_step([batch]); tz.train_step([batch]); tu.train_step([batch])
    out["resume_equal"] = all(torch.equal(a, b) for a, b in zip(t2.buffers.params, tz.buffers.params))
    out["still_equal"] = all(torch.equal(a, b) for a, b in zip(tz.buffers.params, tu.buffers.params))
    out["param"] = torch.cat([p.detach().reshape(-1) for p in tz.buffers.params]).numpy()
    q.put(out)
    dist.destroy_process_group()


def test_zero_sharded_optimizer_equals_the_unsharded_update_gloo_world2(tmp_path):
    """--zero-sharding os (fairseq/trainer.py:241-252): reduce-scattered buckets, Adam on this rank's spans with 1 / world of the
    state, parameters all-gathered — against the all-reduce route on the same batches: every parameter and both moments bit-equal
    after three updates (the update is element-wise; clip-norm 0 here, so the scale is the same number), the gradient norm equal to
    rounding (a sum of the ranks' partial sums), replicas identical, and a checkpoint written from the shards resumes to the same
    next update."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_zero_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert r["zero_on"] and r["params_equal"] and r["state_equal"] and r["resume_equal"] and r["still_equal"], {k: v for k, v in r.items() if k != "param"}
        nz, tz, nu, tu = r["state_numel"]
        assert nz * 2 == tz and nu == tu and tz >= tu, r["state_numel"]        # half of the optimizer state per rank
        for a, b in zip(*r["gnorm"]):
            assert a == pytest.approx(b, rel=1e-6)
    spans = res[0]["spans"] + res[1]["spans"]
    assert sum(hi - lo for lo, hi in spans) == res[0]["state_numel"][1]        # the two ranks' spans tile the flat buffer
    assert len(set(spans)) == len(spans) and all(lo % 8 == 0 and hi % 8 == 0 for lo, hi in spans)
    np.testing.assert_array_equal(res[0]["param"], res[1]["param"])


# ---- world 8 (the node size BASELINE.json asks for) over gloo: every route of the gradient exchange against ONE process ----------------
class _ParallelLinears(torch.nn.Module):
    """Gradients that do NOT depend on the parameters (each Linear reads the input and is summed): with integer-valued batches every
    gradient element is an exact small integer in fp32, so a sum over ranks is exact in ANY association — which is what lets eight
    ranks be compared with one process BIT for bit although gloo's ring adds in its own order.  Parameter sizes are chosen so that the
    buckets are uneven (280 / 168 / 35 / 21 elements against a 131-element cap) and the last one is a sliver."""

    def __init__(self):
        super().__init__()
        self.a, self.b, self.c, self.d = torch.nn.Linear(6, 40), torch.nn.Linear(6, 24), torch.nn.Linear(6, 5), torch.nn.Linear(6, 3)

    def forward(self, x, use_b=True):
        y = self.a(x).sum() + 3 * self.c(x).sum() + 5 * self.d(x).sum()
        return y + 2 * self.b(x).sum() if use_b else y

    def upgrade_state_dict(self, sd):
        pass


class _SumCriterion(torch.nn.Module):
    skip_b = False  # the update where `b` is dropped on every rank (what wav2vec2's layerdrop does: same np.random seed everywhere)

    def forward(self, model, sample):
        if self.skip_b:
            import_module("chimera-st_amd.distributed").notify_unused_parameters(model.b.parameters())
        loss = model(sample["x"], use_b=not self.skip_b)
        n = sample["x"].shape[0]
        return loss, n, {"loss": loss.detach(), "sample_size": n, "ntokens": n, "nsentences": n}

    @staticmethod
    def logging_keys():
        return ("loss", "sample_size", "ntokens", "nsentences")


def _world8_batches(update):
    """Per update the eight ranks' integer batches (4 / 2 / 4 ... rows: sample_size 28 or 24, never a power of two); in update 2 rank 5's
    shard of the epoch has run out (ShardedIterator pads with an empty batch: data/iterators.py:470)."""
    g = torch.Generator().manual_seed(900 + update)
    out = []
    for r in range(8):
        x = torch.randint(-3, 4, (2 if r == 1 else 4, 6), generator=g).float()
        out.append({} if (update == 2 and r == 5) else {"x": x})
    return out


def _world8_trainer(zero, collective="allreduce"):
    Trainer = import_module("chimera-st_amd.trainer").Trainer
    torch.manual_seed(0)
    net = _ParallelLinears()
    args = Namespace(bf16=False, lr=[1e-2], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.0, clip_norm=0.0, seed=1,
                     bucket_cap_mb=0.0005, zero_sharding="os" if zero else "none")
    os.environ["CST_DDP_COLLECTIVE"] = collective
    tr = Trainer(args, tasks.FairseqTask(args), net, _SumCriterion(), device="cpu")
    opt = tr.optimizer

    def sumsq_span(grad, out):
        out.add_(grad.double().pow(2).sum().float())

    def adam_span(master, m, v, grad, param):
        b1, b2, t = opt.betas[0], opt.betas[1], opt.num_updates
        g = grad.float() * opt._scale
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(opt.eps)
        master.addcdiv_(m, denom, value=-opt.lr / (1 - b1 ** t))
        param.copy_(master)

    opt._sumsq_span, opt._adam_span = sumsq_span, adam_span
    return tr


def _world8_run(tr, batches_of, n_updates=4):
    """batches_of(update) -> the list of micro-batches THIS trainer sees in that update."""
    gn = []
    for u in range(n_updates):
        tr.criterion.skip_b = u == 1
        o = tr.train_step(batches_of(u))
        gn.append((o["gnorm"], o["sample_size"]))
    return gn


def _world8_worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {"rank": rank}
    for route, (zero, coll) in {"allreduce": (False, "allreduce"), "rs_ag": (False, "rs_ag"), "zero": (True, "allreduce")}.items():
        tr = _world8_trainer(zero, coll)
        assert tr.ddp and tr.world == 8 and tr.zero == zero and len(tr.model.reducer.buckets) >= 4
        assert tr.model.reducer.collective == ("rs" if zero else coll)
        gn = _world8_run(tr, lambda u: [_world8_batches(u)[rank]])
        st = tr.optimizer.fairseq_state_dict()  # (a collective on the sharded route: every rank calls it)
        out[route] = dict(gnorm=gn, params=[p.detach().clone().numpy() for p in tr.buffers.params],
                          m=[st["state"][i]["exp_avg"].numpy() for i in sorted(st["state"])],
                          v=[st["state"][i]["exp_avg_sq"].numpy() for i in sorted(st["state"])],
                          buckets=[(b["lo"], b["hi"]) for b in tr.model.reducer.buckets])
        if zero:
            out["spans"] = list(tr.optimizer.segments)
            out["state_numel"] = (tr.optimizer.exp_avg.numel(), tr.buffers.total)
            path = os.path.join(tmp, "ckpt8.pt")
            tr.save_checkpoint(path)  # assembled from the eight ranks' shards, written by rank 0
            dist.barrier()
            t2 = _world8_trainer(True)
            t2.load_checkpoint(path)
            t2.criterion.skip_b = tr.criterion.skip_b = False
            t2.train_step([_world8_batches(7)[rank]]); tr.train_step([_world8_batches(7)[rank]])
            out["resume_equal"] = all(torch.equal(a, b) for a, b in zip(t2.buffers.params, tr.buffers.params))
    q.put(out)
    dist.destroy_process_group()


def test_world8_every_route_equals_one_process_bit_for_bit(tmp_path):
    """Eight gloo ranks — the node size of BASELINE configs 3 / 4, which no hardware has run yet — through the three routes of the
    gradient exchange (bucketed all-reduce, reduce-scatter + all-gather per bucket, --zero-sharding os) on uneven buckets, with an
    update whose layer is skipped on every rank (its bucket travels as zeros: legacy_distributed_data_parallel.py:155-156) and an
    update where one rank's shard has run out (dummy batch, loss zeroed: trainer.py:469-477): parameters and both Adam moments after
    four updates are BIT-equal on every rank, on every route, and equal to ONE process accumulating the same batches
    (trainer.py:393-394 sharding <-> update_freq); the sharded route keeps 1 / 8 of the state per rank, its spans tile the buffer,
    and the checkpoint assembled from eight shards loads into one process and into eight."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t["rank"])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # ONE process, no process group: the same batches as accumulated micro-batches of one update (the empty one does not exist there)
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    one = _world8_trainer(False)
    assert not one.ddp
    gn1 = _world8_run(one, lambda u: [b for b in _world8_batches(u) if b])
    st1 = one.optimizer.fairseq_state_dict()
    assert [s for _, s in gn1] == [30.0, 30.0, 26.0, 30.0]
    for r in res:
        for route in ("allreduce", "rs_ag", "zero"):
            got = r[route]
            for a, b in zip(got["params"], one.buffers.params):
                np.testing.assert_array_equal(a, b.detach().numpy(), err_msg="%s rank %d" % (route, r["rank"]))
            for i, key in enumerate(sorted(st1["state"])):
                np.testing.assert_array_equal(got["m"][i], st1["state"][key]["exp_avg"].numpy())
                np.testing.assert_array_equal(got["v"][i], st1["state"][key]["exp_avg_sq"].numpy())
            for (ga, sa), (gb, sb) in zip(got["gnorm"], gn1):
                assert sa == sb and ga == pytest.approx(gb, rel=1e-6)
        assert r["resume_equal"]
        n, total = r["state_numel"]
        assert n * 8 == total  # an eighth of the optimizer state per rank
    sizes = sorted(hi - lo for lo, hi in res[0]["allreduce"]["buckets"])
    assert len(sizes) >= 4 and sizes[0] < sizes[-1] // 2, sizes  # uneven buckets, a sliver among them
    spans = [s for r in res for s in r["spans"]]
    assert sum(hi - lo for lo, hi in spans) == res[0]["state_numel"][1] and len(set(spans)) == len(spans)
    assert all(lo % 8 == 0 and hi % 8 == 0 for lo, hi in spans)
    # the checkpoint written from eight shards, read by ONE process: both moments are the single-process ones
    fresh = _world8_trainer(False)
    fresh.load_checkpoint(os.path.join(str(tmp_path), "ckpt8.pt"))
    st2 = fresh.optimizer.fairseq_state_dict()
    for key in st1["state"]:
        assert torch.equal(st2["state"][key]["exp_avg"], st1["state"][key]["exp_avg"])
        assert torch.equal(st2["state"][key]["exp_avg_sq"], st1["state"][key]["exp_avg_sq"])
    assert fresh.num_updates == 4 and all(torch.equal(a, b) for a, b in zip(fresh.buffers.params, one.buffers.params))


def test_roctx_phase_ranges_follow_the_reference_scopes(tmp_path):
    import subprocess
    import sys
    """CST_ROCTX=1: one update emits the reference's record_function scopes (fairseq_cli/train.py:225-227, fairseq_task.py:439-444,
    trainer.py:601-627) as roctx push / pop pairs, properly nested, in the reference's order; without the variable nothing is
    loaded and `scope()` is the shared no-op.  (Run in a child process: the switch is read at import.)"""
    prog = tmp_path / "ranges.py"
    prog.write_text('''
import sys, importlib
from argparse import Namespace
sys.path.insert(0, %r)
import torch
P = importlib.import_module("chimera-st_amd.profiling")
seen = []
if P.enabled():
    class Fake:
        def roctxRangePushA(self, name): seen.append("+" + name.decode()); return 0
        def roctxRangePop(self): seen.append("-"); return 0
    P._lib = Fake()
sys.path.insert(0, %r)
import test_host_cpu as T
tr = T._cpu_trainer(0)
tr.train_step([{"x": torch.randn(5, 6)}])
tr.train_step([{"x": torch.randn(5, 6)}])
print("RANGES " + " ".join(seen))
print("NULL", P.scope("x") is P.scope("y"))
''' % (ROOT, os.path.join(ROOT, "tests")))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "CST_ROCTX")}
    r = subprocess.run([sys.executable, str(prog)], capture_output=True, text=True, timeout=300, env=dict(env, CST_ROCTX="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RANGES")][0].split()[1:]
    # (_cpu_trainer swaps optimizer.step for plain SGD, so "multiply-grads" — which lives inside the real step — is not in this list)
    one = lambda n: ["+train_step-%d" % n, "+forward", "-", "+backward", "-", "+reduce-grads", "-", "+clip-grads", "-", "+optimizer", "-", "-"]
    assert line == one(0) + one(1), line
    r = subprocess.run([sys.executable, str(prog)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "RANGES" in r.stdout and [l for l in r.stdout.splitlines() if l.startswith("RANGES")][0].split()[1:] == []
    assert "NULL True" in r.stdout
    # the real library loads and takes a push / pop (no tracer attached: the calls are no-ops inside roctx)
    r = subprocess.run([sys.executable, "-c", "import sys, importlib; sys.path.insert(0, %r); P = importlib.import_module('chimera-st_amd.profiling');\n"
                        "with P.scope('forward'): pass\nprint('ok', P._lib is not None)" % ROOT], capture_output=True, text=True, timeout=300,
                       env=dict(env, CST_ROCTX="1"))
    assert r.returncode == 0 and "ok True" in r.stdout, r.stderr[-1500:]


def test_host_thread_budget_follows_the_cgroup_quota(tmp_path, monkeypatch):
    """hostcfg: torch's intra-op pool is cut to what the container may run (cgroup v2 cpu.max / v1 cfs quota, inside the affinity
    mask), shared between the ranks of a node, never raised; CST_HOST_THREADS overrides (0 = hands off)."""
    import importlib
    import os
    import torch
    H = importlib.import_module("chimera-st_amd.hostcfg")
    aff = len(os.sched_getaffinity(0))
    v2 = tmp_path / "v2"; v2.mkdir()
    (v2 / "cpu.max").write_text("250000 100000\n")
    assert H.usable_cpus(str(v2)) == min(aff, 3)
    (v2 / "cpu.max").write_text("max 100000\n")
    assert H.usable_cpus(str(v2)) == aff
    v1 = tmp_path / "v1"; (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("150000\n"); (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert H.usable_cpus(str(v1)) == min(aff, 2)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert H.usable_cpus(str(v1)) == aff and H.usable_cpus(str(tmp_path / "none")) == aff
    before = torch.get_num_threads()
    try:
        monkeypatch.delenv("CST_HOST_THREADS", raising=False)
        monkeypatch.setattr(H, "usable_cpus", lambda root="/sys/fs/cgroup": 4)
        torch.set_num_threads(max(before, 2))
        assert H.limit_host_threads(2) == 2 and torch.get_num_threads() == 2     # 4 usable CPUs, two ranks on the node
        assert H.limit_host_threads(1) == 2                                        # never raised
        assert H.limit_host_threads(64) == 1
        monkeypatch.setenv("CST_HOST_THREADS", "0")
        torch.set_num_threads(3)
        assert H.limit_host_threads(64) == 3                                       # hands off
        monkeypatch.setenv("CST_HOST_THREADS", "2")
        assert H.limit_host_threads() == 2
    finally:
        torch.set_num_threads(before)
