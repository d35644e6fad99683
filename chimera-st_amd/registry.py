"""The reference's plugin API for this path: fairseq's registries
(fairseq/models/__init__.py:61-164, fairseq/tasks/__init__.py:29-80, fairseq/registry.py:16-84).
Same decorator names, same lookup tables, same `build_*` entry points, so the model / task /
criterion names used by chimera/scripts/*.sh resolve to the MI355X-native classes."""
import argparse

MODEL_REGISTRY = {}
ARCH_MODEL_REGISTRY = {}
ARCH_MODEL_INV_REGISTRY = {}
ARCH_CONFIG_REGISTRY = {}
TASK_REGISTRY = {}
CRITERION_REGISTRY = {}


def register_model(name):
    def wrap(cls):
        if name in MODEL_REGISTRY:
            raise ValueError("Cannot register duplicate model ({})".format(name))
        MODEL_REGISTRY[name] = cls
        return cls

    return wrap


def register_model_architecture(model_name, arch_name):
    def wrap(fn):
        if model_name not in MODEL_REGISTRY:
            raise ValueError("Cannot register model architecture for unknown model type ({})".format(model_name))
        if arch_name in ARCH_MODEL_REGISTRY:
            raise ValueError("Cannot register duplicate model architecture ({})".format(arch_name))
        if not callable(fn):
            raise ValueError("Model architecture must be callable ({})".format(arch_name))
        ARCH_MODEL_REGISTRY[arch_name] = MODEL_REGISTRY[model_name]
        ARCH_MODEL_INV_REGISTRY.setdefault(model_name, []).append(arch_name)
        ARCH_CONFIG_REGISTRY[arch_name] = fn
        return fn

    return wrap


def register_task(name):
    def wrap(cls):
        if name in TASK_REGISTRY:
            raise ValueError("Cannot register duplicate task ({})".format(name))
        TASK_REGISTRY[name] = cls
        return cls

    return wrap


def register_criterion(name):
    def wrap(cls):
        if name in CRITERION_REGISTRY:
            raise ValueError("Cannot register duplicate criterion ({})".format(name))
        CRITERION_REGISTRY[name] = cls
        return cls

    return wrap


def build_model(args, task):
    """fairseq.models.build_model (models/__init__.py:49-58): arch -> class.build_model(args, task)."""
    return ARCH_MODEL_REGISTRY[args.arch].build_model(args, task)


def setup_task(args, **kwargs):
    return TASK_REGISTRY[args.task].setup_task(args, **kwargs)


def build_criterion(args, task):
    return CRITERION_REGISTRY[args.criterion].build_criterion(args, task)


def parse_args_and_arch(argv, extra_defaults=None):
    """Two-pass argparse like fairseq/options.py:77-209: discover --arch/--task/--criterion, add their
    add_args, re-parse, then let the arch function fill defaults (ARCH_CONFIG_REGISTRY[arch](args))."""
    base = argparse.ArgumentParser(add_help=False, allow_abbrev=False)
    base.add_argument("--arch", "-a", required=True)
    base.add_argument("--task", default="speech_to_text")
    base.add_argument("--criterion", default="label_smoothed_cross_entropy")
    known, _ = base.parse_known_args(argv)
    parser = argparse.ArgumentParser(allow_abbrev=False, parents=[base])
    add_common_args(parser)
    # options.py:137-145: model-specific flags live in a group with argument_default=SUPPRESS so that the arch function's
    # getattr(args, name, default) sees only what the user actually passed
    model_group = parser.add_argument_group("Model-specific configuration", argument_default=argparse.SUPPRESS)
    ARCH_MODEL_REGISTRY[known.arch].add_args(model_group)
    TASK_REGISTRY[known.task].add_args(parser)  # (the positional `data` belongs to the task, tasks/speech_to_text.py:27)
    CRITERION_REGISTRY[known.criterion].add_args(parser)
    if not any(a.dest == "data" for a in parser._actions):
        parser.add_argument("data", nargs="?", default=None)
    if extra_defaults:
        parser.set_defaults(**extra_defaults)
    args = parser.parse_args(argv)
    ARCH_CONFIG_REGISTRY[args.arch](args)
    return args


def add_common_args(parser):
    """The trainer/optimizer/distributed flags chimera/scripts/train-en2any-ST.sh passes."""
    g = parser
    g.add_argument("--seed", type=int, default=1)
    g.add_argument("--fp16", action="store_true")
    g.add_argument("--bf16", action="store_true")
    g.add_argument("--memory-efficient-fp16", action="store_true")
    g.add_argument("--max-tokens", type=int, default=None)
    g.add_argument("--max-sentences", "--batch-size", type=int, default=None, dest="batch_size")
    g.add_argument("--update-freq", type=int, nargs="+", default=[1])
    g.add_argument("--max-update", type=int, default=0)
    g.add_argument("--max-epoch", type=int, default=0)
    g.add_argument("--optimizer", default="adam")
    g.add_argument("--adam-betas", default="(0.9, 0.999)")
    g.add_argument("--adam-eps", type=float, default=1e-8)
    g.add_argument("--weight-decay", type=float, default=0.0)
    g.add_argument("--lr", type=float, nargs="+", default=[0.25])
    g.add_argument("--lr-scheduler", default="inverse_sqrt")
    g.add_argument("--warmup-updates", type=int, default=4000)
    g.add_argument("--warmup-init-lr", type=float, default=-1)
    g.add_argument("--clip-norm", type=float, default=0.0)
    g.add_argument("--ddp-backend", default="no_c10d")
    g.add_argument("--distributed-world-size", type=int, default=None, help="ranks to start when launched plainly (default: one per visible GPU, options.py:310)")
    g.add_argument("--bucket-cap-mb", type=int, default=25)
    g.add_argument("--zero-sharding", default="none", choices=["none", "os"], help="os: optimizer state sharded over the data-parallel "
                   "ranks (fairseq/trainer.py:241-252; here: reduce-scattered gradient buckets, per-shard Adam, all-gathered parameters)")
    g.add_argument("--ddp-reserve-cus", type=int, default=None, help="CUs the persistent GEMMs leave to the gradient all-reduce while "
                   "buckets are in flight under the backward pass (distributed.BucketedGradAllReduce.reserve_cus)")
    g.add_argument("--nonfinite-tolerance", type=int, default=20, help="consecutive updates with NaN / Inf gradients that are skipped "
                   "before FloatingPointError (the reference's DynamicLossScaler gives up after ~20 halvings)")
    g.add_argument("--num-workers", type=int, default=1)
    g.add_argument("--save-dir", default="checkpoints")
    g.add_argument("--tensorboard-logdir", default=None)
    g.add_argument("--log-format", default=None)
    g.add_argument("--log-interval", type=int, default=100)
    g.add_argument("--skip-invalid-size-inputs-valid-test", action="store_true")
    g.add_argument("--train-subset", default="train")
    g.add_argument("--valid-subset", default="valid")
    g.add_argument("--save-interval-updates", type=int, default=0)
    g.add_argument("--keep-interval-updates", type=int, default=-1)
    g.add_argument("--keep-last-epochs", type=int, default=-1)
    g.add_argument("--keep-best-checkpoints", type=int, default=-1)
    g.add_argument("--reset-optimizer", action="store_true")
    g.add_argument("--reset-dataloader", action="store_true")
    g.add_argument("--reset-meters", action="store_true")
    g.add_argument("--sentence-avg", action="store_true")
    return parser
