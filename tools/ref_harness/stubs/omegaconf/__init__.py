"""Minimal omegaconf stand-in so `import fairseq` succeeds in the golden-vector
harness (tools/ref_harness). Harness-only: never imported by the product or tests."""
from . import _utils  # noqa: F401

MISSING = "???"


def II(x):
    return "${" + str(x) + "}"


class DictConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class ListConfig(list):
    pass


class _OmegaConf:
    @staticmethod
    def create(x=None):
        return DictConfig(x or {})

    @staticmethod
    def set_struct(cfg, flag):
        return None

    @staticmethod
    def is_config(x):
        return isinstance(x, DictConfig)

    @staticmethod
    def to_container(x, resolve=False):
        return dict(x)


OmegaConf = _OmegaConf()


class open_dict:
    def __init__(self, cfg):
        self.cfg = cfg

    def __enter__(self):
        return self.cfg

    def __exit__(self, *a):
        return False
