"""No-GPU checks of the drop-in boundary: the C-ABI shared library loads and exports every symbol include/cst.h
declares, the ctypes binding agrees with the header, and the product fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

from conftest import ROOT, load_pkg

HEADER = os.path.join(ROOT, "include", "cst.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cst_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    lib = load_pkg().lib
    if not os.path.exists(lib.LIB_PATH):
        ge.build()
    return lib


def test_header_declares_expected_entry_points():
    fns = header_functions()
    for name in ("cst_gemm", "cst_attn_fwd", "cst_attn_bwd", "cst_layernorm_fwd", "cst_layernorm_bwd", "cst_conv0_gn_gelu_fwd",
                 "cst_conv0_gn_gelu_bwd", "cst_ls_ce_fwd", "cst_ls_ce_bwd", "cst_adam_step", "cst_sumsq", "cst_last_error"):
        assert name in fns


def test_library_exports_every_declared_symbol(built):
    lib = built.load()
    out = subprocess.check_output(["nm", "-D", "--defined-only", built.LIB_PATH], text=True)
    exported = set(re.findall(r" T (cst_[a-z0-9_]+)", out))
    declared = set(header_functions())
    assert declared <= exported, "declared but not exported: %s" % sorted(declared - exported)
    assert exported <= declared, "exported but not declared in include/cst.h: %s" % sorted(exported - declared)
    bound = {name for name, _, _ in built.SYMBOLS}
    assert bound == declared, "ctypes binding out of sync with the header: %s" % sorted(bound ^ declared)
    assert lib.cst_version() >= 1


def test_descriptor_structs_match_header_field_order(built):
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    bodies = {name: body for body, name in re.findall(r"typedef struct \{([^{}]*)\} (\w+);", src)}
    for struct, cls in (("cst_gemm_desc", built.GemmDesc), ("cst_attn_desc", built.AttnDesc), ("cst_beam_desc", built.BeamDesc)):
        body = bodies[struct]
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            first, *rest = decl.split(",")
            names.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", first)[-1])
            names += [re.findall(r"[A-Za-z_][A-Za-z0-9_]*", r)[-1] for r in rest]
        assert names == [f[0] for f in cls._fields_], struct


def test_error_reporting_without_gpu(built):
    """A bad descriptor is rejected by argument validation (before any HIP call) with a message, never a crash."""
    lib = built.load()
    d = built.GemmDesc()
    rc = lib.cst_gemm(ctypes.byref(d), None)
    assert rc == -1
    assert b"null operand" in lib.cst_last_error()
    a = built.AttnDesc()
    a.dtype = 7
    assert lib.cst_attn_fwd(ctypes.byref(a), None) == -1
    assert b"bad dtype" in lib.cst_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_ops_fail_loudly_on_cpu_tensors(built):
    CF = __import__("importlib").import_module("chimera-st_amd.functional")
    x = torch.randn(4, 16)
    w = torch.randn(8, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback|not on the GPU|No HIP GPUs"):
        CF.linear(x, w)
    with pytest.raises(RuntimeError, match="no CPU fallback|not on the GPU|No HIP GPUs"):
        CF.layer_norm(x, torch.ones(16), torch.zeros(16))
