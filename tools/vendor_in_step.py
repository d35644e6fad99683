#!/usr/bin/env python3
"""Prices the in-step ceiling of the GEMM class: inside ONE training update of the bench workload every bf16
GEMM launch of this library is SHADOWED by the vendor GEMM (torch.matmul -> hipBLASLt) of the same product — same operands where the
layout allows (plain strided matrices), a contiguous copy of the operand made outside the timed events where it does not (the conv
stack's overlapping-row implicit GEMMs) — and both are timed with an event pair on the launch stream.  The vendor call computes the bare
product into a scratch output (no bias / activation / residual / dropout / act' epilogue, no dead-tile skipping): it is an UPPER bound of
what a vendor-backed step could reach for that launch, since the epilogue work this library fuses would be extra launches there.  The
order alternates per launch (even ordinal: vendor first, odd: this library first) so that neither side always finds the operands warm.

Tools only: nothing here is on the product path.  usage (GPU box): python tools/vendor_in_step.py > profiles/r05_vendor_in_step.txt"""
import argparse, collections, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
K = importlib.import_module("chimera-st_amd.kernels")
L = importlib.import_module("chimera-st_amd.lib")

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="s2t_w2v2")
ap.add_argument("--max-copy-gb", type=float, default=6.0, help="largest operand copy made for a vendor call (overlapping-row A)")
ap.add_argument("--probe-log", default=None, help="(child) synchronise around every vendor call and log TRY / OK lines: finds a twin that faults")
ap.add_argument("--skip", default=None, help="(child) file of launch tags whose vendor twin is not to be run")
ap.add_argument("--child", action="store_true")
a = ap.parse_args()
if not a.child:
    # parent (never touches the GPU): a vendor call on an unusual operand view may take the process down with a memory fault, so the
    # twins are first tried one by one in probe children (a tag left at TRY is put on the skip list), then one clean timed child runs
    import subprocess, tempfile
    tmp = tempfile.mkdtemp()
    skip, log = os.path.join(tmp, "skip.txt"), os.path.join(tmp, "probe.log")
    open(skip, "w").close()
    for attempt in range(6):
        open(log, "w").close()
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--child", "--model", a.model, "--max-copy-gb", str(a.max_copy_gb),
                              "--probe-log", log, "--skip", skip], stdout=subprocess.DEVNULL)
        lines = open(log).read().splitlines()
        if rc == 0:
            break
        tries = [l[4:] for l in lines if l.startswith("TRY ")]
        oks = set(l[3:] for l in lines if l.startswith("OK "))
        bad = [t for t in tries if t not in oks]
        sys.stderr.write("probe child exited %d; vendor twin faulted on: %s\n" % (rc, bad[-1:] or "?"))
        if not bad:
            break
        with open(skip, "a") as f:
            f.write(bad[-1] + "\n")
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "--child", "--model", a.model, "--max-copy-gb", str(a.max_copy_gb), "--skip", skip]))
SKIP = set(open(a.skip).read().splitlines()) if a.skip else set()
PROBE = open(a.probe_log, "a") if a.probe_log else None
PROBED = set()
args = argparse.Namespace(batch=32, seconds=30.0, lengths="uniform", dtype="bf16", model=a.model, dropout=0.1, layerdrop=0.0)
device = torch.device("cuda", 0)
L.load()
trainer, task, tasks, ns = bench.build(args, device)
sample = bench.make_batch(tasks, task, args, 0, device)
for _ in range(3):
    trainer.train_step([sample])
torch.cuda.synchronize()

orig_gemm = K.gemm
records = []   # (tag, flops, ours event pair, vendor event pair or None, note)
ordinal = [0]
scratch = {}


def _scratch(n, dtype):
    t = scratch.get(dtype)
    if t is None or t.numel() < n:
        t = scratch[dtype] = torch.empty(int(n * 1.1) + 1024, dtype=dtype, device=device)
    return t[:n]


def _view(T, off, rows, cols, ld, kmajor, nb, stride):
    """Operand [nb, rows(mn), cols(k)] of a cst_gemm launch as a strided torch view: k-major = row r at r * ld, k contiguous;
    mn-major = element (r, k) at k * ld + r."""
    st = (stride, ld, 1) if kmajor else (stride, 1, ld)
    return torch.as_strided(T, (nb, rows, cols), st, T.storage_offset() + off)


def shadow(A, B, C, M, N, Kd, **kw):
    nb = kw.get("batch0", 1) * kw.get("batch1", 1)
    plain_batch = kw.get("batch1", 1) == 1 and not kw.get("a_seg") and not kw.get("b_seg")
    ok = A.dtype == torch.bfloat16 and plain_batch
    tag = "M=%d N=%d K=%d b=%d %s %s%s%s%s%s%s%s" % (
        M, N, Kd, nb, "ak" if kw["a_kmajor"] else "am", "bk" if kw["b_kmajor"] else "bm",
        " bias" if kw.get("bias") is not None else "", " act" if kw.get("act", 0) else "", " aux_out" if kw.get("aux_out") is not None else "",
        " dact" if kw.get("dact", 0) else "", " resid" if kw.get("resid") is not None else "", " drop" if kw.get("drop_p", 0.0) > 0 else "")
    tag += " live" if (kw.get("m_live") is not None or kw.get("k_live") is not None or kw.get("m_len") is not None or kw.get("k_len") is not None) else ""
    vendor = None
    note = ""
    try:
        if not ok:
            raise ValueError("not bf16 / segmented / two-level batch")
        if tag in SKIP:
            raise ValueError("the vendor call faults on this operand view (probe)")
        sa, sb = kw.get("sa", (0, 0))[0], kw.get("sb", (0, 0))[0]
        Av = _view(A, kw.get("a_off", 0), M, Kd, kw["lda"], kw["a_kmajor"], nb, sa)
        Bv = _view(B, kw.get("b_off", 0), N, Kd, kw["ldb"], kw["b_kmajor"], nb, sb)
        if kw["a_kmajor"] and kw["lda"] < Kd:  # overlapping rows (implicit-GEMM conv): the vendor needs a materialised operand
            if nb * M * Kd * 2 > a.max_copy_gb * 2 ** 30:
                raise ValueError("operand copy too large")
            Av = Av.contiguous()
            note = "A copied (overlapping rows)"
        if nb > 1 and sb == 0 and Av.is_contiguous():  # one weight for every batch and a dense A: fold the batch into the rows
            Av, Bv, nbv = Av.reshape(1, nb * M, Kd), Bv[:1], 1
        else:
            nbv = nb
        out = _scratch(nb * M * N, torch.bfloat16).view(nbv, nb * M // nbv, N)
        vendor = (Av, Bv.transpose(1, 2), out)
    except (ValueError, RuntimeError) as ex:
        vendor, note = None, "no twin: " + str(ex).splitlines()[0][:80]
    first_vendor = ordinal[0] % 2 == 0
    ordinal[0] += 1
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def run_vendor():
        probing = PROBE is not None and tag not in PROBED
        if probing:
            torch.cuda.synchronize()
            PROBE.write("TRY " + tag + "\n"); PROBE.flush(); os.fsync(PROBE.fileno())
        ev[2].record()
        torch.bmm(vendor[0], vendor[1], out=vendor[2]) if vendor[0].shape[0] > 1 else torch.mm(vendor[0][0], vendor[1][0], out=vendor[2][0])
        ev[3].record()
        if probing:
            torch.cuda.synchronize()
            PROBED.add(tag)
            PROBE.write("OK " + tag + "\n"); PROBE.flush()

    if vendor is not None and first_vendor:
        run_vendor()
    ev[0].record()
    r = orig_gemm(A, B, C, M, N, Kd, **kw)
    ev[1].record()
    if vendor is not None and not first_vendor:
        run_vendor()
    records.append((tag, 2.0 * M * N * Kd * nb, (ev[0], ev[1]), (ev[2], ev[3]) if vendor is not None else None, note, first_vendor))
    return r


K.gemm = shadow
trainer.train_step([sample])   # first shadowed update: the vendor library picks (and on some shapes times) its kernels here
torch.cuda.synchronize()
if PROBE is None:
    records.clear()
    ordinal[0] = 0
    trainer.train_step([sample])   # the update that is reported
    torch.cuda.synchronize()
K.gemm = orig_gemm

agg = collections.OrderedDict()
for tag, fl, ours, vend, note, fv in records:
    e = agg.setdefault(tag, dict(n=0, fl=0.0, ours=0.0, vend=0.0, nv=0, ours_first=0.0, ours_second=0.0, vend_first=0.0, vend_second=0.0, note=note))
    t = ours[0].elapsed_time(ours[1])
    e["n"] += 1; e["fl"] += fl; e["ours"] += t
    if vend is not None:
        tv = vend[0].elapsed_time(vend[1])
        e["vend"] += tv; e["nv"] += 1
        if fv:
            e["vend_first"] += tv; e["ours_second"] += t
        else:
            e["ours_first"] += t; e["vend_second"] += tv
tot_o = sum(e["ours"] for e in agg.values())
tot_both = sum(e["ours"] for e in agg.values() if e["nv"])
tot_v = sum(e["vend"] for e in agg.values())
print("bench workload (B = 32 x <= 30 s, bf16, dropout 0.1), ONE update, every cst_gemm launch shadowed by torch.mm / torch.bmm (hipBLASLt) on the same product")
print("TF/s on NOMINAL 2*M*N*K*batch (dead-tile skipping of this library is NOT credited here; the vendor computes every row and no epilogue)")
print("%d launches; this library %.2f ms in total, %.2f ms on the %d launches that have a vendor twin; vendor %.2f ms on those"
      % (len(records), tot_o, tot_both, sum(e["nv"] for e in agg.values()), tot_v))
print("%5s %9s %9s %8s %8s %7s  %s" % ("calls", "ours ms", "vendor ms", "ours TF", "vend TF", "ratio", "launch"))
for tag, e in sorted(agg.items(), key=lambda kv: -kv[1]["ours"]):
    if e["nv"]:
        print("%5d %9.3f %9.3f %8.0f %8.0f %7.2f  %s%s" % (e["n"], e["ours"], e["vend"], e["fl"] / e["ours"] / 1e9, e["fl"] / e["vend"] / 1e9,
                                                        e["ours"] / e["vend"], tag, ("  [" + e["note"] + "]") if e["note"] else ""))
    else:
        print("%5d %9.3f %9s %8.0f %8s %7s  %s  [no vendor twin%s]" % (e["n"], e["ours"], "-", e["fl"] / e["ours"] / 1e9, "-", "-", tag,
                                                                      (": " + e["note"]) if e["note"] else ""))
print()
print("order effect (sum over launches with a twin): this library when it ran FIRST %.2f ms / SECOND %.2f ms; vendor FIRST %.2f / SECOND %.2f"
      % (sum(e["ours_first"] for e in agg.values()), sum(e["ours_second"] for e in agg.values()),
         sum(e["vend_first"] for e in agg.values()), sum(e["vend_second"] for e in agg.values())))
