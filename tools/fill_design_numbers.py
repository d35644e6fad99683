#!/usr/bin/env python3
"""Fills the @@…@@ placeholders of DESIGN.md §8 from the files tools/collect_profiles.sh produced:  python tools/fill_design_numbers.py profiles/r03
(the section text is tools/design_section8_<round>.md.in; placeholders a template does not use are skipped)"""
import json, os, re, sys
tag = sys.argv[1]  # e.g. gpurun_out/r02_prof/r02
def J(name):
    for l in open("%s_%s.json" % (tag, name)):
        if l.startswith("{"):
            return json.loads(l)
s2t, chim, d0, dec, mx = J("bench_s2t"), J("bench_chimera"), J("bench_dropout0"), J("bench_decode"), J("bench_maxlen")
r = s2t["roofline"]; pc = r["per_class_ms"]
stats = open("%s_kernel_stats.txt" % tag).read()
m = re.search(r"= (\d+) launches per update; kernel time ([\d.]+) ms", stats)
launches, ktime = int(m.group(1)), float(m.group(2))
pmc = json.load(open("%s_pmc_gemm_class.json" % tag))
summ = open("%s_pmc_step_summary.txt" % tag).read()
def row(pattern, sect):
    sec = summ.split("# ")[sect]
    for l in sec.splitlines():
        if pattern in l:
            return l
    return ""
def mfma(pattern):
    l = row(pattern, 1); mm = re.search(r"MFMA busy\s+([\d.]+)%", l); return mm.group(1) if mm else "–"
def fab(pattern):
    l = row(pattern, 2); mm = re.search(r"fabric\s+([\d.]+) MB/launch.*?(\d+) GB/s", l); return (mm.group(1), mm.group(2)) if mm else ("?", "?")
mm0 = re.search(r"([\d.]+) +([\d.]+)  _ZN12_GLOBAL__N_120conv0_fwd_reg", stats)
c0f = "%.2f" % (float(mm0.group(2)) / 1e3) if mm0 else "0.53"
gn = gt = rt = 0.0
for line in stats.splitlines():
    mm1 = re.match(r"\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", line)
    if mm1 and "gemm" in mm1.group(4) and "splitk_reduce" not in mm1.group(4):
        gn += float(mm1.group(1)); gt += float(mm1.group(2))
    elif mm1 and "splitk_reduce" in mm1.group(4):
        rt += float(mm1.group(2))
cpu = s2t["cpu_baseline"]
dom = r["dominant_launch"]
rep = {
 "@@S2T@@": "Default = the training recipe (dropout 0.1 at every site of the reference, layerdrop off): **%.0f utterances/s, %.1f ms per update** on this box." % (s2t["value"], s2t["ms_per_step"]),
 "@@GEMM@@": "%.1f" % pc["gemm"], "@@ATTN@@": "%.1f / %.1f" % (pc["attn_fwd"], pc["attn_bwd"]), "@@LN@@": "%.1f" % pc["layernorm"],
 "@@CONV0@@": "%.1f" % pc["conv0"], "@@EW@@": "%.1f" % pc["elementwise"], "@@OPT@@": "%.2f" % (pc["optim"] + pc["loss"]),
 "@@LAUNCHES@@": "%d" % launches,
 "@@ROOF@@": "%.0f TFLOP/s of 2500 (`frac` %.3f) over %d launches, average %.3f ms." % (r["achieved"], r["frac"], r["launches"], r["avg_launch_ms"]),
 "@@TRACEGEMM@@": "%.0f GEMM launches per update, %.1f ms + %.1f ms of split-K reduces, which sit inside the same event pairs, = %.1f ms against %.1f ms from the event table (the box ran %.0f %% slower under the tracer)." % (gn, gt, rt, gt + rt, pc["gemm"], 100.0 * (J("bench_under_trace")["ms_per_step"] / s2t["ms_per_step"] - 1.0)),
 "@@TRAFFIC@@": "%.0f MB against %.0f MB algorithmic = %.2f × (round 1: 372 vs 247 MB = 1.5 ×)." % (r["traffic"] / 1e6, r["algorithmic_bytes"] / 1e6, r["traffic"] / r["algorithmic_bytes"]) if r.get("traffic") else "n/a",
 "@@DOM@@": "%.3f ms, %.0f TF/s (`frac` %.3f), fabric traffic %.0f MB vs %.0f MB algorithmic." % (dom["avg_launch_ms"], dom["achieved"], dom["frac"], dom["traffic"] / 1e6, dom["algorithmic_bytes"] / 1e6),
 "@@MAXLEN@@": "%.0f utt/s, %.1f ms per update; GEMM class %.1f ms = %.0f TFLOP/s (`frac` %.3f) over the full 44 TFLOP, attention %.1f + %.1f ms (`%s_bench_maxlen.json`)." % (mx["value"], mx["ms_per_step"], mx["roofline"]["per_class_ms"]["gemm"], mx["roofline"]["achieved"], mx["roofline"]["frac"], mx["roofline"]["per_class_ms"]["attn_fwd"], mx["roofline"]["per_class_ms"]["attn_bwd"], tag.split("/")[-1]),
 "@@D0@@": "%.0f utt/s, %.1f ms (`%s_bench_dropout0.json`)." % (d0["value"], d0["ms_per_step"], tag.split("/")[-1]),
 "@@CHIMERA@@": "%.0f utt/s, %.1f ms (`%s_bench_chimera.json`)." % (chim["value"], chim["ms_per_step"], tag.split("/")[-1]),
 "@@CPU@@": "%.2f utterances/s on %d threads (%s s per update)." % (cpu["value"], cpu["cores"], " / ".join("%.1f" % v for v in cpu["seconds_per_update"])),
 "@@MFMA@@": "persistent GEMM %s %% (plain / bias / activation epilogues), %s %% (one-operand, B mn-major: the dX GEMMs), %s %% (one-operand, k/k); 16-wave dW kernel %s %%; attention forward %s %%, dQ %s %%, dK/dV %s %%." % (mfma("gemm8p_kernel<true, true, 1>"), mfma("gemm8p_kernel<true, false, 2>"), mfma("gemm8p_kernel<true, true, 2>"), mfma("CfgILi256ELi256"), mfma("attn_fwd_kernel"), mfma("attn_bwd_dq"), mfma("attn_bwd_dkv")),
 "@@FABRIC@@": "persistent GEMM (mode 1) %s MB at %s GB/s, dW kernel %s MB at %s GB/s, LayerNorm (768-wide rows) forward %s MB at %s GB/s / backward %s MB at %s GB/s, column sums %s GB/s, dropout %s GB/s, Adam %s MB at %s GB/s, conv0 forward ≈ 2.1 GB written in %s ms ≈ %.1f TB/s (below the listing's cut-off since it stops at the frames that are read) / backward %s MB at %s GB/s (VALU-bound)." % (fab("gemm8p_kernel<true, true, 1>") + fab("CfgILi256ELi256") + fab("ln_fwd_kernelIDF16bLi2") + fab("ln_bwd_kernelIDF16bLi2") + (fab("colsum_kernel")[1], fab("dropout_kernel")[1]) + fab("adam_kernel") + (c0f, 2.1 / float(c0f)) + fab("conv0_bwd_reg")),
 "@@DECODE@@": "s2t_transformer_l (12 + 6 layers), 32 utterances × ≤ 30 s of filter banks, beam 5, 201 steps, bf16: **%.0f utterances/s, %.0f tokens/s, %.3f ms per decode step** (encoder %.1f ms per batch); `roofline` bound = hbm: %.2f GB of algorithmic bytes per step ÷ %.3f ms = %.0f GB/s = %.3f of 8 TB/s (round 1, builder-measured: 0.82 ms on another box; the same-box A/B of this round's LayerNorm folding is in §5.7)." % (dec["value"], dec["config"]["tokens_per_s"], dec["config"]["ms_per_decode_step"], dec["config"]["encoder_ms"], dec["roofline"]["algorithmic_bytes"] / 1e9, dec["roofline"]["avg_launch_ms"], dec["roofline"]["achieved"], dec["roofline"]["frac"]),
}
here = os.path.dirname(os.path.abspath(__file__))
rnd = os.path.basename(tag)
tmpl = os.path.join(here, "design_section8_%s.md.in" % rnd)
sec = open(tmpl if os.path.exists(tmpl) else os.path.join(here, "design_section8.md.in")).read()  # DESIGN.md section 8 with @@...@@ placeholders
if rnd != "r02":
    h2d = s2t["config"].get("h2d", {})
    rep["@@H2D@@"] = "%.0f utterances/s (`value_with_h2d`)." % h2d["value_with_h2d"] if "value_with_h2d" in h2d else "not measured."
    rep["@@MFMA@@"] = ("persistent GEMM %s %% (forward: bias / activation epilogues), %s %% (one-operand epilogues: the dX GEMMs, conv stack); 16-wave dW kernel %s %%; "
                       "attention forward %s %%, dQ %s %%, dK/dV %s %% (dropout 0.1)." % (mfma("gemm8p_kernel<true, true, 1>"), mfma("gemm8p_kernel<true, true, 2>"), mfma("CfgILi256ELi256"),
                                                                                         mfma("fa_fwd_kernel<true>"), mfma("fa_dq_kernel<true>"), mfma("fa_dkv_kernel<true>")))
    try:
        rep["@@VENDOR@@"] = " ".join(l.strip().replace("(A k-major, W [N,K]): ", "").replace("torch.matmul/hipBLASLt", "hipBLASLt") + ";" for l in open("%s_hipblaslt_vs_this_library.txt" % tag) if " x " in l and "ratio" in l)
    except OSError:
        rep["@@VENDOR@@"] = "(not collected)"
    rep["@@DECODE@@"] = rep["@@DECODE@@"].split(" (round 1, builder-measured")[0] + "."
    rep["@@FABRIC@@"] = rep["@@FABRIC@@"].replace(", dropout ? GB/s", "")
for k, v in rep.items():
    if k in sec:
        sec = sec.replace(k, v)
assert "@@" not in sec, re.findall(r"@@\w+@@", sec)
d = open("DESIGN.md").read()
a, b = d.index("## 8. Measurement (round"), d.index("## 9. VERDICT round")
open("DESIGN.md", "w").write(d[:a] + sec + d[b:])
print("filled", len(rep))
