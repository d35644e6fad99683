#!/usr/bin/env python3
"""Instruction histogram of the MFMA-carrying basic blocks of one kernel in a hipcc -save-temps .s file.

usage: isa_loop_stats.py file.s kernel_substring [--min-mfma N]
Classes: mfma, valu (v_* except mfma / accvgpr moves), trans (v_exp/v_log/v_rcp/v_rsq/v_sqrt/v_sin/v_cos), salu, lds (ds_*),
vmem (global_/buffer_/flat_/scratch_), wait (s_waitcnt / s_barrier / s_nop)."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_accvgpr"):
        return "accmov"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[sys.argv.index("--min-mfma") + 1]) if "--min-mfma" in sys.argv else 4
    verbose = "-v" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z0-9]+:", l) and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], ("entry", [])
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            blocks.append(cur)
            cur = (m.group(1), [])
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        cur[1].append(t.split()[0])
    blocks.append(cur)
    for name, ops in blocks:
        cls = collections.Counter(classify(o) for o in ops)
        if cls["mfma"] < min_mfma:
            continue
        print("%s: %d instr  " % (name, len(ops)) + "  ".join("%s=%d" % kv for kv in sorted(cls.items())))
        if verbose:
            hist = collections.Counter(ops)
            print("   " + ", ".join("%s x%d" % kv for kv in hist.most_common(40)))
    for l in lines[end:end + 400]:
        if key in l and ".name" in l:
            break
    meta = [l.strip() for l in lines if re.search(r"\.(vgpr_count|sgpr_count|agpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):", l)]
    # print the metadata of this kernel: locate its .name entry in the yaml block
    txt = "\n".join(lines)
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
        if key in m.group(0):
            for f in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
                mm = re.search(r"\.%s:\s+(\d+)" % f, m.group(0))
                if mm:
                    print("   .%s %s" % (f, mm.group(1)))
            break


if __name__ == "__main__":
    main()
