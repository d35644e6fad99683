"""CPU tests of the measurement tooling that cannot be exercised on the single-GPU box: the per-bucket overlap analysis of
tools/ddp_overlap_trace.py on a synthetic two-rank kernel trace (the CSV layout of `rocprofv3 --kernel-trace --output-format csv`)."""
import csv
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("ddp_overlap_trace", os.path.join(ROOT, "tools", "ddp_overlap_trace.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _write(path, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kind", "Agent_Id", "Queue_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for name, s, e in rows:
            w.writerow(["KERNEL_DISPATCH", 1, 1, name, s, e])


def test_overlap_analysis_on_a_synthetic_trace(tmp_path):
    m = _load()
    ms = 1_000_000
    for rank in (0, 1):
        rows = []
        t = 0
        for u in range(2):  # two updates: forward 10 ms, loss, backward 3 x 10 ms, optimizer 1 ms
            rows.append(("gemm8p_kernel<fwd>", t, t + 10 * ms)); t += 10 * ms
            rows.append(("ls_ce_fwd_kernel", t, t + 1 * ms)); b0 = t; t += 1 * ms
            for _ in range(3):
                rows.append(("gemm_kernel<bwd>", t, t + 10 * ms)); t += 10 * ms
            b1 = t
            # bucket 0 runs entirely under the backward (hidden), bucket 1 starts 2 ms before its end and runs 6 ms (4 ms exposed)
            rows.append(("ncclDevKernel_AllReduce_Sum_bf16", b0 + 12 * ms, b0 + 17 * ms))
            rows.append(("ncclDevKernel_AllReduce_Sum_bf16", b1 - 2 * ms, b1 + 4 * ms))
            t = b1 + 4 * ms
            rows.append(("adam_kernel", t, t + 1 * ms)); t += 1 * ms
        _write(str(tmp_path / ("rank%d" % rank) / "host" / "1_kernel_trace.csv"), rows)
    out = m.report(str(tmp_path), verbose=False)
    assert sorted(out) == ["rank0", "rank1"]
    for ups in out.values():
        assert len(ups) == 2
        for u in ups:
            assert abs(u["backward_ms"] - 31.0) < 1e-6          # loss kernel + three backward kernels
            assert len(u["collectives"]) == 2
            assert abs(u["collective_ms"] - 11.0) < 1e-6
            assert abs(u["hidden_ms"] - 7.0) < 1e-6 and abs(u["exposed_ms"] - 4.0) < 1e-6
            assert abs(u["tail_after_backward_ms"] - 4.0) < 1e-6
            assert abs(u["collectives"][0]["start_ms"] - 12.0) < 1e-6
