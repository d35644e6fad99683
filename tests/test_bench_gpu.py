"""bench.py's output contract (the driver parses it): ONE JSON line on stdout, last, with the metric / roofline / cpu_baseline
objects.  Run on a reduced workload so the test takes seconds; the numbers themselves are not asserted."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4", "--seconds", "3",
                        "--cpu-seconds", "1"] + extra, capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    assert lines and lines[-1].startswith("{"), "the JSON line must be the last line on stdout: %r" % lines[-3:]
    assert sum(1 for l in lines if l.startswith("{")) == 1
    return json.loads(lines[-1])


def test_bench_line_contract():
    d = _run([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["unit"] == "utterances/s" and d["value"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert abs(d["value"] - 4 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    assert r["unit"] in ("TFLOP/s", "GB/s") and r["launches"] > 0 and r["avg_launch_ms"] > 0
    if r["bound"] == "mfma":
        # `frac` is the ALGORITHMIC figure; the executed one (tile rounding included) can only be larger
        assert r["frac_algorithmic"] == r["frac"] and r["frac_executed"] >= r["frac"] * 0.999 and r["algorithmic_tflop_per_update"]["gemm"] > 0
        assert 0 < r["attention"]["frac"] < 1 and 0 < r["step_mfu_algorithmic"] < 1
        assert 0 < r["hbm_bound_classes"]["layernorm"]["frac"] < 1 and r["hbm_bound_classes"]["optim"]["achieved_GBps"] > 0
    assert set(d["config"]["parity"]) == {"fp32", "bf16"}
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "utterances/s" and c["sample"]
    assert c["reference_equivalent"]["value"] > c["value"] and "profiles/" in c["reference_equivalent"]["source"] and c["reference_equivalent"]["approximate"] is True
    assert d["config"]["h2d"]["included_in_value"] is False and d["config"]["h2d"]["value_with_h2d"] > 0
    # the default 1-GPU run also carries BASELINE configs 4 (Chimera) and 5 (beam-search decode) measured on the same box
    x = d["extra"]
    # ... and the all-max-length variant of the headline workload (the configuration BASELINE.md section 3 prices)
    assert x["maxlen"]["value"] > 0 and x["maxlen"]["steps"] == 4 and 0 < x["maxlen"]["mfu"] < 1 and x["maxlen"]["per_class_ms"]["gemm"] > 0
    assert x["chimera"]["value"] > 0 and x["chimera"]["unit"] == "utterances/s" and x["chimera"]["roofline"]["frac"] > 0
    assert x["decode"]["value"] > 0 and x["decode"]["ms_per_decode_step"] > 0 and x["decode"]["roofline"]["bound"] == "hbm"


def test_bench_under_a_process_group():
    """The N > 1 launch path on the one configuration a 1-GPU box offers: a 1-rank RCCL group with the collective path forced on
    (RANK / WORLD_SIZE / MASTER_* from the environment, as torch.distributed.run sets them)."""
    d = _run(["--no-cpu-baseline", "--no-roofline", "--no-extra"], env=dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                                                                MASTER_PORT="29541", CST_DDP_FORCE="1"))
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"


def test_bench_two_ranks_share_the_gpu_over_gloo():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` as far as a 1-GPU box can follow it: two processes with
    RANK 0 / 1 on cuda:0, gloo instead of RCCL (CST_DIST_BACKEND).  Rank 0 alone prints the line; value = the work of BOTH ranks over
    the slowest rank's time; every rank trains on the same lengths (fixed work per GPU)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        e = dict(os.environ)
        e.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CST_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                                       "--seconds", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    lines0 = [l for l in outs[0][0].strip().splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and not [l for l in outs[1][0].strip().splitlines() if l.startswith("{")]
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 8
    assert abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    assert d["cpu_baseline"] is None and not d.get("extra")  # the CPU baseline and the extra legs belong to the N = 1 line


def test_bench_launches_its_own_ranks_when_started_plainly():
    """`python3 bench.py --gpus 2` with NO rank variables — the way the driver starts the N = 1 run: the parent (which never touches the
    GPU) starts the two rank processes itself (distributed.launch_ranks; fairseq/distributed_utils.py:286-303), relays rank 0's line and
    exits 0.  gloo on the shared GPU stands in for RCCL on two GPUs (CST_DIST_BACKEND)."""
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    e["CST_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--seconds", "3"], capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 8 and d["value"] > 0
    # a rank that fails takes the job down with its exit code instead of leaving the others in a collective
    e["CST_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2",
                        "--seconds", "2"], capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 7 and "rank 1 exited with code 7" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_eight_ranks_started_plainly_share_the_gpu_over_gloo():
    """`python3 bench.py --gpus 8` with no rank variables — the command of the first 8-GPU session — as far as a 1-GPU box can follow
    it: the launcher starts EIGHT rank processes (LOCAL_RANK 0..7 all map to cuda:0 here, gloo stands in for RCCL), every rank runs
    the full-size model on one 2 s utterance, the gradient buckets of 166.8 M parameters travel through the eight-rank collective,
    rank 0 alone prints the line with n_gpus = 8.  No scaling number is read off this (eight processes time-slice one GPU)."""
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    e["CST_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--batch", "1",
                        "--seconds", "2"], capture_output=True, text=True, timeout=1500, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "dp8" and d["config"]["global_batch"] == 8 and d["scaling"] == "weak"
    assert abs(d["value"] - 8 * 1 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6 and d["cpu_baseline"] is None and not d.get("extra")


def test_bench_decode_line_contract():
    """--mode decode (BASELINE configs[4]) on a reduced workload: the same one-line contract, an HBM-bound roofline object."""
    e = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "decode", "--steps", "1", "--warmup", "1", "--batch", "4",
                        "--seconds", "3", "--max-len", "8"], capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    assert lines and lines[-1].startswith("{") and sum(1 for l in lines if l.startswith("{")) == 1
    d = json.loads(lines[-1])
    assert d["unit"] == "utterances/s" and d["value"] > 0 and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert d["config"]["beam"] == 5 and d["config"]["max_len"] == 8 and d["config"]["ms_per_decode_step"] > 0
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["algorithmic_bytes"] > 0 and "53 graph nodes" in rf["kernel"]  # 47 (the query projection rides in the cross-attention launch) + the six split-K reduces of fc2
