"""`not gpu`: bench.py's own logic -- the JSON line it assembles -- at world sizes the 1-GPU boxes cannot run.  `--emulated` runs the real kernel
sources on the CPU emulator of tests/hostemu with gloo in place of RCCL; the figures mean nothing, the STRUCTURE of the line is what is tested:
at N > 1 the line must carry a non-null `parity` (every rank's sampled documents against the oracle, reduced), `cpu_baseline` (rank 0's bounded
sample) and `roofline` (the slowest rank's kernels) -- the round-4 review's point: without them a scaling line earns nothing."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, extra, timeout=900):
    args = ["--emulated", "--gpus", str(world), "--steps", "1", "--warmup", "1", "--pipelined-steps", "0", "--min-len", "20", "--max-len", "300"] + extra
    env = dict(os.environ, OMP_NUM_THREADS="1")
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [x for x in out.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]           # ONE JSON line on stdout, from rank 0
    return json.loads(lines[0])


def test_world_2_line_has_parity_cpu_baseline_and_roofline():
    line = _run(2, ["--docs", "240", "--parity-sample-docs", "50", "--cpu-sample-docs", "120"])
    assert line["emulated"] is True and line["data"].startswith("EMULATED")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 1 and line["warmup"] == 1
    assert line["config"]["job_docs"] == 480 and line["config"]["docs_per_gpu"] == 240
    # parity: both ranks checked the first and the last 50 documents of their own shard
    assert line["parity"].startswith("bit-exact vs oracle on 2 x 100 = 200 sampled docs"), line["parity"]
    # cpu_baseline: rank 0's bounded sample, non-null, with its thread sweep and the one-thread figure
    cpu = line["cpu_baseline"]
    assert cpu and cpu["kind"] == "port" and cpu["value"] > 0 and cpu["value_1_thread"] > 0 and cpu["cores"] >= 1 and "120 documents" in cpu["sample"]
    # roofline: the slowest rank's kernels, priced with that rank's algorithmic bytes
    r = line["roofline"]
    # (the emulator's GB/s round to zero: present and numeric is what is asserted)
    assert r and r["bound"] == "hbm" and r["achieved"] >= 0 and 0 <= r["frac"] < 1 and r["peak"] == 8000.0 and r["rank"] in (0, 1) and r["avg_launch_ms"] > 0
    assert r["algorithmic_bytes_per_launch"] > 0 and set(r["kernels_ms"]) >= {"k_probe", "k_place", "k_pretok"}
    assert line["rank_ms_per_step"]["max"] >= line["rank_ms_per_step"]["min"] > 0
    assert line["value"] > 0 and line["ms_per_step"] == pytest.approx(line["rank_ms_per_step"]["max"], rel=1e-3)


def test_world_2_shard_smaller_than_two_samples_is_checked_whole():
    line = _run(2, ["--docs", "60", "--parity-sample-docs", "50", "--cpu-sample-docs", "60", "--kind", "2"])
    assert line["parity"].startswith("bit-exact vs oracle on 2 x 60 = 120 sampled docs (every document of every rank's shard"), line["parity"]
    assert line["cpu_baseline"] and line["roofline"]


def test_world_1_line_and_real_text_kind(tmp_path):
    # the default shape at N = 1 (full-batch parity) ...
    line = _run(1, ["--docs", "150", "--cpu-sample-docs", "100", "--real-text-mb", "0", "--heldout-steps", "0"])
    assert line["parity"].startswith("bit-exact vs oracle on all 150 docs") and line["cpu_baseline"] and line["roofline"] and line["n_gpus"] == 1
    # ... and kind 6: real files of the box, every file once, cut at character boundaries; the memo emptied before every step
    line = _run(1, ["--kind", "6", "--real-text-mb", "1", "--vocab", "gpt2", "--pattern", "1", "--cpu-sample-docs", "500"])
    rt = line["config"]["real_text"]
    assert rt["bytes"] >= 1 << 20 and rt["files"] > 10 and len(rt["sha256"]) == 64 and rt["docs"] == line["config"]["docs_per_gpu"]
    assert line["parity"].startswith("bit-exact vs oracle on all %d docs" % rt["docs"]), line["parity"]
    assert "EMPTIED before every step" in line["config"]["piece_memo"] and line["value_warm_memo"] and line["value_no_memo"]
    assert line["config"]["vocab"] == "gpt2" and line["config"]["pattern"].startswith("pattern 1")
