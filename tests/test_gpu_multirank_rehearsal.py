"""GPU (one-GPU box): the N > 1 legs of bench.py and tools/bench_c5.py, rehearsed with two processes on the one device over
gloo (--rehearse): rank/shard arithmetic, barriers, the max-reduce of the elapsed time, the digest all-reduce and the single
JSON line on rank 0 are the code the driver's 8-GPU run executes; only the backend string and the device index differ."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run2(script, extra, port):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, script), "--gpus", "2", "--rehearse"] + extra
    pr = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, (pr.stdout[-2000:], pr.stderr[-3000:])
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]            # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_rehearsal():
    r = run2("bench.py", ["--steps", "3", "--warmup", "1", "--blocks", "65536"], 29611)
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["config"]["global_blocks"] == 2 * 65536 and r["value"] > 0 and r["roofline"]["frac"] > 0
    assert "cpu_baseline" not in r                     # rank 0 at N = 1 only


def test_bench_c5_two_ranks_rehearsal_digest_matches_single_rank():
    """4 GOPs of 2 small frames: ranks 0 / 1 own GOPs {0, 2} / {1, 3}; the all-reduced digest equals the one-process digest"""
    small = ["--frames", "8", "--gop", "2", "--width", "640", "--height", "384", "--sizes", "64,32,16,8,4"]
    r2 = run2("tools/bench_c5.py", small, 29612)
    assert r2["n_gpus"] == 2 and r2["gop_owner"] == {"0": 0, "1": 1, "2": 0, "3": 1} and r2["frames"] == 8
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools/bench_c5.py")] + small, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-2000:]
    r1 = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][0])
    assert r1["digest_blocks_eob_qchk_recon"] == r2["digest_blocks_eob_qchk_recon"]
    assert r1["frames"] == 8 and r1["n_gpus"] == 1
