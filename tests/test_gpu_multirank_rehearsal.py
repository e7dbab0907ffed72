"""GPU (one-GPU box): the N > 1 legs of bench.py and tools/bench_c5.py, invoked the way the driver invokes them (`python bench.py
--gpus 2`, no launcher: the script starts its own ranks) and rehearsed with two processes on the one device over gloo (--rehearse): rank/shard arithmetic, barriers, the max-reduce of the elapsed time, the digest all-reduce and the single
JSON line on rank 0 are the code the driver's 8-GPU run executes; only the backend string and the device index differ."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(pr):
    assert pr.returncode == 0, (pr.stdout[-2000:], pr.stderr[-3000:])
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]            # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def run2(script, extra, port, launcher=False):
    """launcher=False: the way the driver issues the command — `python <script> --gpus 2 ...`, nothing in front; the script
    starts its own ranks.  launcher=True: under torch.distributed.run (the N > 1 contract of the prompt)."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, script), "--gpus", "2", "--rehearse"] + extra
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    return _json_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600))


@pytest.mark.parametrize("launcher", [False, True])
def test_bench_two_ranks_rehearsal(launcher):
    r = run2("bench.py", ["--steps", "3", "--warmup", "1", "--blocks", "65536"], 29611, launcher)
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["config"]["global_blocks"] == 2 * 65536 and r["value"] > 0 and r["roofline"]["frac"] > 0
    assert [p["rank"] for p in r["roofline"]["per_rank"]] == [0, 1] and all(p["frac"] > 0 for p in r["roofline"]["per_rank"])
    assert r["digest_blocks_eob_sad_qchk"][0] == 2 * 65536            # all-reduced over the two ranks
    assert "cpu_baseline" not in r                     # rank 0 at N = 1 only


def test_bench_digest_of_two_ranks_is_the_sum_of_the_two_seeds():
    """rank r draws its blocks from seed 13596 + r: the all-reduced digest equals one-process runs' digests added"""
    r2 = run2("bench.py", ["--steps", "1", "--warmup", "1", "--blocks", "16384", "--no-probes"], 29613)
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    tot = [0, 0, 0, 0]
    for rank in (0, 1):
        e = dict(env, SVT_BENCH_SEED_RANK=str(rank))
        r1 = _json_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--blocks", "16384",
                                        "--no-probes", "--no-cpu-baseline"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600))
        assert r1["n_gpus"] == 1
        tot = [a + b for a, b in zip(tot, r1["digest_blocks_eob_sad_qchk"])]
    tot[3] %= (1 << 31) - 1
    assert r2["digest_blocks_eob_sad_qchk"] == tot


def test_bench_c5_two_ranks_rehearsal_digest_matches_single_rank():
    """4 GOPs of 2 small frames: ranks 0 / 1 own GOPs {0, 2} / {1, 3}; the all-reduced digest equals the one-process digest"""
    small = ["--frames", "8", "--gop", "2", "--width", "640", "--height", "384", "--sizes", "64,32,16,8,4"]
    r2 = run2("tools/bench_c5.py", small, 29612)
    assert r2["n_gpus"] == 2 and r2["gop_owner"] == {"0": 0, "1": 1, "2": 0, "3": 1} and r2["frames"] == 8
    r1 = _json_line(subprocess.run([sys.executable, os.path.join(ROOT, "tools/bench_c5.py")] + small, cwd=ROOT, capture_output=True, text=True, timeout=600))
    assert r1["digest_blocks_eob_qchk_recon"] == r2["digest_blocks_eob_qchk_recon"]
    assert r1["frames"] == 8 and r1["n_gpus"] == 1
