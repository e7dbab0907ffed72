"""CPU: the self-launch path of bench.py / tools/bench_c5.py (`python bench.py --gpus N` with no launcher in front starts N ranks
as child processes, before torch or the HIP library are loaded), world size 2 over gloo; a `--gpus N` that disagrees with
WORLD_SIZE is refused; a failing rank fails the run and takes the others with it."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update(extra)
    return env


def test_gpus_flag_and_world_size_must_agree():
    """a `--gpus 8` request never silently becomes a one-rank run (checked before anything touches a GPU)"""
    env = _clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    for script in ("bench.py", "tools/bench_c5.py"):
        pr = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", "2"], cwd=ROOT, env=env, capture_output=True,
                            text=True, timeout=120)
        assert pr.returncode != 0 and "WORLD_SIZE" in pr.stderr
        assert not [l for l in pr.stdout.splitlines() if l.startswith("{")]


RANK_SCRIPT = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import __graft_entry__ as ge
    ge.load_package()
    from cidana_svt_av1_amd import launcher, sharding
    gpus = int(sys.argv[sys.argv.index("--gpus") + 1])
    if launcher.needs_spawn(gpus):
        sys.exit(launcher.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], gpus, timeout_s=120))
    world = launcher.check_world(gpus)
    rank = int(os.environ["RANK"])
    if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
        sys.exit(7)
    import numpy as np
    import torch.distributed as dist
    dist.init_process_group("gloo")
    lo, hi = sharding.shard_range(1000, rank, world)
    d = sharding.allreduce_digest(np.array([hi - lo, rank + 1, 0, lo], dtype=np.int64))
    dist.barrier()
    if rank == 0:
        print(json.dumps({"n_gpus": world, "digest": d.tolist(), "local_rank": os.environ["LOCAL_RANK"]}), flush=True)
    else:
        print("noise from rank", rank, flush=True)        # must not reach the parent's stdout
    dist.destroy_process_group()
""") % ROOT


def test_spawn_ranks_world_size_two_gloo(tmp_path):
    script = tmp_path / "ranks.py"
    script.write_text(RANK_SCRIPT)
    pr = subprocess.run([sys.executable, str(script), "--gpus", "2"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [l for l in pr.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, pr.stdout                      # rank 0's single line, nothing else
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["digest"] == [1000, 3, 0, 500]


def test_spawn_ranks_failing_rank_fails_the_run(tmp_path):
    script = tmp_path / "ranks.py"
    script.write_text(RANK_SCRIPT)
    pr = subprocess.run([sys.executable, str(script), "--gpus", "2", "--fail-rank", "1"], env=_clean_env(), capture_output=True,
                        text=True, timeout=300)
    assert pr.returncode == 7                              # rank 0 sat in init_process_group: terminated by the parent, by PID
    assert not [l for l in pr.stdout.splitlines() if l.startswith("{")]
