"""One process per GPU, started by the program itself.

`python bench.py --gpus N` (no launcher in front, WORLD_SIZE unset) must still run N ranks: `spawn_ranks` starts N fresh
copies of the calling script as CHILD processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, waits
for them and returns the worst exit code.  It is called before the parent has imported torch or loaded the HIP library — a
process that has touched the GPU must neither fork ranks nor be replaced by another program, so nothing here execs.
Standard library only.  (ranks own GOPs / block ranges; there is nothing to exchange but the timing and a digest.)
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time


def world_from_env() -> int | None:
    v = os.environ.get("WORLD_SIZE")
    return int(v) if v not in (None, "") else None


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def needs_spawn(gpus: int) -> bool:
    """True in the parent of a `--gpus N` run that nobody launched as ranks."""
    return gpus > 1 and world_from_env() is None


def check_world(gpus: int) -> int:
    """Inside a rank (or a single process): the world the environment describes must be the one asked for.  A `--gpus 8`
    request never silently becomes a one-rank run."""
    world = world_from_env() or 1
    if world != gpus:
        sys.stderr.write(f"error: --gpus {gpus} but WORLD_SIZE is {world}: launch {gpus} ranks "
                         f"(python -m torch.distributed.run --nproc-per-node {gpus} ...) or call the script without a launcher\n")
        sys.exit(2)
    return world


def spawn_ranks(script: str, argv: list[str], gpus: int, timeout_s: float | None = None) -> int:
    """Start `gpus` ranks of `script argv`; rank 0's stdout is relayed line by line by a thread (its ONE JSON line reaches our stdout),
    the other ranks' stdout goes to stderr.  Returns 0 when every rank returned 0, else the first non-zero code seen
    (remaining ranks are terminated by PID)."""
    port = int(os.environ.get("MASTER_PORT") or free_port())
    procs = []
    for r in range(gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(gpus), "LOCAL_WORLD_SIZE": str(gpus),
                    "MASTER_ADDR": os.environ.get("MASTER_ADDR") or "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    def relay():
        for line in procs[0].stdout:          # JSON lines are the result; library chatter on stdout ("[Gloo] Rank 0 is connected ...") is not
            out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            out.write(line)
            out.flush()

    relay_thread = threading.Thread(target=relay, daemon=True)
    relay_thread.start()
    rc = 0
    t0 = time.monotonic()
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0:
                break                       # a rank failed: the others may be waiting for it in a collective
            if timeout_s is not None and time.monotonic() - t0 > timeout_s:
                rc = 124
                break
            if pending:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        relay_thread.join(timeout=5)
    return rc
