"""`python3 bench.py --gpus N` as a PLAIN command (no torchrun, no WORLD_SIZE): bench.py starts its own N ranks before
anything touches the GPU and relays rank 0's single JSON line.  --dry_run stops after the rendezvous and the known-answer
collectives (metalign_amd/distributed.py::selfcheck_collectives), so the plumbing is checked here without a GPU; the same
command without --dry_run runs under `-m gpu` in tests/test_pipeline_gpu.py / tests/dist_two_ranks_one_gpu.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCH_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")


def run_bench(*argv, **extra_env):
    env = {k: v for k, v in os.environ.items() if k not in LAUNCH_VARS}
    env.update({"MG_DIST_BACKEND": "gloo", **extra_env})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_plain_command_spawns_its_own_ranks(world):
    p = run_bench("--gpus", str(world), "--dry_run")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout  # ONE JSON line on stdout, whatever the launcher and gloo print elsewhere
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["dry_run"] is True and d["collective_selfcheck"] == "ok" and d["backend"] == "gloo"


def test_one_gpu_dry_run_does_not_launch_anything():
    p = run_bench("--gpus", "1", "--dry_run")
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["collective_selfcheck"].startswith("skipped")


def test_a_failing_rank_fails_the_command():
    # an unknown backend makes init_process_group raise in every rank: the plain command must not exit 0 or print a line
    p = run_bench("--gpus", "2", "--dry_run", MG_DIST_BACKEND="no_such_backend")
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]


def test_selfcheck_catches_misdelivered_data():
    """The known-answer check itself: a transport that swaps two ranks' all-gather contributions must be caught."""
    import numpy as np
    import torch

    from metalign_amd import distributed as mgd

    class OneRank:  # world 1 "transport" that corrupts the all-reduce
        class ReduceOp:
            SUM = 0

        def get_backend(self):
            return "gloo"

        def all_gather(self, outs, t):
            outs[0].copy_(t)

        def all_reduce(self, t, op=None):
            t[0] = t[0] & 0xffffffff  # a 32-bit reduction

    with pytest.raises(RuntimeError, match="all_reduce"):
        mgd.selfcheck_collectives(OneRank(), torch, 0, 1, "cpu")
    assert np  # (numpy is what the check compares with)
