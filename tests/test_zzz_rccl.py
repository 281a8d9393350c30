"""GPU: the RCCL code path of the multi-GPU layer executed on ONE MI355X -- backend "nccl" with
world_size 1 and the collectives forced (``force_collective``): communicator set-up, the packed fp64 all-reduce on
RCCL's stream, ``Work.wait()`` semantics, the side stream next to hipGraph replays of the env step.  It runs in a child
process (a process group is process-global state) and is collected last.  tests/test_distributed_gloo.py covers
world_size 2 on CPU; an 8-GPU node is the driver's to run."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_world_size_one_runs_every_collective():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "workers", "rccl_world1.py"), str(port)],
                       capture_output=True, text=True, timeout=280, env=env)
    assert p.returncode == 0, p.stderr[-4000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["backend"] == "nccl"
    # the four families of SURVEY 8e(2)
    assert out["families"] == ["Curriculum", "Episode_Reward", "Episode_Termination", "Metrics"]
    assert any(k.startswith("Episode_Termination/") for k in out["stats_names"])
    assert "Curriculum/terrain_levels" in out["stats_names"]
    # one rank: the all-reduce is the identity -> merged == local, bit for bit for the statistics
    assert out["stats_max_abs_diff"] == 0.0 and out["fused_stats_max_abs_diff"] == 0.0
    # the additive row about the running mean against the local Chan merge: fp64 rounding only
    assert out["scaler_mean_max_rel"] < 1e-12 and out["scaler_var_max_rel"] < 1e-10
    assert out["scaler_row_path_max_rel"] < 1e-10
    assert out["scaler_count"][0] == out["scaler_count"][1]
    a, b = out["consecutive_successes"]
    assert a == b and a != 1.0          # the all-reduced partial sums moved the average exactly as the local ones
