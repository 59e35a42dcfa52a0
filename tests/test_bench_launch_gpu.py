"""`python bench.py --gpus N` exactly as a driver without a launcher of its own calls it: no torchrun, no rank
variables in the environment.  The parent process must start the ranks itself, and rank 0's single JSON line must say
how many ranks met and where each of them ran.  On a box with one GPU the ranks share it (gloo for the rendezvous, the
library's own transport for the halo messages)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _plain_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE")}
    env.update(OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", SIH_HALO_TIMEOUT_MS="5000")
    return env


@pytest.mark.parametrize("ranks,mesh", [(2, "QU240"), (4, "QU240")])
def test_plain_bench_command_starts_its_own_ranks(ranks, mesh):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1",
           "--mesh", mesh]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=_plain_env(), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == 3 and d["value"] > 0
    w = d["world"]
    assert w["ranks_at_rendezvous"] == ranks and w["launcher"] == "bench.py itself"
    assert sorted(r["rank"] for r in w["ranks"]) == list(range(ranks))
    assert all(r["device"] == r["rank"] % w["devices_visible_per_rank"] for r in w["ranks"])
    assert len({r["pid"] for r in w["ranks"]}) == ranks  # one process per rank
    assert d["exchange"]["transport"] in ("p2p", "torch")
    assert d["roofline"]["frac"] > 0 and "cpu_baseline" not in d  # the CPU leg is rank 0's at N = 1 only


def test_a_transport_that_fails_cleanly_is_retried_once_with_the_conservative_settings():
    """The library's own transport made to fail from the host (a time limit no message can meet is not needed: the
    set-up is told to refuse) must not end the run: the ranks fall back to torch.distributed for the halo messages in
    the same launch, and the line says which transport ran."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--mesh", "QU480"]
    env = _plain_env()
    env["SIH_TEST_REFUSE_P2P"] = "1"
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["exchange"]["transport"] == "torch" and "refused" in d["config"]["parallelism"]
