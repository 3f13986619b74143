"""GPU: bench.py's own rank launcher.  `python bench.py --gpus 2` (no WORLD_SIZE in the environment) must start two ranks
itself and print ONE JSON line with n_gpus == 2; with APTAI_BENCH_BACKEND=gloo both ranks share the one card of the test
box (the RCCL branch needs two cards and is what the driver's 8-GPU run exercises).  Also rehearsed: a hipGraph capture that
fails on ONE rank only must switch EVERY rank to the eager loop (different collective sequences would hang)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *flags, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(APTAI_BENCH_BACKEND="gloo", **extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "2", "--seconds", "1", "--no-cpu-baseline", *flags], env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0]), p.stderr


def test_bench_launches_its_own_ranks():
    res, err = _run({})
    assert res["n_gpus"] == 2 and res["config"]["ranks"] == 2 and res["config"]["global_batch"] == 4
    assert res["config"]["execution"].startswith("hipGraph")
    assert res["value"] > 0 and res["scaling"] == "weak"
    assert "gloo" in res["config"]["collective"]


def test_capture_failure_on_one_rank_moves_every_rank_to_eager():
    res, err = _run({"APTAI_GRAPH_FAIL_CAPTURE": "rank1"})
    assert res["n_gpus"] == 2 and res["config"]["execution"] == "eager autograd loop"
    assert "ALL ranks run the eager loop" in err
