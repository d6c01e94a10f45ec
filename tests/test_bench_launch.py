"""bench.py honours --gpus: with N > 1 and no launcher around it (WORLD_SIZE unset) it starts its own N ranks as child
processes; a launcher that set a different WORLD_SIZE is an error line and a non-zero exit, never a silently different run.
CPU part: the launch logic (there is no GPU here, so the ranks that are started report exactly that, as rank 0's JSON error
line with n_gpus = N).  GPU part: the real thing on the one card of the GPU box -- `python bench.py --gpus 2` bare, ranks
sharing device 0 under gloo (FMMBEM_BENCH_BACKEND=gloo: a rehearsal of the N > 1 path, not a measurement)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_world_size_that_differs_from_gpus_is_refused():
    r, lines = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2
    assert len(lines) == 1 and lines[0]["value"] is None and "WORLD_SIZE=4" in lines[0]["error"] and lines[0]["stage"] == "launch"
    # a non-zero rank stays silent but exits with the same code
    r, lines = _run(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "1", "LOCAL_RANK": "1"})
    assert r.returncode == 2 and lines == []


def test_gpus_1_with_a_launcher_world_of_2_is_refused():
    r, lines = _run(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in lines[0]["error"]


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="the CPU form of the launch test: on a GPU box the gpu-marked test runs the real thing")
def test_bare_gpus_2_starts_two_ranks_itself():
    """No GPU here: the two ranks bench.py starts each find none; rank 0 says so in ONE JSON line that carries n_gpus = 2 --
    which is the evidence that two ranks with WORLD_SIZE = 2 were started by `python bench.py --gpus 2` alone."""
    r, lines = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--recursions", "3"])
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["rank"] == 0 and "needs a GPU" in lines[0]["error"]


@pytest.mark.gpu
def test_bare_gpus_2_runs_the_sharded_path_on_one_card():
    r, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--recursions", "6", "--no-cpu-baseline"],
                    {"FMMBEM_BENCH_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert out["preflight"]["paths_equal"] is True and out["preflight"]["used"] == "default"
    assert out["preflight"]["exchange_symmetric"] is True
    assert out["replicas_equal"] is True
    assert len(out["per_rank"]) == 2
    assert out["rel_l2_vs_direct_sample"] < 1e-5


@pytest.mark.gpu
def test_one_process_form_over_a_device_list():
    """`--gpus 2 --one-process`: one plan over two list entries (the one GPU named twice here) inside this process -- the same line
    format, n_gpus = 2, the result checked against the Direct sum as ever."""
    r, lines = _run(["--gpus", "2", "--one-process", "--steps", "2", "--warmup", "1", "--recursions", "6", "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = lines[-1]
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["one_process"]["devices"] == [0, 0]
    assert out["rel_l2_vs_direct_sample"] < 1e-5
