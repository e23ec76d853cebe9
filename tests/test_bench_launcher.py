"""bench.py's own multi-rank launcher (`python bench.py --gpus 2` without torchrun) exercised on CPU: two gloo ranks, the
oracle stand-in engine, a tiny workload.  Checks the plumbing only -- rank spawning before any device call, rendezvous on
127.0.0.1, max-over-ranks timing, both population layouts, ONE JSON line from rank 0 -- never a performance number."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra_args, extra_env):
    env = dict(os.environ, LENV_BENCH_PLUMBING_ENGINE="oracle.engine_standin:OracleNesEngine", OMP_NUM_THREADS="1",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"] + extra_args,
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout        # exactly one JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.timeout(400)
def test_bench_self_launches_two_ranks():
    line = _run_bench(["--gpus", "2"], {})
    assert line["n_gpus"] == 2 and line["ranks"]["world_size"] == 2
    assert line["ranks"]["launcher"] == "bench.py child processes"
    assert line["scaling"] == "weak" and line["steps"] == 2 and line["warmup"] == 1
    # weak layout: 64 workers per rank; strong layout: global pop 64 split over the ranks
    assert line["weak"]["global_pop"] == 128 and line["weak"]["workers_per_gpu"] == 64
    assert line["strong"]["global_pop"] == 64 and line["strong"]["workers_per_gpu"] == 32
    assert line["value"] == line["weak"]["value"] > 0 and line["strong"]["value"] > 0
    assert "PLUMBING" in line["data"]


@pytest.mark.timeout(400)
def test_bench_single_rank_line_shape():
    line = _run_bench([], {})
    assert line["n_gpus"] == 1 and line["strong"]["value"] == line["value"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "timed_region_s"):
        assert k in line


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_driver_command_rehearsed_with_two_ranks_on_one_gpu():
    """The driver's N > 1 command rehearsed on the one GPU a test box has: `bench.py --gpus 2` starts two child ranks that share cuda:0 and
    talk through gloo (LENV_BENCH_BACKEND), with the PRODUCT's HIP engine -- the real weak / strong population layouts, two captured graphs
    around the eager all-gather per generation, max-over-ranks timing, one JSON line.  (The numbers are two processes time-slicing one GPU:
    a plumbing check of the multi-rank bench path, never a measurement; RCCL itself is covered by the one-rank communicator test.)"""
    env = dict(os.environ, LENV_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LENV_BENCH_PLUMBING_ENGINE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-configs", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"]["world_size"] == 2 and line["ranks"]["backend"] == "gloo"
    assert line["ranks"]["collective_ran"] is True and line["ranks"]["collectives_per_generation"] == 1
    assert line["ranks"]["launcher"] == "bench.py child processes"
    assert line["config"]["graphs_per_generation"] == 2
    assert line["weak"]["global_pop"] == 128 and line["weak"]["workers_per_gpu"] == 64
    assert line["strong"]["global_pop"] == 64 and line["strong"]["workers_per_gpu"] == 32
    import math
    assert math.isfinite(line["strong"]["value"]) and line["strong"]["value"] > 0 and line["value"] == line["weak"]["value"] > 0
    assert line["data"] == "synthetic" and line["dtype"] == "f32" and line["scaling"] == "weak"
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["kernel_ms"] > 0
