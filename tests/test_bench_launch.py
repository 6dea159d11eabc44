"""bench.py's launch contract (VERDICT r3 missing 3): ``--gpus N`` never prints a line for fewer ranks than asked for.

CPU part: without devices every form exits non-zero and prints no JSON line.  GPU part: a WORLD_SIZE that contradicts
``--gpus`` is refused before any GPU work; ``--gpus 2`` without a launcher starts its two ranks itself (they time-share the
box's GPU over gloo under the development switch ULTRA_BENCH_SHARE_GPU=1) and the line carries ``n_gpus: 2`` and config 4's
N-rank numbers -- the pretrain_3g-shaped step with the bucketed gradient all-reduce."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=1500):
    env = dict(os.environ, **(env or {}))
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        if env.get(key) == "":
            env.pop(key)
    run = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [line for line in run.stdout.splitlines() if line.startswith("{")]
    return run, lines


@pytest.mark.skipif(torch.cuda.device_count() > 0, reason="the no-device behaviour")
@pytest.mark.parametrize("gpus", ["1", "2", "8"])
def test_without_devices_bench_exits_non_zero_and_prints_no_line(gpus):
    run, lines = _run(["--gpus", gpus, "--steps", "2", "--warmup", "1"], env={"WORLD_SIZE": "", "RANK": "", "LOCAL_RANK": ""},
                      timeout=300)
    assert run.returncode != 0 and not lines, (run.returncode, run.stdout[-500:], run.stderr[-500:])


@pytest.mark.gpu
def test_world_size_that_contradicts_gpus_is_refused():
    run, lines = _run(["--gpus", "8", "--steps", "2", "--warmup", "1"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"},
                      timeout=300)
    assert run.returncode != 0 and not lines and "WORLD_SIZE" in run.stderr
    if torch.cuda.device_count() < 8:            # and without a launcher: too few devices is an error, not an n_gpus: 1 line
        run, lines = _run(["--gpus", "8", "--steps", "2", "--warmup", "1"], env={"WORLD_SIZE": "", "RANK": "", "LOCAL_RANK": ""},
                          timeout=300)
        assert run.returncode != 0 and not lines and "visible" in run.stderr


@pytest.mark.gpu
def test_gpus_2_launches_its_own_ranks_and_times_the_pretraining_step_with_the_reducer():
    share = {} if torch.cuda.device_count() >= 2 else {"ULTRA_BENCH_SHARE_GPU": "1"}
    # (config 5 runs on a graph of 1 / 30 the size here: two ranks share one GPU on this box)
    run, lines = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--stress-shape", "300000,1500000,40", "--no-cpu-baseline",
                       "--mrr-queries", "0"], env=dict(share, WORLD_SIZE="", RANK="", LOCAL_RANK=""))
    assert run.returncode == 0 and len(lines) == 1, (run.returncode, run.stdout[-1000:], run.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["scaling"] == "weak" and line["value"] > 0
    cfg = line["config"]
    assert len(cfg["per_rank_ms_per_step"]) == 2
    assert cfg["cfg4_n_gpus"] == 2 and cfg["cfg4_step_ms_max_over_ranks"] > 0
    assert "cfg4_allreduce_exposed_ms_per_step" in cfg and cfg["cfg4_edge_messages_per_s_nominal"] > 0
    four = [c for c in cfg["configs"] if c["config"] == 4][0]
    assert len(four["per_rank_step_ms"]) == 2 and "GradientReducer" in four["gradient_allreduce"]
    # the record says what ran the collectives and on how many different devices (never a gloo number read as xGMI)
    two_gpus = torch.cuda.device_count() >= 2
    assert cfg["collective_backend"] == ("nccl" if two_gpus else "gloo") == cfg["cfg4_collective_backend"]
    assert cfg["ranks_distinct_devices"] == (2 if two_gpus else 1)
    assert cfg["cfg4_step_modes_agree"] is True and cfg["cfg4_step_mode"] in ("phased", "after")
    assert "[bench rank 0" in run.stderr and "[bench rank 1" in run.stderr          # per-phase progress of every rank
    # round 6: the first N-rank line also says who ran what -- per-rank exposed all-reduce and step modes -- and carries config 5 on
    # every rank (replicas, each rank its own triples, no collective in the data path)
    assert len(four["allreduce_exposed_ms_per_step_per_rank"]) == 2 and len(four["step_modes_per_rank"]) == 2
    five = [c for c in cfg["configs"] if c["config"] == 5][0]
    assert five["all_ranks"]["n_gpus"] == 2 == cfg["cfg5_n_gpus"] and len(five["all_ranks"]["per_rank_ms_per_triple"]) == 2
    assert cfg["cfg5_triples_per_s_all_ranks"] > 0 and five["predict_ms"] > 0


def test_a_phase_that_overruns_still_leaves_the_headline_line_of_rank_0():
    """world > 1: a leg after the timed region that hangs (the first real RCCL run) must not cost the headline -- rank 0 prints
    the reduced headline as a line marked `partial` before the watchdog exits with code 3; other ranks print nothing."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "p = bench.Phases(int(sys.argv[1]), 2)\n"
            "p.fallback = {'metric': 'm', 'value': 1.5, 'n_gpus': 2}\n"
            "p.enter('config 4', 0.2)\n"
            "time.sleep(5)\n" % ROOT)
    for rank, lines in ((0, 1), (1, 0)):
        run = subprocess.run([sys.executable, "-c", code, str(rank)], capture_output=True, text=True, timeout=120)
        assert run.returncode == 3, run.stderr[-500:]
        out = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
        assert len(out) == lines
        if lines:
            line = json.loads(out[0])
            assert line["value"] == 1.5 and line["n_gpus"] == 2 and "config 4" in line["partial"]
    assert "exceeded" in run.stderr
