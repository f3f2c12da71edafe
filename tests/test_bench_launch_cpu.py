"""`python bench.py --gpus N` must start its own N ranks (the driver runs the plain command): the launch path is
exercised here without a GPU through --selftest-launch (gloo rendezvous on 127.0.0.1, one all-reduce)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, gpus=2):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MIRX_BENCH_CHILD")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--selftest-launch"],
                          env=env, capture_output=True, text=True, timeout=300)


def test_plain_command_starts_its_own_ranks_and_prints_one_line():
    r = _run()
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] == 3.0        # both ranks took part in the all-reduce
    # the line proves by itself which backend carried the collectives and who the ranks were
    assert d["collective_backend"] == "gloo" and d["rehearsal"] is False
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and [r["local_rank"] for r in d["ranks"]] == [0, 1]
    assert len({r["pid"] for r in d["ranks"]}) == 2


def test_rehearsal_switch_is_refused_where_more_than_one_gpu_is_visible():
    r = _run({"MIRX_BENCH_REHEARSE": "1", "MIRX_BENCH_SELFTEST_VISIBLE_GPUS": "8"})
    assert r.returncode == 4 and "MIRX_BENCH_REHEARSE" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    # on a one-GPU box the rehearsal stays available (the builder's only way to run two ranks)
    r = _run({"MIRX_BENCH_REHEARSE": "1", "MIRX_BENCH_SELFTEST_VISIBLE_GPUS": "1"})
    assert r.returncode == 0, r.stderr[-2000:]


def test_a_failing_rank_makes_the_plain_command_fail():
    r = _run({"MIRX_BENCH_SELFTEST_FAIL": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_world_size_mismatch_is_an_error_not_a_relaunch():
    env = {k: v for k, v in os.environ.items() if k != "MIRX_BENCH_CHILD"}
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr
