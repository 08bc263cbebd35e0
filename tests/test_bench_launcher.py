"""bench.py's own launcher (the reference is single-process, train.py:229-233: starting N ranks is entirely ours).
`python bench.py --gpus N` with no launcher in the environment must start N ranks itself and report the rank count the
collective library counted; a launcher/flag mismatch is an error, never a silent N = 1 run.  --dry-run stops after the
rendezvous and the counting all-reduce, so this runs on a box without GPUs over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env_over):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(dict({"MQGAN_BENCH_BACKEND": "gloo"}, **env_over))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def _line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out.stdout, out.stderr[-2000:])
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_starts_two_ranks():
    out = _run(["--gpus", "2", "--dry-run"])
    assert out.returncode == 0, out.stderr[-2000:]
    line = _line(out)
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["self_launched"] is True and line["backend"] == "gloo"
    assert line["value"] is None


def test_gpus_1_runs_in_process():
    out = _run(["--gpus", "1", "--dry-run"])
    assert out.returncode == 0, out.stderr[-2000:]
    line = _line(out)
    assert line["n_gpus"] == 1 and line["self_launched"] is False and line["backend"] is None


def test_launcher_and_flag_disagreeing_is_an_error():
    out = _run(["--gpus", "2", "--dry-run"], RANK="0", WORLD_SIZE="3", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29431")
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    out = _run(["--gpus", "8", "--dry-run"], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")       # r2: this ran silently on one GPU
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def test_more_ranks_than_devices_is_an_error_for_rccl():
    out = _run(["--gpus", "2"], MQGAN_BENCH_BACKEND="nccl")          # this box has no GPU: refuse instead of running something else
    assert out.returncode != 0 and "GPU(s) are visible" in out.stderr
