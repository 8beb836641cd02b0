"""bench.py's job-size rules, checked without a GPU: a launcher environment of another size is refused before anything is
imported, and `python bench.py --gpus 2` without a launcher starts two ranks itself (here they stop at "no HIP device" --
the parent relays their exit code instead of printing a one-rank line)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra, drop=()):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") + tuple(drop):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=300)


def test_world_size_mismatch_is_refused_before_any_work():
    for ws, gpus in (("1", "2"), ("4", "2"), ("2", "1")):
        r = _bench(["--gpus", gpus], {"RANK": "0", "WORLD_SIZE": ws})
        assert r.returncode != 0 and "WORLD_SIZE" in r.stderr, (ws, gpus, r.stderr[-500:])
        assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_no_launcher_starts_the_ranks_as_a_child():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-side check (tests/test_gpu_bench_multirank.py runs the real thing)")
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--extra", "none", "--no-cpu-baseline"], {})
    assert r.returncode != 0
    # both ranks came up through torch.distributed.run and each said why it stopped
    assert r.stderr.count("bench.py needs an MI355X") >= 2, r.stderr[-2000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
