"""bench.py's N > 1 control flow on the box the driver has: two ranks on ONE GPU, launched exactly as the driver launches
them (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), collectives staged through the host
(HDK_BENCH_BACKEND=gloo: the gather + fold of perfect-hash tables, the tuple exchange of the open-addressing group-by and
its forced fallback, the exchange of partial tables).  Every `checks` entry of the printed line must hold.  The nccl
branch differs in the transport only (Comm.all_gather / all_reduce / the all-to-all calls)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(config, rows, extra_env=None, scaling="strong", ranks=2):
    env = dict(os.environ, HDK_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1",
           "--config", config, "--rows", str(rows), "--scaling", scaling, "--extra", "none", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    return _headline(r.stdout, lines[0])


def _headline(stdout, text):
    """The driver's contract for the line: the LAST stdout line, compact, with the roofline object in it."""
    assert stdout.rstrip("\n").splitlines()[-1] == text
    assert len(text) < 4096, len(text)
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "checks", "detail"):
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert os.path.exists(os.path.join(ROOT, line["detail"]))
    return line


def _all_checks_hold(line):
    assert line["checks"], line
    for k, v in line["checks"].items():
        assert v is not False, (k, line["checks"])


def test_c2_two_ranks_gather_and_fold():
    line = _run("c2", 64_000_000)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert "all-gather" in line["config"]["parallelism"] and line["config"]["fragments_per_gpu"] == 1
    _all_checks_hold(line)
    assert line["checks"]["groups"] == 64 and line["checks"]["oracle_bit_exact_on_sample"] is True


def test_q3_two_ranks_weak_scaling():
    line = _run("q3", 48_000_000, scaling="weak")
    assert line["scaling"] == "weak" and line["config"]["rows_per_gpu"] == 48_000_000
    _all_checks_hold(line)


def test_c5_two_ranks_tuple_exchange():
    line = _run("c5", 64_000_000, {"HDK_BENCH_EXCHANGE": "tuples"})
    assert "tuples scattered to owner segments" in line["config"]["parallelism"], line["config"]
    ex = line["exchange"]
    assert ex["tuple_bytes"] == 8 and all(ex["ms"][k] > 0 for k in ("scatter", "all_to_all", "aggregate"))
    assert ex["bytes_sent_over_xgmi_per_gpu"] > 0
    _all_checks_hold(line)
    assert line["checks"]["sum_of_sums"] is True and line["checks"]["idempotent"] is True
    assert line["exchange_model"]["mode"] in ("tuples", "tables") and line["exchange_model"]["link_gbps_assumed"] == 64.0


def test_c5_two_ranks_forced_table_exchange():
    line = _run("c5", 64_000_000, {"HDK_BENCH_EXCHANGE": "tables"})
    assert "owner partition of the partial table" in line["config"]["parallelism"], line["config"]
    assert line["merge"]["ms"] > 0 and line["merge"]["bytes_sent_over_xgmi_per_gpu"] > 0
    _all_checks_hold(line)


def _run_rccl_single_rank(config, rows, extra_env=None):
    """ONE rank, backend nccl (= RCCL), the N > 1 step forced (HDK_BENCH_SINGLE_RANK_COLLECTIVES): process-group init with
    a device id, all-gather / all-reduce / all-to-all / barrier on RCCL next to the library's launches on the explicit
    stream -- everything of the RCCL branch that one GPU can run."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HDK_BENCH_SINGLE_RANK_COLLECTIVES="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env.pop("HDK_BENCH_BACKEND", None)
    env.update(extra_env or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--config", config,
           "--rows", str(rows), "--extra", "none", "--no-cpu-baseline", "--no-multi-gpu-emulation"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return _headline(r.stdout, lines[0])


def test_rccl_branch_with_one_rank_gather_and_fold():
    line = _run_rccl_single_rank("c2", 64_000_000)
    assert line["n_gpus"] == 1 and "all-gather" in line["config"]["parallelism"], line["config"]
    _all_checks_hold(line)
    assert line["checks"]["groups"] == 64


def test_rccl_branch_with_one_rank_table_exchange():
    """The owner partition of the rank's partial table, RCCL all-to-all with unequal splits, owner re-insert."""
    line = _run_rccl_single_rank("c5", 32_000_000, {"HDK_BENCH_EXCHANGE": "tables"})
    assert "owner partition of the partial table" in line["config"]["parallelism"], line["config"]
    assert line["merge"]["ms"] > 0
    _all_checks_hold(line)


def test_rccl_branch_with_one_rank_tuple_exchange():
    """One rank, one owner: scatter to the (single) owner segment, the EQUAL-split all-to-all on RCCL, owner aggregation --
    the tuple exchange's collective on the real transport, as far as one GPU allows."""
    line = _run_rccl_single_rank("c5", 32_000_000, {"HDK_BENCH_EXCHANGE": "tuples"})
    assert "tuples scattered to owner segments" in line["config"]["parallelism"], line["config"]
    ex = line["exchange"]
    assert ex["tuple_bytes"] == 8 and all(ex["ms"][k] > 0 for k in ("scatter", "all_to_all", "aggregate"))
    _all_checks_hold(line)
    assert line["checks"]["sum_of_sums"] is True and line["checks"]["idempotent"] is True
    assert line["checks"]["groups"] > 0 and line["checks"]["row_count"] is True


def test_plain_python_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (the form the driver uses at N = 1): bench.py starts the two ranks itself
    as a child torch.distributed.run and the line says n_gpus 2, confirmed by an all-reduce of ones."""
    env = dict(os.environ, HDK_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "c2",
           "--rows", "64000000", "--extra", "none", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = _headline(r.stdout, lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen_by_collective"] == 2
    _all_checks_hold(line)


def test_world_size_mismatch_is_refused():
    """A launcher environment for another job size: no line, non-zero exit (never a 1-GPU number under --gpus 2)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--rows", "32000000",
           "--extra", "none", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout[-1000:]
    assert "WORLD_SIZE" in r.stderr


def test_default_command_prints_a_parsable_headline():
    """`python bench.py` exactly as the driver runs it at N = 1 (BASELINE size, every secondary config, CPU baseline, end-to-end
    leg, emulations): the last stdout line is the compact headline with `roofline` and `cpu_baseline` filled in."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HDK_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = _headline(r.stdout, r.stdout.rstrip("\n").splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["rows"] == 1_000_000_000 and line["dtype"] == "int64"
    assert 0.5 < line["roofline"]["frac"] <= 1.0 and line["roofline"]["traffic"]
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["unit"] == "rows/s"
    _all_checks_hold(line)
    assert len(line["configs"]) >= 13 and all(ok for _, _, ok in line["configs"].values()), line["configs"]
    with open(os.path.join(ROOT, line["detail"])) as f:
        full = json.load(f)
    assert full["value"] == line["value"] and len(full["configs"]) == len(line["configs"])
