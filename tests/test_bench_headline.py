"""bench.py's last stdout line must stay something the driver can parse: a compact (< 4 KB) headline carrying BASELINE.json's
metric, `roofline` and `cpu_baseline`, with the secondary configs reduced to name -> [ms_per_step, frac, checks_ok] and
everything else in a side file.  Built here from a canned full result (round 5's own 20 KB line, which the driver could not
parse) -- no GPU needed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (importing bench touches neither torch nor HIP)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU = ("value", "unit", "cores", "kind", "sample")


def _canned():
    with open(os.path.join(ROOT, "profiles", "r05_bench_default.json")) as f:
        return json.load(f)


def test_headline_is_compact_and_complete():
    full = _canned()
    assert len(json.dumps(full)) > 16000  # (the shape that broke the driver's parser)
    h = bench.headline(full, "gpurun_out/bench_detail.json")
    s = json.dumps(h)
    assert len(s) < bench.HEADLINE_MAX_BYTES, len(s)
    assert "\n" not in s
    for k in REQUIRED:
        assert k in h, k
    for k in ROOFLINE:
        assert k in h["roofline"], k
    for k in CPU:
        assert k in h["cpu_baseline"], k
    assert h["value"] == full["value"] and h["ms_per_step"] == full["ms_per_step"]
    assert h["config"]["workload"].startswith("C2") and "model" not in h["config"]
    assert abs(h["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-4
    assert h["roofline"]["bound"] == "hbm" and h["roofline"]["peak"] == 8000.0
    assert h["detail"] == "gpurun_out/bench_detail.json"
    # every secondary config of the full line survives as a triple
    assert len(h["configs"]) == len(full["configs"])
    for name, (ms, frac, ok) in h["configs"].items():
        assert ms > 0 and 0 < frac <= 1 and ok is True, name
    assert set(h["checks"]) == set(full["checks"])


def test_headline_never_exceeds_the_limit_even_when_inflated():
    full = _canned()
    big = dict(full, configs=full["configs"] * 12)
    for i, c in enumerate(big["configs"]):
        big["configs"][i] = dict(c, metric=f"rows/sec, cfg{i:03d}", config=dict(c["config"], name=f"cfg{i:03d}_with_a_long_name_" + "x" * 20))
    h = bench.headline(big, "gpurun_out/bench_detail.json")
    assert len(json.dumps(h)) < bench.HEADLINE_MAX_BYTES
    for k in REQUIRED:
        assert k in h, k


def test_emit_writes_the_side_file_and_prints_one_last_line(tmp_path, capsys, monkeypatch):
    full = _canned()
    monkeypatch.setenv("HDK_BENCH_DETAIL", str(tmp_path / "d" / "detail.json"))
    bench.emit(full)
    out = capsys.readouterr().out
    assert out.endswith("\n") and out.count("\n") == 1
    h = json.loads(out)
    assert h["roofline"]["frac"] and h["cpu_baseline"]["value"]
    with open(tmp_path / "d" / "detail.json") as f:
        assert json.load(f) == full


def test_a_failed_check_shows_in_the_triple():
    full = _canned()
    full["configs"][0]["checks"]["sum_equals_torch_gather_sum"] = False
    h = bench.headline(full)
    assert list(h["configs"].values())[0][2] is False
