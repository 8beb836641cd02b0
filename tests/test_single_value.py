"""SINGLE_VALUE(x) (hdk::ir::AggType::kSingleValue): checked_single_agg_id in the row function (QE/RuntimeFunctions.cpp:
489-506,567-583,743-760; the *_shared forms, QE/cuda_mapd_rt.cu:670-782) and reduceOneSlotSingleValue over partial
results (QE/ResultSetReduction.cpp:1186-1230).  One value per group: the value; a second, different value: error 15
(Execute::ERR_SINGLE_VALUE_FOUND_MULTIPLE_VALUES).  The oracle's four runtime functions are pinned to the reference's
in test_oracle_vs_ref.py / test_oracle_golden.py; here the plan level: oracle on the CPU, the device against it."""
import ctypes as C

import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd import result_set as rs
from hdk_amd.ir import Agg, Cmp, ColRef, KeyRef, Lit, QueryUnit
from hdk_amd.storage import ArrowStorage

from util import assert_buffers_equal, run_oracle

ERR = A.ERR_SINGLE_VALUE_FOUND_MULTIPLE_VALUES


def _storage(n=200_000, seed=3, frag=60_000):
    """k: 500 groups.  one/one32/oned/onef: a function of k, with NULLs (every group sees its value many times).
    two: like `one`, except that group 123 holds two different values."""
    rng = np.random.default_rng(seed)
    k = rng.integers(0, 500, n).astype(np.int64)
    one = k * 7 - 100
    one[rng.random(n) < 0.2] = A.NULL_BIGINT
    one32 = (k * 3 + 1).astype(np.int32)
    one32[rng.random(n) < 0.2] = A.NULL_INT
    oned = k * 0.5
    oned[rng.random(n) < 0.2] = np.frombuffer(np.int64(A.NULL_DOUBLE_BITS).tobytes(), dtype=np.float64)[0]
    onef = (k * 0.25).astype(np.float32)
    onef[rng.random(n) < 0.2] = np.frombuffer(np.int32(A.NULL_FLOAT_BITS).tobytes(), dtype=np.float32)[0]
    two = k * 7 - 100
    two[rng.random(n) < 0.2] = A.NULL_BIGINT
    hit = np.flatnonzero(k == 123)
    two[hit[len(hit) // 2]] = 4242
    st = ArrowStorage()
    st.import_numpy("t", {"k": k, "kb": k * 3_000_000_019 - 2**40, "one": one, "one32": one32, "oned": oned, "onef": onef,
                          "two": two, "v": rng.integers(-100, 100, n).astype(np.int64)}, fragment_size=frag)
    return st


def _queries(col):
    sv = Agg("single_value", ColRef(col), "sv")
    return {
        "perfect_hash": QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0, "k"), sv, Agg("count", None, "c"), Agg("sum", ColRef("v"), "s")]),
        "perfect_hash_columnar": QueryUnit("t", groupby=[ColRef("k")], output_columnar=True, targets=[KeyRef(0, "k"), Agg("min", ColRef("v"), "m"), sv]),
        "baseline_hash": QueryUnit("t", groupby=[ColRef("kb")], force_baseline=True, baseline_entry_count=2003,
                                   targets=[KeyRef(0, "k"), Agg("count", None, "c"), sv]),
        "non_grouped": QueryUnit("t", quals=[Cmp(ColRef("k"), "=", Lit(77))], targets=[Agg("count", None, "c"), sv]),
    }


@pytest.mark.parametrize("col", ["one", "one32", "oned", "onef"])
def test_oracle_single_value(oracle, col):
    st = _storage()
    for name, q in _queries(col).items():
        cp, buf, err = run_oracle(oracle, st, q)
        assert err == 0, (name, col)
        assert any(t.agg == A.AGG_SINGLE_VALUE and t.skip_null == 0 for t in cp.plan.targets[:cp.plan.num_targets])
        cols = rs.to_columns(cp, buf)
        if name == "non_grouped":
            want = {"one": 77 * 7 - 100, "one32": 77 * 3 + 1, "oned": 77 * 0.5, "onef": 77 * 0.25}[col]
            assert cols["sv"][0] == want
        else:
            keys = np.asarray(cols["k"])
            kk = keys if name != "baseline_hash" else (keys + 2**40) // 3_000_000_019
            want = {"one": kk * 7 - 100, "one32": kk * 3 + 1, "oned": kk * 0.5, "onef": kk * 0.25}[col]
            assert len(kk) == 500 and np.array_equal(np.asarray(cols["sv"], dtype=np.float64), np.asarray(want, dtype=np.float64))


def test_oracle_two_values_is_error_15_and_partials_reduce(oracle):
    st = _storage()
    for name, q in _queries("two").items():
        if name == "non_grouped":
            q = QueryUnit("t", quals=[Cmp(ColRef("k"), "=", Lit(123))], targets=q.targets)
        cp, buf, err = run_oracle(oracle, st, q)
        assert err == ERR, name
    # reduceOneSlotSingleValue: per-fragment partials of a clean column fold to the whole result; partials that disagree
    # are error 15 at the fold even though every partial on its own is clean
    nfrag = len(st.get("t").frag_rows)
    for col, want_err in (("one", 0), ("two", ERR)):
        q = _queries(col)["perfect_hash"]
        cp, whole, err = run_oracle(oracle, st, q)
        parts = []
        for f in range(nfrag):
            _, b, e = run_oracle(oracle, st, cp, frag_ids=[f])
            parts.append((b, e))
        this = parts[0][0].copy()
        codes = [oracle.reduce(cp.plan, this, cp.entry_count, b, cp.entry_count, cp.init_vals) for b, _ in parts[1:]]
        if want_err == 0:
            assert not any(codes) and not any(e for _, e in parts)
            assert_buffers_equal(cp, this, whole)
        else:
            assert ERR in codes or any(e == ERR for _, e in parts)


@pytest.mark.gpu
@pytest.mark.parametrize("col", ["one", "one32", "oned", "onef"])
def test_device_single_value_matches_the_oracle(oracle, gpu_executor_factory, col):
    from test_gpu_baseline import _check_rows
    st = _storage()
    ex = gpu_executor_factory(st)
    for name, q in _queries(col).items():
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0
        step = ex.prepare(cp)
        assert step.kernel_names() == "hdk_scan_agg_global", (name, step.kernel_names())  # the final table only
        res = step.run()
        step.free()
        if cp.plan.query_kind == A.Q_BASELINE_HASH:
            _check_rows(cp, res.buffer, want)
        else:
            assert_buffers_equal(cp, res.buffer, want)


@pytest.mark.gpu
def test_device_two_values_is_error_15(oracle, gpu_executor_factory):
    import torch
    from hdk_amd import distributed as D
    from hdk_amd._lib import HdkHipError
    st = _storage()
    ex = gpu_executor_factory(st)
    for name, q in _queries("two").items():
        if name == "non_grouped":
            q = QueryUnit("t", quals=[Cmp(ColRef("k"), "=", Lit(123))], targets=q.targets)
        cp, _, err = run_oracle(oracle, st, q)
        assert err == ERR
        with pytest.raises(HdkHipError) as ei:
            ex.execute(cp)
        assert ei.value.code == ERR, name
    # hdk_hip_reduce_buffers over per-fragment partials computed on the device (the multi-GPU fold of a perfect-hash
    # table): clean partials fold to the oracle's whole result, partials that disagree leave 15 in dev_error
    nfrag = len(st.get("t").frag_rows)
    for col, want_err in (("one", 0), ("two", ERR)):
        cp, whole, _ = run_oracle(oracle, st, _queries(col)["perfect_hash"])
        parts = []
        for f in range(nfrag):
            step = ex.prepare(cp, frag_ids=[f])
            try:
                parts.append(np.array(step.run().buffer, copy=True))
            except HdkHipError as e:  # the fragment that holds both values of group 123
                assert col == "two" and e.code == ERR
                parts = None
                break
            finally:
                step.free()
        if parts is None:
            continue
        gathered = torch.from_numpy(np.concatenate(parts)).cuda()
        d_err = torch.zeros(1, dtype=torch.int32, device="cuda")
        merged = D.merge_gathered_on_device(cp, gathered, nfrag, 0, d_err=d_err)
        torch.cuda.synchronize()
        assert int(d_err.item()) == want_err
        if not want_err:
            assert_buffers_equal(cp, merged.cpu().numpy(), whole)
