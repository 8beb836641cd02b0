"""GPU parity for the join variants of tests/join_variants_cases.py: bucketized (DATE) tables,
IS NOT DISTINCT FROM builds and probes, SEMI / ANTI joins -- the device builds against the oracle's builds, and every
query through every kernel family against the oracle's result."""
import ctypes as C

import numpy as np
import pytest

from hdk_amd import _abi as A

from join_variants_cases import make_variants
from test_gpu_projection import _sorted_rows
from test_oracle_golden import decode_keyed
from test_projection import run_projection_oracle
from util import assert_buffers_equal, oracle_join_tables, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case():
    return make_variants(seed=6, nf=150_000, nd=3_000)


def _slot_of_row(st, info):
    """slot of every inner row as the fill computes it (fill_hash_join_buff_impl, HashJoinRuntime.cpp:197-240); -1 = dropped"""
    col = st.get(info["inner_table"]).columns[info["inner_cols"][0]]
    a = np.concatenate(col.fragments).astype(np.int64)
    isn = a == info["null_val"]
    if info["col_types"][0] == A.JC_SMALL_DATE:
        a = a * 86400
    a = np.where(isn, info["translated_null_build"], a)
    slot = (a - info["min"]) // info["bucket"]
    if not info["uses_bw_eq"]:
        slot[isn] = -1
    return slot


def test_variant_builds_match_the_oracle(oracle, gpu_executor_factory, case):
    """hdk_hip_fill_hash_join_buff[_bucketized] (for_semi_join 0 / 1, uses_bw_eq 0 / 1), hdk_hip_fill_one_to_many_hash_table
    [_bucketized], the keyed semi fill -- against orc_fill_* (HashJoinRuntime.cpp:197-293,770-853; JoinHashImpl.h:55-97)."""
    st, _, cases, _ = case
    ex = gpu_executor_factory(st)
    seen = set()
    for name, q, _ in cases:
        cp = ex.compile(q)
        info = cp.join_infos[0]
        want = oracle_join_tables(oracle, st, cp)[0]
        table = ex._build_join_table(cp, 0)
        got = ex.mgr.to_host(table.ptr, want.nbytes, 0, want.dtype)
        kind, n = info["kind"], info["entry_count"]
        seen.add((kind, bool(info["bucketized"]), info["uses_bw_eq"], info["for_semi_join"]))
        if kind == A.JOIN_ONE_TO_ONE and not info["for_semi_join"]:
            assert np.array_equal(got, want), name
        elif kind == A.JOIN_ONE_TO_ONE:
            # first row wins -- on the device ANY row of the key may be first: same slots filled, each with a row of that slot
            assert np.array_equal(got >= 0, want >= 0), name
            slot = _slot_of_row(st, info)
            filled = np.nonzero(got >= 0)[0]
            assert filled.size and np.array_equal(slot[got[filled]], filled), name
        elif kind == A.JOIN_ONE_TO_MANY:
            assert np.array_equal(got[:2 * n], want[:2 * n]), name  # pos and count are deterministic
            gp, gc, gi, wi = got[:n], got[n:2 * n], got[2 * n:], want[2 * n:]
            for k in np.nonzero(gp >= 0)[0]:
                assert sorted(gi[gp[k]:gp[k] + gc[k]]) == sorted(wi[gp[k]:gp[k] + gc[k]]), (name, k)
        else:
            assert kind == A.JOIN_KEYED_ONE_TO_ONE and info["for_semi_join"], name
            kc, w = len(info["inner_cols"]), info["key_width"]
            g = decode_keyed(got.view(np.uint8), n, kc, w, True, info["num_elems"])
            wnt = decode_keyed(want.view(np.uint8), n, kc, w, True, info["num_elems"])
            assert set(g) == set(wnt), name
            inner = st.get(info["inner_table"])
            cols = [np.concatenate(inner.columns[c].fragments) for c in info["inner_cols"]]
            for key, (rid,) in g.items():
                assert tuple(int(c[rid]) for c in cols) == key, name
    # every build variant of the ABI was driven
    assert {(A.JOIN_ONE_TO_ONE, True, 0, 0), (A.JOIN_ONE_TO_MANY, True, 0, 0), (A.JOIN_ONE_TO_ONE, False, 1, 0),
            (A.JOIN_ONE_TO_MANY, False, 1, 0), (A.JOIN_ONE_TO_MANY, True, 1, 0), (A.JOIN_ONE_TO_ONE, False, 0, 1),
            (A.JOIN_ONE_TO_ONE, True, 0, 1), (A.JOIN_KEYED_ONE_TO_ONE, False, 0, 1)} <= seen


def test_variant_queries_on_every_kernel_family(oracle, gpu_executor_factory, case):
    st, _, cases, _ = case
    ex = gpu_executor_factory(st)
    kernels = set()
    for name, q, _ in cases:
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0, name
        for flags in (0, A.LAUNCH_FORCE_GENERIC, A.LAUNCH_FORCE_SCALAR, A.LAUNCH_FORCE_GLOBAL_ATOMICS):
            step = ex.prepare(cp, flags=flags)
            kernels.add(step.kernel_names().split(",")[0])
            res = step.run()
            assert_buffers_equal(cp, res.buffer, want)
            step.free()
    # inner one-to-one probes with a bucket / a translated NULL ride the batched interpreter, everything with loops the
    # row-at-a-time one
    assert {"hdk_scan_agg_vec_join", "hdk_scan_agg_generic", "hdk_scan_agg_global"} <= kernels, kernels


def test_variant_queries_without_fused_tables(oracle, gpu_executor_factory, case):
    """the batched probe of the REFERENCE table layout (probe_join_g) with a bucket and with a translated NULL"""
    st, _, cases, _ = case
    ex = gpu_executor_factory(st)
    ex.fuse_join_tables = False
    for name, q, _ in cases:
        if name not in ("date_one_to_one", "date_one_to_one_group", "bw_eq_one_to_one", "semi_dups", "semi_date"):
            continue
        cp, want, err = run_oracle(oracle, st, q)
        step = ex.prepare(cp)
        assert step.kernel_names().startswith("hdk_scan_agg_vec_join"), (name, step.kernel_names())
        assert_buffers_equal(cp, step.run().buffer, want)
        step.free()


def test_variant_projections(oracle, gpu_executor_factory, case):
    st, _, _, proj_cases = case
    ex = gpu_executor_factory(st)
    for name, q, _ in proj_cases:
        cp, want, err, nrows = run_projection_oracle(oracle, st, q)
        assert err == 0 and nrows > 0, name
        for flags in (0, A.LAUNCH_FORCE_SCALAR):
            step = ex.prepare(cp, flags=flags)
            res = step.run()
            assert res.total_matched == nrows, (name, step.kernel_names())
            assert np.array_equal(_sorted_rows(cp, res.buffer, nrows), _sorted_rows(cp, want, nrows)), name
            step.free()


def test_small_date_columns_in_filters_keys_and_targets(oracle, gpu_executor_factory, case):
    """HDK_COL_SMALL_DATE outside joins: a DATE column as a filter operand, as a perfect-hash key (bucket 86400:
    get_group_value_fast with a bucket) and as MIN / MAX / COUNT argument, on every strategy."""
    from hdk_amd.ir import Agg, Cmp, ColRef, ExtractYear, KeyRef, Lit, QueryUnit
    st = case[0]
    ex = gpu_executor_factory(st)
    F = ColRef
    qs = [QueryUnit("fact", quals=[Cmp(F("day"), ">", Lit(18100 * 86400))], groupby=[F("b")],
                    targets=[KeyRef(0, "b"), Agg("count", F("day"), "c"), Agg("min", F("day"), "mn"), Agg("max", F("day16"), "mx")]),
          QueryUnit("ddup", groupby=[F("day")], targets=[KeyRef(0, "day"), Agg("count", None, "c"), Agg("sum", F("v"), "s")]),
          QueryUnit("fact", groupby=[ExtractYear(F("day"))], targets=[KeyRef(0, "y"), Agg("count", None, "c")])]
    for q in qs:
        cp, want, err = run_oracle(oracle, st, q)
        assert err == 0
        for flags in (0, A.LAUNCH_FORCE_SCALAR, A.LAUNCH_FORCE_GLOBAL_ATOMICS):
            assert_buffers_equal(cp, ex.execute(cp, flags=flags).buffer, want)
    assert ex.compile(qs[1]).plan.key_bucket[0] == 86400
