"""Differential fuzz, GPU half: seeded random plans (filters, all join kinds, 0-2 keys, perfect and
baseline hash, row-wise and columnar, every aggregate) through the default kernels AND the forced
alternatives, each compared with the oracle (bit-exact integers, 1e-6 relative fp sums)."""
import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd import result_set as rs
from hdk_amd.ir import QueryMustRunOnCpu
from hdk_amd.plan import compile_query

from fuzz_queries import make_tables, make_tables_wide, random_query, random_query_wide
from test_gpu_baseline import _check_rows
from test_gpu_projection import _sorted_rows
from test_projection import run_projection_oracle
from util import assert_buffers_equal, run_oracle

import os

pytestmark = pytest.mark.gpu

# HDK_FUZZ_SEEDS="100:160" adds a range of seeds for a soak run (scripts/gpu/soak.sh); the default set stays small
_EXTRA = os.environ.get("HDK_FUZZ_SEEDS", "")
_EXTRA_SEEDS = list(range(*map(int, _EXTRA.split(":")))) if _EXTRA else []
_ROWS = int(os.environ.get("HDK_FUZZ_ROWS", "60000"))  # soak runs also scale the fact table (many tiles per block)


def _compare(cp, got, want, float32_atol=0.0):
    # (a one-to-many join multiplies the rows: at soak sizes a float accumulator sees 1e8 additions, and the oracle's
    # row-order float sum itself is only good to a few 1e-3 relative)
    f32_rtol = 2e-4 if _ROWS <= 60_000 else 1e-2  # (soak at 1 M rows: a float sum of -1.75e8 over ~1e7 addends of |x| < 100 was 3.5e-3 off in the ORACLE)
    if cp.plan.query_kind == A.Q_BASELINE_HASH:
        _check_rows(cp, got, want, float32_rtol=f32_rtol, float32_atol=float32_atol)  # slot placement is insertion-order dependent
    else:
        assert_buffers_equal(cp, got, want, float32_rtol=f32_rtol, float32_atol=float32_atol)


@pytest.mark.parametrize("seed", [1, 2, 3] + _EXTRA_SEEDS)
def test_random_aggregate_plans(oracle, gpu_executor_factory, seed):
    _aggregate_fuzz(oracle, gpu_executor_factory, seed, make_tables, random_query, strict=seed in (1, 2, 3))


@pytest.mark.parametrize("seed", [11, 12] + _EXTRA_SEEDS)
def test_random_wide_aggregate_plans(oracle, gpu_executor_factory, seed):
    """The same differential loop over the wider generator: float32 measures (float accumulators), OR / NOT filter
    trees, extract(year) / decimal-cast keys, 64-bit COUNT."""
    _aggregate_fuzz(oracle, gpu_executor_factory, seed + 5000, make_tables_wide, random_query_wide, strict=False)


def _aggregate_fuzz(oracle, gpu_executor_factory, seed, make, random_query, strict):
    rng = np.random.default_rng(seed)
    st = make(rng, _ROWS, 700 if _ROWS <= 60_000 else 20_000)
    # float accumulators: the oracle adds in float in row order (error ~ rows x 2^-24 x |partial sum|), the device rounds
    # once; the f32 column is N(0, 30), so sums wander by a few thousand at most over the (joined) rows of a query
    f32_atol = 2e-5 * _ROWS if _ROWS <= 60_000 else 1e-3 * _ROWS  # (the keyed one-to-many join of the generator matches ~185 rows each)
    ex = gpu_executor_factory(st)
    ran, kernels, div0 = 0, set(), 0
    for i in range(40):
        q = random_query(rng)
        try:
            cp, want, err = run_oracle(oracle, st, q)
        except QueryMustRunOnCpu:
            continue
        if err in (A.ERR_DIV_BY_ZERO, A.ERR_OVERFLOW_OR_UNDERFLOW):
            # the device must report the same error (record_error_code, QE/RuntimeFunctions.cpp:1123-1135), whatever
            # kernel runs the plan
            from hdk_amd._lib import HdkHipError
            for flags in (0, A.LAUNCH_FORCE_GLOBAL_ATOMICS, A.LAUNCH_FORCE_SCALAR):
                exd = gpu_executor_factory(st)
                exd.fuse_join_tables = flags == 0
                with pytest.raises(HdkHipError) as ei:
                    exd.execute(cp, flags=flags)
                assert ei.value.code == err, (seed, i, q, flags)
            div0 += 1
            continue
        assert err == 0, (seed, i, q)
        step = ex.prepare(cp)
        kernels.add(step.kernel_names().split(",")[0])
        res = step.run()
        step.free()
        try:
            _compare(cp, res.buffer, want, f32_atol)
            for flags in (A.LAUNCH_FORCE_GLOBAL_ATOMICS, A.LAUNCH_FORCE_SCALAR, A.LAUNCH_FORCE_PARTITIONED):
                if flags == A.LAUNCH_FORCE_PARTITIONED and cp.plan.query_kind != A.Q_BASELINE_HASH:
                    continue  # (the radix-partitioned path only exists for open-addressing plans)
                if flags == A.LAUNCH_FORCE_SCALAR and any(j["kind"] != A.JOIN_ONE_TO_ONE for j in cp.join_infos):
                    continue  # already row-at-a-time
                ex2 = gpu_executor_factory(st)
                ex2.fuse_join_tables = False
                _compare(cp, ex2.execute(cp, flags=flags).buffer, want, f32_atol)
        except AssertionError as e:
            raise AssertionError(f"seed {seed} query {i}: {q}\n{e}") from e
        ran += 1
    if strict:  # (soak seeds and the wide generator only have to agree with the oracle)
        assert ran >= 25 and len(kernels) >= 3, (ran, kernels)
    else:
        assert ran >= 10, ran


@pytest.mark.parametrize("seed", [77, -78] + _EXTRA_SEEDS + [-s for s in _EXTRA_SEEDS])
def test_random_projection_plans(oracle, gpu_executor_factory, seed):
    """(negative seeds: the wider generator)"""
    rng = np.random.default_rng(abs(seed))
    st = (make_tables_wide if seed < 0 else make_tables)(rng, 50_000, 500)
    ex = gpu_executor_factory(st)
    ran = 0
    for i in range(25):
        q = (random_query_wide if seed < 0 else random_query)(rng, projection=True)
        try:
            cp, want, err, nrows = run_projection_oracle(oracle, st, q)
        except QueryMustRunOnCpu:
            continue
        if err:
            continue
        res = ex.execute(cp)
        assert res.total_matched == nrows, (i, q)
        assert np.array_equal(_sorted_rows(cp, res.buffer, nrows), _sorted_rows(cp, want, nrows)), (i, q)
        ran += 1
    assert ran >= (15 if seed == 77 else 4), ran  # (the wider generator draws float projections, which the library rejects)
