"""The C ABI against malformed plans (a C++ caller with a bad plan must get HDK_HIP_ERR_INVALID_ARG, not an
out-of-bounds read).  CPU half: hdk_hip_validate_plan -- host only, no device -- accepts every plan the planner
produces for a few hundred seeded random queries and rejects single-field corruptions of them, naming the field.  The
GPU half (tests/test_gpu_abi_negative.py) sends corrupted plans through the entry points that launch kernels."""
import ctypes as C

import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd._lib import lib
from hdk_amd.ir import Agg, ColRef, JoinSpec, KeyRef, QueryMustRunOnCpu, QueryUnit
from hdk_amd.plan import compile_query

from fuzz_queries import make_tables, random_query


def _plans(n=250, seed=7):
    rng = np.random.default_rng(seed)
    st = make_tables(rng, 3000, 120)
    out = []
    for i in range(n):
        q = random_query(rng, allow_join=True, projection=(i % 5 == 4))
        try:
            out.append(compile_query(st, q))
        except QueryMustRunOnCpu:
            continue
    return out


def corruptions(cp):
    """(description, mutator) pairs: each breaks exactly one field of a valid plan."""
    p = cp.plan
    muts = [
        ("abi_version", lambda q: setattr(q, "abi_version", 2)),
        ("num_cols", lambda q: setattr(q, "num_cols", A.MAX_COLS + 1)),
        ("num_targets", lambda q: setattr(q, "num_targets", 0)),
        ("num_targets high", lambda q: setattr(q, "num_targets", A.MAX_TARGETS + 3)),
        ("query_kind", lambda q: setattr(q, "query_kind", 9)),
        ("num_quals", lambda q: setattr(q, "num_quals", -1)),
        ("num_joins", lambda q: setattr(q, "num_joins", A.MAX_JOINS + 1)),
        ("key_count", lambda q: setattr(q, "key_count", A.MAX_KEYS + 1)),
    ]
    if p.num_cols:
        muts += [("column width", lambda q: setattr(q.cols[0], "width", 3)),
                 ("column kind", lambda q: setattr(q.cols[0], "kind", 7)),
                 ("column table", lambda q: setattr(q.cols[0], "table", 5)),
                 ("column buf_idx", lambda q: setattr(q.cols[0], "buf_idx", -2))]
    for t in range(p.num_targets):
        if p.targets[t].has_arg:
            muts += [(f"target {t} leaf column", lambda q, t=t: (setattr(q.targets[t].arg.leaf0, "kind", A.LEAF_COL),
                                                                 setattr(q.targets[t].arg.leaf0, "col", 200))),
                     (f"target {t} nsteps", lambda q, t=t: setattr(q.targets[t].arg, "nsteps", 9)),
                     (f"target {t} leaf kind", lambda q, t=t: setattr(q.targets[t].arg.leaf0, "kind", 11))]
            if p.targets[t].arg.nsteps:
                muts += [(f"target {t} step op", lambda q, t=t: setattr(q.targets[t].arg.steps[0], "op", 99)),
                         (f"target {t} check_width", lambda q, t=t: setattr(q.targets[t].arg.steps[0], "check_width", 3))]
            break
    muts += [("target agg", lambda q: setattr(q.targets[0], "agg", 17)),
             ("target arg_is_fp", lambda q: setattr(q.targets[0], "arg_is_fp", 5))]
    if p.query_kind in (A.Q_PERFECT_HASH, A.Q_BASELINE_HASH) and not p.output_columnar:
        for t in range(p.num_targets):
            if p.targets[t].slot_width:
                muts.append((f"target {t} slot_off", lambda q, t=t: setattr(q.targets[t], "slot_off", int(q.row_size_quad) * 8)))
                break
    if p.query_kind != A.Q_NON_GROUPED and p.query_kind != A.Q_PROJECTION:
        muts += [("entry_count", lambda q: setattr(q, "entry_count", 0)),
                 ("key_width", lambda q: setattr(q, "key_width", 3)),
                 ("key leaf column", lambda q: (setattr(q.keys[0].leaf0, "kind", A.LEAF_COL), setattr(q.keys[0].leaf0, "col", -1)))]
    if p.num_quals:
        muts += [("qual cmp", lambda q: setattr(q.quals[0], "cmp", 0)),
                 ("qual lhs column", lambda q: (setattr(q.quals[0].lhs.leaf0, "kind", A.LEAF_COL), setattr(q.quals[0].lhs.leaf0, "col", 99))),
                 ("qual rhs kind", lambda q: setattr(q.quals[0].rhs, "kind", 0)),
                 ("filter program", lambda q: (setattr(q, "num_filter_ops", 2), q.filter_ops.__setitem__(0, 0), q.filter_ops.__setitem__(1, 0)))]
    if p.num_joins:
        muts += [("join kind", lambda q: setattr(q.joins[0], "kind", 8)),
                 ("join table_idx", lambda q: setattr(q.joins[0], "table_idx", 3)),
                 ("join null_mode", lambda q: setattr(q.joins[0], "null_mode", 4)),
                 ("join key column", lambda q: (setattr(q.joins[0].outer_key.leaf0, "kind", A.LEAF_COL),
                                                setattr(q.joins[0].outer_key.leaf0, "col", 77)))]
    if p.keyless:
        muts.append(("idx_target_as_key", lambda q: setattr(q, "idx_target_as_key", 40)))
    return muts


def test_planner_output_validates_and_corruptions_do_not():
    L = lib()
    plans = _plans()
    assert len(plans) > 120
    kinds = set()
    rejected = 0
    for cp in plans:
        assert L.hdk_hip_validate_plan(C.byref(cp.plan), 0) == A.OK, L.hdk_hip_last_error()
        kinds.add((int(cp.plan.query_kind), int(cp.plan.num_joins) > 0, bool(cp.plan.output_columnar)))
        for what, mutate in corruptions(cp):
            bad = A.Plan.from_buffer_copy(cp.plan)
            mutate(bad)
            st = L.hdk_hip_validate_plan(C.byref(bad), 0)
            assert st in (A.ERR_INVALID_ARG, A.ERR_UNSUPPORTED), (what, st)
            assert L.hdk_hip_last_error()  # a message names the field
            rejected += 1
    assert len(kinds) >= 6 and rejected > 2000


def test_single_value_targets_validate_and_their_corruptions_do_not():
    L = lib()
    rng = np.random.default_rng(5)
    st = make_tables(rng, 3000, 120)
    cp = None
    for name, tab in st.tables.items():
        cols = [c for c, col in tab.columns.items() if col.type.kind == "int" and col.type.size == 8]
        if len(cols) >= 2:
            cp = compile_query(st, QueryUnit(name, groupby=[ColRef(cols[0])], force_baseline=True, baseline_entry_count=4099,
                                            targets=[KeyRef(0, "k"), Agg("single_value", ColRef(cols[1]), "sv"), Agg("count", None, "c")]))
            break
    assert cp is not None
    assert L.hdk_hip_validate_plan(C.byref(cp.plan), 0) == A.OK, L.hdk_hip_last_error()
    sv = [t for t in range(cp.plan.num_targets) if cp.plan.targets[t].agg == A.AGG_SINGLE_VALUE][0]
    for what, mutate in (("no argument", lambda q: setattr(q.targets[sv], "has_arg", 0)),
                         ("2-byte slot", lambda q: setattr(q.targets[sv], "slot_width", 2)),
                         ("aggregate kind past the last", lambda q: setattr(q.targets[sv], "agg", A.AGG_SINGLE_VALUE + 1))):
        bad = A.Plan.from_buffer_copy(cp.plan)
        mutate(bad)
        assert L.hdk_hip_validate_plan(C.byref(bad), 0) in (A.ERR_INVALID_ARG, A.ERR_UNSUPPORTED), what


def test_null_and_garbage_plans():
    L = lib()
    assert L.hdk_hip_validate_plan(None, 0) == A.ERR_INVALID_ARG
    rng = np.random.default_rng(3)
    for _ in range(200):  # random bytes: must be rejected (or, by miracle, valid) -- never crash
        raw = rng.integers(0, 256, C.sizeof(A.Plan), dtype=np.uint8).tobytes()
        p = A.Plan.from_buffer_copy(raw)
        p.abi_version = A.PLAN_ABI
        assert L.hdk_hip_validate_plan(C.byref(p), 0) in (A.OK, A.ERR_INVALID_ARG, A.ERR_UNSUPPORTED)
        assert L.hdk_hip_validate_plan(C.byref(p), 1) in (A.OK, A.ERR_INVALID_ARG, A.ERR_UNSUPPORTED)
    # host-only entry points that take a plan run the same check
    q = C.c_int64(0)
    bad = A.Plan()
    assert L.hdk_hip_baseline_table_quads(C.byref(bad), 10, C.byref(q)) == A.ERR_INVALID_ARG


def test_layout_only_plans_of_the_reduction_matrices_validate():
    """hdk_hip_reduce_buffers takes plans that only describe the output layout (the reference's reduction works from
    the QueryMemoryDescriptor alone): the reference's Reduce.* / ReduceRandomGroups.* descriptors pass the layout
    check, and corrupting the layout half is still caught."""
    import rs_matrix as M
    from test_resultset_matrices import DOC, RANDOM, REDUCE
    L = lib()
    n = 0
    for case in REDUCE + RANDOM:
        lay = M.make_layout(DOC, case)
        if not M.supported_by_library(lay):
            continue
        plan = M.make_plan(lay)
        assert L.hdk_hip_validate_plan(C.byref(plan), 1) == A.OK, (case["name"], L.hdk_hip_last_error())
        bad = A.Plan.from_buffer_copy(plan)
        bad.targets[0].agg = 23
        assert L.hdk_hip_validate_plan(C.byref(bad), 1) == A.ERR_INVALID_ARG
        bad = A.Plan.from_buffer_copy(plan)
        bad.entry_count = 0
        assert L.hdk_hip_validate_plan(C.byref(bad), 1) == A.ERR_INVALID_ARG
        n += 1
    assert n >= 70
