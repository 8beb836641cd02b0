"""Shared helpers for the parity tests: run a QueryUnit through the oracle (CPU restatement of the
reference's row function) and compare it with the HIP path bit for bit."""
import numpy as np

from hdk_amd import _abi as A
from hdk_amd.plan import compile_query, init_buffer_host


def host_fragments(O, storage, cp, frag_ids=None):
    outer = storage.get(cp.query.table)
    if frag_ids is None:
        frag_ids = list(range(outer.num_fragments))
    frags = []
    for f in frag_ids:
        cols = []
        for (tn, cn, slot) in cp.input_cols:
            t = storage.get(tn)
            if slot == 0:
                cols.append(t.columns[cn].fragments[f])
            else:
                fr = t.columns[cn].fragments
                cols.append(fr[0] if len(fr) == 1 else np.concatenate(fr))
        frags.append(cols)
    ntab = 1 + len(cp.inner_tables)
    num_rows = np.zeros(max(len(frag_ids) * ntab, 1), dtype=np.int64)
    for i, f in enumerate(frag_ids):
        num_rows[i * ntab] = outer.frag_rows[f]
        for ti, tn in enumerate(cp.inner_tables):
            num_rows[i * ntab + 1 + ti] = storage.get(tn).num_rows
    return O.HostFragments(frags, num_rows[:len(frag_ids) * ntab], ntab)


def oracle_join_tables(O, storage, cp):
    """Build the perfect one-to-one join tables with the oracle's restatement of the CPU build."""
    import ctypes as C
    tables = []
    for info in cp.join_infos:
        inner = storage.get(info["inner_table"])
        entries = info["max"] - info["min"] + 1
        buff = np.empty(entries, dtype=np.int32)
        O.lib().orc_init_hash_join_buff(buff.ctypes.data, entries, -1)
        frs = inner.columns[info["inner_col"]].fragments
        chunks = O.make_join_chunks(frs)
        ti = A.JoinColumnTypeInfo(info["elem_sz"], info["min"], info["max"], info["null_val"], 0, A.JC_SIGNED, 0)
        rc = O.lib().orc_fill_hash_join_buff(buff.ctypes.data, -1, 0, C.cast(chunks, C.c_void_p), len(frs),
                                             C.byref(ti), 1)
        assert rc == 0
        tables.append(buff)
    return tables


def run_oracle(O, storage, q_or_cp, frag_ids=None):
    cp = q_or_cp if hasattr(q_or_cp, "plan") else compile_query(storage, q_or_cp)
    buf = init_buffer_host(cp)
    hf = host_fragments(O, storage, cp, frag_ids)
    jt = oracle_join_tables(O, storage, cp)
    err = O.run_plan(cp.plan, hf, buf, jt)
    return cp, buf, err


def assert_buffers_equal(cp, got, want, fp_rtol=1e-6):
    """Bit-exact for every integer slot / key; fp SUM/AVG slots within `fp_rtol` relative
    (north_star tolerance: GPU summation order differs from the CPU's row order)."""
    from hdk_amd import result_set as rs
    n = cp.entry_count
    got = np.ascontiguousarray(got[:cp.buffer_quads])
    want = np.ascontiguousarray(want[:cp.buffer_quads])
    if not any(cp.plan.targets[t].arg_is_fp and cp.plan.targets[t].agg in (A.AGG_SUM, A.AGG_AVG)
               for t in range(cp.plan.num_targets)):
        if not np.array_equal(got, want):
            bad = np.nonzero(got != want)[0]
            raise AssertionError(f"buffers differ at {bad[:8]} (of {bad.size}): got {got[bad[:8]]} want {want[bad[:8]]}")
        return
    gk, wk = rs._key_arrays(cp, got, n), rs._key_arrays(cp, want, n)
    for a, b in zip(gk, wk):
        assert np.array_equal(a, b)
    gs, ws = rs._slot_arrays(cp, got, n), rs._slot_arrays(cp, want, n)
    s = 0
    for t in range(cp.plan.num_targets):
        tg = cp.plan.targets[t]
        nsl = 2 if tg.agg == A.AGG_AVG else 1
        if tg.arg_is_fp and tg.agg in (A.AGG_SUM, A.AGG_AVG):
            a, b = gs[s].view(np.float64), ws[s].view(np.float64)
            np.testing.assert_allclose(a, b, rtol=fp_rtol, atol=0)
            for k in range(1, nsl):
                assert np.array_equal(gs[s + k], ws[s + k])
        else:
            for k in range(nsl):
                assert np.array_equal(gs[s + k], ws[s + k]), f"target {t} slot {k}"
        s += nsl
