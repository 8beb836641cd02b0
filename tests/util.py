"""Shared helpers for the parity tests: run a QueryUnit through the oracle (CPU restatement of the
reference's row function) and compare it with the HIP path bit for bit."""
import numpy as np

from hdk_amd import _abi as A
from hdk_amd.plan import (align8, columnar_init_vals, columnar_slot_offsets, compact_init_vals, compile_query,
                          eff_key_count)


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
    """Build the join tables of a plan with the oracle's restatement of the CPU builders: perfect
    one-to-one / one-to-many (HashJoinRuntime.cpp:197-293,770-853) or keyed (:357-573,723-950)."""
    import ctypes as C
    L = O.lib()
    tables = []
    for ji, info in enumerate(cp.join_infos):
        inner = storage.get(info["inner_table"])
        kind = info["kind"]
        cols = info["inner_cols"]
        semi = info["for_semi_join"]
        if kind in (A.JOIN_ONE_TO_ONE, A.JOIN_ONE_TO_MANY):
            entries = info["entry_count"]  # the normalised slot count (HashEntryInfo::getNormalizedHashEntryCount)
            frs = inner.columns[cols[0]].fragments
            chunks = O.make_join_chunks(frs)
            ti = A.JoinColumnTypeInfo(info["elem_sz"], info["min"], info["max"], info["null_val"], info["uses_bw_eq"],
                                      info["col_types"][0], info["translated_null_build"])
            if kind == A.JOIN_ONE_TO_ONE:
                buff = np.empty(entries, dtype=np.int32)
                L.orc_init_hash_join_buff(buff.ctypes.data, entries, -1)
                rc = L.orc_fill_hash_join_buff(buff.ctypes.data, -1, semi, C.cast(chunks, C.c_void_p), len(frs),
                                               C.byref(ti), info["bucket"])
                assert rc == 0
            else:
                buff = np.empty(2 * entries + inner.num_rows, dtype=np.int32)
                L.orc_init_hash_join_buff(buff.ctypes.data, buff.size, -1)
                L.orc_fill_one_to_many_hash_table(buff.ctypes.data, entries, -1, C.cast(chunks, C.c_void_p), len(frs),
                                                  C.byref(ti), info["bucket"])
            tables.append(buff)
            continue
        kc, w, entries = len(cols), info["key_width"], info["entry_count"]
        keep = []
        jcs = (A.JoinColumn * kc)()
        tis = (A.JoinColumnTypeInfo * kc)()
        for k, c in enumerate(cols):
            frs = inner.columns[c].fragments
            chunks = O.make_join_chunks(frs)
            keep.append(chunks)
            jcs[k] = A.JoinColumn(C.cast(chunks, C.c_void_p).value, C.sizeof(chunks), len(frs), inner.num_rows,
                                  info["elem_szs"][k])
            tis[k] = A.JoinColumnTypeInfo(info["elem_szs"][k], info["mins"][k], info["maxs"][k], info["null_vals"][k],
                                          0, info["col_types"][k], 0)
        if kind == A.JOIN_KEYED_ONE_TO_ONE:
            buff = np.empty(entries * (kc + 1) * w, dtype=np.uint8)
            L.orc_init_baseline_hash_join_buff(buff.ctypes.data, entries, kc, w, 1, -1)
            rc = L.orc_fill_baseline_hash_join_buff_semi(buff.ctypes.data, entries, -1, semi, kc, w, jcs, tis)
            assert rc == 0, rc
        else:
            buff = np.empty(entries * kc * w + (2 * entries + inner.num_rows) * 4, dtype=np.uint8)
            L.orc_init_baseline_hash_join_buff(buff.ctypes.data, entries, kc, w, 0, -1)
            rc = L.orc_fill_one_to_many_baseline_hash_table(buff.ctypes.data, entries, -1, kc, w, jcs, tis)
            assert rc == 0, rc
        tables.append(buff)
    return tables


def oracle_init_buffer(O, cp, entry_count=None, fill=0):
    """A freshly initialised output buffer written by the ORACLE's restatement of the reference's init
    kernels (orc_init_group_by_buffer / orc_init_columnar_group_by_buffer <- QE/GpuInitGroups.cu:17-166),
    called with the arguments QueryMemoryInitializer hands them (QE/QueryMemoryInitializer.cpp:1054-1156).
    Nothing of the product's own buffer image (hdk_amd.plan.init_buffer_host) is involved.  Padding bytes
    (behind 4-byte keys, between columnar columns) are written by neither the reference nor the oracle;
    they keep `fill` (0, which is also what the product's init kernels leave there)."""
    p = cp.plan
    n = int(entry_count if entry_count is not None else p.entry_count)
    L = O.lib()
    if p.query_kind == A.Q_NON_GROUPED:
        # out_vec slots start at init_agg_vals (QueryExecutionContext.cpp:452-458)
        return np.ascontiguousarray(cp.init_vals, dtype=np.int64).copy()
    if p.output_columnar:
        offs = columnar_slot_offsets(cp, n)
        nk = eff_key_count(p)
        total = offs[-1] + n * cp.slot_widths[-1] if offs else nk * align8(n * 8)
        buf = np.full(align8(total) // 8, fill, dtype=np.int64)
        iv = np.ascontiguousarray(columnar_init_vals(cp), dtype=np.int64)
        sizes = np.ascontiguousarray(cp.slot_widths, dtype=np.int8)
        L.orc_init_columnar_group_by_buffer(buf.ctypes.data, iv.ctypes.data, n, nk, len(cp.slot_widths),
                                            sizes.ctypes.data, 1, int(p.keyless), 8)
        return buf
    iv = np.ascontiguousarray(compact_init_vals(cp), dtype=np.int64)
    buf = np.full(n * int(p.row_size_quad), fill, dtype=np.int64)
    L.orc_init_group_by_buffer(buf.ctypes.data, iv.ctypes.data, n, eff_key_count(p), int(p.key_width),
                               int(p.row_size_quad), int(p.keyless), 1)
    return buf


def run_oracle(O, storage, q_or_cp, frag_ids=None):
    cp = q_or_cp if hasattr(q_or_cp, "plan") else compile_query(storage, q_or_cp)
    buf = oracle_init_buffer(O, cp)
    hf = host_fragments(O, storage, cp, frag_ids)
    jt = oracle_join_tables(O, storage, cp)
    err = O.run_plan(cp.plan, hf, buf, jt)
    return cp, buf, err


def assert_buffers_equal(cp, got, want, fp_rtol=1e-6, float32_rtol=2e-4, float32_atol=0.0):
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
        if tg.arg_is_fp == A.FP_SLOT_FLOAT and tg.agg in (A.AGG_SUM, A.AGG_AVG):
            # float accumulator in the low 4 bytes: the reference adds row by row in float, the device folds block
            # partials held in double and rounds once -- compared within float32 summation error; the slot's other
            # bytes (the init pattern) are compared exactly
            gi, wi = gs[s].astype(np.int64), ws[s].astype(np.int64)
            assert np.array_equal(gi >> 32, wi >> 32)
            a = (gi & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
            b = (wi & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
            np.testing.assert_allclose(a, b, rtol=float32_rtol, atol=float32_atol)
            for k in range(1, nsl):
                assert np.array_equal(gs[s + k], ws[s + k])
        elif tg.arg_is_fp and tg.agg in (A.AGG_SUM, A.AGG_AVG):
            a, b = gs[s].view(np.float64), ws[s].view(np.float64)
            np.testing.assert_allclose(a, b, rtol=fp_rtol, atol=0)
            for k in range(1, nsl):
                assert np.array_equal(gs[s + k], ws[s + k])
        else:
            for k in range(nsl):
                assert np.array_equal(gs[s + k], ws[s + k]), f"target {t} slot {k}"
        s += nsl
