"""GPU parity of the C-ABI primitives around the scan kernel: buffer init, join-table build
(one-to-one / one-to-many, reference KATs of Tests/JoinHashTableTest.cpp:133-260), device reduction."""
import ctypes as C

import numpy as np
import pytest

from hdk_amd import _abi as A
from hdk_amd._lib import check, lib
from hdk_amd.hip_mgr import HipMgr
from hdk_amd.ir import Agg, ColRef, KeyRef, QueryUnit
from hdk_amd.plan import compile_query
from hdk_amd.storage import ArrowStorage

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mgr():
    return HipMgr()


def test_hipmgr_contract(mgr):
    assert mgr.getDeviceCount() >= 1 and mgr.getSubGroupSize() == 64
    props = mgr.getDeviceProperties(0)
    assert props.num_cu >= 64 and props.arch_name.decode().startswith("gfx9")
    a = np.arange(1000, dtype=np.int64)
    d = mgr.to_device(a, 0)
    d2 = mgr.alloc(a.nbytes, 0)
    mgr.copyDeviceToDevice(d2.ptr, d.ptr, a.nbytes, 0, 0)
    assert np.array_equal(mgr.to_host(d2.ptr, a.nbytes, 0), a)
    mgr.setDeviceMem(d2.ptr, 0xFF, a.nbytes, 0)
    assert (mgr.to_host(d2.ptr, a.nbytes, 0) == -1).all()
    mgr.zeroDeviceMem(d2.ptr, a.nbytes, 0)
    assert (mgr.to_host(d2.ptr, a.nbytes, 0) == 0).all()
    # pinned host memory (CudaMgr::allocatePinnedHostMem) as the source of an asynchronous upload
    host, hptr = mgr.pinned_array((4096,), np.int64)
    host[:] = np.arange(4096) * 3
    d3 = mgr.alloc(host.nbytes, 0)
    mgr.copyHostToDeviceAsync(d3.ptr, hptr, host.nbytes, 0)
    mgr.synchronizeStream(0)
    assert np.array_equal(mgr.to_host(d3.ptr, host.nbytes, 0), np.arange(4096) * 3)
    del host
    mgr.freePinnedHostMem(hptr)


def test_measurement_helpers(mgr):
    """hdk_hip_mgr_measure_hbm (what bench.py reports as roofline.peak_measured) and the HIP-event log of
    HDK_HIP_LAUNCH_RECORD_EVENTS (hdk_hip_collect_scan_times: DeviceClock, QE/DeviceKernel.cpp:25-43)."""
    from hdk_amd.executor import Executor
    L = lib()
    copy, read = C.c_double(0), C.c_double(0)
    check(L.hdk_hip_mgr_measure_hbm(0, 256 << 20, 3, C.byref(copy), C.byref(read)))
    assert 500 < copy.value < 9000 and 500 < read.value < 9000 and read.value > copy.value * 0.8
    st = ArrowStorage()
    rng = np.random.default_rng(0)
    st.import_numpy("t", {"k": rng.integers(0, 64, 1 << 20), "v": rng.integers(-5, 5, 1 << 20)}, fragment_size=1 << 18)
    ex = Executor(st, 0, mgr)
    n = C.c_int32(0)
    check(L.hdk_hip_collect_scan_times(0, None, 0, C.byref(n)))  # drain what earlier tests may have left
    step = ex.prepare(QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("sum", ColRef("v"))]),
                      flags=A.LAUNCH_RECORD_EVENTS)
    for _ in range(3):
        step.run()
    ms = (C.c_float * 8)()
    check(L.hdk_hip_collect_scan_times(0, ms, 8, C.byref(n)))
    assert n.value == 3 and all(0 < ms[i] < 50 for i in range(3))
    check(L.hdk_hip_collect_scan_times(0, ms, 8, C.byref(n)))
    assert n.value == 0  # collected once
    step.free()


@pytest.mark.parametrize("columnar", [False, True])
@pytest.mark.parametrize("nkeys", [1, 2])
def test_init_kernels_match_oracle_init(mgr, oracle, columnar, nkeys):
    """hdk_hip_init_[columnar_]group_by_buffer against the oracle's restatement of QE/GpuInitGroups.cu:17-166
    (not against anything the product computes)."""
    st = ArrowStorage()
    st.import_numpy("t", {"a": np.arange(50, dtype=np.int64) % 7, "b": (np.arange(50, dtype=np.int64) % 3) + (2**40),
                          "v": np.arange(50, dtype=np.int32)})
    gb = [ColRef("a"), ColRef("b")][:nkeys]
    q = QueryUnit("t", groupby=gb, output_columnar=columnar, force_baseline=(nkeys == 2),
                  targets=[KeyRef(0), Agg("count"), Agg("min", ColRef("v")), Agg("avg", ColRef("v"))])
    cp = compile_query(st, q)
    from hdk_amd.executor import Executor
    step = Executor(st, 0, mgr).prepare(cp) if cp.plan.query_kind != A.Q_BASELINE_HASH else None
    from util import oracle_init_buffer
    want = oracle_init_buffer(oracle, cp)
    if step is None:  # baseline plans cannot launch yet on every build; initialise through the ABI directly
        from hdk_amd.plan import columnar_init_vals, compact_init_vals
        L = lib()
        out = mgr.alloc(cp.buffer_bytes, 0)
        p = cp.plan
        if columnar:
            d_init = mgr.to_device(columnar_init_vals(cp), 0)  # zero-width slots have no init value
            d_sz = mgr.to_device(np.array(cp.slot_widths, dtype=np.int8), 0)
            check(L.hdk_hip_init_columnar_group_by_buffer(out.ptr, d_init.ptr, p.entry_count, p.key_count,
                                                          len(cp.slot_widths), d_sz.ptr, 1, p.keyless, 8, 256, 1024, 0, None))
        else:
            d_init = mgr.to_device(compact_init_vals(cp), 0)
            check(L.hdk_hip_init_group_by_buffer(out.ptr, d_init.ptr, p.entry_count, p.key_count, p.key_width,
                                                 p.row_size_quad, p.keyless, 1, 256, 1024, 0, None))
        mgr.synchronizeStream(0)
        got = mgr.to_host(out.ptr, cp.buffer_bytes, 0)
    else:
        step.init_output()
        mgr.synchronizeStream(0)
        got = mgr.to_host(step.out_ptr, cp.buffer_bytes, 0)
    assert np.array_equal(got[:cp.buffer_quads], want[:cp.buffer_quads])


def _device_join_column(mgr, arrays, elem_sz):
    bufs = [mgr.to_device(a, 0) for a in arrays]
    chunks = (A.JoinChunk * len(arrays))()
    rid = 0
    for i, a in enumerate(arrays):
        chunks[i].col_buff = bufs[i].ptr
        chunks[i].num_elems = a.size
        chunks[i].row_id = rid
        rid += a.size
    raw = np.frombuffer(bytes(chunks), dtype=np.uint8)
    d_chunks = mgr.to_device(raw, 0)
    return A.JoinColumn(d_chunks.ptr, raw.nbytes, len(arrays), rid, elem_sz), (bufs, d_chunks)


def test_join_build_kats_and_random(mgr, oracle):
    L = lib()
    O = oracle
    rng = np.random.default_rng(3)
    perm = rng.permutation(100_000).astype(np.int64) + 17
    cases = [([np.arange(10, dtype=np.int32)], True), ([np.array([0, 1, 2, 4, 5, 6, 7, 9], dtype=np.int32)], True),
             ([np.array([0, 1, 2, 3, 4, 0, 1, 2, 3, 4], dtype=np.int32)], False),
             ([np.ascontiguousarray(perm[i::3]) for i in range(3)], True),
             ([rng.integers(5, 5000, 20_000).astype(np.int32), rng.integers(5, 5000, 7_001).astype(np.int32)], False)]
    for arrays, unique in cases:
        allk = np.concatenate(arrays)
        lo, hi = int(allk.min()), int(allk.max())
        esz = arrays[0].dtype.itemsize
        nullv = A.NULL_INT if esz == 4 else A.NULL_BIGINT
        ti = A.JoinColumnTypeInfo(esz, lo, hi, nullv, 0, A.JC_SIGNED, 0)
        jc, keep = _device_join_column(mgr, arrays, esz)
        n = hi - lo + 1
        # one-to-one
        table = mgr.alloc(n * 4, 0)
        d_err = mgr.to_device(np.zeros(1, dtype=np.int32), 0)
        check(L.hdk_hip_init_hash_join_buff(table.ptr, n, -1, 0, None))
        check(L.hdk_hip_fill_hash_join_buff(table.ptr, -1, 0, d_err.ptr, jc, ti, 0, None))
        mgr.synchronizeStream(0)
        err = int(mgr.to_host(d_err.ptr, 4, 0, np.int32)[0])
        want = np.empty(n, dtype=np.int32)
        O.lib().orc_init_hash_join_buff(want.ctypes.data, n, -1)
        chunks = O.make_join_chunks(arrays)
        rc = O.lib().orc_fill_hash_join_buff(want.ctypes.data, -1, 0, C.cast(chunks, C.c_void_p), len(arrays), C.byref(ti), 1)
        assert (err == 0) == unique == (rc == 0)
        if unique:
            assert np.array_equal(mgr.to_host(table.ptr, n * 4, 0, np.int32), want)
        # one-to-many: [pos | count | ids]; ids inside a bin are an unordered set (JoinHashTableTest toSet())
        total = 2 * n + allk.size
        t2 = mgr.alloc(total * 4, 0)
        check(L.hdk_hip_init_hash_join_buff(t2.ptr, 2 * n, -1, 0, None))
        check(L.hdk_hip_fill_one_to_many_hash_table(t2.ptr, A.HashEntryInfo(n, 1), -1, jc, ti, 0, None))
        mgr.synchronizeStream(0)
        got = mgr.to_host(t2.ptr, total * 4, 0, np.int32)
        w2 = np.empty(total, dtype=np.int32)
        O.lib().orc_init_hash_join_buff(w2.ctypes.data, 2 * n, -1)
        O.lib().orc_fill_one_to_many_hash_table(w2.ctypes.data, n, -1, C.cast(chunks, C.c_void_p), len(arrays), C.byref(ti), 1)
        assert np.array_equal(got[:2 * n], w2[:2 * n])  # pos and count are deterministic
        gp, gc, gi = got[:n], got[n:2 * n], got[2 * n:]
        wi = w2[2 * n:]
        for k in np.nonzero(gp >= 0)[0][:2000]:
            assert sorted(gi[gp[k]:gp[k] + gc[k]]) == sorted(wi[gp[k]:gp[k] + gc[k]])


def _fill_one_to_one(mgr, oracle, arrays, ti, bucket=1, payload=None, two_levels=None, monkeypatch=None):
    """Device table (and fused table) against orc_fill_hash_join_buff + a by-hand fused form.  Returns (err, rc)."""
    L, O = lib(), oracle
    esz = arrays[0].dtype.itemsize
    jc, keep = _device_join_column(mgr, arrays, esz)
    rng_ = ti.max_val - ti.min_val + 1 + (1 if ti.uses_bw_eq else 0)
    n = (rng_ + bucket - 1) // bucket if bucket > 1 else rng_
    table = mgr.alloc(n * 4, 0)
    d_err = mgr.to_device(np.zeros(1, dtype=np.int32), 0)
    check(L.hdk_hip_init_hash_join_buff(table.ptr, n, -1, 0, None))
    fused = None
    if payload is None:
        check(L.hdk_hip_fill_hash_join_buff_bucketized(table.ptr, -1, 0, d_err.ptr, jc, ti, bucket, 0, None))
    else:
        nc = len(payload)
        dcols = [mgr.to_device(c, 0) for c in payload]
        ptrs = (C.c_void_p * max(nc, 1))(*[d.ptr for d in dcols])
        widths = (C.c_int32 * max(nc, 1))(*[c.dtype.itemsize for c in payload])
        kinds = (C.c_int32 * max(nc, 1))(*[A.COL_DOUBLE if c.dtype == np.float64 else A.COL_INT for c in payload])
        fused = mgr.alloc(n * (1 + nc) * 8, 0)
        rows = sum(a.size for a in arrays)
        sb = L.hdk_hip_join_build_scratch_bytes(rows, n, nc)
        scratch = mgr.alloc(max(sb, 8), 0) if nc != 2 else None  # two payload columns: the library's own scratch
        check(L.hdk_hip_fill_hash_join_buff_fused(table.ptr, -1, 0, d_err.ptr, jc, ti, bucket, ptrs, widths, kinds, nc, fused.ptr,
                                                  scratch.ptr if scratch else None, sb if scratch else 0, 0, None))
    mgr.synchronizeStream(0)
    err = int(mgr.to_host(d_err.ptr, 4, 0, np.int32)[0])
    want = np.empty(n, dtype=np.int32)
    O.lib().orc_init_hash_join_buff(want.ctypes.data, n, -1)
    chunks = O.make_join_chunks(arrays)
    rc = O.lib().orc_fill_hash_join_buff(want.ctypes.data, -1, 0, C.cast(chunks, C.c_void_p), len(arrays), C.byref(ti), bucket)
    if rc == 0 and err == 0:
        got = mgr.to_host(table.ptr, n * 4, 0, np.int32)
        assert np.array_equal(got, want)
        if payload is not None:
            gf = mgr.to_host(fused.ptr, n * (1 + len(payload)) * 8, 0, np.int64).reshape(n, 1 + len(payload))
            assert np.array_equal(gf[:, 0], want.astype(np.int64))
            ok = want >= 0
            for c, col in enumerate(payload):
                w = np.zeros(n, dtype=np.int64)
                w[ok] = col[want[ok]].view(np.int64) if col.dtype == np.float64 else col[want[ok]].astype(np.int64)
                assert np.array_equal(gf[:, 1 + c], w), c
    return err, rc


def test_partitioned_join_build(mgr, oracle, monkeypatch):
    """The one-to-one build by slot-range partitions (join_build_part.h) against orc_fill_hash_join_buff
    (HashJoinRuntime.cpp:197-293): forced on small inputs; one slice, several slices, two scatter levels; chunked key
    columns of every width; NULL keys skipped, or filed under the translated NULL of a kBwEq join; DATE buckets; sparse
    ranges; duplicates (-1) and keys outside the statistics (-2); a key distribution that overflows a sub-slab (the armed
    atomic kernels build the table); the fused form with one to three payload columns from the caller's or the library's
    scratch, and with four through the gather."""
    monkeypatch.setenv("HDK_HIP_BUILD_PARTITION_MIN_ROWS", "1")
    L = lib()
    rng = np.random.default_rng(77)

    def ti_of(allk, esz, nullv, bw=0, ctype=None, bucket=1, lo=None, hi=None):
        live = allk[allk != nullv]
        lo = int(live.min()) if lo is None else lo
        hi = int(live.max()) if hi is None else hi
        tr = (hi // bucket + 1) if (bw and bucket > 1) else hi + 1
        return A.JoinColumnTypeInfo(esz, lo, hi, nullv, bw, A.JC_SIGNED if ctype is None else ctype, tr if bw else 0)

    for two in (None, "1", "3"):
        if two:
            monkeypatch.setenv("HDK_HIP_BUILD_TWO_LEVELS", two)
        else:
            monkeypatch.delenv("HDK_HIP_BUILD_TWO_LEVELS", raising=False)
        # a permutation in three chunks, 100 K keys (4 slices), int64
        perm = rng.permutation(100_000).astype(np.int64) + 17
        arrays = [np.ascontiguousarray(perm[i::3]) for i in range(3)]
        assert _fill_one_to_one(mgr, oracle, arrays, ti_of(perm, 8, A.NULL_BIGINT)) == (0, 0)
        # sparse: 30 K keys over a range of 300 K, int32, NULLs skipped
        k32 = (rng.permutation(30_000) * 10 - 5000).astype(np.int32)
        k32[rng.random(k32.size) < 0.02] = A.NULL_INT
        assert _fill_one_to_one(mgr, oracle, [k32[:11_111], k32[11_111:]], ti_of(k32, 4, A.NULL_INT)) == (0, 0)
        # int16 keys, one NULL row filed under max + 1 (kBwEq)
        k16 = rng.permutation(3_000).astype(np.int16)
        k16[5] = A.NULL_SMALLINT
        assert _fill_one_to_one(mgr, oracle, [k16], ti_of(k16, 2, A.NULL_SMALLINT, bw=1)) == (0, 0)
        # DATE in seconds, bucket 86400
        days = rng.permutation(40_000).astype(np.int64)[:25_000]
        secs = days * 86400
        assert _fill_one_to_one(mgr, oracle, [secs], ti_of(secs, 8, A.NULL_BIGINT, bucket=86400), bucket=86400) == (0, 0)
        # duplicates: -1 on both sides
        dup = rng.integers(0, 50_000, 60_000).astype(np.int64)
        err, rc = _fill_one_to_one(mgr, oracle, [dup], ti_of(dup, 8, A.NULL_BIGINT))
        assert err == -1 and rc != 0
        # fused forms
        pay = [rng.integers(-2**40, 2**40, perm.size, dtype=np.int64), rng.integers(-100, 100, perm.size).astype(np.int32),
               rng.normal(size=perm.size), rng.integers(0, 9, perm.size).astype(np.int16)]
        for nc in (0, 1, 2, 3, 4):  # (0: a join none of whose inner columns is read still gets [row id] entries)
            assert _fill_one_to_one(mgr, oracle, arrays, ti_of(perm, 8, A.NULL_BIGINT), payload=pay[:nc]) == (0, 0), nc
    monkeypatch.delenv("HDK_HIP_BUILD_TWO_LEVELS", raising=False)
    # stale statistics: a key above max
    st_k = np.arange(5_000, dtype=np.int64)
    jc, keep = _device_join_column(mgr, [st_k], 8)
    ti = A.JoinColumnTypeInfo(8, 0, 3_999, A.NULL_BIGINT, 0, A.JC_SIGNED, 0)
    table = mgr.alloc(4_000 * 4, 0)
    d_err = mgr.to_device(np.zeros(1, dtype=np.int32), 0)
    check(L.hdk_hip_init_hash_join_buff(table.ptr, 4_000, -1, 0, None))
    check(L.hdk_hip_fill_hash_join_buff(table.ptr, -1, 0, d_err.ptr, jc, ti, 0, None))
    mgr.synchronizeStream(0)
    assert int(mgr.to_host(d_err.ptr, 4, 0, np.int32)[0]) == -2
    assert np.array_equal(mgr.to_host(table.ptr, 16_000, 0, np.int32), np.arange(4_000, dtype=np.int32))
    # nearly every row in the first of eight level-1 bins (8 slices each): its sub-slabs overflow, the armed atomic kernels
    # build the table
    monkeypatch.setenv("HDK_HIP_BUILD_TWO_LEVELS", "3")
    skew = rng.permutation(8 * 32768)[:200_000].astype(np.int64)
    skew = np.concatenate([skew, np.array([64 * 32768 - 1], dtype=np.int64)])
    pay = [rng.integers(0, 1000, skew.size, dtype=np.int64)]
    assert _fill_one_to_one(mgr, oracle, [skew], ti_of(skew, 8, A.NULL_BIGINT), payload=pay) == (0, 0)
    assert _fill_one_to_one(mgr, oracle, [skew], ti_of(skew, 8, A.NULL_BIGINT)) == (0, 0)
    # (with payloads a slice has 4 096 slots: 100 K rows in the first of sixteen 131 K-slot bins overflow its sub-slabs)
    monkeypatch.setenv("HDK_HIP_BUILD_TWO_LEVELS", "5")
    skew = np.concatenate([rng.permutation(32 * 4096)[:100_000].astype(np.int64), np.array([64 * 32768 - 1], dtype=np.int64)])
    pay = [rng.integers(0, 1000, skew.size, dtype=np.int64), rng.integers(0, 1000, skew.size).astype(np.int32)]
    assert _fill_one_to_one(mgr, oracle, [skew], ti_of(skew, 8, A.NULL_BIGINT), payload=pay) == (0, 0)
    monkeypatch.delenv("HDK_HIP_BUILD_TWO_LEVELS", raising=False)
    assert L.hdk_hip_join_build_scratch_bytes(0, 100, 1) == 0 and L.hdk_hip_join_build_scratch_bytes(10**6, 10**6, 9) == 0


def test_device_reduce_perfect_hash(mgr, oracle):
    """hdk_hip_reduce_buffers == the oracle's ResultSetReduction restatement, partials in order."""
    rng = np.random.default_rng(21)
    st = ArrowStorage()
    n = 40_000
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    v[rng.random(n) < 0.3] = A.NULL_BIGINT
    f = rng.normal(size=n)
    st.import_numpy("t", {"k": rng.integers(0, 50, n).astype(np.int64), "v": v, "f": f}, fragment_size=5_000)
    from hdk_amd.executor import Executor
    from util import run_oracle
    for columnar in (False, True):
        q = QueryUnit("t", groupby=[ColRef("k")], output_columnar=columnar,
                      targets=[KeyRef(0), Agg("sum", ColRef("v")), Agg("min", ColRef("v")), Agg("count"),
                               Agg("avg", ColRef("v")), Agg("max", ColRef("f"))])
        cp = compile_query(st, q)
        ex = Executor(st, 0, mgr)
        # partial per fragment (GPU), then device merge vs oracle merge of the same partials
        partials = [ex.execute(cp, frag_ids=[fr]).buffer.copy() for fr in range(8)]
        want = partials[0].copy()
        for pbuf in partials[1:]:
            assert oracle.reduce(cp.plan, want, cp.entry_count, pbuf, cp.entry_count, cp.init_vals) == 0
        d_parts = [mgr.to_device(pb, 0) for pb in partials]
        that = (C.c_void_p * 7)(*[d.ptr for d in d_parts[1:]])
        counts = (C.c_uint32 * 7)(*([cp.entry_count] * 7))
        d_err = mgr.to_device(np.zeros(1, dtype=np.int32), 0)
        iv = np.ascontiguousarray(cp.init_vals)
        check(lib().hdk_hip_reduce_buffers(C.byref(cp.plan), d_parts[0].ptr, cp.entry_count, that, counts, 7,
                                           iv.ctypes.data, d_err.ptr, 0, None))
        mgr.synchronizeStream(0)
        got = mgr.to_host(d_parts[0].ptr, cp.buffer_bytes, 0)
        assert np.array_equal(got, want)  # same order of partials => bit-identical, fp included
        # and the merged partials equal the single-pass result
        cp2, full, err = run_oracle(oracle, st, cp)
        from util import assert_buffers_equal
        assert_buffers_equal(cp, got, full)


@pytest.mark.parametrize("width", [4, 8])
def test_keyed_join_build_on_device(mgr, oracle, width):
    """hdk_hip_{init,fill}_baseline_hash_join_buff / fill_one_to_many_baseline_hash_table: the decoded table
    ({key tuple: row ids}, JoinHashTableTest's toSet()) equals the oracle's; the reference KATs
    (Tests/JoinHashTableTest.cpp:355-440) and a larger random case with duplicates and NULLs."""
    from test_oracle_golden import decode_keyed
    L = lib()
    dt = np.int32 if width == 4 else np.int64
    rng = np.random.default_rng(width)
    big_a = rng.integers(0, 3000, 50_000).astype(dt)
    big_b = rng.integers(-20, 20, 50_000).astype(dt)
    big_a[rng.random(50_000) < 0.01] = np.iinfo(dt).min
    uniq = rng.permutation(200_000)[:60_000].astype(dt)
    cases = [([np.array([0, 1, 3], dtype=dt)] * 2, True), ([np.array([0, 1, 3, 3], dtype=dt)] * 2, False),
             ([big_a, big_b], False), ([uniq, (uniq % 7).astype(dt), (uniq // 3).astype(dt)], True)]
    for cols, unique in cases:
        kc, n = len(cols), len(cols[0])
        entries = 2 * n
        parts = [[c[: n // 3], c[n // 3:]] for c in cols]
        jcs, tis, keep = (A.JoinColumn * kc)(), (A.JoinColumnTypeInfo * kc)(), []
        ojcs, otis, okeep = (A.JoinColumn * kc)(), (A.JoinColumnTypeInfo * kc)(), []
        for k in range(kc):
            jc, kp = _device_join_column(mgr, parts[k], width)
            keep.append(kp)
            jcs[k] = jc
            tis[k] = A.JoinColumnTypeInfo(width, int(cols[k].min()), int(cols[k].max()), int(np.iinfo(dt).min), 0, A.JC_SIGNED, 0)
            chunks = oracle.make_join_chunks(parts[k])
            okeep.append(chunks)
            ojcs[k] = A.JoinColumn(C.cast(chunks, C.c_void_p).value, C.sizeof(chunks), 2, n, width)
            otis[k] = tis[k]
        d_err = mgr.to_device(np.zeros(1, dtype=np.int32), 0)
        one_bytes = entries * (kc + 1) * width
        t1 = mgr.alloc(one_bytes, 0)
        check(L.hdk_hip_init_baseline_hash_join_buff(t1.ptr, entries, kc, width, 1, -1, 0, None))
        check(L.hdk_hip_fill_baseline_hash_join_buff(t1.ptr, entries, -1, 0, kc, width, 1, d_err.ptr, jcs, tis, 0, None))
        mgr.synchronizeStream(0)
        err = int(mgr.to_host(d_err.ptr, 4, 0, np.int32)[0])
        want1 = np.empty(one_bytes, dtype=np.uint8)
        oracle.lib().orc_init_baseline_hash_join_buff(want1.ctypes.data, entries, kc, width, 1, -1)
        rc = oracle.lib().orc_fill_baseline_hash_join_buff(want1.ctypes.data, entries, -1, kc, width, ojcs, otis)
        assert (err == 0) == unique == (rc == 0), (err, rc)
        if unique:
            assert decode_keyed(mgr.to_host(t1.ptr, one_bytes, 0, np.uint8), entries, kc, width, True, n) == \
                decode_keyed(want1, entries, kc, width, True, n)
        dict_bytes = entries * kc * width
        many_bytes = dict_bytes + (2 * entries + n) * 4
        t2 = mgr.alloc(many_bytes, 0)
        check(mgr_zero(mgr, d_err))
        check(L.hdk_hip_init_baseline_hash_join_buff(t2.ptr, entries, kc, width, 0, -1, 0, None))
        check(L.hdk_hip_fill_baseline_hash_join_buff(t2.ptr, entries, -1, 0, kc, width, 0, d_err.ptr, jcs, tis, 0, None))
        check(L.hdk_hip_fill_one_to_many_baseline_hash_table(t2.ptr + dict_bytes, t2.ptr, entries, -1, kc, width, jcs, tis,
                                                             0, None))
        mgr.synchronizeStream(0)
        assert int(mgr.to_host(d_err.ptr, 4, 0, np.int32)[0]) == 0
        want2 = np.empty(many_bytes, dtype=np.uint8)
        oracle.lib().orc_init_baseline_hash_join_buff(want2.ctypes.data, entries, kc, width, 0, -1)
        assert oracle.lib().orc_fill_one_to_many_baseline_hash_table(want2.ctypes.data, entries, -1, kc, width, ojcs, otis) == 0
        got = decode_keyed(mgr.to_host(t2.ptr, many_bytes, 0, np.uint8), entries, kc, width, False, n)
        assert got == decode_keyed(want2, entries, kc, width, False, n)
        if n <= 4:
            assert got == ({(0, 0): [0], (1, 1): [1], (3, 3): [2]} if unique else {(0, 0): [0], (1, 1): [1], (3, 3): [2, 3]})
        for b in (t1, t2, d_err):
            b.free()


def mgr_zero(mgr, buf):
    mgr.zeroDeviceMem(buf.ptr, 4, 0)
    return 0


@pytest.mark.timeout(900)
def test_merge_gathered_on_device_orders_with_torch(mgr, oracle):
    """distributed.merge_gathered_on_device on torch's default stream: the fold must be ordered after the
    torch ops that produced `gathered` (it used to run on the library's private stream) and before the
    torch ops that read the result -- no explicit synchronisation in between."""
    import torch
    from hdk_amd import distributed as D
    from hdk_amd.executor import Executor
    rng = np.random.default_rng(5)
    n = 300_000
    st = ArrowStorage()
    v = rng.integers(-10**6, 10**6, n).astype(np.int64)
    v[rng.random(n) < 0.1] = A.NULL_BIGINT
    st.import_numpy("t", {"k": rng.integers(0, 200, n).astype(np.int64), "v": v}, fragment_size=50_000)
    q = QueryUnit("t", groupby=[ColRef("k")], targets=[KeyRef(0), Agg("sum", ColRef("v")), Agg("count"), Agg("avg", ColRef("v"))])
    ex = Executor(st, 0, mgr)
    cp = ex.compile(q)
    world = 3
    parts = [ex.execute(cp, frag_ids=D.shard_fragments(6, world, r)).buffer.view(np.int64)[:cp.buffer_quads] for r in range(world)]
    want = parts[0].copy()
    for r in range(1, world):
        assert oracle.reduce(cp.plan, want, cp.entry_count, parts[r], cp.entry_count, cp.init_vals) == 0
    for _ in range(5):
        big = torch.randn(1 << 24, device="cuda")  # keep the default stream busy before the gather lands
        big = big * 2 + 1
        gathered = torch.cat([torch.from_numpy(p.copy()).cuda() for p in parts]) + (big[:1] * 0).to(torch.int64)
        merged = D.merge_gathered_on_device(cp, gathered, world, 0)
        got = (merged + 0).cpu().numpy()
        assert np.array_equal(got, want)
