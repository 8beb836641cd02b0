"""BASELINE.json's configurations as device-resident synthetic workloads, shared by bench.py
(`--config`) and the full-size GPU tests (tests/test_gpu_fullsize.py).

  c2   1 B rows, 64 keys:  SELECT key, SUM(val) ... GROUP BY key                   (headline; perfect hash)
  c3   1 B-row fact JOIN 10 M-row dim on int64 key:  SELECT SUM(fact.val + dim.dval)
  c3g  the same join, then GROUP BY dim.dval / 15625 (64 groups on the joined column), SUM(fact.val)
  c3m  the same join with another target list: SUM(fact.val), COUNT(*), MAX(dim.dval)
  c5   1 B rows, 100 M keys:  SELECT key, SUM(val) ... GROUP BY key                (open addressing, 200 M entries)
  c5s  the shard one GPU of eight sees in C5: 125 M rows drawn from the 100 M-key domain (200 M entries)
  q1..q4  taxi Q1-Q4 (Benchmarks/taxi/taxi_reduced_bench.cpp:51-89) over a 1 B-row table with the sample's domains

Columns are generated ON THE DEVICE with torch (seed = SEED + global fragment index, so any rank regenerates any
fragment), wrapped as the executor's resident chunks -- the same state HDK reaches once GpuBufferMgr has cached the
chunks.  Nothing here computes query results except `reference_checks`, which uses plain torch ops on the same
tensors as an independent cross-check of size-independent properties (sum of sums, row counts, distinct keys).
"""
import numpy as np

SEED = 20261002
FRAGMENT_ROWS = 32_000_000  # ArrowStorage default (omniscidb/ArrowStorage/ArrowStorage.h:40)


class TensorChunk:
    """A torch CUDA tensor standing in for a DeviceBuffer of the executor's chunk cache."""

    def __init__(self, tensor):
        self.tensor = tensor
        self.ptr = tensor.data_ptr()
        self.nbytes = tensor.numel() * tensor.element_size()

    def free(self):
        self.tensor = None
        self.ptr = 0


def fragment_rows(rows, fragment_size=FRAGMENT_ROWS):
    out = []
    while rows > 0:
        out.append(min(fragment_size, rows))
        rows -= out[-1]
    return out or [0]


CONFIGS = {
    # name: (default rows, algorithmic bytes per row, description)
    "c2": (1_000_000_000, 16, "C2: SELECT key, SUM(val) GROUP BY key; int64, 64 uniform keys"),
    "c3": (1_000_000_000, 16, "C3: SELECT SUM(fact.val + dim.dval) FROM fact JOIN dim(10 M rows) ON fact.fk = dim.key"),
    "c3g": (1_000_000_000, 16, "C3g: SELECT dim.dval / 15625, SUM(fact.val) FROM fact JOIN dim(10 M rows) ON fact.fk = dim.key GROUP BY 1 "
                               "(SURVEY.md 8d's star-schema variant of C3: 64 groups on the joined column)"),
    "c3m": (1_000_000_000, 16, "C3m: SELECT SUM(fact.val), COUNT(*), MAX(dim.dval) FROM fact JOIN dim(10 M rows) ON fact.fk = dim.key"),
    "c3gm": (1_000_000_000, 16, "C3gm: SELECT dim.dval % 64, SUM(fact.val) FROM fact JOIN dim(10 M rows) ON fact.fk = dim.key GROUP BY 1 "
                                "(SURVEY.md 8d's variant of C3 as written: a modulo has no expression range, so the 128-entry table is "
                                "GroupByBaselineHash)"),
    # the reference's own benchmark for small open-addressing tables (Benchmarks/synthetic_benchmark/queries/BaselineHash/
    # BH001-005.sql over create_table.py's INT columns): cast(x AS double) key, five aggregates of one column; 8 bytes per row
    "bh1": (1_000_000_000, 8, "BH001: SELECT cast(x10 AS double), count(y10), sum(y10), max(y10), min(y10), avg(y10) GROUP BY 1; 10 groups"),
    "bh2": (1_000_000_000, 8, "BH002: the same by cast(x100 AS double); 100 groups"),
    "bh3": (1_000_000_000, 8, "BH003: the same by cast(x1k AS double); 1 K groups"),
    "bh4": (1_000_000_000, 8, "BH004: the same by cast(x10k AS double); 10 K groups"),
    "bh5": (1_000_000_000, 8, "BH005: the same by cast(x100k AS double); 100 K groups"),
    # the same suite's other families (tests/syn_queries.py holds the queries): NonGroupedAgg, MultiStep, PerfectHashMultiCol
    "nga2": (1_000_000_000, 24, "NGA02: SELECT SUM(x10), SUM(y10), SUM(z10), SUM(x100), SUM(y100), SUM(z100); six INT columns, no key"),
    "nga5": (1_000_000_000, 24, "NGA05: SELECT AVG(x10), AVG(y10), AVG(z10), AVG(x100), AVG(y100), AVG(z100)"),
    "msbs1": (1_000_000_000, 12, "MSBS001: SELECT cast(x1k AS float), count(*), max(x100), max(x10), max(x10 + 1), sum(x100), sum(x10 + 1) "
                                 "GROUP BY 1; 1 K groups, open addressing (the query's post-aggregate arithmetic is above the hot path)"),
    "msphs1": (1_000_000_000, 12, "MSPHS001: the same by x1k itself; 1 K groups, perfect hash"),
    "phm2": (1_000_000_000, 12, "PHM002: SELECT x100, y10, count(z10), sum(z10), max(z10), min(z10), avg(z10) GROUP BY 1, 2; 1 K groups, "
                                "two-column perfect hash"),
    "msphs1w": (1_000_000_000, 24, "MSPHS001 over BIGINT columns (an Arrow table of int64): the same kernel, narrowing in registers"),
    "msphs1f": (1_000_000_000, 12, "MSPHS001 WHERE x10 < 8 (about 70 % of the rows pass; the filter column is one of the arguments)"),
    "msphs1o": (1_000_000_000, 12, "MSPHS001 WHERE x10 < 4 OR NOT (x100 <= 60) (an AND / OR / NOT program; 58 % of the rows pass)"),
    "c5": (1_000_000_000, 16, "C5: SELECT key, SUM(val) GROUP BY key; int64, 100 M uniform keys (open addressing)"),
    "c5s": (125_000_000, 16, "C5 per-GPU shard of 8: 125 M rows drawn from the 100 M-key domain"),
    "q1": (1_000_000_000, 4, "taxi Q1: SELECT cab_type, COUNT(*) GROUP BY cab_type"),
    "q2": (1_000_000_000, 10, "taxi Q2: SELECT passenger_count, AVG(total_amount) GROUP BY passenger_count"),
    "q3": (1_000_000_000, 10, "taxi Q3: SELECT passenger_count, extract(year from pickup_datetime), COUNT(*) GROUP BY 1, 2"),
    "q4": (1_000_000_000, 18, "taxi Q4: SELECT passenger_count, extract(year ...), cast(trip_distance as int), COUNT(*) GROUP BY 1, 2, 3"),
}


# name -> query (built from tests/syn_queries.py: the reference's synthetic benchmark suite beyond BaselineHash)
SYN_SUITE = {
    "nga2": lambda SQ: SQ.nga(2),
    "nga5": lambda SQ: SQ.nga(5),
    "msbs1": lambda SQ: SQ.msbs(1),  # (cast(x1k AS float), as the suite writes it)
    "msphs1": lambda SQ: SQ.msphs(1),
    "phm2": lambda SQ: SQ.phm(2),
    "msphs1w": lambda SQ: SQ.msphs(1),
    "msphs1f": lambda SQ: SQ.filtered(SQ.msphs(1), "x10", "<", 8),
    "msphs1o": lambda SQ: SQ.filtered_or(SQ.msphs(1), ("x10", "<", 4), ("x100", ">", 60)),
}
SYN_WIDE = ("msphs1w",)  # 8-byte columns


def _expr_columns(e, out):
    from hdk_amd.ir import ColRef
    if isinstance(e, ColRef):
        out.add(e.name)
    for attr in ("arg", "lhs", "rhs"):
        x = getattr(e, attr, None)
        if x is not None and not isinstance(x, (int, float, str)):
            _expr_columns(x, out)


def _query_columns(q):
    out = set()
    def leaves(c):
        from hdk_amd.ir import Cmp
        if isinstance(c, Cmp):
            return [c.lhs]
        return [x for attr in ("lhs", "rhs", "arg", "left", "right", "operand", "a", "b") if hasattr(c, attr) and getattr(c, attr) is not None
                for x in leaves(getattr(c, attr))]
    for e in list(q.groupby) + [t.arg for t in q.targets if getattr(t, "arg", None) is not None] + [x for c in q.quals for x in leaves(c)]:
        _expr_columns(e, out)
    return out


class Workload:
    """Tables resident on `device`, the query, and what the checks need."""

    def __init__(self, name, rows, device, mgr, frag_ids=None, fragment_size=FRAGMENT_ROWS, dim_rows=10_000_000,
                 key_domain=100_000_000, seed_offset=0, generators=None, dim_key_stride=1):
        import torch
        from hdk_amd.executor import Executor
        from hdk_amd.ir import Agg, Cast, ColRef, ExtractYear, FP64, INT32, JoinSpec, KeyRef, QueryUnit, Type
        from hdk_amd.storage import ArrowStorage, ChunkStats, Column, Table
        if name not in CONFIGS:
            raise ValueError(f"unknown config {name}; one of {sorted(CONFIGS)}")
        self.name, self.rows, self.device = name, int(rows), device
        self.alg_bytes_per_row = CONFIGS[name][1]
        self.description = CONFIGS[name][2]
        self.frag_rows = fragment_rows(self.rows, fragment_size)
        self.nfrag = len(self.frag_rows)
        # fragments this process holds (strong scaling: fragment f -> rank f mod G)
        self.frag_ids = list(range(self.nfrag)) if frag_ids is None else list(frag_ids)
        self.local_rows = int(sum(self.frag_rows[f] for f in self.frag_ids))
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.storage = ArrowStorage()
        self.ex = Executor(self.storage, device, mgr)
        self.cols = {}  # (table, column) -> {fragment: tensor}
        I64 = Type("int", 8, True)

        def table(tname, specs, frag_rows, local):
            """specs: {column: (Type, generator(f, n) -> tensor, (min, max))}"""
            columns = []
            for cname, (ctype, gen, (lo, hi)) in specs.items():
                col = Column(cname, ctype, [None] * len(frag_rows), [ChunkStats(lo, hi, False)] * len(frag_rows))
                columns.append(col)
                self.cols[(tname, cname)] = {}
                for f in local:
                    t = gen(f, frag_rows[f])
                    self.cols[(tname, cname)][f] = t
                    self.ex.cache.put((tname, cname, f), TensorChunk(t))
                    if len(frag_rows) == 1:  # (what ColumnFetcher::linearizeColumnFragments would hand out)
                        self.ex.cache.put((tname, cname, "all"), TensorChunk(t))
            self.storage.add_table(Table(tname, columns, frag_rows))

        def uniform(lo, hi, salt, dtype=torch.int64):
            def gen(f, n):
                g = torch.Generator(device=self.dev)
                g.manual_seed(SEED + seed_offset + 1000 * salt + f)
                return torch.randint(lo, hi, (n,), dtype=dtype, device=self.dev, generator=g)
            return gen

        val = (I64, uniform(-2**31, 2**31, 1), (-2**31, 2**31 - 1))
        if name == "c2":
            kgen = generators("key") if generators else uniform(0, 64, 0)
            vgen = generators("val") if generators else val[1]
            table("t", {"key": (I64, kgen, (0, 63)), "val": (I64, vgen, val[2])}, self.frag_rows, self.frag_ids)
            self.query = QueryUnit("t", groupby=[ColRef("key")], targets=[KeyRef(0, "key"), Agg("sum", ColRef("val"), "s")])
            self.key_col, self.val_col = ("t", "key"), ("t", "val")
        elif name in ("c5", "c5s"):
            self.key_domain = int(key_domain)
            table("t", {"key": (I64, uniform(0, key_domain, 0), (0, key_domain - 1)), "val": val}, self.frag_rows,
                  self.frag_ids)
            self.query = QueryUnit("t", groupby=[ColRef("key")], targets=[KeyRef(0, "key"), Agg("sum", ColRef("val"), "s")])
            self.key_col, self.val_col = ("t", "key"), ("t", "val")
        elif name in ("bh1", "bh2", "bh3", "bh4", "bh5"):
            self.bh_groups = {"bh1": 10, "bh2": 100, "bh3": 1000, "bh4": 10_000, "bh5": 100_000}[name]
            I32 = Type("int", 4, True)
            g = self.bh_groups
            table("syn", {"x": (I32, uniform(1, g + 1, 11, torch.int32), (1, g)), "y10": (I32, uniform(1, 11, 12, torch.int32), (1, 10))},
                  self.frag_rows, self.frag_ids)
            y = ColRef("y10")
            self.query = QueryUnit("syn", groupby=[Cast(ColRef("x"), FP64)],
                                   targets=[KeyRef(0, "key0"), Agg("count", y, "c"), Agg("sum", y, "s"), Agg("max", y, "mx"),
                                            Agg("min", y, "mn"), Agg("avg", y, "a")])
            self.key_col, self.val_col = ("syn", "x"), ("syn", "y10")
        elif name in SYN_SUITE:
            import os
            import sys
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
            import syn_queries as SQ
            self.query = SYN_SUITE[name](SQ)
            ctype, tdtype = (Type("int", 8, True), torch.int64) if name in SYN_WIDE else (Type("int", 4, True), torch.int32)
            names = sorted(_query_columns(self.query))
            table("syn", {c: (ctype, uniform(1, SQ.SYN_COLUMNS[c] + 1, 20 + sorted(SQ.SYN_COLUMNS).index(c), tdtype), (1, SQ.SYN_COLUMNS[c]))
                          for c in names}, self.frag_rows, self.frag_ids)
            self.syn_hi = {c: SQ.SYN_COLUMNS[c] for c in names}
        elif name in ("c3", "c3g", "c3gm", "c3m"):
            self.dim_rows = int(dim_rows)
            self.description = self.description.replace("dim(10 M rows)", f"dim({self.dim_rows / 1e6:g} M rows)")
            self.dim_key_stride = stride = int(dim_key_stride)  # > 1: a SPARSE dimension (keys k * stride: the table's range is stride x its rows)

            def dim_key(f, n):
                g = torch.Generator(device=self.dev)
                g.manual_seed(SEED + 7)
                return torch.randperm(n, dtype=torch.int64, device=self.dev, generator=g) * stride

            def fact_fk(f, n):
                return uniform(0, self.dim_rows, 2)(f, n) * stride
            table("dim", {"key": (I64, dim_key, (0, (self.dim_rows - 1) * stride)),
                          "dval": (I64, uniform(0, 10**6, 8), (0, 10**6 - 1))}, [self.dim_rows], [0])
            # the planner decides one-to-one from the inner table's data (plan._inner_keys_unique): a permutation is unique
            self.storage.get("dim").__dict__["_unique_keys_cache"] = {("key", False, 1): 1}  # (cols..., nulls_match, bucket)
            table("fact", {"fk": (I64, fact_fk, (0, (self.dim_rows - 1) * stride)), "val": val}, self.frag_rows,
                  self.frag_ids)
            j = [JoinSpec("dim", ColRef("fk"), "key")]
            dval = ColRef("dval", "dim")
            self.query = {
                "c3": QueryUnit("fact", joins=j, targets=[Agg("sum", ColRef("val") + dval, "s")]),
                "c3g": QueryUnit("fact", joins=j, groupby=[dval / 15625], targets=[KeyRef(0, "g"), Agg("sum", ColRef("val"), "s")]),
                "c3gm": QueryUnit("fact", joins=j, groupby=[dval % 64], targets=[KeyRef(0, "g"), Agg("sum", ColRef("val"), "s")]),
                "c3m": QueryUnit("fact", joins=j, targets=[Agg("sum", ColRef("val"), "s"), Agg("count", None, "c"), Agg("max", dval, "mx")]),
            }[name]
        else:  # taxi-shaped table (taxi_reduced_bench.cpp:13-24 column types, the sample's value domains)
            table("trips", {
                "cab_type": (Type("dict", 4), uniform(0, 2, 3, torch.int32), (0, 1)),
                "passenger_count": (Type("int", 2), uniform(0, 7, 4, torch.int16), (0, 6)),
                "pickup_datetime": (Type("timestamp", 8, unit="s"), uniform(1230768000, 1451606400, 5), (1230768000, 1451606399)),
                "trip_distance": (Type("decimal", 8, scale=2), uniform(0, 5000, 6), (0, 4999)),
                "total_amount": (Type("decimal", 8, scale=2), uniform(0, 20000, 9), (0, 19999)),
            }, self.frag_rows, self.frag_ids)
            self.storage.get("trips").columns["cab_type"].dictionary = ["green", "yellow"]
            pc, ts = ColRef("passenger_count"), ColRef("pickup_datetime")
            self.query = {
                "q1": QueryUnit("trips", groupby=[ColRef("cab_type")], targets=[KeyRef(0, "cab_type"), Agg("count", None, "cnt")]),
                "q2": QueryUnit("trips", groupby=[pc], targets=[KeyRef(0, "passenger_count"),
                                                             Agg("avg", ColRef("total_amount"), "avg_amount")]),
                "q3": QueryUnit("trips", groupby=[pc, ExtractYear(ts)],
                                targets=[KeyRef(0, "passenger_count"), KeyRef(1, "year"), Agg("count", None, "cnt")]),
                "q4": QueryUnit("trips", groupby=[pc, ExtractYear(ts), Cast(ColRef("trip_distance"), INT32)],
                                targets=[KeyRef(0, "passenger_count"), KeyRef(1, "year"), KeyRef(2, "distance"),
                                         Agg("count", None, "cnt")]),
            }[name]
        torch.cuda.synchronize(self.dev)
        self.compiled = self.ex.compile(self.query)

    # ---- host copies (for the oracle, on a bounded sample) ------------------------------------------------------
    def host_fragment(self, table, f):
        """numpy copies of one fragment's columns, attached to the storage so that the oracle can read them."""
        t = self.storage.get(table)
        for cname in t.column_order:
            if t.columns[cname].fragments[f] is None:
                t.columns[cname].fragments[f] = self.cols[(table, cname)][f].cpu().numpy()
        return {c: t.columns[c].fragments[f] for c in t.column_order}

    def sample_storage(self, rows=2_000_000):
        """The first `rows` rows of the outer table's first local fragment (and the whole inner table) as a small
        host-resident storage: the oracle runs the same query on it, the device result must match it bit for bit."""
        from hdk_amd.storage import ArrowStorage
        st = ArrowStorage()
        outer = self.query.table
        f0 = self.frag_ids[0]
        t = self.storage.get(outer)
        n = min(rows, t.frag_rows[f0])
        cols = {c: self.cols[(outer, c)][f0][:n].cpu().numpy() for c in t.column_order}
        st.import_numpy(outer, cols, fragment_size=max(n // 4, 1), types={c: t.columns[c].type for c in t.column_order})
        for c in t.column_order:
            st.get(outer).columns[c].dictionary = t.columns[c].dictionary
        for j in getattr(self.query, "joins", []):
            it = self.storage.get(j.inner_table)
            icols = {c: self.cols[(j.inner_table, c)][0].cpu().numpy() for c in it.column_order}
            st.import_numpy(j.inner_table, icols, types={c: it.columns[c].type for c in it.column_order})
        return st

    # ---- independent properties computed with plain torch on the same tensors -----------------------------------
    def reference_checks(self):
        """Size-independent facts about the LOCAL rows, from torch ops only (wrap-around int64 like the engine):
        c2/c5: sum of val, number of distinct keys; c3: SUM(val + dval[fk]); taxi: per-group counts of Q1/Q2 keys."""
        torch = self.torch
        out = {"rows": self.local_rows}
        if self.name in ("c2", "c5", "c5s"):
            total = 0
            for f in self.frag_ids:
                total = (total + int(self.cols[self.val_col][f].sum().item())) % (1 << 64)
            out["sum_val"] = total
        elif self.name in SYN_SUITE:
            out["syn"] = self._syn_reference()
        elif self.name.startswith("bh"):
            g = self.bh_groups
            cnt = torch.zeros(g + 1, dtype=torch.int64, device=self.dev)
            sm = torch.zeros(g + 1, dtype=torch.int64, device=self.dev)
            mx = torch.full((g + 1,), -(1 << 62), dtype=torch.int64, device=self.dev)
            mn = torch.full((g + 1,), 1 << 62, dtype=torch.int64, device=self.dev)
            for f in self.frag_ids:
                x = self.cols[("syn", "x")][f].to(torch.int64)
                y = self.cols[("syn", "y10")][f].to(torch.int64)
                cnt += torch.bincount(x, minlength=g + 1)
                sm.index_add_(0, x, y)
                mx = torch.maximum(mx, torch.zeros_like(mx).scatter_reduce_(0, x, y, "amax", include_self=False))
                mn = torch.minimum(mn, torch.full_like(mn, 1 << 62).scatter_reduce_(0, x, y, "amin", include_self=True))
                del x, y
            out["bh"] = {"count": cnt.cpu().tolist(), "sum": sm.cpu().tolist(), "max": mx.cpu().tolist(), "min": mn.cpu().tolist()}
        elif self.name in ("c3", "c3g", "c3gm", "c3m"):
            key = self.cols[("dim", "key")][0]
            dval = self.cols[("dim", "dval")][0]
            st_ = getattr(self, "dim_key_stride", 1)
            by_key = torch.empty_like(dval)
            by_key[key // st_] = dval  # dval of the dim row whose key is k * stride
            total = 0
            sums = torch.zeros(64, dtype=torch.int64, device=self.dev)
            counts = torch.zeros(64, dtype=torch.int64, device=self.dev)
            sum_val, mx = 0, -(1 << 63)
            for f in self.frag_ids:
                fk, v = self.cols[("fact", "fk")][f], self.cols[("fact", "val")][f]
                d = by_key[fk // st_]
                if self.name == "c3":
                    total = (total + int((v + d).sum().item())) % (1 << 64)
                elif self.name in ("c3g", "c3gm"):
                    g = torch.div(d, 15625, rounding_mode="trunc") if self.name == "c3g" else d % 64
                    sums.index_add_(0, g, v)
                    counts += torch.bincount(g, minlength=64)
                else:
                    sum_val = (sum_val + int(v.sum().item())) % (1 << 64)
                    mx = max(mx, int(d.max().item()))
                del d
            out["sum_val_plus_dval"] = total
            if self.name in ("c3g", "c3gm"):
                out["group_sums"] = [int(x) for x in sums.cpu().tolist()]
                out["group_counts"] = [int(x) for x in counts.cpu().tolist()]
            if self.name == "c3m":
                out["sum_val"], out["max_dval"] = sum_val, mx
        else:
            kcol = "cab_type" if self.name == "q1" else "passenger_count"
            counts = None
            for f in self.frag_ids:
                c = torch.bincount(self.cols[("trips", kcol)][f].to(torch.int64), minlength=8)
                counts = c if counts is None else counts + c
            out["key_counts"] = [int(x) for x in counts.cpu().tolist()]
            if self.name == "q2":
                sums = torch.zeros(8, dtype=torch.int64, device=self.dev)
                for f in self.frag_ids:
                    sums.index_add_(0, self.cols[("trips", "passenger_count")][f].to(torch.int64),
                                    self.cols[("trips", "total_amount")][f])
                out["key_sums"] = [int(x) for x in sums.cpu().tolist()]
        return out

    # ---- the suite's other families: every group and target from plain torch ops over the same tensors ---------------------------
    def _syn_eval(self, e, f):
        """int64 tensor of expression `e` (a column, column + / - / * literal, cast(column AS ...)) over fragment f."""
        from hdk_amd.ir import BinOp, Cast, ColRef, Lit
        torch = self.torch
        if isinstance(e, ColRef):
            return self.cols[("syn", e.name)][f].to(torch.int64)
        if isinstance(e, Cast):
            return self._syn_eval(e.arg, f)
        if isinstance(e, BinOp) and isinstance(e.rhs, Lit):
            a = self._syn_eval(e.lhs, f)
            return {"+": a + int(e.rhs.value), "-": a - int(e.rhs.value), "*": a * int(e.rhs.value)}[e.op]
        raise ValueError(f"not a suite expression: {e}")

    def _syn_reference(self):
        """{'dims': [...], 'targets': {name: list by dense group id}} -- bincount / index_add_ / scatter_reduce_ per fragment."""
        from hdk_amd.ir import Agg
        torch = self.torch
        q = self.query
        dims = []
        for k in q.groupby:
            cs = set()
            _expr_columns(k, cs)
            dims.append(self.syn_hi[cs.pop()])
        G = 1
        for d in dims:
            G *= d
        aggs = [t for t in q.targets if isinstance(t, Agg)]
        cnt = torch.zeros(G, dtype=torch.int64, device=self.dev)
        acc = {}
        for t in aggs:
            if t.arg is not None:
                init = {"max": -(1 << 62), "min": 1 << 62}.get(t.kind, 0)
                acc[t.name] = torch.full((G,), init, dtype=torch.int64, device=self.dev)
        for f in self.frag_ids:
            gid = None
            stride = 1
            for k, d in zip(q.groupby, dims):
                term = (self._syn_eval(k, f) - 1) * stride
                gid = term if gid is None else gid + term
                stride *= d
            if gid is None:
                gid = torch.zeros(self.frag_rows[f], dtype=torch.int64, device=self.dev)
            keep = None  # the suite's filters: `column cmp literal` leaves under AND / OR / NOT, over columns without NULLs

            def cond(c):
                from hdk_amd.ir import And, Cmp, Not, Or
                if isinstance(c, Cmp):
                    a, b = self._syn_eval(c.lhs, f), int(c.rhs.value)
                    return {"<": a < b, "<=": a <= b, ">": a > b, ">=": a >= b, "=": a == b, "<>": a != b}[c.op]
                kids = [cond(getattr(c, attr)) for attr in vars(c) if getattr(c, attr) is not None]
                if isinstance(c, Not):
                    return ~kids[0]
                return kids[0] & kids[1] if isinstance(c, And) else kids[0] | kids[1]
            for c in q.quals:
                m = cond(c)
                keep = m if keep is None else keep & m
            if keep is not None:
                gid = gid[keep]
            cnt += torch.bincount(gid, minlength=G)
            for t in aggs:
                if t.arg is None:
                    continue
                v = self._syn_eval(t.arg, f)
                if keep is not None:
                    v = v[keep]
                if t.kind in ("sum", "avg"):
                    acc[t.name].index_add_(0, gid, v)
                elif t.kind == "max":
                    acc[t.name] = torch.maximum(acc[t.name], torch.full_like(acc[t.name], -(1 << 62)).scatter_reduce_(0, gid, v, "amax", include_self=True))
                elif t.kind == "min":
                    acc[t.name] = torch.minimum(acc[t.name], torch.full_like(acc[t.name], 1 << 62).scatter_reduce_(0, gid, v, "amin", include_self=True))
                del v
            del gid
        return {"dims": dims, "count": cnt.cpu().tolist(), "targets": {n: a.cpu().tolist() for n, a in acc.items()}}

    def check_syn(self, cols, ref):
        """The launch's result columns (ExecutionResult.to_columns) against _syn_reference(): every group, every target; AVG
        within 1e-6 relative (a double quotient), everything else exact."""
        from hdk_amd.ir import Agg, KeyRef
        q = self.query
        dims, cnt = ref["dims"], ref["count"]
        keys = [t for t in q.targets if isinstance(t, KeyRef)]
        aggs = [t for t in q.targets if isinstance(t, Agg)]
        nrows = len(cols[aggs[0].name])
        if nrows != sum(1 for c in cnt if c):
            return False
        for i in range(nrows):
            gid, stride = 0, 1
            for t, d in zip(sorted(keys, key=lambda t: t.idx), dims):
                kv = cols[t.name][i]
                if kv is None or kv != int(kv):
                    return False
                gid += (int(kv) - 1) * stride
                stride *= d
            n = cnt[gid]
            for t in aggs:
                got = cols[t.name][i]
                if t.kind == "count":
                    ok = got == n
                elif t.kind == "avg":
                    want = ref["targets"][t.name][gid] / n
                    ok = got is not None and abs(got - want) <= 1e-6 * max(abs(want), 1e-300)
                else:
                    ok = got == ref["targets"][t.name][gid]
                if not ok:
                    return False
        return True

    def distinct_keys(self):
        """Number of distinct group keys over the local rows (c2 / c5): torch.unique on the concatenated key column."""
        torch = self.torch
        keys = torch.cat([self.cols[self.key_col][f] for f in self.frag_ids])
        n = int(torch.unique(keys).numel())
        del keys
        return n
