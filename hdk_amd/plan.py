"""QueryUnit -> (memory layout, init values, POD plan).

Host-side mirror of what the reference does between `RelAlgExecutionUnit` and the JIT:
  * expression ranges           QueryEngine/ExpressionRange.cpp:380-420,796-830
  * perfect vs baseline hash    QueryEngine/MemoryLayoutBuilder.cpp:91-237, ColRangeInfo.cpp:24-66
  * keyless hash                QueryEngine/MemoryLayoutBuilder.cpp:249-413
  * slot widths / compaction    QueryEngine/MemoryLayoutBuilder.cpp:559-652, ResultSet/ColSlotContext.cpp
  * row / column offsets        ResultSet/QueryMemoryDescriptor.cpp:240-258,301-372,457-478
  * init values                 QueryEngine/OutputBufferInitialization.cpp:24-77,112-258
Instead of emitting LLVM IR the work unit is pattern-matched into `hdk_hip_plan`; shapes outside the
fixed kernel library raise QueryMustRunOnCpu (reference RelAlgExecutor.cpp:183-192 retry).
"""
import math
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import _abi as A
from .ir import (Agg, And, BinOp, Cast, Cmp, ColRef, Expr, ExtractYear, KeyRef, Lit, Not, Or, QueryMustRunOnCpu,
                 QueryUnit, Type)
from .storage import ArrowStorage, Table

BASELINE_THRESHOLD = 1_000_000  # Config.exec.group_by.baseline_threshold (Shared/Config.h:51)
DEFAULT_MAX_GROUPS_BUFFER_ENTRY_GUESS = 16384  # Shared/Config.h:42
BIG_GROUP_THRESHOLD = 16384  # Config.exec.group_by.big_group_threshold (Shared/Config.h:43)
MAX_BUFFER_SIZE = 1 << 30  # MemoryLayoutBuilder.cpp:176


def _dbits(x: float) -> int:
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


def align8(x: int) -> int:
    return (x + 7) & ~7


# ---------------------------------------------------------------------------------------------
# expression ranges (ExpressionRange.cpp)
# ---------------------------------------------------------------------------------------------
@dataclass
class Range:
    kind: str  # 'int' | 'fp' | 'invalid'
    lo: float = 0
    hi: float = 0
    bucket: int = 0
    has_nulls: bool = False


def _extract_year(ts: int) -> int:
    # host restatement of omniscidb/Utils/ExtractFromTime.cpp:260-272 (general path)
    day = ts // 86400
    era = (day - 11017) // 146097
    doe = day - 11017 - era * 146097
    yoe = (doe - doe // 1460 + doe // 36524 - (1 if doe == 146096 else 0)) // 365
    doy = doe - (365 * yoe + yoe // 4 - yoe // 100)
    return 2000 + era * 400 + yoe + (1 if 306 <= doy else 0)


def _scale_down(v: int, scale: int) -> int:
    tmp = scale >> 1
    tmp = v + tmp if v >= 0 else v - tmp
    return int(tmp / scale) if scale else v  # C truncation


class _Binder:
    """Resolves columns and types for one QueryUnit."""

    def __init__(self, storage: ArrowStorage, q: QueryUnit):
        self.storage = storage
        self.q = q
        self.outer: Table = storage.get(q.table)
        self.inner: List[Table] = [storage.get(j.inner_table) for j in q.joins]
        self.cols: List[Tuple[str, str, int]] = []  # (table_name, col_name, table_slot)

    def resolve(self, c: ColRef):
        if c.table in (None, self.q.table):
            if c.name in self.outer.columns:
                return 0, self.outer, self.outer.columns[c.name]
            if c.table is not None:
                raise KeyError(f"column {c.name} not in table {c.table}")
        for ji, t in enumerate(self.inner):
            if (c.table in (None, t.name)) and c.name in t.columns:
                return ji + 1, t, t.columns[c.name]
        raise KeyError(f"cannot resolve column {c}")

    def col_index(self, c: ColRef) -> int:
        slot, t, _ = self.resolve(c)
        key = (t.name, c.name, slot)
        if key not in self.cols:
            if len(self.cols) >= A.MAX_COLS:
                raise QueryMustRunOnCpu("too many input columns for the fixed kernel library")
            self.cols.append(key)
        return self.cols.index(key)

    # ----- types -------------------------------------------------------------------------
    def _left_joined(self, slot: int) -> bool:
        return slot > 0 and self.q.joins[slot - 1].type == "left"

    def type_of(self, e: Expr) -> Type:
        if isinstance(e, ColRef):
            slot, _, col = self.resolve(e)
            # a LEFT join without a match yields NULL for every column of the inner table; a DATE in days is an 8-byte
            # DATE in seconds once decoded (Type.logical)
            lt = col.type.logical()
            return lt.with_nullable(True) if self._left_joined(slot) else lt
        if isinstance(e, Lit):
            return Type("fp", 8, False) if isinstance(e.value, float) else Type("int", 8, False)
        if isinstance(e, BinOp):
            lt, rt = self.type_of(e.lhs), self.type_of(e.rhs)
            nullable = lt.nullable or rt.nullable
            if lt.kind == "decimal" or rt.kind == "decimal":
                raise QueryMustRunOnCpu("decimal arithmetic is outside the fixed kernel library")
            if lt.kind == "date" or rt.kind == "date":
                raise QueryMustRunOnCpu("date arithmetic is outside the fixed kernel library")
            if lt.is_fp or rt.is_fp:
                if e.op == "%":
                    raise QueryMustRunOnCpu("fp modulo")
                return Type("fp", 8, nullable)
            return Type("int", 8, nullable)
        if isinstance(e, ExtractYear):
            return Type("int", 8, self.type_of(e.arg).nullable)
        if isinstance(e, Cast):
            return e.to.with_nullable(self.type_of(e.arg).nullable)
        raise QueryMustRunOnCpu(f"unsupported expression {e!r}")

    # ----- ranges ------------------------------------------------------------------------
    def range_of(self, e: Expr) -> Range:
        if isinstance(e, ColRef):
            slot, _, col = self.resolve(e)
            st = col.table_stats()
            has_nulls = st.has_nulls or self._left_joined(slot)
            if st.min is None:
                return Range("invalid", has_nulls=has_nulls)
            if col.type.is_fp:
                return Range("fp", st.min, st.max, 0, has_nulls)
            # a DATE column's range carries the day bucket (getLeafColumnRange, QE/ExpressionRange.cpp:553-558)
            return Range("int", st.min, st.max, 86400 if col.type.kind == "date" else 0, has_nulls)
        if isinstance(e, Lit):
            if isinstance(e.value, float):
                return Range("fp", e.value, e.value)
            return Range("int", int(e.value), int(e.value))
        if isinstance(e, BinOp):
            a, b = self.range_of(e.lhs), self.range_of(e.rhs)
            if a.kind == "invalid" or b.kind == "invalid":
                return Range("invalid")
            hn = a.has_nulls or b.has_nulls
            kind = "fp" if "fp" in (a.kind, b.kind) else "int"
            if e.op == "+":
                return Range(kind, a.lo + b.lo, a.hi + b.hi, 0, hn)
            if e.op == "-":
                return Range(kind, a.lo - b.hi, a.hi - b.lo, 0, hn)
            if e.op == "*":
                c = [a.lo * b.lo, a.lo * b.hi, a.hi * b.lo, a.hi * b.hi]
                return Range(kind, min(c), max(c), 0, hn)
            if e.op == "/" and kind == "int" and b.lo == b.hi and b.lo > 0:
                return Range("int", int(a.lo / b.lo), int(a.hi / b.lo), 0, hn)
            return Range("invalid")  # modulo etc.: ExpressionRange.cpp:416-419
        if isinstance(e, ExtractYear):
            a = self.range_of(e.arg)
            if a.kind != "int":
                return Range("invalid")
            return Range("int", _extract_year(int(a.lo)), _extract_year(int(a.hi)), 0, a.has_nulls)
        if isinstance(e, Cast):
            a = self.range_of(e.arg)
            st = self.type_of(e.arg)
            if a.kind == "invalid":
                return a
            if st.kind == "decimal" and e.to.kind == "int":
                s = 10 ** st.scale
                return Range("int", _scale_down(int(a.lo), s), _scale_down(int(a.hi), s), 0, a.has_nulls)
            if e.to.is_fp:
                return Range("fp", float(a.lo), float(a.hi), 0, a.has_nulls)
            if st.is_fp:
                return Range("int", math.floor(a.lo), math.ceil(a.hi), 0, a.has_nulls)
            return a
        return Range("invalid")


def _ndv_bound(b: "_Binder", e: Expr, rows: int) -> int:
    """Upper bound on the distinct values of a group-by expression from the column statistics (+ 1 for NULL)."""
    if isinstance(e, Cast):
        return _ndv_bound(b, e.arg, rows)
    if isinstance(e, BinOp) and e.op == "%" and isinstance(e.rhs, Lit) and isinstance(e.rhs.value, int) and e.rhs.value:
        m = abs(int(e.rhs.value))
        a = b.range_of(e.lhs)
        signed = a.kind != "int" or a.lo < 0
        return min((2 * m - 1) if signed else m, _ndv_bound(b, e.lhs, rows)) + (1 if a.kind != "int" or a.has_nulls else 0)
    if isinstance(e, BinOp) and e.op in "+-*" and (isinstance(e.rhs, Lit) or isinstance(e.lhs, Lit)):
        return _ndv_bound(b, e.lhs if isinstance(e.rhs, Lit) else e.rhs, rows)
    r = b.range_of(e)
    if r.kind == "int":
        c = int(r.hi) - int(r.lo)
        if r.bucket:
            c //= r.bucket
        return min(c + 1 + (1 if r.has_nulls else 0), rows)
    return rows


# ---------------------------------------------------------------------------------------------
# expression flattening into hdk_hip_expr chains
# ---------------------------------------------------------------------------------------------
_OPS = {"+": A.OP_ADD, "-": A.OP_SUB, "*": A.OP_MUL, "/": A.OP_DIV, "%": A.OP_MOD}


def _result_null(t: Type) -> int:
    """Null sentinel of an expression RESULT: computed values are carried as int64 / double."""
    return A.NULL_DOUBLE_BITS if t.is_fp else A.NULL_BIGINT


def _make_leaf(b: _Binder, e: Expr) -> A.Leaf:
    leaf = A.Leaf()
    if isinstance(e, ColRef):
        t = b.type_of(e)
        leaf.kind = A.LEAF_COL
        leaf.col = b.col_index(e)
        leaf.null_val = A.to_i64(t.null_as_int64_or_double_bits())
        leaf.nullable = 1 if t.nullable else 0
    elif isinstance(e, Lit):
        if isinstance(e.value, float):
            leaf.kind = A.LEAF_FP
            leaf.ival = _dbits(e.value)
        else:
            leaf.kind = A.LEAF_INT
            leaf.ival = A.to_i64(int(e.value))
        leaf.nullable = 0
    else:
        raise QueryMustRunOnCpu("expression too deep for the fixed kernel library (right operand must "
                                "be a column or literal)")
    return leaf


def _flatten(b: _Binder, e: Expr):
    """-> (leaf0_expr, [(op, rhs_expr_or_None, literal_param, out_type)])."""
    if isinstance(e, (ColRef, Lit)):
        return e, []
    if isinstance(e, BinOp):
        l0, steps = _flatten(b, e.lhs)
        if not isinstance(e.rhs, (ColRef, Lit)):
            raise QueryMustRunOnCpu("right operand must be a column or literal")
        return l0, steps + [(_OPS[e.op], e.rhs, b.type_of(e))]
    if isinstance(e, ExtractYear):
        l0, steps = _flatten(b, e.arg)
        at = b.type_of(e.arg)
        if at.kind not in ("timestamp", "date") or at.unit != "s":
            raise QueryMustRunOnCpu("extract(year) needs a TIMESTAMP(0) or DATE argument")
        return l0, steps + [(A.OP_EXTRACT_YEAR, None, b.type_of(e))]
    if isinstance(e, Cast):
        l0, steps = _flatten(b, e.arg)
        at = b.type_of(e.arg)
        if at.kind == "decimal" and e.to.kind == "int":
            return l0, steps + [(A.OP_SCALE_DOWN, Lit(10 ** at.scale), b.type_of(e))]
        if at.is_integer_like and e.to.is_fp:
            return l0, steps + [(A.OP_CAST_INT_TO_FP, None, b.type_of(e))]
        if at.is_fp and e.to.is_integer_like:
            return l0, steps + [(A.OP_CAST_FP_TO_INT, None, b.type_of(e))]
        if at.is_integer_like and e.to.is_integer_like and at.kind != "decimal":
            return l0, steps  # widening int cast: values are carried as int64 already
        raise QueryMustRunOnCpu(f"unsupported cast {at} -> {e.to}")
    raise QueryMustRunOnCpu(f"unsupported expression {e!r}")


def _sql_int_width(b: _Binder, e) -> int:
    """Byte width of an integer operand's SQL type, for the overflow check of + - * (the reference checks in the
    operation's own type: the wider of the two operand types, QE/ArithmeticIR.cpp:277-520; an integer literal is
    INTEGER when it fits 32 bits, else BIGINT).  0 for floating point."""
    if isinstance(e, Lit):
        if isinstance(e.value, float):
            return 0
        return 4 if -(2**31) <= int(e.value) < 2**31 else 8
    t = b.type_of(e)
    return 0 if t.is_fp else t.size


def make_expr(b: _Binder, e: Expr) -> A.Expr:
    l0, steps = _flatten(b, e)
    if len(steps) > A.MAX_EXPR_STEPS:
        raise QueryMustRunOnCpu("expression chain too long for the fixed kernel library")
    x = A.Expr()
    x.leaf0 = _make_leaf(b, l0)
    x.nsteps = len(steps)
    cur_w = _sql_int_width(b, l0)
    for i, (op, rhs, out_t) in enumerate(steps):
        st = x.steps[i]
        st.op = op
        st.out_class = A.VC_FP if out_t.is_fp else A.VC_INT
        if rhs is not None:
            st.rhs = _make_leaf(b, rhs)
        st.null_out = A.to_i64(_result_null(out_t))
        # the SQL type the step computes in: the wider operand for binary integer ops; what a unary op yields
        if op in (A.OP_ADD, A.OP_SUB, A.OP_MUL, A.OP_DIV, A.OP_MOD):
            rw = _sql_int_width(b, rhs)
            cur_w = 0 if (out_t.is_fp or not cur_w or not rw) else max(cur_w, rw)
            st.check_width = cur_w if op in (A.OP_ADD, A.OP_SUB, A.OP_MUL) else 0
        else:
            cur_w = 0 if out_t.is_fp else 8  # extract / scale-down / casts: BIGINT results
    t = b.type_of(e)
    x.vclass = A.VC_FP if t.is_fp else A.VC_INT
    if steps:
        x.null_val = A.to_i64(_result_null(t))
        x.nullable = 1  # every step may produce NULL (eval tracks it in-band)
        # a chain over non-nullable inputs can still never be NULL; keep `nullable` precise
        x.nullable = 1 if t.nullable else 0
    else:
        x.null_val = x.leaf0.null_val
        x.nullable = x.leaf0.nullable
    return x


# ---------------------------------------------------------------------------------------------
# layout + plan
# ---------------------------------------------------------------------------------------------
@dataclass
class OutCol:
    name: str
    kind: str  # 'key' | 'agg'
    type: Type
    target_idx: int = -1
    key_idx: int = -1
    agg: str = ""
    dictionary: Optional[list] = None
    scale: int = 0  # decimal scale of the aggregate's argument (values are scaled int64)


@dataclass
class CompiledPlan:
    plan: A.Plan
    query: QueryUnit
    init_vals: np.ndarray  # int64 per slot (init_agg_val_vec)
    slot_widths: List[int]
    input_cols: List[Tuple[str, str, int]]  # (table, column, table_slot) per buf_idx
    inner_tables: List[str]
    join_infos: list
    out_cols: List[OutCol]
    key_types: List[Type]
    buffer_bytes: int
    key_ranges: List[Range] = field(default_factory=list)

    @property
    def entry_count(self):
        return int(self.plan.entry_count)

    @property
    def buffer_quads(self):
        return self.buffer_bytes // 8


_CMP = {"=": A.CMP_EQ, "==": A.CMP_EQ, "<>": A.CMP_NE, "!=": A.CMP_NE, "<": A.CMP_LT, ">": A.CMP_GT,
        "<=": A.CMP_LE, ">=": A.CMP_GE}
_AGG = {"count": A.AGG_COUNT, "sum": A.AGG_SUM, "min": A.AGG_MIN, "max": A.AGG_MAX, "avg": A.AGG_AVG,
        "single_value": A.AGG_SINGLE_VALUE}


def _bits_to_double(v: int) -> float:
    import struct
    return struct.unpack("<d", struct.pack("<q", int(v)))[0]


def _agg_init_val(kind: str, arg_t: Optional[Type], nullable: bool, width: int) -> int:
    """get_agg_initial_val (OutputBufferInitialization.cpp:112-258) for an 8/4-byte slot."""
    is_fp = arg_t is not None and arg_t.is_fp
    if kind in ("count", "avg_count"):
        return 0
    if kind in ("sum", "avg"):
        if nullable:
            if is_fp:
                return A.NULL_DOUBLE_BITS if width == 8 else A.NULL_FLOAT_BITS
            return A.NULL_BIGINT if width == 8 else A.NULL_INT
        return 0  # 0 / 0.0 share the all-zero pattern
    if kind == "min":
        if nullable:
            return (A.NULL_DOUBLE_BITS if is_fp else A.NULL_BIGINT) if width == 8 else (
                A.NULL_FLOAT_BITS if is_fp else A.NULL_INT)
        if is_fp:
            return _dbits(np.finfo(np.float64).max) if width == 8 else int(
                np.array([np.finfo(np.float32).max], dtype=np.float32).view(np.int32)[0])
        return (2**63 - 1) if width == 8 else (2**31 - 1)
    if kind == "max":
        if nullable:
            return (A.NULL_DOUBLE_BITS if is_fp else A.NULL_BIGINT) if width == 8 else (
                A.NULL_FLOAT_BITS if is_fp else A.NULL_INT)
        if is_fp:
            return _dbits(-np.finfo(np.float64).max) if width == 8 else int(
                np.array([-np.finfo(np.float32).max], dtype=np.float32).view(np.int32)[0])
        return -(2**63) if width == 8 else -(2**31)
    raise ValueError(kind)


def _expr_refs_inner(b: "_Binder", e) -> bool:
    """Does the expression read a column of a joined (inner) table?"""
    if isinstance(e, ColRef):
        return b.resolve(e)[0] > 0
    if isinstance(e, BinOp):
        return _expr_refs_inner(b, e.lhs) or _expr_refs_inner(b, e.rhs)
    if isinstance(e, (ExtractYear, Cast)):
        return _expr_refs_inner(b, e.arg)
    return False


def _inner_keys_unique(inner: Table, cols: List[str], nulls_match: bool = False, bucket: int = 1) -> bool:
    """Would the one-to-one fill succeed (no two rows in one slot)?  The reference finds out by trying
    (fill_one_to_one_hashtable returns -1 -> NeedsOneToManyHash, PerfectHashTableBuilder.h:134-141); the storage layer
    answers from the data so that the plan names the table kind up front.  For a perfect table (one key column) the slots
    are worked out exactly as the fill does (fill_hash_join_buff_impl, HashJoinRuntime.cpp:197-240): decoded element,
    NULL rows dropped -- or, for a kBwEq join, filed under max + 1 --, slot = (elem - min) / bucket."""
    cache = inner.__dict__.setdefault("_unique_keys_cache", {})
    key = tuple(cols) + (bool(nulls_match), int(bucket))
    if key not in cache:
        arrs = [np.concatenate(inner.columns[c].fragments) if inner.columns[c].fragments else np.zeros(0, np.int64)
                for c in cols]
        if len(cols) == 1:
            c0 = inner.columns[cols[0]]
            st = c0.table_stats()
            a0 = arrs[0].astype(np.int64)
            isn = a0 == c0.type.null_value()
            if c0.type.is_date_in_days:
                a0 = a0 * 86400
            if nulls_match:
                a0 = np.where(isn, int(st.max) + 1, a0)
            else:
                a0 = a0[~isn]
            slots = (a0 - int(st.min)) // max(int(bucket), 1)
            _, cnt = np.unique(slots, return_counts=True)
        else:
            keep = np.ones(len(arrs[0]), dtype=bool)
            for c, a in zip(cols, arrs):
                keep &= a != inner.columns[c].type.null_value()
            m = np.stack([a[keep].astype(np.int64) for a in arrs], axis=1)
            _, cnt = np.unique(m, axis=0, return_counts=True)
        cache[key] = int(cnt.max()) if cnt.size else 1  # largest matching set
    return cache[key] <= 1


def _inner_max_matches(inner: Table, cols: List[str], nulls_match: bool = False, bucket: int = 1) -> int:
    _inner_keys_unique(inner, cols, nulls_match, bucket)
    return inner.__dict__["_unique_keys_cache"][tuple(cols) + (bool(nulls_match), int(bucket))]


def _compile_joins_and_quals(b: "_Binder", q: QueryUnit, p: A.Plan):
    """Join descriptors (joins first: inner-table columns need their table slot), then the filter
    conjuncts, each tagged with the stage it can run at."""
    join_infos = []
    p.num_joins = len(q.joins)
    for ji, j in enumerate(q.joins):
        inner = b.inner[ji]
        okeys, icols = j.outer_keys, j.inner_cols
        if len(okeys) != len(icols) or not 1 <= len(okeys) <= A.MAX_JOIN_KEYS:
            raise QueryMustRunOnCpu("join key lists must have 1..%d matching entries" % A.MAX_JOIN_KEYS)
        for c in icols:
            if not inner.columns[c].type.is_integer_like:
                raise QueryMustRunOnCpu("join keys must be integers")
        for k in okeys:
            if _expr_refs_inner(b, k) and not all(b.resolve(r)[0] <= ji for r in _colrefs(k)):
                raise QueryMustRunOnCpu("a join key may only read the outer table and earlier joins")
        icol = inner.columns[icols[0]]
        st = icol.table_stats()
        if st.min is None:
            raise QueryMustRunOnCpu("empty join inner table")
        if j.type not in ("inner", "left", "semi", "anti"):
            raise QueryMustRunOnCpu(f"join type {j.type!r}")
        jn = p.joins[ji]
        jn.outer_key = make_expr(b, okeys[0])
        for k in range(1, len(okeys)):
            jn.extra_keys[k - 1] = make_expr(b, okeys[k])
        jn.min_key = int(st.min)
        jn.max_key = int(st.max)
        okt = b.type_of(okeys[0])
        if okt.is_fp:
            raise QueryMustRunOnCpu("join keys must be integers")
        jn.null_val = A.to_i64(okt.null_as_int64_or_double_bits())
        bw_eq = bool(j.null_safe)
        # a DATE key is bucketized by day: bucket_normalization = the range's bucket (get_bucketized_hash_entry_info,
        # QE/JoinHashTable/PerfectJoinHashTable.cpp:45-85; the bucket comes from getLeafColumnRange)
        is_date = icol.type.kind == "date"
        if is_date != (okt.kind == "date"):
            raise QueryMustRunOnCpu("a DATE join key needs a DATE on both sides")
        bucket = 86400 if is_date else 1
        # key_col nullable or kBwEq: the probe gets the NULL (getHashJoinArgs, PerfectJoinHashTable.cpp:798-801) and is the
        # _bitwise form for kBwEq, else _nullable for a nullable key (codegenSlot, :1018-1031)
        jn.null_mode = A.JOIN_NULL_BITWISE if bw_eq else (A.JOIN_NULL_NULLABLE if okt.nullable else A.JOIN_NULL_NONE)
        jn.bucket = bucket if is_date else 0
        jn.type = {"inner": A.JOIN_INNER, "left": A.JOIN_LEFT, "semi": A.JOIN_SEMI, "anti": A.JOIN_ANTI}[j.type]
        semi = j.type in ("semi", "anti")  # for_semi_anti_join: the fill lets the first row of a key win
        jn.table_idx = ji
        # HashEntryInfo: hash_entry_count = max - min + 1 (+ 1 slot for the NULLs of a kBwEq join), normalised by the bucket
        hash_entry_count = int(st.max) - int(st.min) + 1 + (1 if bw_eq else 0)
        range_entries = -(-hash_entry_count // bucket)  # getNormalizedHashEntryCount (HashJoinRuntime.h:46-55)
        keyed = len(okeys) > 1 or range_entries > 2**31 - 1  # TooManyHashEntries -> keyed table
        unique = semi or _inner_keys_unique(inner, icols, bw_eq and not keyed, 1 if keyed else bucket)
        if keyed:
            if bw_eq or is_date:
                raise QueryMustRunOnCpu("null-safe / DATE keys on a keyed join table are outside the fixed kernel library")
            jn.kind = A.JOIN_KEYED_ONE_TO_ONE if unique else A.JOIN_KEYED_ONE_TO_MANY
            # BaselineJoinHashTable::getKeyComponentWidth: 8 if any key column is 8 bytes wide
            jn.key_component_width = 8 if any(inner.columns[c].type.size == 8 for c in icols) or \
                any(b.type_of(k).size == 8 for k in okeys) else 4
            jn.key_component_count = len(okeys)
            jn.entry_count = max(2 * inner.num_rows, 2)  # 2 x tuple-count upper bound
            if jn.entry_count > 2**31 - 1:
                raise QueryMustRunOnCpu("keyed join table too large")
        else:
            jn.kind = A.JOIN_ONE_TO_ONE if unique else A.JOIN_ONE_TO_MANY
            jn.key_component_count = 1
            jn.key_component_width = 8
            jn.entry_count = range_entries
            if bw_eq:
                if int(st.max) >= 2**63 - 1:
                    raise QueryMustRunOnCpu("Cannot translate null value for kBW_EQ")  # PerfectJoinHashTable.cpp:165-168
                # what the probe is handed for a NULL key (getHashJoinArgs, PerfectJoinHashTable.cpp:803-810): max + 1, and
                # for a DATE key max / bucket + 1 -- while the BUILD files NULL rows under max + 1 in both cases
                # (JoinColumnTypeInfo::translated_null_val, Builders/PerfectHashTableBuilder.h:100-106).  For a DATE key the
                # two differ, and the probe's value normally fails its own `key >= min_key` test: NULLs then never match.
                # Restated as it is; a translated value whose slot would fall outside the table is refused.
                tr = int(st.max) + 1
                if is_date:
                    q_ = abs(int(st.max)) // bucket  # C++ integer division truncates toward zero
                    tr = (q_ if int(st.max) >= 0 else -q_) + 1
                if tr >= int(st.min) and (tr - int(st.min)) // bucket >= range_entries:
                    raise QueryMustRunOnCpu("translated NULL key falls outside the join table")
                jn.translated_null = tr
        join_infos.append({"inner_table": inner.name, "inner_col": icols[0], "inner_cols": list(icols),
                           "min": int(st.min), "max": int(st.max), "null_val": icol.type.null_value(),
                           "elem_sz": icol.type.size,
                           "null_vals": [inner.columns[c].type.null_value() for c in icols],
                           "elem_szs": [inner.columns[c].type.size for c in icols],
                           "mins": [int(inner.columns[c].table_stats().min) for c in icols],
                           "maxs": [int(inner.columns[c].table_stats().max) for c in icols],
                           "col_types": [A.JC_SMALL_DATE if inner.columns[c].type.is_date_in_days else A.JC_SIGNED
                                         for c in icols],
                           "kind": int(jn.kind), "entry_count": int(jn.entry_count),
                           "key_width": int(jn.key_component_width), "num_elems": inner.num_rows,
                           # the build's arguments (PerfectHashTableBuilder.h:87-130): HashEntryInfo, uses_bw_eq with its
                           # translated NULL (always max + 1), the day bucket of a DATE key, first-row-wins fill
                           "hash_entry_count": hash_entry_count, "bucket": bucket, "bucketized": bool(is_date),
                           "uses_bw_eq": 1 if bw_eq else 0, "translated_null_build": int(st.max) + 1,
                           "for_semi_join": 1 if semi else 0,
                           "max_matches": 1 if semi else _inner_max_matches(inner, icols, bw_eq and not keyed, 1 if keyed else bucket)})
    # filter: a plain conjunction of comparisons keeps the per-conjunct staging; anything with OR / NOT becomes a
    # postfix program over the comparisons (hdk_hip_plan::filter_ops), evaluated at one stage
    leaves: List[Cmp] = []
    prog: List[int] = []

    def emit(c):
        if isinstance(c, Cmp):
            if c not in leaves:
                leaves.append(c)
            prog.append(leaves.index(c))
        elif isinstance(c, Not):
            emit(c.arg)
            prog.append(A.F_NOT)
        elif isinstance(c, (And, Or)):
            emit(c.lhs)
            emit(c.rhs)
            prog.append(A.F_AND if isinstance(c, And) else A.F_OR)
        else:
            raise QueryMustRunOnCpu(f"unsupported filter condition {c!r}")

    for i, c in enumerate(q.quals):
        emit(c)
        if i:
            prog.append(A.F_AND)
    if len(leaves) > A.MAX_QUALS:
        raise QueryMustRunOnCpu("too many filter comparisons for the fixed kernel library")
    p.num_quals = len(leaves)
    for qi, c in enumerate(leaves):
        ql = p.quals[qi]
        ql.lhs = make_expr(b, c.lhs)
        ql.rhs = _make_leaf(b, c.rhs)
        ql.cmp = _CMP[c.op]
        ql.after_joins = 1 if (_expr_refs_inner(b, c.lhs) or _expr_refs_inner(b, c.rhs)) else 0
    if all(isinstance(c, Cmp) for c in q.quals):
        p.num_filter_ops = 0
    else:
        if len(prog) > A.MAX_FILTER_OPS:
            raise QueryMustRunOnCpu("filter expression too long for the fixed kernel library")
        p.num_filter_ops = len(prog)
        for i, op in enumerate(prog):
            p.filter_ops[i] = op
        p.filter_after_joins = 1 if any(p.quals[i].after_joins for i in range(len(leaves))) else 0
    return join_infos


def _colrefs(e) -> list:
    if isinstance(e, ColRef):
        return [e]
    if isinstance(e, BinOp):
        return _colrefs(e.lhs) + _colrefs(e.rhs)
    if isinstance(e, (ExtractYear, Cast)):
        return _colrefs(e.arg)
    return []


def _fill_cols(storage: ArrowStorage, b: "_Binder", p: A.Plan):
    p.num_cols = len(b.cols)
    for i, (tn, cn, slot) in enumerate(b.cols):
        ct = storage.get(tn).columns[cn].type
        c = p.cols[i]
        c.buf_idx = i
        c.table = slot
        c.width = ct.size
        _fill_col_kind(storage, tn, cn, ct, c)


def _fill_col_kind(storage, tn, cn, ct, c):
    if ct.is_fp:
        c.kind = A.COL_DOUBLE if ct.size == 8 else A.COL_FLOAT
    elif ct.is_date_in_days:
        if ct.size not in (2, 4):
            raise QueryMustRunOnCpu("a DATE in days is 2 or 4 bytes wide")
        c.kind = A.COL_SMALL_DATE  # get_col_decoder -> FixedWidthSmallDate (QE/ColumnIR.cpp:46-49)
    else:
        c.kind = A.COL_INT
        # ChunkStats of the column (the reference's planner reads them through getExpressionRange,
        # QE/ExpressionRange.cpp): lets multi-pass strategies move the column in fewer bytes than its width
        st = storage.get(tn).columns[cn].table_stats()
        if st.min is not None and st.max is not None and -(2**63) <= int(st.min) <= int(st.max) < 2**63:
            c.has_stats, c.has_nulls = 1, 1 if st.has_nulls else 0
            c.min_val, c.max_val = int(st.min), int(st.max)


def compile_projection(storage: ArrowStorage, q: QueryUnit) -> CompiledPlan:
    """Filter/project step: QueryDescriptionType::Projection.  Layout (QueryMemoryDescriptor.cpp:314-342,
    457-478): row-wise [int64 row position | 8-byte slots]; columnar [positions | columns at their
    logical widths] (isLogicalSizedColumnsAllowed, :496-500).  entry_count = scan_limit, else the number
    of input rows (the reference guesses max_groups_buffer_entry_count and retries on -pos)."""
    from .ir import Proj
    b = _Binder(storage, q)
    p = A.Plan()
    p.abi_version = A.PLAN_ABI
    if len(q.targets) > A.MAX_TARGETS or len(q.joins) > A.MAX_JOINS:
        raise QueryMustRunOnCpu("query exceeds the fixed kernel library's limits")
    join_infos = _compile_joins_and_quals(b, q, p)
    p.query_kind = A.Q_PROJECTION
    p.key_count = 0
    p.keyless = 0
    p.idx_target_as_key = -1
    p.key_width = 8
    columnar = bool(q.output_columnar)
    p.output_columnar = 1 if columnar else 0
    # one output row per surviving row combination: outer rows x the largest matching set of each join
    # (the reference guesses a size and re-runs with a bigger buffer when the kernel reports -pos)
    fanout = 1
    for info in join_infos:
        fanout *= max(int(info["max_matches"]), 1)
    entry_count = int(q.scan_limit) if q.scan_limit else max(b.outer.num_rows * fanout, 1)
    if entry_count >= 2**31:
        raise QueryMustRunOnCpu("projection of more than 2^31-1 rows")
    p.entry_count = entry_count
    p.num_targets = len(q.targets)
    out_cols, slot_widths, init_vals = [], [], []
    row_off = 8
    for ti, t in enumerate(q.targets):
        assert isinstance(t, Proj)
        tt = b.type_of(t.expr)
        if tt.is_fp and tt.size == 4:
            raise QueryMustRunOnCpu("float32 projections are outside the fixed kernel library")
        tg = p.targets[ti]
        tg.agg = A.AGG_ID
        tg.has_arg = 1
        tg.key_idx = -1
        tg.arg = make_expr(b, t.expr)
        tg.arg_is_fp = 1 if tt.is_fp else 0
        w = tt.size if (columnar and not (tg.arg.nsteps and not tt.is_fp)) else 8
        if columnar and tg.arg.nsteps and not tt.is_fp:
            w = 8  # computed integer expressions are carried as int64
        tg.slot_width = w
        tg.slot2_width = w
        tg.slot_off = row_off
        row_off += 8
        slot_widths.append(w)
        init_vals.append(0)
        src = t.expr
        dic = b.resolve(src)[2].dictionary if isinstance(src, ColRef) else None
        name = t.name or (src.name if isinstance(src, ColRef) else f"expr_{ti}")
        rt = tt if w == tt.size else Type("fp" if tt.is_fp else "int", 8, tt.nullable)
        if isinstance(src, ColRef) and tt.kind in ("decimal", "timestamp", "date", "dict", "bool") and w == tt.size:
            rt = tt
        out_cols.append(OutCol(name, "proj", rt, ti, dictionary=dic,
                               scale=(tt.scale if tt.kind == "decimal" else 0)))
    p.row_size_quad = 0 if columnar else row_off // 8
    if columnar:
        total = align8(entry_count * 8)
        for w in slot_widths:
            total = align8(total) + entry_count * w
        buffer_bytes = align8(total)
    else:
        buffer_bytes = row_off * entry_count
    _fill_cols(storage, b, p)
    return CompiledPlan(plan=p, query=q, init_vals=np.array(init_vals, dtype=np.int64), slot_widths=slot_widths,
                        input_cols=list(b.cols), inner_tables=[t.name for t in b.inner], join_infos=join_infos,
                        out_cols=out_cols, key_types=[], buffer_bytes=buffer_bytes, key_ranges=[])


def compile_query(storage: ArrowStorage, q: QueryUnit) -> CompiledPlan:
    from .ir import Proj
    if q.targets and all(isinstance(t, Proj) for t in q.targets):
        if q.groupby:
            raise QueryMustRunOnCpu("projection targets with GROUP BY")
        return compile_projection(storage, q)
    if any(isinstance(t, Proj) for t in q.targets):
        raise QueryMustRunOnCpu("mixing projections and aggregates")
    b = _Binder(storage, q)
    p = A.Plan()
    p.abi_version = A.PLAN_ABI
    if len(q.joins) > A.MAX_JOINS or len(q.groupby) > A.MAX_KEYS or len(q.targets) > A.MAX_TARGETS:
        raise QueryMustRunOnCpu("query exceeds the fixed kernel library's limits")

    join_infos = _compile_joins_and_quals(b, q, p)

    # ---- group-by keys + hash type (get_col_range_info) ---------------------------------
    nkeys = len(q.groupby)
    key_types = [b.type_of(k) for k in q.groupby]
    key_ranges = [b.range_of(k) for k in q.groupby]
    p.key_count = nkeys
    for ki, k in enumerate(q.groupby):
        # (a floating-point key is its bit pattern from here on: groupByColumnCodegen bit-casts the double to the 8-byte
        # key word, QE/IRCodegen.cpp:1219-1221; its range is not an integer range, so the layout is GroupByBaselineHash)
        # A FLOAT key is widened to double before it becomes the key word (CgenState::castToTypeIn(group_key, 64) in
        # groupByColumnCodegen, QE/IRCodegen.cpp:1219-1221; read back as static_cast<float>(double), RS/ResultSetIteration.cpp
        # makeTargetValue, case 8), its NULL the FLOAT sentinel widened.  A FLOAT column is carried as a double already
        # (fixed_width_float_decode).  cast(<integer> AS FLOAT): the step library has the conversion to double only, which is the
        # same value exactly when the argument's statistics lie inside +-2^24 (MultiStep/MSBS001-005: cast(x1k AS float)) --
        # wider arguments would need the rounding to float and stay out.
        if key_types[ki].is_fp and key_types[ki].size != 8:
            if isinstance(k, Cast) and b.type_of(k.arg).is_integer_like:
                ar = b.range_of(k.arg)
                if ar.kind != "int" or int(ar.lo) < -(1 << 24) or int(ar.hi) > (1 << 24):
                    raise QueryMustRunOnCpu("cast(<integer> AS FLOAT) group-by key whose argument may exceed 2^24 (the rounding to float "
                                            "is outside the fixed kernel library)")
            elif not isinstance(k, ColRef):
                raise QueryMustRunOnCpu("4-byte floating-point group-by key expressions are outside the fixed kernel library")
        p.keys[ki] = make_expr(b, k)
        if key_types[ki].is_fp and key_types[ki].size == 4 and p.keys[ki].nsteps:
            # the cast's NULL is the FLOAT sentinel, and the key word holds it widened to double (computed doubles carry
            # NULL_DOUBLE otherwise: _result_null)
            fnull = A.to_i64(key_types[ki].null_as_int64_or_double_bits())
            p.keys[ki].steps[p.keys[ki].nsteps - 1].null_out = fnull
            p.keys[ki].null_val = fnull

    def card(r: Range) -> int:  # ColRangeInfo::getBucketedCardinality
        c = int(r.hi) - int(r.lo)
        if r.bucket:
            c //= r.bucket
        return c + 1 + (1 if r.has_nulls else 0)

    if nkeys == 0:
        kind = A.Q_NON_GROUPED
        entry_count = 1
    else:
        perfect = all(r.kind == "int" for r in key_ranges) and not q.force_baseline
        entry_count = 0
        if perfect and nkeys == 1:
            col_count = nkeys + len(q.targets)
            max_entry_count = MAX_BUFFER_SIZE // (col_count * 8)
            r = key_ranges[0]
            if int(r.hi) - int(r.lo) >= max_entry_count and not r.bucket:
                perfect = False
            else:
                entry_count = max(card(r), 1)
        elif perfect:
            total = 1
            for r in key_ranges:
                total *= card(r)
            if total == 0 or total > BASELINE_THRESHOLD:
                perfect = False
            else:
                entry_count = total
        kind = A.Q_PERFECT_HASH if perfect else A.Q_BASELINE_HASH
        if not perfect:
            if q.baseline_entry_count:
                entry_count = int(q.baseline_entry_count)
            else:
                # RelAlgExecutor.cpp:1553-1557: 2 x the cardinality estimate; the estimate here is
                # min(row count, product of key ranges)
                est = b.outer.num_rows
                prod = 1
                for r in key_ranges:
                    if r.kind != "int":
                        prod = None
                        break
                    prod *= card(r)
                if prod is not None:
                    est = min(est, prod)
                    entry_count = max(2 * est, DEFAULT_MAX_GROUPS_BUFFER_ENTRY_GUESS)
                else:
                    # keys without an integer range (cast(x as double), x % m: QE/ExpressionRange.cpp:391-419).  The reference
                    # takes the default guess for small inputs and runs its NDV estimator query otherwise
                    # (groups_approx_upper_bound <= big_group_threshold, RelAlgExecutor.cpp:1533-1557); the stand-in for that
                    # query is a bound on the distinct values from the column statistics (callers with a real estimate pass
                    # baseline_entry_count)
                    upper = max([b.outer.num_rows] + [t.num_rows for t in b.inner])
                    if upper <= BIG_GROUP_THRESHOLD:
                        entry_count = DEFAULT_MAX_GROUPS_BUFFER_ENTRY_GUESS
                    else:
                        ndv = 1
                        for k in q.groupby:
                            ndv *= _ndv_bound(b, k, est)
                        entry_count = 2 * max(min(ndv, est), 1)
        if entry_count >= 2**31:
            raise QueryMustRunOnCpu("group-by buffer entry count does not fit int32")
    p.query_kind = kind
    p.entry_count = entry_count

    for ki, r in enumerate(key_ranges):
        if kind == A.Q_PERFECT_HASH:
            p.key_min[ki] = int(r.lo)
            p.key_bucket[ki] = int(r.bucket)
            p.key_card[ki] = card(r)
            p.key_has_nulls[ki] = 1 if r.has_nulls else 0
            p.key_null_translated[ki] = int(r.hi) + (r.bucket if r.bucket else 1)

    # ---- slots: widths (pick_target_compact_width) ----------------------------------------
    all_small = nkeys == 1 and all(
        (isinstance(t, Agg) and t.kind == "count" and t.arg is None) or
        (isinstance(t, KeyRef) and key_types[t.idx].size <= 4)
        for t in q.targets)
    total_tuples = b.outer.num_rows + sum(t.num_rows for t in b.inner)
    W = 4 if (all_small and not q.bigint_count and total_tuples <= 0xFFFFFFFF) else 8
    if kind == A.Q_NON_GROUPED:
        W = 8

    # ---- keyless (get_keyless_info), single-column perfect hash only ----------------------
    keyless, idx_target_as_key = False, -1
    if kind == A.Q_PERFECT_HASH and nkeys == 1 and not key_ranges[0].bucket:
        found, index, ok = False, 0, True
        for t in q.targets:
            if isinstance(t, Agg) and not found:
                at = b.type_of(t.arg) if t.arg is not None else None
                ar = b.range_of(t.arg) if t.arg is not None else None
                if t.kind == "avg":
                    index += 1
                    if not (at.nullable and (ar.kind == "invalid" or ar.has_nulls)):
                        found = True
                elif t.kind == "count":
                    if not (at is not None and at.nullable and (ar.kind == "invalid" or ar.has_nulls)):
                        found = True
                elif t.kind == "sum":
                    if at.nullable:
                        if ar.kind != "invalid" and not ar.has_nulls:
                            found = True
                    elif ar.kind != "invalid" and (ar.hi < 0 or ar.lo > 0):
                        found = True
                elif t.kind in ("min", "max"):
                    # the reference compares the range with get_agg_initial_val of the (nullable) argument type --
                    # for a nullable argument that is the NULL sentinel -- and, for fp ranges, with that int64
                    # REINTERPRETED as a double, 4-byte float patterns included (MemoryLayoutBuilder.cpp:326-394)
                    fa = at.is_fp and at.size == 4
                    init = A.to_i64(_agg_init_val(t.kind, at, at.nullable, 4 if fa else 8))
                    bound = _bits_to_double(init) if ar.kind == "fp" else init
                    if t.kind == "min":
                        if ar.kind != "invalid" and ar.hi < bound:
                            found = True
                    elif ar.kind != "invalid" and not ar.has_nulls and ar.lo > bound:
                        found = True
                else:
                    ok = False
            if not ok:
                break
            if not found:
                index += 1
        keyless = ok and found
        idx_target_as_key = index
    p.keyless = 1 if keyless else 0
    p.idx_target_as_key = idx_target_as_key if keyless else -1

    # ---- key width / columnar ---------------------------------------------------------------
    columnar = bool(q.output_columnar) and kind != A.Q_NON_GROUPED
    if kind == A.Q_BASELINE_HASH and not columnar:
        # pick_baseline_key_width (MemoryLayoutBuilder.cpp:654-690)
        kw = 4
        for r, t in zip(key_ranges, key_types):
            if r.kind != "int" or (t.size == 8 and r.has_nulls) or not (
                    r.lo > -(2**31) and r.hi < A.EMPTY_KEY_32 - 1):
                kw = 8
        key_width = kw
    else:
        key_width = 8
    p.key_width = key_width
    p.output_columnar = 1 if columnar else 0

    # ---- targets ------------------------------------------------------------------------------
    p.num_targets = len(q.targets)
    init_vals: List[int] = []
    slot_widths: List[int] = []
    out_cols: List[OutCol] = []
    keys_bytes = 0 if (keyless or kind == A.Q_NON_GROUPED) else align8(nkeys * key_width)
    row_off = keys_bytes
    for ti, t in enumerate(q.targets):
        tg = p.targets[ti]
        tg.slot_width = W
        tg.slot2_width = W
        if isinstance(t, KeyRef):
            if kind == A.Q_NON_GROUPED:
                raise QueryMustRunOnCpu("key projection without GROUP BY")
            tg.agg = A.AGG_ID
            tg.has_arg = 1
            tg.key_idx = t.idx
            tg.arg = make_expr(b, q.groupby[t.idx])
            tg.skip_null = 0
            tg.arg_is_fp = 0
            tg.null_val = 0
            init_vals.append(0)  # non-agg targets init to 0 (OutputBufferInitialization.cpp:45-47)
            if kind == A.Q_BASELINE_HASH:
                # GroupByBaselineHash: a projected group key is read back from the key columns, its slot
                # has size 0 (target_groupby_indices -> ColSlotContext::addSlotForColumn(0, 0),
                # QE/MemoryLayoutBuilder.cpp:921-927, RS/ColSlotContext.cpp:43-48)
                tg.slot_width = 0
                tg.slot2_width = 0
            slot_widths.append(int(tg.slot_width))
            src = q.groupby[t.idx]
            dic = b.resolve(src)[2].dictionary if isinstance(src, ColRef) else None
            out_cols.append(OutCol(t.name or (src.name if isinstance(src, ColRef) else f"key{t.idx}"),
                                   "key", key_types[t.idx], ti, t.idx, dictionary=dic))
        else:
            tg.agg = _AGG[t.kind]
            tg.has_arg = 0 if t.arg is None else 1
            at = None
            if t.arg is not None:
                tg.arg = make_expr(b, t.arg)
                at = b.type_of(t.arg)
                if at.kind in ("dict", "bool") and t.kind != "count":
                    raise QueryMustRunOnCpu(f"{t.kind} over {at.kind}")
            elif t.kind != "count":
                raise ValueError(f"{t.kind} needs an argument")
            # takes_float_argument (Shared/TargetInfo.h:170-179): SUM / MIN / MAX / AVG over a FLOAT argument
            # accumulate a float in the slot's low 4 bytes, whatever the padded slot width
            float_acc = at is not None and at.is_fp and at.size == 4 and t.kind != "count"  # (SINGLE_VALUE included: :176)
            arg_nullable = at.nullable if at is not None else False
            # group-by: the declared nullability decides; non-grouped: always nullable and always
            # *_skip_val (OutputBufferInitialization.cpp:57-60, TargetExprBuilder.cpp:546-551)
            eff_nullable = True if kind == A.Q_NON_GROUPED else arg_nullable
            tg.skip_null = 1 if (t.arg is not None and eff_nullable) else 0
            # SINGLE_VALUE: checked_single_agg_id takes the argument type's NULL whatever the nullability and is never a
            # *_skip_val call (QE/TargetExprBuilder.cpp:429-445,542-546); its slot starts like MAX's
            # (get_agg_initial_val, QE/OutputBufferInitialization.cpp:211-213)
            single = t.kind == "single_value"
            init_kind = "max" if single else ("sum" if t.kind == "avg" else t.kind)
            if single:
                tg.skip_null = 0
            tg.arg_is_fp = A.FP_SLOT_FLOAT if float_acc else (A.FP_SLOT_DOUBLE if (at is not None and at.is_fp) else 0)
            if t.kind == "count":
                # agg_count[_skip_val]: skip value = the argument type's NULL (widened)
                tg.null_val = A.to_i64(at.null_as_int64_or_double_bits()) if at is not None else 0
                init_vals.append(0)
                slot_widths.append(W)
            else:
                if t.kind in ("min", "max", "single_value"):
                    # domain-range-equivalent aggregates keep the ARGUMENT type's NULL
                    # (get_agg_initial_val on the arg type: inline_int_null_value(type) in an 8-byte slot)
                    nullv = at.null_as_int64_or_double_bits()
                else:
                    # SUM / AVG: integer sums are BIGINT (TargetInfo.h:121-135), fp sums DOUBLE
                    nullv = A.NULL_DOUBLE_BITS if at.is_fp else A.NULL_BIGINT
                tg.null_val = A.to_i64(nullv)
                if float_acc:
                    # the scan compares doubles (the decoder widens): null_val = the float sentinel widened; the
                    # slot starts at the 4-byte pattern, sign-extended (OutputBufferInitialization.cpp:52-65)
                    tg.null_val = A.to_i64(at.null_as_int64_or_double_bits())
                    iv = _agg_init_val(init_kind, at, eff_nullable, 4)
                elif eff_nullable:
                    iv = nullv
                else:
                    iv = _agg_init_val(init_kind, at, False, 8)
                init_vals.append(A.to_i64(iv))
                slot_widths.append(W)
                if t.kind == "avg":
                    init_vals.append(0)
                    slot_widths.append(W)
            if t.kind == "count":
                rt = Type("int", 4 if W == 4 else (8 if q.bigint_count else 4), False)
            elif t.kind == "avg":
                rt = Type("fp", 8, True)
            elif t.kind == "sum":
                rt = Type("fp", at.size, True) if at.is_fp else Type("int", 8, True)
            else:
                rt = at
            out_cols.append(OutCol(t.name or f"{t.kind}_{ti}", "agg", rt, ti, agg=t.kind,
                                   scale=(at.scale if (at is not None and at.kind == "decimal") else 0)))
        # row-wise offsets (ColSlotContext::getColOnlyOffInBytes)
        if tg.slot_width == 0:
            tg.slot_off = row_off
            continue
        if W == 8:
            row_off = align8(row_off)
        tg.slot_off = row_off
        row_off += W
        if tg.agg == A.AGG_AVG:
            if W == 8:
                row_off = align8(row_off)
            tg.slot2_off = row_off
            row_off += W
    row_bytes = align8(row_off)
    p.row_size_quad = row_bytes // 8 if not columnar else 0

    if kind == A.Q_NON_GROUPED:
        buffer_bytes = 8 * len(init_vals)
        for ti in range(p.num_targets):  # out_vec: one int64 slot per agg column
            pass
    elif columnar:
        if key_width != 8:
            raise QueryMustRunOnCpu("columnar output needs 8-byte keys")
        total = 0 if keyless else nkeys * align8(entry_count * 8)
        for w in slot_widths:
            total = align8(total) + entry_count * w
        buffer_bytes = align8(total)
    else:
        buffer_bytes = row_bytes * entry_count

    # ---- input columns (buf_idx order == order of first use) ----------------------------------
    p.num_cols = len(b.cols)
    for i, (tn, cn, slot) in enumerate(b.cols):
        ct = storage.get(tn).columns[cn].type
        c = p.cols[i]
        c.buf_idx = i
        c.table = slot
        c.width = ct.size
        _fill_col_kind(storage, tn, cn, ct, c)
    return CompiledPlan(plan=p, query=q, init_vals=np.array(init_vals, dtype=np.int64),
                        slot_widths=slot_widths, input_cols=list(b.cols),
                        inner_tables=[t.name for t in b.inner], join_infos=join_infos,
                        out_cols=out_cols, key_types=key_types, buffer_bytes=buffer_bytes,
                        key_ranges=key_ranges)


# ---------------------------------------------------------------------------------------------
# host-side helpers shared by the executor and the tests
# ---------------------------------------------------------------------------------------------
def eff_key_count(p) -> int:
    """Key columns in front of the slots: the group-by keys, or the row-position column of a projection."""
    return 1 if p.query_kind == A.Q_PROJECTION else int(p.key_count)


def columnar_slot_offsets(cp: CompiledPlan, entry_count: Optional[int] = None) -> List[int]:
    """Byte offset of every slot column (QueryMemoryDescriptor::getColOffInBytes, columnar)."""
    p = cp.plan
    n = int(entry_count if entry_count is not None else p.entry_count)
    off = 0 if p.keyless else p.key_count * align8(n * 8)
    if p.query_kind == A.Q_PROJECTION:
        off = align8(n * 8)  # the row-position column
    offs = []
    for w in cp.slot_widths:
        off = align8(off)
        offs.append(off)
        off += n * w
    return offs


def columnar_init_vals(cp: CompiledPlan) -> np.ndarray:
    """init_agg_val_vec for the columnar init kernel: zero-width slots have no entry
    (OutputBufferInitialization.cpp:45-47; init_columnar_group_by_buffer_gpu skips them without
    consuming an init value, QE/GpuInitGroups.cu:133-166)."""
    return np.array([v for v, w in zip(cp.init_vals, cp.slot_widths) if w], dtype=np.int64)


def compact_init_vals(cp: CompiledPlan) -> np.ndarray:
    """Row-wise init values as the kernel's INIT_AGG_VALS param: one int64 word per quad of the
    row's slot region (QueryExecutionContext.cpp:829-836 compact_init_vals)."""
    p = cp.plan
    if p.query_kind == A.Q_NON_GROUPED or p.output_columnar:
        return cp.init_vals.copy()
    keys_bytes = 0 if p.keyless else align8(eff_key_count(p) * p.key_width)
    nbytes = p.row_size_quad * 8 - keys_bytes
    raw = np.zeros(nbytes, dtype=np.uint8)
    s = 0
    for ti in range(p.num_targets):
        tg = p.targets[ti]
        n = 2 if tg.agg == A.AGG_AVG else 1
        for k in range(n):
            off = (tg.slot_off if k == 0 else tg.slot2_off) - keys_bytes
            w = cp.slot_widths[s]
            v = int(cp.init_vals[s])
            if w:  # zero-width slot: a projected key of a baseline table, nothing to initialise
                raw[off:off + w] = np.frombuffer(
                    (v & ((1 << (8 * w)) - 1)).to_bytes(w, "little"), dtype=np.uint8)
            s += 1
    return raw.view(np.int64).copy()


def init_buffer_host(cp: CompiledPlan, entry_count: Optional[int] = None) -> np.ndarray:
    """Host copy of a freshly initialised output buffer (what hdk_hip_init_*_group_by_buffer
    writes on the device) -- numpy restatement used by tests and by CPU-side reduction."""
    p = cp.plan
    n = int(entry_count if entry_count is not None else p.entry_count)
    if p.query_kind == A.Q_NON_GROUPED:
        return cp.init_vals.copy()
    if p.output_columnar:
        offs = columnar_slot_offsets(cp, n)
        nk = eff_key_count(p)
        total = offs[-1] + n * cp.slot_widths[-1] if offs else nk * align8(n * 8)
        buf = np.zeros(align8(total), dtype=np.uint8)
        if not p.keyless:
            buf[:nk * align8(n * 8)].view(np.int64)[:] = A.EMPTY_KEY_64
        for off, w, v in zip(offs, cp.slot_widths, cp.init_vals):
            if not w:
                continue
            dt = {8: np.int64, 4: np.int32, 2: np.int16, 1: np.int8}[w]
            buf[off:off + n * w].view(dt)[:] = np.int64(v).astype(dt)
        return buf.view(np.int64).copy()
    rq = int(p.row_size_quad)
    buf = np.zeros((n, rq), dtype=np.int64)
    keys_quads = 0
    if not p.keyless:
        nk = eff_key_count(p)
        keys_quads = align8(nk * p.key_width) // 8
        kb = np.zeros((n, keys_quads * 8), dtype=np.uint8)
        if p.key_width == 8:
            kb.view(np.int64)[:, :nk] = A.EMPTY_KEY_64
        else:
            kb.view(np.int32)[:, :nk] = A.EMPTY_KEY_32
        buf[:, :keys_quads] = kb.view(np.int64)
    buf[:, keys_quads:] = compact_init_vals(cp)[None, :]
    return buf.reshape(-1).copy()
