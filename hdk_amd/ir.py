"""Minimal plan IR for the hot path: types, expressions and the query unit the executor consumes.

It is the flattened stand-in for HDK's `RelAlgExecutionUnit` (reference
omniscidb/QueryEngine/RelAlgExecutionUnit.h): input columns, simple_quals/quals, join quals,
groupby_exprs, target_exprs.  The full `hdk::ir` DAG (omniscidb/IR/) stays out of scope; a real
integration pattern-matches its work unit into this shape.
"""
from dataclasses import dataclass, field
from typing import Sequence, List, Optional, Union

from . import _abi as A


class QueryMustRunOnCpu(Exception):
    """Plan shape outside the fixed kernel library (reference QueryEngine/ErrorHandling.h;
    caught by RelAlgExecutor::executeRelAlgQuery, RelAlgExecutor.cpp:183-192)."""


# ---------------------------------------------------------------------------------------------
# types
# ---------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Type:
    """Subset of hdk::ir::Type (omniscidb/IR/Type.h): fixed-width, in-band-null columns."""
    kind: str  # 'int' | 'fp' | 'decimal' | 'timestamp' | 'date' | 'dict' | 'bool'
    size: int  # bytes
    nullable: bool = True
    scale: int = 0  # decimal scale
    unit: str = "s"  # timestamp / date unit; a DATE in days (unit 'd', 2 or 4 bytes: Arrow date32, hdk::ir::DateType with
    #                  TimeUnit::kDay) is read as epoch seconds by fixed_width_small_date_decode (QE/ColumnIR.cpp:46-49)

    @property
    def is_fp(self):
        return self.kind == "fp"

    @property
    def is_integer_like(self):
        return self.kind in ("int", "decimal", "timestamp", "date", "dict", "bool")

    @property
    def is_date_in_days(self):
        return self.kind == "date" and self.unit == "d"

    def logical(self) -> "Type":
        """The type expressions see (hdk::ir::Type::canonicalize): a DATE in days is an 8-byte DATE in seconds once
        decoded, with NULL_BIGINT for the column's narrow NULL (FixedWidthSmallDate, QE/Codec.cpp:86-102)."""
        return Type("date", 8, self.nullable, 0, "s") if self.is_date_in_days else self

    def with_nullable(self, n):
        return Type(self.kind, self.size, n, self.scale, self.unit)

    def null_value(self) -> int:
        """In-band null sentinel (omniscidb/Shared/InlineNullValues.h:33-39): the int value, or the
        bit pattern for fp."""
        if self.is_fp:
            return A.NULL_DOUBLE_BITS if self.size == 8 else A.NULL_FLOAT_BITS
        return -(1 << (8 * self.size - 1))

    def null_as_int64_or_double_bits(self) -> int:
        """Null sentinel after the column has been widened by the decoder: ints sign-extend, floats
        are widened to double (FLT_MIN as a double)."""
        if self.is_fp:
            if self.size == 8:
                return A.NULL_DOUBLE_BITS
            import struct
            f = struct.unpack("<f", struct.pack("<I", A.NULL_FLOAT_BITS))[0]
            return struct.unpack("<q", struct.pack("<d", float(f)))[0]
        return self.null_value()


def int_type(size, nullable=True):
    return Type("int", size, nullable)


INT64 = int_type(8)
INT32 = int_type(4)
INT16 = int_type(2)
INT8 = int_type(1)
DATE32 = Type("date", 4, True, 0, "d")  # Arrow date32: days, 4 bytes (ArrowStorageUtils.cpp:899-914)
DATE16 = Type("date", 2, True, 0, "d")
DATE64 = Type("date", 8, True, 0, "s")  # a DATE held in seconds: plain integer decode, still bucketized by day
FP64 = Type("fp", 8)
FP32 = Type("fp", 4)


# ---------------------------------------------------------------------------------------------
# expressions
# ---------------------------------------------------------------------------------------------
class Expr:
    def __add__(self, o):
        return BinOp("+", self, _lit(o))

    def __sub__(self, o):
        return BinOp("-", self, _lit(o))

    def __mul__(self, o):
        return BinOp("*", self, _lit(o))

    def __truediv__(self, o):
        return BinOp("/", self, _lit(o))

    def __mod__(self, o):
        return BinOp("%", self, _lit(o))


def _lit(v):
    return v if isinstance(v, Expr) else Lit(v)


@dataclass(frozen=True)
class ColRef(Expr):
    """Column of the outer table (table=None / scan table name) or of a joined inner table."""
    name: str
    table: Optional[str] = None


@dataclass(frozen=True)
class Lit(Expr):
    value: Union[int, float]


@dataclass(frozen=True)
class BinOp(Expr):
    op: str  # + - * / %
    lhs: Expr
    rhs: Expr


@dataclass(frozen=True)
class ExtractYear(Expr):
    """extract(year from <timestamp>) -- omniscidb/Utils/ExtractFromTime.cpp:260-272."""
    arg: Expr


@dataclass(frozen=True)
class Cast(Expr):
    """cast(<expr> as <to>): decimal->int (rounded scale down), int->fp, fp->int."""
    arg: Expr
    to: Type


@dataclass(frozen=True)
class Cmp:
    """Filter conjunct lhs <op> rhs with rhs a literal or a column."""
    lhs: Expr
    op: str  # = <> < > <= >=
    rhs: Expr


@dataclass(frozen=True)
class And:
    """lhs AND rhs over filter conditions, three-valued (logical_and, QE/RuntimeFunctions.cpp:361-372)."""
    lhs: "Cond"
    rhs: "Cond"


@dataclass(frozen=True)
class Or:
    """lhs OR rhs, three-valued (logical_or, QE/RuntimeFunctions.cpp:374-384)."""
    lhs: "Cond"
    rhs: "Cond"


@dataclass(frozen=True)
class Not:
    """NOT arg, three-valued (logical_not, QE/RuntimeFunctions.cpp:355-358)."""
    arg: "Cond"


Cond = Union[Cmp, And, Or, Not]


@dataclass(frozen=True)
class Agg:
    """Aggregate target: kind in count/sum/min/max/avg/single_value; arg None = COUNT(*)."""
    kind: str
    arg: Optional[Expr] = None
    name: Optional[str] = None


@dataclass(frozen=True)
class KeyRef:
    """Non-aggregate target projecting group-by key #idx."""
    idx: int
    name: Optional[str] = None


@dataclass(frozen=True)
class Proj:
    """Projected expression of a filter/project step (no GROUP BY, no aggregates)."""
    expr: Expr
    name: Optional[str] = None


@dataclass(frozen=True)
class JoinSpec:
    """Equi-join of the outer table with `inner_table` on outer_key == inner_col.  Both sides may be
    lists of equal length (composite key -> keyed/"baseline" hash table, BaselineJoinHashTable); the
    table kind (one-to-one / one-to-many, perfect / keyed) is chosen from the inner table's data the
    way PerfectJoinHashTable::reify / HashJoin::getInstance do (QE/JoinHashTable/HashJoin.cpp:258-330)."""
    inner_table: str
    outer_key: Union[Expr, Sequence[Expr]]
    inner_col: Union[str, Sequence[str]]
    type: str = "inner"  # inner | left | semi | anti  (JoinType, Shared/sqldefs.h:33)
    # outer_key IS NOT DISTINCT FROM inner_col (hdk::ir::OpType::kBwEq): NULL keys match NULL keys -- the build files
    # them under a translated value and the probe is hash_join_idx_bitwise (PerfectJoinHashTable.cpp:798-816)
    null_safe: bool = False

    @property
    def outer_keys(self) -> list:
        return list(self.outer_key) if isinstance(self.outer_key, (list, tuple)) else [self.outer_key]

    @property
    def inner_cols(self) -> list:
        return list(self.inner_col) if isinstance(self.inner_col, (list, tuple)) else [self.inner_col]


@dataclass
class QueryUnit:
    """What one execution step needs (cf. RelAlgExecutionUnit)."""
    table: str
    quals: List[Cond] = field(default_factory=list)  # conjunction of conditions (each a Cmp or an And / Or / Not tree)
    joins: List[JoinSpec] = field(default_factory=list)
    groupby: List[Expr] = field(default_factory=list)
    targets: List[Union[Agg, KeyRef, Proj]] = field(default_factory=list)
    scan_limit: Optional[int] = None  # projection: max output rows (RelAlgExecutionUnit::scan_limit)
    # hints (ExecutionOptions / Config.exec.group_by)
    output_columnar: bool = False
    bigint_count: bool = False  # Config.exec.group_by.bigint_count (omniscidb/Shared/Config.h:44)
    baseline_entry_count: Optional[int] = None  # max_groups_buffer_entry_count override
    force_baseline: bool = False
