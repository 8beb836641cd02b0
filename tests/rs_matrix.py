"""Port of the reference's layout / reduction test matrices (Tests/ResultSetTest.cpp, Tests/ResultSetTestUtils.cpp).

The cases themselves are data (tests/golden/resultset_matrices.json).  This module restates, INDEPENDENTLY of
hdk_amd/plan.py:
  * the descriptor -> layout rules the cases rely on: slot widths of the descriptor builders
    (ResultSetTestUtils.cpp:484-601), slot offsets / row size / columnar offsets
    (RS/ColSlotContext.cpp:129-211, RS/QueryMemoryDescriptor.cpp:240-256,314-371), init values of a result
    storage (RS/ResultSetStorage.cpp:173-197);
  * the fill procedures (ResultSetTestUtils.cpp:101-470, ResultSetTest.cpp:300-572), the number generators
    (ResultSetTestUtils.h:30-66) and the emulator that predicts the reduced rows (ResultSetTest.cpp:262-298,
    690-860);
  * the expected values of test_reduce (ResultSetTest.cpp:1109-1148).
From a layout it fills an `hdk_hip_plan` BY HAND (no plan compiler involved), so the oracle's reducer and
hdk_hip_reduce_buffers are checked against the reference's expectations, and plan.py's own layouts can be compared
with `layout()` where a QueryUnit produces the same descriptor.
"""
import json
import os
import struct

import numpy as np

from hdk_amd import _abi as A

HERE = os.path.dirname(os.path.abspath(__file__))
EMPTY_KEY_64 = 2**63 - 1
DEADBEEF = 0xdeadbeef


def load_matrices():
    with open(os.path.join(HERE, "golden", "resultset_matrices.json")) as f:
        return json.load(f)


# ---- types -------------------------------------------------------------------------------------------------
TYPES = {  # name -> (class, bytes, nullable)
    "int8": ("int", 1, True), "int16": ("int", 2, True), "int32": ("int", 4, True), "int64": ("int", 8, True),
    "int32nn": ("int", 4, False), "fp64": ("fp", 8, True), "fp64nn": ("fp", 8, False), "dict32": ("dict", 4, True),
}


def _dbits(x):
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


def null_int(nbytes):  # inline_int_null_value
    return -(1 << (8 * nbytes - 1))


NULL_DOUBLE_BITS = A.NULL_DOUBLE_BITS


def align8(x):
    return (x + 7) & ~7


class Target:
    def __init__(self, is_agg, agg, type_name, arg_name):
        self.is_agg, self.agg, self.type_name, self.arg_name = is_agg, agg, type_name, arg_name
        self.cls, self.size, self.nullable = TYPES[type_name]
        self.arg = TYPES[arg_name] if arg_name else None

    @property
    def nslots(self):
        return 2 if (self.is_agg and self.agg == "avg") else 1

    def compact_type(self):
        """get_compact_type (Shared/SqlTypesLayout.h:36-55) -> (class, bytes, nullable)."""
        if not self.is_agg:
            return (self.cls, self.size, self.nullable)
        if self.arg is None:
            return (self.cls, self.size, self.nullable)
        if self.agg in ("min", "max"):
            return self.arg
        return (self.cls, self.size, self.arg[2])

    def init_vals(self):
        """initialize_target_values_for_storage (RS/ResultSetStorage.cpp:173-197)."""
        if self.agg == "count" and self.is_agg:
            return [0]
        if self.nullable:
            pat = NULL_DOUBLE_BITS if self.cls == "fp" else (0 if self.cls == "str" else null_int(self.size))
            v = [pat if self.is_agg else 0]
        else:
            v = [DEADBEEF if self.is_agg else 0]
        if self.is_agg and self.agg == "avg":
            v.append(0)
        return v


class Layout:
    """What a QueryMemoryDescriptor built by the test helpers says about the buffer."""

    def __init__(self, desc, targets, num_bytes, columnar, keyless, target_idx_for_key):
        self.desc, self.targets = desc, targets
        self.kind = desc["kind"]
        self.entry_count = int(desc["entry_count"])
        self.nkeys = len(desc["group_col_widths"])
        self.columnar, self.keyless = bool(columnar), bool(keyless)
        self.idx_target_as_key = target_idx_for_key if keyless else -1
        # slot widths: slot_bytes = max(num_bytes, type size); AVG gets two such slots (ResultSetTestUtils.cpp:508-520)
        self.slot_widths, self.slot_target = [], []
        for ti, t in enumerate(targets):
            w = max(num_bytes, t.size)
            for _ in range(t.nslots):
                self.slot_widths.append(w)
                self.slot_target.append(ti)
        self.init_vals = [v for t in targets for v in t.init_vals()]
        # row-wise: ColSlotContext::getColOnlyOffInBytes / getAlignedPaddedSizeForRange (8-byte slots aligned to 8)
        off, self.slot_off = 0, []
        for w in self.slot_widths:
            if w == 8:
                off = align8(off)
            self.slot_off.append(off)
            off += w
        cols = off
        self.key_bytes = 0 if self.keyless else align8(self.nkeys * 8)
        self.row_bytes = align8(self.key_bytes + cols)  # QueryMemoryDescriptor::getRowSize
        # columnar: getPrependedGroupBufferSizeInBytes + align_to_int64(width * entry_count) per slot
        n = self.entry_count
        coff = 0 if self.keyless else self.nkeys * align8(8 * n)
        self.col_off = []
        for w in self.slot_widths:
            self.col_off.append(coff)
            coff += align8(w * n)
        self.buffer_bytes = coff if self.columnar else self.row_bytes * n

    def with_entry_count(self, n):
        d = dict(self.desc)
        d["entry_count"] = n
        return Layout(d, self.targets, 0, self.columnar, self.keyless, self.idx_target_as_key)._copy_widths(self)

    def _copy_widths(self, other):
        # rebuild with the same slot widths (num_bytes is folded into them)
        self.slot_widths = list(other.slot_widths)
        self.slot_target = list(other.slot_target)
        off, self.slot_off = 0, []
        for w in self.slot_widths:
            if w == 8:
                off = align8(off)
            self.slot_off.append(off)
            off += w
        self.row_bytes = align8(self.key_bytes + off)
        n = self.entry_count
        coff = 0 if self.keyless else self.nkeys * align8(8 * n)
        self.col_off = []
        for w in self.slot_widths:
            self.col_off.append(coff)
            coff += align8(w * n)
        self.buffer_bytes = coff if self.columnar else self.row_bytes * n
        return self

    # ---- raw slot access -------------------------------------------------------------------------------
    def slot_addr(self, entry, slot):
        if self.columnar:
            return self.col_off[slot] + entry * self.slot_widths[slot]
        return entry * self.row_bytes + self.key_bytes + self.slot_off[slot]

    def key_addr(self, entry, k):
        if self.columnar:
            return k * align8(8 * self.entry_count) + entry * 8
        return entry * self.row_bytes + k * 8

    def write_int(self, buf, entry, slot, v):
        w = self.slot_widths[slot]
        a = self.slot_addr(entry, slot)
        buf[a:a + w] = np.frombuffer(int(v & ((1 << (8 * w)) - 1)).to_bytes(w, "little"), dtype=np.uint8)

    def write_fp(self, buf, entry, slot, v):
        w = self.slot_widths[slot]
        a = self.slot_addr(entry, slot)
        buf[a:a + w] = np.frombuffer(struct.pack("<d" if w == 8 else "<f", float(v)), dtype=np.uint8)

    def read_int(self, buf, entry, slot):
        w = self.slot_widths[slot]
        a = self.slot_addr(entry, slot)
        return int.from_bytes(bytes(buf[a:a + w]), "little", signed=True)

    def write_key(self, buf, entry, k, v):
        a = self.key_addr(entry, k)
        buf[a:a + 8] = np.frombuffer(int(v & (2**64 - 1)).to_bytes(8, "little"), dtype=np.uint8)

    def read_key(self, buf, entry, k):
        a = self.key_addr(entry, k)
        return int.from_bytes(bytes(buf[a:a + 8]), "little", signed=True)

    def first_slot(self, ti):
        return self.slot_target.index(ti)

    def new_buffer(self):
        return np.zeros(self.buffer_bytes, dtype=np.uint8)

    def is_empty(self, buf, entry):
        """ResultSetStorage::isEmptyEntry (RS/ResultSetStorage.cpp:439-521)."""
        if self.keyless:
            s = self.idx_target_as_key
            iv = self.init_vals[s]
            if self.slot_widths[s] == 4:
                iv = int(np.int32(np.uint32(iv & 0xffffffff)))
            return self.read_int(buf, entry, s) == iv
        return self.read_key(buf, entry, 0) == EMPTY_KEY_64

    # ---- decoded row (what ResultSet::getRowAt gives, as far as the checks look at it) -------------------
    def decode_row(self, buf, entry):
        out = []
        for ti, t in enumerate(self.targets):
            s = self.first_slot(ti)
            cls = t.compact_type()[0]
            raw = self.read_int(buf, entry, s)
            if t.is_agg and t.agg == "avg":
                cnt = self.read_int(buf, entry, s + 1)
                if cnt == 0:
                    out.append(("fp", None))
                else:
                    total = struct.unpack("<d", struct.pack("<q", raw))[0] if cls == "fp" else raw
                    out.append(("fp", total / cnt))
            elif cls == "fp":
                w = self.slot_widths[s]
                val = struct.unpack("<d", struct.pack("<q", raw))[0] if w == 8 else \
                    struct.unpack("<f", struct.pack("<i", raw))[0]
                out.append(("fp", val))
            else:
                out.append((cls, raw))
        return out


def make_layout(doc, case):
    targets = [Target(*t) for t in doc["target_sets"][case["targets"]]]
    return Layout(doc["descriptors"][case["desc"]], targets, int(case["num_bytes"]), case["columnar"],
                  case.get("keyless", False), case.get("target_idx_for_key"))


# ---- hdk_hip_plan filled by hand from a layout --------------------------------------------------------------
_AGG = {"count": A.AGG_COUNT, "sum": A.AGG_SUM, "min": A.AGG_MIN, "max": A.AGG_MAX, "avg": A.AGG_AVG}


def supported_by_library(lay: Layout):
    """4- and 8-byte slots, plus the 1- and 2-byte MIN / MAX slots of columnar buffers with logical-sized columns
    (include/hdk_hip.h: hdk_hip_target.slot_width; the reference's small-slot runtime covers those two aggregates only)."""
    for ti, t in enumerate(lay.targets):
        w = lay.slot_widths[lay.first_slot(ti)]
        if w in (1, 2) and not (lay.columnar and t.is_agg and t.agg in ("min", "max")):
            return False
        if w not in (1, 2, 4, 8):
            return False
    return True


def make_plan(lay: Layout) -> A.Plan:
    p = A.Plan()
    p.abi_version = A.PLAN_ABI
    p.query_kind = A.Q_PERFECT_HASH if lay.kind == "perfect" else A.Q_BASELINE_HASH
    p.num_cols = 0
    p.key_count = lay.nkeys
    p.entry_count = lay.entry_count
    p.key_width = 8
    p.keyless = 1 if lay.keyless else 0
    p.idx_target_as_key = lay.idx_target_as_key
    p.output_columnar = 1 if lay.columnar else 0
    p.row_size_quad = 0 if lay.columnar else lay.row_bytes // 8
    p.num_targets = len(lay.targets)
    for ti, t in enumerate(lay.targets):
        tg = p.targets[ti]
        s = lay.first_slot(ti)
        cls = t.compact_type()[0]
        tg.agg = _AGG[t.agg] if t.is_agg else A.AGG_ID
        tg.has_arg = 1
        tg.skip_null = 1  # TargetInfo::skip_null_val is true in every case of the matrices
        tg.slot_width = lay.slot_widths[s]
        tg.slot_off = lay.key_bytes + lay.slot_off[s]
        tg.arg_is_fp = 1 if cls == "fp" else 0
        tg.key_idx = 0
        tg.null_val = A.to_i64(lay.init_vals[s])
        if t.nslots == 2:
            tg.slot2_width = lay.slot_widths[s + 1]
            tg.slot2_off = lay.key_bytes + lay.slot_off[s + 1]
        else:
            tg.slot2_width = tg.slot_width
            tg.slot2_off = tg.slot_off
    return p


# ---- number generators (ResultSetTestUtils.h:30-66) ----------------------------------------------------------
class EvenNumberGenerator:
    def __init__(self):
        self.crt = 0

    def next(self):
        v = self.crt
        self.crt += 2
        return v

    def reset(self):
        self.crt = 0


class ReverseOddOrEvenNumberGenerator:
    def __init__(self, init):
        self.crt = self.init = init

    def next(self):
        v = self.crt
        self.crt -= 2
        return v

    def reset(self):
        self.crt = self.init


def make_generator(name, entry_count):
    return EvenNumberGenerator() if name == "even" else ReverseOddOrEvenNumberGenerator(2 * entry_count - 1)


# ---- fills ---------------------------------------------------------------------------------------------------
def _fill_entry(lay: Layout, buf, entry, v, empty, null_val=False, baseline=False):
    """fill_one_entry_no_collisions / fill_one_entry_one_col / fill_one_entry_baseline
    (ResultSetTestUtils.cpp:101-160,162-250,600-660): the slots of one entry from the value `v`."""
    for ti, t in enumerate(lay.targets):
        s = lay.first_slot(ti)
        cls = t.cls
        if t.is_agg and t.agg == "count":  # (TargetInfo.agg_kind == kCount, also the test's non-agg kMin never is)
            vv = 0 if (empty or null_val) else v
        elif t.nullable and null_val:      # isNullable && skip_null_val && null_val
            vv = null_int(t.size) if cls != "fp" else NULL_DOUBLE_BITS
        elif baseline and empty:
            vv = null_int(t.size)
        else:
            vv = v
        if empty and not baseline:
            lay.write_int(buf, entry, s, 0 if lay.keyless else vv)
        elif cls == "fp" and not (null_val and t.nullable):
            lay.write_fp(buf, entry, s, vv)
        else:
            lay.write_int(buf, entry, s, vv)
        if t.is_agg and t.agg == "avg":
            lay.write_int(buf, entry, s + 1, 0 if (empty or (t.nullable and null_val)) else 1)


def _baseline_insert(O, lay: Layout, buf, key_vals):
    """get_group_value / get_group_value_columnar on the raw buffer, as the reference's fills do
    (ResultSetTestUtils.cpp:399-470): returns the entry the key now owns."""
    L = O.lib()
    key = np.array(key_vals, dtype=np.int64)
    i64 = buf.view(np.int64)
    n = lay.entry_count
    if lay.columnar:
        slot_ptr = L.orc_get_group_value_columnar(i64.ctypes.data, n, key.ctypes.data, len(key_vals))
        assert slot_ptr, "table full"
        # returns &buffer[key_count * entry_count + entry]
        return (slot_ptr - i64.ctypes.data) // 8 - len(key_vals) * n
    rq = lay.row_bytes // 8
    slot_ptr = L.orc_get_group_value(i64.ctypes.data, n, key.ctypes.data, len(key_vals), 8, rq)
    assert slot_ptr, "table full"
    return ((slot_ptr - i64.ctypes.data) // 8) // rq


def init_storage(lay: Layout, buf, baseline_init=None):
    """Empty storage as the fills leave it: EMPTY keys; perfect hash: 0xdeadbeef (0 when keyless) through
    fill_one_entry(empty); baseline: per-slot values given by the caller."""
    for e in range(lay.entry_count):
        if not lay.keyless:
            for k in range(lay.nkeys):
                lay.write_key(buf, e, k, EMPTY_KEY_64)
        if baseline_init is not None:
            for s, v in enumerate(baseline_init):
                lay.write_int(buf, e, s, v)


def fill_storage(O, lay: Layout, gen, step):
    """fill_storage_buffer (ResultSetTestUtils.cpp:250-482): every `step`-th entry holds the next generated value."""
    buf = lay.new_buffer()
    n = lay.entry_count
    if lay.kind == "perfect":
        init_storage(lay, buf)
        for i in range(n):
            if i % step == 0:
                v = gen.next()
                if not lay.keyless:
                    for k in range(lay.nkeys):
                        lay.write_key(buf, i, k, v)
                _fill_entry(lay, buf, i, v, empty=False)
            else:
                _fill_entry(lay, buf, i, 0 if lay.keyless else DEADBEEF, empty=True)
        return buf
    # baseline: slots start at 0 (COUNT) / 0xdeadbeef, keys go through get_group_value (:399-470)
    init = []
    for t in lay.targets:
        init.append(0 if (t.is_agg and t.agg == "count") else DEADBEEF)
        if t.nslots == 2:
            init.append(DEADBEEF)
    init_storage(lay, buf, init)
    for i in range(0, n, step):
        v = gen.next()
        e = _baseline_insert(O, lay, buf, [v] * lay.nkeys)
        _fill_entry(lay, buf, e, v, empty=False, baseline=True)
    return buf


def result_storage(O, lay: Layout, entry_count=None):
    """An empty result storage (ResultSetStorage::initializeRowWise / initializeColWise,
    RS/ResultSetStorage.cpp + ResultSetReduction.cpp:120-170): EMPTY keys, slots at their init values."""
    rl = lay if entry_count is None else lay.with_entry_count(entry_count)
    buf = rl.new_buffer()
    init_storage(rl, buf, rl.init_vals)
    return rl, buf


# ---- expected values -----------------------------------------------------------------------------------------
def expected_reduce_row(lay: Layout, row_idx, step):
    """test_reduce's assertions (ResultSetTest.cpp:1109-1148): SUM and COUNT columns hold step * row_idx, the
    others row_idx; dictionary columns are not checked."""
    out = []
    for t in lay.targets:
        cls = "fp" if (t.is_agg and t.agg == "avg") else t.cls
        if cls == "dict":
            out.append(None)
            continue
        ref = step * row_idx if (t.is_agg and t.agg in ("sum", "count")) else row_idx
        out.append(float(ref) if cls == "fp" else int(ref))
    return out


class Emulator:
    """ResultSetEmulator (ResultSetTest.cpp:78-298,690-860): two storages with random group membership and the
    reduced rows they must give.  Membership comes from a seeded generator (the reference uses random_device)."""

    def __init__(self, O, lay: Layout, prct1, prct2, flow, seed):
        self.lay, self.flow = lay, flow
        n = lay.entry_count
        rng = np.random.default_rng(seed)
        self.groups = []
        for pct in (prct1, prct2):
            idx = rng.permutation(n)
            g = np.zeros(n, dtype=bool)
            g[idx[:n * pct // 100]] = True
            self.groups.append(g)
        self.values = [np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)]
        self.bufs = [self._fill(O, 0), self._fill(O, 1)]
        self.null_val = null_int(lay.targets[0].size)

    def _fill(self, O, which):
        """rse_fill_storage_buffer_* (:300-572): the generator advances for EVERY entry; in the NULL flow the last
        four entries hold NULLs."""
        lay, flow = self.lay, self.flow
        n = lay.entry_count
        gen = EvenNumberGenerator()
        buf = lay.new_buffer()
        groups, values = self.groups[which], self.values[which]
        if lay.kind == "perfect":
            init_storage(lay, buf)
            for i in range(n):
                v = gen.next()
                last4 = flow == 2 and i >= n - 4
                if groups[i]:
                    values[i] = -1 if last4 else v
                    for k in range(lay.nkeys):
                        lay.write_key(buf, i, k, v)
                    _fill_entry(lay, buf, i, v, empty=False, null_val=last4)
                else:
                    if last4:
                        values[i] = -1
                    _fill_entry(lay, buf, i, 0 if lay.keyless else DEADBEEF, empty=True, null_val=(flow == 2))
            return buf
        init = []
        for t in lay.targets:  # :527-548
            if t.is_agg and t.agg == "count":
                init.append(0)
            elif t.nullable and flow == 2:
                init.append(null_int(t.size))
            else:
                init.append(DEADBEEF)
            if t.nslots == 2:
                init.append(0)
        init_storage(lay, buf, init)
        for i in range(n):
            v = gen.next()
            if groups[i]:
                last4 = flow == 2 and i >= n - 4
                values[i] = -1 if last4 else v
                e = _baseline_insert(O, lay, buf, [v] * lay.nkeys)
                _fill_entry(lay, buf, e, v, empty=False, null_val=last4, baseline=True)
        return buf

    # rseAggregateK* (:700-860)
    def _agg(self, kind, i):
        g1, g2 = self.groups[0][i], self.groups[1][i]
        v1, v2 = int(self.values[0][i]), int(self.values[1][i])
        nullv = self.null_val
        if kind == "min":
            if g1 and g2:
                if v1 == -1 and v2 == -1:
                    return nullv
                return min(v1, v2) if (v1 != -1 and v2 != -1) else max(v1, v2)
            v = v1 if g1 else v2
            return v if v != -1 else nullv
        if kind == "max":
            if g1 and g2:
                return nullv if (v1 == -1 and v2 == -1) else max(v1, v2)
            v = v1 if g1 else v2
            return v if v != -1 else nullv
        if kind == "sum":
            if g1 and g2:
                if v1 == -1 and v2 == -1:
                    return nullv
                return (v1 if v1 != -1 else 0) + (v2 if v2 != -1 else 0)
            v = v1 if g1 else v2
            return v if v != -1 else nullv
        if kind == "count":
            if g1 and g2:
                return (v1 if v1 != -1 else 0) + (v2 if v2 != -1 else 0)
            v = v1 if g1 else v2
            return v if v != -1 else 0
        # avg -> double or None (NULL_DOUBLE)
        if g1 and g2:
            if v1 == -1 and v2 == -1:
                return None
            vals = [v for v in (v1, v2) if v != -1]
            return float(sum(vals)) / len(vals) if len(vals) > 1 else float(vals[0])
        v = v1 if g1 else v2
        return float(v) if v != -1 else None

    def expected(self):
        """{group index -> reference row} (mergeResultSets, :262-298)."""
        out = {}
        for i in range(self.lay.entry_count):
            if self.groups[0][i] or self.groups[1][i]:
                out[i] = [self._agg(t.agg, i) for t in self.lay.targets]
        return out
