// CPU-only check of hdk_amd/glue/HipPlanExtractor.h: work units written as (stand-in) hdk::ir trees are run through
// PlanExtractor + make_plan and the resulting hdk_hip_plan is dumped, one file per query, into the directory named on
// the command line.  tests/test_plan_extractor.py builds the SAME queries with hdk_amd/plan.py (the Python model of the
// mapping, which every GPU parity test goes through) and compares the expression half byte for byte: filters, the
// filter program, join descriptors, group-by keys, targets.  No device, no library call.
#include <cstdio>
#include <cstring>
#include <string>

#include "hdk_decls.h"
#include "qmd_standin.h"

#include "HipPlanExtractor.h"
#include "ir_standin.h"

using namespace standin;
using hdk::ir::AggType;
using hdk::ir::OpType;
using hip_rt::TypeDesc;

namespace {

std::string g_dir;

void dump(const char* name, const hdk_hip_plan& p) {
  const std::string path = g_dir + "/" + name + ".plan";
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f || std::fwrite(&p, sizeof(p), 1, f) != 1) throw std::runtime_error("cannot write " + path);
  std::fclose(f);
}

hip_rt::PlanExtractor<StandInIr>::ColumnResolver by_column_id() {
  return [](const Expr* e) { return e->column_id; };  // the stand-in columns are numbered in COL_BUFFERS order
}

hdk_hip_plan run(std::vector<hip_rt::HipInputCol> cols, const hip_rt::UnitView<StandInIr>& unit, const QmdStandIn& qmd,
                 const std::vector<hip_rt::JoinTableDesc>& joins = {}, std::vector<hip_rt::HipWorkUnit::KeyRange> ranges = {}) {
  hip_rt::PlanExtractor<StandInIr> ex(std::move(cols), by_column_id());
  hip_rt::HipWorkUnit wu = ex.extract(unit, joins, qmd.getQueryDescriptionType() == QueryDescriptionType::Projection);
  wu.key_ranges = std::move(ranges);
  return hip_rt::make_plan(wu, qmd);
}

TypeDesc int_t(int size, bool nullable) { return type_of(TypeDesc::Integer, size, nullable); }

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  g_dir = argv[1];
  try {
    {  // c2: SELECT key, SUM(val), COUNT(*) FROM t GROUP BY key            (key BIGINT NOT NULL, val BIGINT)
      auto key = column(bigint(false), 0, 0), val = column(bigint(true), 0, 1);
      hip_rt::UnitView<StandInIr> u;
      u.groupby = {key.get()};
      auto sum = agg_expr(bigint(true), AggType::kSum, val), cnt = agg_expr(bigint(false), AggType::kCount, nullptr);
      u.targets = {key.get(), sum.get(), cnt.get()};
      QmdStandIn q;
      q.query_desc_type_ = QueryDescriptionType::GroupByPerfectHash;
      q.group_col_widths_ = {8};
      q.padded_slot_widths_ = {8, 8, 8};
      q.entry_count_ = 64;
      q.max_val_ = 63;
      dump("c2", run({{0, 8, HDK_COL_INT}, {0, 8, HDK_COL_INT}}, u, q));
    }
    {  // taxi q4: GROUP BY passenger_count, extract(year from pickup_datetime), cast(trip_distance as int) -> COUNT(*)
      auto pc = column(int_t(2, true), 0, 0), ts = column(type_of(TypeDesc::Timestamp, 8, true), 0, 1),
           dist = column(type_of(TypeDesc::Decimal, 8, true, 2), 0, 2);
      auto year = extract(hdk::ir::DateExtractField::kYear, ts);
      auto idist = u_oper(int_t(4, true), OpType::kCast, dist);
      hip_rt::UnitView<StandInIr> u;
      u.groupby = {pc.get(), year.get(), idist.get()};
      auto cnt = agg_expr(int_t(4, false), AggType::kCount, nullptr);
      u.targets = {pc.get(), year.get(), idist.get(), cnt.get()};
      QmdStandIn q;
      q.query_desc_type_ = QueryDescriptionType::GroupByPerfectHash;
      q.group_col_widths_ = {8, 8, 8};
      q.padded_slot_widths_ = {8, 8, 8, 4};
      q.entry_count_ = 8 * 8 * 52;
      dump("q4", run({{0, 2, HDK_COL_INT}, {0, 8, HDK_COL_INT}, {0, 8, HDK_COL_INT}}, u, q, {},
                     {{0, 6, 0, true}, {2009, 2015, 0, true}, {0, 50, 0, true}}));
    }
    {  // filters: WHERE (v < 5 OR NOT (w >= 2.5)) AND k = 2 ; SELECT COUNT(*), MIN(v + 3), AVG(w)
      auto v = column(bigint(true), 0, 0), w = column(fp64(true), 0, 1), k = column(int_t(4, true), 0, 2);  // (COL_BUFFERS order: first use)
      auto boolean = type_of(TypeDesc::Boolean, 1, true);
      auto lt = bin_oper(boolean, OpType::kLt, v, int_literal(5));
      auto ge = bin_oper(boolean, OpType::kGe, w, fp_literal(2.5));
      auto nt = u_oper(boolean, OpType::kNot, ge);
      auto orr = bin_oper(boolean, OpType::kOr, lt, nt);
      auto eq = bin_oper(boolean, OpType::kEq, k, int_literal(2));
      hip_rt::UnitView<StandInIr> u;
      u.quals = {orr.get(), eq.get()};
      auto cnt = agg_expr(int_t(4, false), AggType::kCount, nullptr);
      auto vp3 = bin_oper(bigint(true), OpType::kPlus, v, int_literal(3));
      auto mn = agg_expr(bigint(true), AggType::kMin, vp3);
      auto av = agg_expr(fp64(true), AggType::kAvg, w);
      u.targets = {cnt.get(), mn.get(), av.get()};
      QmdStandIn q;
      q.query_desc_type_ = QueryDescriptionType::NonGroupedAggregate;
      q.padded_slot_widths_ = {8, 8, 8, 8};
      dump("filters", run({{0, 8, HDK_COL_INT}, {0, 8, HDK_COL_DOUBLE}, {0, 4, HDK_COL_INT}}, u, q));
    }
    {  // join: SELECT SUM(val + dval), COUNT(*) FROM fact JOIN dim ON fk = key   (one-to-one perfect table on [0, 999])
      auto fk = column(bigint(true), 0, 0), val = column(bigint(true), 0, 1), dkey = column(bigint(false), 1, 3),
           dval = column(bigint(false), 1, 2);
      auto on = bin_oper(type_of(TypeDesc::Boolean, 1, true), OpType::kEq, fk, dkey);
      hip_rt::UnitView<StandInIr> u;
      u.joins.push_back({{on.get()}, HDK_JOIN_INNER});
      auto add = bin_oper(bigint(true), OpType::kPlus, val, dval);
      auto sum = agg_expr(bigint(true), AggType::kSum, add), cnt = agg_expr(bigint(false), AggType::kCount, nullptr);
      u.targets = {sum.get(), cnt.get()};
      QmdStandIn q;
      q.query_desc_type_ = QueryDescriptionType::NonGroupedAggregate;
      q.padded_slot_widths_ = {8, 8};
      hip_rt::JoinTableDesc jt;
      jt.kind = HDK_JOIN_ONE_TO_ONE;
      jt.min_key = 0;
      jt.max_key = 999;
      dump("join", run({{0, 8, HDK_COL_INT}, {0, 8, HDK_COL_INT}, {1, 8, HDK_COL_INT}}, u, q, {jt}));
    }
    {  // c5 shape: SELECT key, SUM(val) GROUP BY key, open addressing, 4-byte table key
      auto key = column(bigint(false), 0, 0), val = column(bigint(true), 0, 1);
      hip_rt::UnitView<StandInIr> u;
      u.groupby = {key.get()};
      auto sum = agg_expr(bigint(true), AggType::kSum, val);
      u.targets = {key.get(), sum.get()};
      QmdStandIn q;
      q.query_desc_type_ = QueryDescriptionType::GroupByBaselineHash;
      q.group_col_widths_ = {8};
      q.group_col_compact_width_ = 4;
      q.padded_slot_widths_ = {0, 8};
      q.entry_count_ = 4096;
      dump("c5", run({{0, 8, HDK_COL_INT}, {0, 8, HDK_COL_INT}}, u, q));
    }
    {  // projection: SELECT key, val * 2 FROM t WHERE val < 100
      auto val = column(bigint(true), 0, 0), key = column(bigint(false), 0, 1);  // (COL_BUFFERS order: first use)
      auto lt = bin_oper(type_of(TypeDesc::Boolean, 1, true), OpType::kLt, val, int_literal(100));
      auto twice = bin_oper(bigint(true), OpType::kMul, val, int_literal(2));
      hip_rt::UnitView<StandInIr> u;
      u.quals = {lt.get()};
      u.targets = {key.get(), twice.get()};
      QmdStandIn q;
      q.query_desc_type_ = QueryDescriptionType::Projection;
      q.group_col_widths_ = {8};
      q.padded_slot_widths_ = {8, 8};
      q.entry_count_ = 1000;
      dump("projection", run({{0, 8, HDK_COL_INT}, {0, 8, HDK_COL_INT}}, u, q));
    }
    {  // float accumulators: SELECT k, SUM(f), AVG(f), MIN(d), COUNT(f) GROUP BY k   (f FLOAT, d DOUBLE)
      auto k = column(int_t(4, false), 0, 0), f = column(type_of(TypeDesc::Fp, 4, true), 0, 1), d = column(fp64(true), 0, 2);
      hip_rt::UnitView<StandInIr> u;
      u.groupby = {k.get()};
      auto sf = agg_expr(type_of(TypeDesc::Fp, 4, true), AggType::kSum, f), af = agg_expr(fp64(true), AggType::kAvg, f),
           md = agg_expr(fp64(true), AggType::kMin, d), cf = agg_expr(int_t(4, false), AggType::kCount, f);
      u.targets = {k.get(), sf.get(), af.get(), md.get(), cf.get()};
      QmdStandIn q;
      q.query_desc_type_ = QueryDescriptionType::GroupByPerfectHash;
      q.group_col_widths_ = {8};
      q.padded_slot_widths_ = {8, 8, 8, 8, 8, 4};
      q.entry_count_ = 300;
      q.max_val_ = 299;
      dump("floats", run({{0, 4, HDK_COL_INT}, {0, 4, HDK_COL_FLOAT}, {0, 8, HDK_COL_DOUBLE}}, u, q));
    }
    {  // SINGLE_VALUE: SELECT k, SINGLE_VALUE(f), SINGLE_VALUE(d), COUNT(*) GROUP BY k   (k INT, f FLOAT, d DOUBLE)
      auto k = column(int_t(4, false), 0, 0), f = column(type_of(TypeDesc::Fp, 4, true), 0, 1), d = column(fp64(true), 0, 2);
      hip_rt::UnitView<StandInIr> u;
      u.groupby = {k.get()};
      auto s1 = agg_expr(type_of(TypeDesc::Fp, 4, true), AggType::kSingleValue, f), s2 = agg_expr(fp64(true), AggType::kSingleValue, d),
           c = agg_expr(int_t(4, false), AggType::kCount, nullptr);
      u.targets = {k.get(), s1.get(), s2.get(), c.get()};
      QmdStandIn q;
      q.query_desc_type_ = QueryDescriptionType::GroupByPerfectHash;
      q.group_col_widths_ = {8};
      q.padded_slot_widths_ = {8, 8, 8, 8};
      q.entry_count_ = 300;
      q.max_val_ = 299;
      dump("single", run({{0, 4, HDK_COL_INT}, {0, 4, HDK_COL_FLOAT}, {0, 8, HDK_COL_DOUBLE}}, u, q));
    }
    {  // DATE join, IS NOT DISTINCT FROM: SELECT COUNT(*), SUM(v) FROM fact JOIN ddup ON fact.day IS NOT DISTINCT FROM ddup.day
       // (day: DATE in days, 4 bytes -> bucketized table: bucket 86400, range in epoch seconds; PerfectJoinHashTable.cpp:798-816)
      auto date4 = type_of(TypeDesc::Date, 4, true);
      date4.seconds = false;
      auto fday = column(date4, 0, 0), dday = column(date4, 1, 2), v = column(bigint(true), 1, 1);
      auto on = bin_oper(type_of(TypeDesc::Boolean, 1, true), OpType::kBwEq, fday, dday);
      hip_rt::UnitView<StandInIr> u;
      u.joins.push_back({{on.get()}, HDK_JOIN_INNER});
      auto cnt = agg_expr(bigint(false), AggType::kCount, nullptr), sum = agg_expr(bigint(true), AggType::kSum, v);
      u.targets = {cnt.get(), sum.get()};
      QmdStandIn q;
      q.query_desc_type_ = QueryDescriptionType::NonGroupedAggregate;
      q.padded_slot_widths_ = {8, 8};
      hip_rt::JoinTableDesc jt;
      jt.kind = HDK_JOIN_ONE_TO_MANY;
      jt.min_key = 18000LL * 86400;
      jt.max_key = 18099LL * 86400;
      jt.bucket = 86400;
      jt.entry_count = 100;  // ceil((max - min + 1 + 1) / 86400)
      dump("date_bw_eq", run({{0, 4, HDK_COL_SMALL_DATE}, {1, 8, HDK_COL_INT}, {1, 4, HDK_COL_SMALL_DATE}}, u, q, {jt}));
    }
    {  // integer IS NOT DISTINCT FROM under a LEFT join: SELECT COUNT(*), SUM(dval) FROM fact LEFT JOIN dim ON fk IS NOT DISTINCT FROM key
      auto fk = column(bigint(true), 0, 0), dkey = column(bigint(true), 1, 2), dval = column(bigint(false), 1, 1);
      auto on = bin_oper(type_of(TypeDesc::Boolean, 1, true), OpType::kBwEq, fk, dkey);
      hip_rt::UnitView<StandInIr> u;
      u.joins.push_back({{on.get()}, HDK_JOIN_LEFT});
      // (a LEFT join makes every inner column nullable: the reference's planner rewrites the ColumnVar's type)
      auto dval_n = column(bigint(true), 1, 1);
      auto cnt = agg_expr(bigint(false), AggType::kCount, nullptr), sum = agg_expr(bigint(true), AggType::kSum, dval_n);
      u.targets = {cnt.get(), sum.get()};
      QmdStandIn q;
      q.query_desc_type_ = QueryDescriptionType::NonGroupedAggregate;
      q.padded_slot_widths_ = {8, 8};
      hip_rt::JoinTableDesc jt;
      jt.kind = HDK_JOIN_ONE_TO_ONE;
      jt.min_key = 0;
      jt.max_key = 999;
      jt.entry_count = 1001;
      dump("bw_eq_left", run({{0, 8, HDK_COL_INT}, {1, 8, HDK_COL_INT}, {1, 8, HDK_COL_INT}}, u, q, {jt}));
    }
    {  // SEMI join: SELECT COUNT(*), SUM(val) FROM fact WHERE EXISTS (SELECT 1 FROM dim WHERE key = fk)
      auto fk = column(bigint(true), 0, 0), val = column(bigint(true), 0, 1), dkey = column(bigint(false), 1, 2);
      auto on = bin_oper(type_of(TypeDesc::Boolean, 1, true), OpType::kEq, fk, dkey);
      hip_rt::UnitView<StandInIr> u;
      u.joins.push_back({{on.get()}, HDK_JOIN_SEMI});
      auto cnt = agg_expr(bigint(false), AggType::kCount, nullptr), sum = agg_expr(bigint(true), AggType::kSum, val);
      u.targets = {cnt.get(), sum.get()};
      QmdStandIn q;
      q.query_desc_type_ = QueryDescriptionType::NonGroupedAggregate;
      q.padded_slot_widths_ = {8, 8};
      hip_rt::JoinTableDesc jt;
      jt.kind = HDK_JOIN_ONE_TO_ONE;
      jt.min_key = 0;
      jt.max_key = 999;
      jt.entry_count = 1000;
      dump("semi", run({{0, 8, HDK_COL_INT}, {0, 8, HDK_COL_INT}, {1, 8, HDK_COL_INT}}, u, q, {jt}));
    }
    {  // MIN / MAX over narrow nullable integers in 8-byte slots: the skip value is the ARGUMENT type's NULL
       // SELECT k, MIN(s), MAX(i), MIN(i + 1), SUM(s) GROUP BY k      (k INT NOT NULL, s SMALLINT, i INT)
      auto k = column(int_t(4, false), 0, 0), sc = column(int_t(2, true), 0, 1), ic = column(int_t(4, true), 0, 2);
      hip_rt::UnitView<StandInIr> u;
      u.groupby = {k.get()};
      auto ip1 = bin_oper(bigint(true), OpType::kPlus, ic, int_literal(1));
      auto mn = agg_expr(int_t(2, true), AggType::kMin, sc), mx = agg_expr(int_t(4, true), AggType::kMax, ic),
           mn2 = agg_expr(bigint(true), AggType::kMin, ip1), sm = agg_expr(bigint(true), AggType::kSum, sc);
      u.targets = {k.get(), mn.get(), mx.get(), mn2.get(), sm.get()};
      QmdStandIn q;
      q.query_desc_type_ = QueryDescriptionType::GroupByPerfectHash;
      q.group_col_widths_ = {8};
      q.padded_slot_widths_ = {8, 8, 8, 8, 8};
      q.entry_count_ = 300;
      q.max_val_ = 299;
      dump("minmax_narrow", run({{0, 4, HDK_COL_INT}, {0, 2, HDK_COL_INT}, {0, 4, HDK_COL_INT}}, u, q));
    }
    // shapes the library does not cover must be refused, not mistranslated
    int refused = 0;
    auto expect_refusal = [&](const hip_rt::UnitView<StandInIr>& u) {
      QmdStandIn q;
      q.query_desc_type_ = QueryDescriptionType::NonGroupedAggregate;
      q.padded_slot_widths_ = {8};
      try {
        run({{0, 8, HDK_COL_INT}, {0, 8, HDK_COL_INT}}, u, q);
      } catch (const hip_rt::QueryMustRunOnCpu&) {
        ++refused;
      }
    };
    {
      auto a = column(bigint(true), 0, 0), b = column(bigint(true), 0, 1);
      auto deep = bin_oper(bigint(true), OpType::kPlus, a, bin_oper(bigint(true), OpType::kMul, a, b));  // right operand is a tree
      auto s1 = agg_expr(bigint(true), AggType::kSum, deep);
      hip_rt::UnitView<StandInIr> u1;
      u1.targets = {s1.get()};
      expect_refusal(u1);
      auto s2 = agg_expr(bigint(false), AggType::kCount, a, /*distinct=*/true);
      hip_rt::UnitView<StandInIr> u2;
      u2.targets = {s2.get()};
      expect_refusal(u2);
      auto s3 = agg_expr(bigint(true), AggType::kApproxQuantile, a);
      hip_rt::UnitView<StandInIr> u3;
      u3.targets = {s3.get()};
      expect_refusal(u3);
      auto bw = bin_oper(bigint(true), OpType::kBwAnd, a, b);
      auto s4 = agg_expr(bigint(true), AggType::kSum, bw);
      hip_rt::UnitView<StandInIr> u4;
      u4.targets = {s4.get()};
      expect_refusal(u4);
      hip_rt::UnitView<StandInIr> u5;  // a non-aggregate target of a non-grouped unit that is not a group-by expression
      u5.targets = {a.get()};
      expect_refusal(u5);
    }
    std::printf("refused %d of 5\n", refused);
    return refused == 5 ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "extract_dump failed: %s\n", e.what());
    return 2;
  }
}
