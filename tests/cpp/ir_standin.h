// A stand-in for the reference's expression tree (omniscidb/IR/Expr.h: ColumnVar :128, Constant :226, UOper :279,
// BinOper :324, AggExpr :788, ExtractExpr :868) with the same shape -- a node kind, a type, operands -- and the
// reference's OWN operator enums (IR/OpTypeEnums.h, IR/DateTimeEnums.h are self-contained headers and are included
// from the reference tree).  IR/Expr.h itself cannot be compiled here (Logger -> Boost).  `StandInIr` is the access
// policy hdk_amd/glue/HipPlanExtractor.h is instantiated with in the harness; HipIrAccessHdk.h is the same policy over
// the real classes.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "HipPlanExtractor.h"

namespace standin {

struct Expr;
using ExprPtr = std::shared_ptr<const Expr>;

struct Expr {
  hip_rt::ExprKind kind{hip_rt::ExprKind::Other};
  hip_rt::TypeDesc type;
  // ColumnVar
  int rte_idx{0};
  int column_id{0};
  // Constant
  bool is_null{false};
  int64_t int_val{0};
  double fp_val{0};
  // UOper / BinOper
  hdk::ir::OpType op{hdk::ir::OpType::kEq};
  ExprPtr left, right;  // UOper / ExtractExpr / AggExpr: `left` is the operand / from / arg
  // ExtractExpr
  hdk::ir::DateExtractField field{hdk::ir::DateExtractField::kYear};
  // AggExpr
  hdk::ir::AggType agg{hdk::ir::AggType::kCount};
  bool distinct{false};
};

inline hip_rt::TypeDesc type_of(hip_rt::TypeDesc::Cls cls, int size, bool nullable, int scale = 0) {
  hip_rt::TypeDesc t;
  t.cls = cls;
  t.size = size;
  t.nullable = nullable;
  t.scale = scale;
  return t;
}
inline hip_rt::TypeDesc bigint(bool nullable) { return type_of(hip_rt::TypeDesc::Integer, 8, nullable); }
inline hip_rt::TypeDesc fp64(bool nullable) { return type_of(hip_rt::TypeDesc::Fp, 8, nullable); }

inline ExprPtr column(hip_rt::TypeDesc t, int rte_idx, int column_id) {
  auto e = std::make_shared<Expr>();
  e->kind = hip_rt::ExprKind::ColumnVar;
  e->type = t;
  e->rte_idx = rte_idx;
  e->column_id = column_id;
  return e;
}
inline ExprPtr int_literal(int64_t v, int size = 0) {  // Constant::make: INTEGER when it fits, else BIGINT
  auto e = std::make_shared<Expr>();
  e->kind = hip_rt::ExprKind::Constant;
  e->type = type_of(hip_rt::TypeDesc::Integer, size ? size : ((v >= INT32_MIN && v <= INT32_MAX) ? 4 : 8), false);
  e->int_val = v;
  return e;
}
inline ExprPtr fp_literal(double v) {
  auto e = std::make_shared<Expr>();
  e->kind = hip_rt::ExprKind::Constant;
  e->type = fp64(false);
  e->fp_val = v;
  return e;
}
inline ExprPtr bin_oper(hip_rt::TypeDesc t, hdk::ir::OpType op, ExprPtr l, ExprPtr r) {
  auto e = std::make_shared<Expr>();
  e->kind = hip_rt::ExprKind::BinOper;
  e->type = t;
  e->op = op;
  e->left = std::move(l);
  e->right = std::move(r);
  return e;
}
inline ExprPtr u_oper(hip_rt::TypeDesc t, hdk::ir::OpType op, ExprPtr operand) {
  auto e = std::make_shared<Expr>();
  e->kind = hip_rt::ExprKind::UOper;
  e->type = t;
  e->op = op;
  e->left = std::move(operand);
  return e;
}
inline ExprPtr extract(hdk::ir::DateExtractField f, ExprPtr from) {  // ExtractExpr: BIGINT, nullable like its argument
  auto e = std::make_shared<Expr>();
  e->kind = hip_rt::ExprKind::Extract;
  e->type = bigint(from->type.nullable);
  e->field = f;
  e->left = std::move(from);
  return e;
}
inline ExprPtr agg_expr(hip_rt::TypeDesc t, hdk::ir::AggType a, ExprPtr arg, bool distinct = false) {
  auto e = std::make_shared<Expr>();
  e->kind = hip_rt::ExprKind::Agg;
  e->type = t;
  e->agg = a;
  e->left = std::move(arg);
  e->distinct = distinct;
  return e;
}

inline bool same_tree(const Expr* a, const Expr* b) {  // Expr::operator==
  if (a == b) return true;
  if (!a || !b || a->kind != b->kind) return false;
  switch (a->kind) {
    case hip_rt::ExprKind::ColumnVar: return a->rte_idx == b->rte_idx && a->column_id == b->column_id;
    case hip_rt::ExprKind::Constant: return a->int_val == b->int_val && a->fp_val == b->fp_val && a->is_null == b->is_null;
    case hip_rt::ExprKind::Extract: return a->field == b->field && same_tree(a->left.get(), b->left.get());
    default: return a->op == b->op && same_tree(a->left.get(), b->left.get()) && same_tree(a->right.get(), b->right.get());
  }
}

struct StandInIr {
  using Expr = standin::Expr;
  static hip_rt::ExprKind kind(const Expr* e) { return e->kind; }
  static hip_rt::TypeDesc type(const Expr* e) { return e->type; }
  static bool same(const Expr* a, const Expr* b) { return same_tree(a, b); }
  static int rte_idx(const Expr* e) { return e->rte_idx; }
  static bool const_is_null(const Expr* e) { return e->is_null; }
  static int64_t const_int(const Expr* e) { return e->int_val; }
  static double const_fp(const Expr* e) { return e->fp_val; }
  static hdk::ir::OpType bin_op(const Expr* e) { return e->op; }
  static const Expr* left(const Expr* e) { return e->left.get(); }
  static const Expr* right(const Expr* e) { return e->right.get(); }
  static hdk::ir::OpType un_op(const Expr* e) { return e->op; }
  static const Expr* operand(const Expr* e) { return e->left.get(); }
  static hdk::ir::DateExtractField extract_field(const Expr* e) { return e->field; }
  static const Expr* extract_from(const Expr* e) { return e->left.get(); }
  static hdk::ir::AggType agg_type(const Expr* e) { return e->agg; }
  static const Expr* agg_arg(const Expr* e) { return e->left.get(); }
  static bool agg_distinct(const Expr* e) { return e->distinct; }
};

}  // namespace standin
