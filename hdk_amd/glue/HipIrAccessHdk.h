// HipIrAccessHdk.h -- the IR access policy of HipPlanExtractor.h over the reference's own expression classes.
//
// Compiled only inside an HDK build (IR/Expr.h pulls in Logger -> Boost, absent from this repository's toolchain); every
// accessor used below exists in the reference at the cited lines of omniscidb/IR/Expr.h and omniscidb/IR/Type.h:
// Expr::is<T>() / as<T>() (:76-84), ColumnVar::rteIdx (:152), Constant::isNull / intVal / fpVal (:243-247),
// BinOper::opType / leftOperand / rightOperand (:334,:363-364), UOper::opType / operand (:291,:300), ExtractExpr::field
// / from (:872-873), AggExpr::aggType / arg / isDistinct (:812-815); Type::size / nullable / isInteger / isDecimal /
// isFloatingPoint / isTimestamp / isBoolean / isExtDictionary (IR/Type.h:52-69), DecimalType::scale (:212),
// TimestampType::unit (:265).  tests/cpp/ir_standin.h implements the same policy over a stand-in tree, and the
// harness runs the extractor through it.
#pragma once
#include "HipPlanExtractor.h"
#include "IR/Expr.h"

namespace hip_rt {
struct HdkIrAccess {
  using Expr = hdk::ir::Expr;
  static ExprKind kind(const Expr* e) {
    if (e->is<hdk::ir::ColumnVar>()) return ExprKind::ColumnVar;  // IR/Expr.h:128
    if (e->is<hdk::ir::Constant>()) return ExprKind::Constant;    // :226
    if (e->is<hdk::ir::BinOper>()) return ExprKind::BinOper;      // :324
    if (e->is<hdk::ir::UOper>()) return ExprKind::UOper;          // :279
    if (e->is<hdk::ir::ExtractExpr>()) return ExprKind::Extract;  // :868
    if (e->is<hdk::ir::AggExpr>()) return ExprKind::Agg;          // :788
    return ExprKind::Other;
  }
  static TypeDesc type(const Expr* e) {
    const hdk::ir::Type* t = e->type();
    TypeDesc d;
    d.size = t->size();
    d.nullable = t->nullable();
    if (t->isInteger()) d.cls = TypeDesc::Integer;
    else if (t->isDecimal()) d.cls = TypeDesc::Decimal, d.scale = t->as<hdk::ir::DecimalType>()->scale();
    else if (t->isFloatingPoint()) d.cls = TypeDesc::Fp;
    else if (t->isTimestamp()) d.cls = TypeDesc::Timestamp, d.seconds = t->as<hdk::ir::TimestampType>()->unit() == hdk::ir::TimeUnit::kSecond;
    else if (t->isDate()) d.cls = TypeDesc::Date, d.seconds = t->as<hdk::ir::DateType>()->unit() != hdk::ir::TimeUnit::kDay;  // IR/Type.h:63,277
    else if (t->isBoolean()) d.cls = TypeDesc::Boolean;
    else if (t->isExtDictionary()) d.cls = TypeDesc::Integer;  // dictionary ids
    return d;
  }
  static bool same(const Expr* a, const Expr* b) { return *a == *b; }
  static int rte_idx(const Expr* e) { return e->as<hdk::ir::ColumnVar>()->rteIdx(); }
  static bool const_is_null(const Expr* e) { return e->as<hdk::ir::Constant>()->isNull(); }
  static int64_t const_int(const Expr* e) { return e->as<hdk::ir::Constant>()->intVal(); }
  static double const_fp(const Expr* e) { return e->as<hdk::ir::Constant>()->fpVal(); }
  static hdk::ir::OpType bin_op(const Expr* e) { return e->as<hdk::ir::BinOper>()->opType(); }
  static const Expr* left(const Expr* e) { return e->as<hdk::ir::BinOper>()->leftOperand(); }
  static const Expr* right(const Expr* e) { return e->as<hdk::ir::BinOper>()->rightOperand(); }
  static hdk::ir::OpType un_op(const Expr* e) { return e->as<hdk::ir::UOper>()->opType(); }
  static const Expr* operand(const Expr* e) { return e->as<hdk::ir::UOper>()->operand(); }
  static hdk::ir::DateExtractField extract_field(const Expr* e) { return e->as<hdk::ir::ExtractExpr>()->field(); }
  static const Expr* extract_from(const Expr* e) { return e->as<hdk::ir::ExtractExpr>()->from(); }
  static hdk::ir::AggType agg_type(const Expr* e) { return e->as<hdk::ir::AggExpr>()->aggType(); }
  static const Expr* agg_arg(const Expr* e) { return e->as<hdk::ir::AggExpr>()->arg(); }
  static bool agg_distinct(const Expr* e) { return e->as<hdk::ir::AggExpr>()->isDistinct(); }
};
}  // namespace hip_rt
