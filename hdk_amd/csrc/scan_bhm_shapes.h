// scan_bhm_shapes.h -- which instantiations of the multi-argument on-chip group-by exist (scan_bhm.h), shared by the four
// translation units that hold them: scan_bhm.hip (4-byte columns, no filter; pass B, matcher and launcher), scan_bhm_q.hip (with
// a plain filter), scan_bhm_w8.hip and scan_bhm_w8q.hip (the same over 8-byte columns).  Four files so that `make -j` compiles
// them side by side.
#pragma once
#include "scan_bhm.h"
#include "scan_bhm_part.h"

namespace hdk {

#ifndef HDK_BHM_U
#define HDK_BHM_U 2  // (A/B builds: -DHDK_BHM_U=4)
#endif
constexpr int kBhmU = HDK_BHM_U;  // 16-byte steps per lane, column and tile over 4-byte columns (8 rows a lane)

// (rows per lane and tile: 8 either way -- two 16-byte steps over 4-byte columns, four over 8-byte ones)
template <class C, int NK, int NS, int W, bool Q>
static const void* bhm_kernel_of(int block) {
  constexpr int U = kBhmU * (W / 4);
  return block == 1024 ? reinterpret_cast<const void*>(hdk_scan_agg_bhm<C, NK, NS, 1024, U, W, Q>)
                       : reinterpret_cast<const void*>(hdk_scan_agg_bhm<C, NK, NS, 256, U, W, Q>);
}
template <int NK, int W, bool Q>
static const void* bhm_dynamic_nk(int ns, int block) {
  return ns == 1 ? bhm_kernel_of<BhmDynamic, NK, 1, W, Q>(block)
                 : (ns == 2 ? bhm_kernel_of<BhmDynamic, NK, 2, W, Q>(block) : bhm_kernel_of<BhmDynamic, NK, 3, W, Q>(block));
}
template <int NK, int W, bool Q>
static const void* bhm_plain_nk(int ns, int block) {
  return ns == 1 ? bhm_kernel_of<BhmPlain<1>, NK, 1, W, Q>(block)
                 : (ns == 2 ? bhm_kernel_of<BhmPlain<2>, NK, 2, W, Q>(block) : bhm_kernel_of<BhmPlain<3>, NK, 3, W, Q>(block));
}
template <int NK, int W, bool Q>
static const void* bhm_scatter_nk(int ns) {
  return ns == 1 ? reinterpret_cast<const void*>(hdk_bhm_scatter<NK, 1, W, Q>)
                 : (ns == 2 ? reinterpret_cast<const void*>(hdk_bhm_scatter<NK, 2, W, Q>) : reinterpret_cast<const void*>(hdk_bhm_scatter<NK, 3, W, Q>));
}

// The shapes with compile-time argument lists (BhmStatic): what the reference's benchmark suite and its neighbours ask for.
// Arguments are numbered in the order the targets name them, columns likewise.  X(NK, NS, code0, code1, code2, code3).
constexpr uint32_t kN = kBhmNone;
#define HDK_BHM_SHAPES(X)                                                                                                          \
  /* MultiStep/MSBS001-005, MSPHS001-...: max(x100) + sum(x100), max(x10), max(x10 + 1) + sum(x10 + 1) */                          \
  X(1, 2, bhm_code(0, true, true, false, 0), bhm_code(1, false, true, false, 0), bhm_code(1, true, true, false, 1), kN)            \
  /* PerfectHashMultiCol/PHM001-006: count / sum / max / min / avg of one column by two keys (and by three); by one key:      */  \
  /* BaselineHash/BH004, PerfectHashSingleCol/PHS004 (10 000 groups: 12 bytes an entry fit one CU's LDS)                      */  \
  X(1, 1, bhm_code(0, true, true, true, 0), kN, kN, kN)                                                                            \
  X(2, 1, bhm_code(0, true, true, true, 0), kN, kN, kN)                                                                            \
  X(3, 1, bhm_code(0, true, true, true, 0), kN, kN, kN)                                                                            \
  /* sums (counts, averages) of two columns: MultiStep/MSBS006-007, MSPHM's SUM(x10), SUM(y10) by one key and by two          */  \
  X(1, 2, bhm_code(0, true, false, false, 0), bhm_code(1, true, false, false, 0), kN, kN)                                          \
  X(2, 2, bhm_code(0, true, false, false, 0), bhm_code(1, true, false, false, 0), kN, kN)                                          \
  /* ... and of three columns (three measures by one key, by two)                                                            */  \
  X(1, 3, bhm_code(0, true, false, false, 0), bhm_code(1, true, false, false, 0), bhm_code(2, true, false, false, 0), kN)          \
  X(2, 3, bhm_code(0, true, false, false, 0), bhm_code(1, true, false, false, 0), bhm_code(2, true, false, false, 0), kN)          \
  /* sum of one column by two keys; sum of `column + literal` by one                                                          */  \
  X(2, 1, bhm_code(0, true, false, false, 0), kN, kN, kN)                                                                          \
  X(1, 1, bhm_code(0, true, false, false, 1), kN, kN, kN)

// One translation unit per (column width, filtered or not): scan_bhm.hip <4, false>, scan_bhm_q.hip <4, true>,
// scan_bhm_w8.hip <8, false>, scan_bhm_w8q.hip <8, true>.  shape_index: the row of HDK_BHM_SHAPES.
template <int W, bool Q>
struct BhmKernels {
  static const void* fixed(int shape_index, int block, bool nulls);  // BhmStatic (nullptr: no such instantiation)
  static const void* dynamic(int nk, int ns, int block);             // BhmDynamic
  static const void* plain(int nk, int ns, int block);               // BhmPlain (nullptr: the filtered quarters have none)
  static const void* scatter(int nk, int ns);                        // pass A of the two-pass form
};
// (the NULL-carrying twins of the compile-time shapes exist for the unfiltered quarters: NULLS_TOO)
#define HDK_BHM_SHAPE_FN(NK, NS, D0, D1, D2, D3) [](int b) -> const void* { return bhm_kernel_of<BhmStatic<D0, D1, D2, D3, false>, NK, NS, kW, kQ>(b); },
#define HDK_BHM_SHAPE_FN_NULLS(NK, NS, D0, D1, D2, D3) [](int b) -> const void* { return bhm_kernel_of<BhmStatic<D0, D1, D2, D3, true>, NK, NS, kW, kQ>(b); },
#define HDK_BHM_SHAPE_FN_NONE(NK, NS, D0, D1, D2, D3) nullptr,
#define HDK_BHM_PLAIN_BODY_YES(W_, Q_) \
  return nk == 1 ? bhm_plain_nk<1, W_, Q_>(ns, block) : (nk == 2 ? bhm_plain_nk<2, W_, Q_>(ns, block) : bhm_plain_nk<3, W_, Q_>(ns, block));
#define HDK_BHM_PLAIN_BODY_NO(W_, Q_) \
  (void)nk;                           \
  (void)ns;                           \
  (void)block;                        \
  return nullptr;
#define HDK_BHM_DEFINE_KERNELS(W_, Q_, NULLS_FN, PLAIN_BODY)                                                                   \
  template <>                                                                                                                  \
  const void* BhmKernels<W_, Q_>::plain(int nk, int ns, int block) {                                                           \
    PLAIN_BODY(W_, Q_)                                                                                                         \
  }                                                                                                                            \
  template <>                                                                                                                  \
  const void* BhmKernels<W_, Q_>::fixed(int shape_index, int block, bool nulls) {                                              \
    using Fn = const void* (*)(int);                                                                                           \
    constexpr int kW = W_;                                                                                                     \
    constexpr bool kQ = Q_;                                                                                                    \
    static const Fn kTable[] = {HDK_BHM_SHAPES(HDK_BHM_SHAPE_FN)};                                                             \
    static const Fn kNulls[] = {HDK_BHM_SHAPES(NULLS_FN)};                                                                     \
    const Fn fn = nulls ? kNulls[shape_index] : kTable[shape_index];                                                           \
    (void)kW;                                                                                                                  \
    (void)kQ;                                                                                                                  \
    return fn ? fn(block) : nullptr;                                                                                           \
  }                                                                                                                            \
  template <>                                                                                                                  \
  const void* BhmKernels<W_, Q_>::dynamic(int nk, int ns, int block) {                                                         \
    return nk == 1 ? bhm_dynamic_nk<1, W_, Q_>(ns, block) : (nk == 2 ? bhm_dynamic_nk<2, W_, Q_>(ns, block) : bhm_dynamic_nk<3, W_, Q_>(ns, block)); \
  }                                                                                                                            \
  template <>                                                                                                                  \
  const void* BhmKernels<W_, Q_>::scatter(int nk, int ns) {                                                                    \
    return nk == 1 ? bhm_scatter_nk<1, W_, Q_>(ns) : (nk == 2 ? bhm_scatter_nk<2, W_, Q_>(ns) : bhm_scatter_nk<3, W_, Q_>(ns)); \
  }
template <> const void* BhmKernels<4, false>::fixed(int, int, bool);
template <> const void* BhmKernels<4, false>::dynamic(int, int, int);
template <> const void* BhmKernels<4, false>::plain(int, int, int);
template <> const void* BhmKernels<4, false>::scatter(int, int);
template <> const void* BhmKernels<4, true>::fixed(int, int, bool);
template <> const void* BhmKernels<4, true>::dynamic(int, int, int);
template <> const void* BhmKernels<4, true>::plain(int, int, int);
template <> const void* BhmKernels<4, true>::scatter(int, int);
template <> const void* BhmKernels<8, false>::fixed(int, int, bool);
template <> const void* BhmKernels<8, false>::dynamic(int, int, int);
template <> const void* BhmKernels<8, false>::plain(int, int, int);
template <> const void* BhmKernels<8, false>::scatter(int, int);
template <> const void* BhmKernels<8, true>::fixed(int, int, bool);
template <> const void* BhmKernels<8, true>::dynamic(int, int, int);
template <> const void* BhmKernels<8, true>::plain(int, int, int);
template <> const void* BhmKernels<8, true>::scatter(int, int);

}  // namespace hdk
