// scan_fast.hip -- the instantiations of the streaming scan/aggregate kernel (scan_agg_fast.h: hdk_scan_agg_direct) and
// the launcher that picks one.  A translation unit of its own: the ~50 instantiations compile in parallel with the rest of
// the library (scan_agg.hip holds the matcher, match_fast, which only needs the argument struct).
#include "host_match.h"
#include "scan_agg_fast.h"

namespace hdk {

template <int KW, int VW, int FIXED, bool Q = false, int XM = 0>
static int32_t launch_direct(const FastArgs& fa, const LaunchShape& shape, hipStream_t s) {
  constexpr int U = (KW != 0 && VW != 0) ? 4 : 8;
  hipLaunchKernelGGL((hdk_scan_agg_direct<KW, VW, U, FIXED, Q, XM>), dim3(shape.grid), dim3(kFastBlock), shape.lds_bytes,
                     s, fa);
  HDK_HIP_CHECK(hipGetLastError());
  return HDK_HIP_OK;
}

// X mode (extra streamed columns: filters between columns, expression arguments): grouped, 8-byte value column; the
// op-list forms of the filtered kernels
template <int KW, int XM>
static int32_t launch_direct_x(const FastArgs& fa, const LaunchShape& shape, hipStream_t s) {
  const int only = fa.nops == 1 ? fa.op_kind[0] : -1;
  if (only == FOP_ADD_U64) return launch_direct<KW, 8, FOP_ADD_U64, false, XM>(fa, shape, s);
  if (fa.nops == 3 && fa.op_kind[0] == FOP_ADD_ONE && fa.op_word[0] == 0 && fa.op_kind[1] == FOP_ADD_U64 &&
      fa.op_kind[2] == FOP_ADD_ONE_IF_NULL) {
    return launch_direct<KW, 8, 100, false, XM>(fa, shape, s);
  }
  return launch_direct<KW, 8, -1, false, XM>(fa, shape, s);
}

// compile-time op list for the single-op shapes (C2: one ds_add_u64 per row; Q1: one +1 per row)
template <int KW>
static int32_t launch_direct_kw(int vw, const FastArgs& fa, const LaunchShape& shape, hipStream_t s) {
  const int only = fa.nops == 1 ? fa.op_kind[0] : -1;
  if (fa.nxq || fa.vform) {  // X mode: KW != 0 and VW == 8 (match_fast)
    if (KW == 0) return HDK_HIP_ERR_UNSUPPORTED;
    constexpr int KX = KW ? KW : 8;
    if (fa.nxprog) return launch_direct_x<KX, 4>(fa, shape, s);  // (match_scan_fast: a program only without extra columns)
    return fa.nx == 2 ? launch_direct_x<KX, 3>(fa, shape, s) : (fa.nx == 1 ? launch_direct_x<KX, 2>(fa, shape, s) : launch_direct_x<KX, 1>(fa, shape, s));
  }
  if (fa.nquals) {  // filtered: KW != 0 and VW == 8 (match_fast); three op-list forms
    if (KW == 0) return HDK_HIP_ERR_UNSUPPORTED;
    constexpr int KQ = KW ? KW : 8;
    if (only == FOP_ADD_U64) return launch_direct<KQ, 8, FOP_ADD_U64, true>(fa, shape, s);
    if (fa.nops == 3 && fa.op_kind[0] == FOP_ADD_ONE && fa.op_word[0] == 0 && fa.op_kind[1] == FOP_ADD_U64 &&
        fa.op_kind[2] == FOP_ADD_ONE_IF_NULL) {
      return launch_direct<KQ, 8, 100, true>(fa, shape, s);
    }
    return launch_direct<KQ, 8, -1, true>(fa, shape, s);
  }
  switch (vw) {
    case 0:
      if (KW != 0 && only == FOP_ADD_ONE) return launch_direct<KW, 0, FOP_ADD_ONE>(fa, shape, s);
      return launch_direct<KW, 0, -1>(fa, shape, s);
    case 4:
      return launch_direct<KW, 4, -1>(fa, shape, s);
    default: {
      if (only == FOP_ADD_U64) return launch_direct<KW, 8, FOP_ADD_U64>(fa, shape, s);
      if (only == FOP_ADD_F64) return launch_direct<KW, 8, FOP_ADD_F64>(fa, shape, s);
      // "row count, sum[, NULL count]" (one AVG, or SUM + COUNT of the same column): compile-time list
      const bool sum_list = (fa.nops == 2 || fa.nops == 3) && fa.op_kind[0] == FOP_ADD_ONE && fa.op_word[0] == 0 &&
                            (fa.op_kind[1] == FOP_ADD_U64 || fa.op_kind[1] == FOP_ADD_F64) &&
                            (fa.nops == 2 || fa.op_kind[2] == FOP_ADD_ONE_IF_NULL);
      if (sum_list && KW != 0) {
        const bool fp = fa.op_kind[1] == FOP_ADD_F64;
        if (fa.nops == 3) {
          return fp ? launch_direct<KW, 8, 101>(fa, shape, s) : launch_direct<KW, 8, 100>(fa, shape, s);
        }
        return fp ? launch_direct<KW, 8, 103>(fa, shape, s) : launch_direct<KW, 8, 102>(fa, shape, s);
      }
      return launch_direct<KW, 8, -1>(fa, shape, s);
    }
  }
}



int32_t launch_fast_direct(int kw, int vw, const FastArgs& fa, const LaunchShape& shape, hipStream_t s) {
  switch (kw) {
    case 0: return launch_direct_kw<0>(vw, fa, shape, s);
    case 1: return launch_direct_kw<1>(vw, fa, shape, s);
    case 2: return launch_direct_kw<2>(vw, fa, shape, s);
    case 4: return launch_direct_kw<4>(vw, fa, shape, s);
    default: return launch_direct_kw<8>(vw, fa, shape, s);
  }
}

}  // namespace hdk
