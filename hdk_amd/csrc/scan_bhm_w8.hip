// scan_bhm_w8.hip -- the multi-argument on-chip group-by (scan_bhm.h, scan_bhm_part.h) over 8-byte columns whose statistics fit
// 32 bits: BIGINT keys and arguments as an Arrow table of int64 brings them.  The rows are narrowed in registers (bhm_narrow),
// everything after that is the 4-byte kernels' code; a value that does not fit raises the stale-statistics flag.
#include "scan_bhm_shapes.h"

namespace hdk {

HDK_BHM_DEFINE_KERNELS(8, false, HDK_BHM_SHAPE_FN_NULLS, HDK_BHM_PLAIN_BODY_YES)

}  // namespace hdk
