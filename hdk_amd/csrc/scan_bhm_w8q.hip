// scan_bhm_w8q.hip -- the multi-argument on-chip group-by (scan_bhm.h, scan_bhm_part.h) over 8-byte columns AND behind a plain filter
// (scan_bhm_w8.hip, scan_bhm_q.hip).
#include "scan_bhm_shapes.h"

namespace hdk {

HDK_BHM_DEFINE_KERNELS(8, true, HDK_BHM_SHAPE_FN_NONE, HDK_BHM_PLAIN_BODY_NO)

}  // namespace hdk
