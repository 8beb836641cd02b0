// scan_bhm_q.hip -- the multi-argument on-chip group-by (scan_bhm.h, scan_bhm_part.h) behind a plain filter (`column cmp literal`, conjunctions,
// AND / OR / NOT programs: plain_quals.h) -- the Q instantiations; the filter's columns ride in the tile's batch of 16-byte loads
// when they are integers of the streamed width.
#include "scan_bhm_shapes.h"

namespace hdk {

HDK_BHM_DEFINE_KERNELS(4, true, HDK_BHM_SHAPE_FN_NONE, HDK_BHM_PLAIN_BODY_YES)

}  // namespace hdk
