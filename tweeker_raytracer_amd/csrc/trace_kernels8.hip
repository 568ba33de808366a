// The builds of the persistent traversal kernel (trace_persistent.h) that walk the compressed 8-ary nodes (device_types.h,
// bvh_wide8.hip): flattened scenes, with and without cutout opacity. A translation unit of its own so that the two sets of
// kernel builds compile side by side.
#include "trace_persistent.h"

namespace twk {

template<bool COUNT, bool CUTOUT, bool PRIMARY>
static void launchTrace8Variant(const LaunchParams& p, int depth, int gridBlocks, hipStream_t stream)
{
  // seven resident blocks per CU where the registers allow it: no cutout opacity, and not the PRIMARY build (device_types.h TWK_PRIMARY_SIX)
  if (!CUTOUT && p.traceWaves == TWK_TRACE_WAVES7 && !(PRIMARY && TWK_PRIMARY_SIX)) launchTraceVariant<COUNT, false, false, true, PRIMARY, true>(p, depth, gridBlocks, stream);
  else launchTraceVariant<COUNT, CUTOUT, false, false, PRIMARY, true>(p, depth, gridBlocks, stream);
}

void launchTrace8(const LaunchParams& p, int depth, bool count, bool primary, int gridBlocks, hipStream_t stream)
{
  if (p.hasCutout)
  {
    if (primary) { if (count) launchTrace8Variant<true, true, true>(p, depth, gridBlocks, stream);  else launchTrace8Variant<false, true, true>(p, depth, gridBlocks, stream); }
    else         { if (count) launchTrace8Variant<true, true, false>(p, depth, gridBlocks, stream); else launchTrace8Variant<false, true, false>(p, depth, gridBlocks, stream); }
    return;
  }
  if (primary) { if (count) launchTrace8Variant<true, false, true>(p, depth, gridBlocks, stream);  else launchTrace8Variant<false, false, true>(p, depth, gridBlocks, stream); }
  else         { if (count) launchTrace8Variant<true, false, false>(p, depth, gridBlocks, stream); else launchTrace8Variant<false, false, false>(p, depth, gridBlocks, stream); }
}

} // namespace twk
