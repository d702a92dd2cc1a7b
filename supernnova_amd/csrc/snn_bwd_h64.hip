// snn_bwd_h64.hip -- instantiations of the BPTT kernel (snn_bwd_small.h) for H = 64.
#include "snn_bwd_small.h"

static const BwdEntry g_bwd_h64[] = {SNN_BWD_ENTRIES(64)};
const BwdEntry* snn_bwd_small_h64(int KKI0, int dirs)
{
    for (const BwdEntry& e : g_bwd_h64)
        if (e.KKI0 == KKI0 && e.dirs == dirs) return &e;
    return nullptr;
}
