// snn_bwd_h32.hip -- instantiations of the BPTT kernel (snn_bwd_small.h) for H = 32.
#include "snn_bwd_small.h"

static const BwdEntry g_bwd_h32[] = {SNN_BWD_ENTRIES(32)};
const BwdEntry* snn_bwd_small_h32(int KKI0, int dirs)
{
    for (const BwdEntry& e : g_bwd_h32)
        if (e.KKI0 == KKI0 && e.dirs == dirs) return &e;
    return nullptr;
}
