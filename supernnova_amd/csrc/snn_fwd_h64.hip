// snn_fwd_h64.hip -- instantiations of the persistent LSTM forward kernel (snn_fwd_small.h) for H = 64.
#include "snn_fwd_small.h"

static const FwdSmallEntry g_fwd_h64[] = {SNN_FWD_ENTRIES(64)};
const FwdSmallEntry* snn_fwd_small_h64(int KKI0, int dirs)
{
    for (const FwdSmallEntry& e : g_fwd_h64)
        if (e.KKI0 == KKI0 && e.dirs == dirs) return &e;
    return nullptr;
}
