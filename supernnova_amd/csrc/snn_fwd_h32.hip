// snn_fwd_h32.hip -- instantiations of the persistent LSTM forward kernel (snn_fwd_small.h) for H = 32.
#include "snn_fwd_small.h"

static const FwdSmallEntry g_fwd_h32[] = {SNN_FWD_ENTRIES(32)};
const FwdSmallEntry* snn_fwd_small_h32(int KKI0, int dirs)
{
    for (const FwdSmallEntry& e : g_fwd_h32)
        if (e.KKI0 == KKI0 && e.dirs == dirs) return &e;
    return nullptr;
}
