// snn_plan.hip -- host-only part of the C ABI: error string, layout / size queries and the batching plan
// (snn_plan_build: the length sort + tiling that replaces utils/training_utils.py:449-467).  No device code and no HIP
// call: this file, snn_featurize.hip and snn_h5.hip also build with a plain host compiler under
// -fsanitize=address,undefined (`make asan`: tests/host_fuzz.cpp drives them with hostile inputs).
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#include "../../include/snn_hip.h"
#include "snn_host.h"

static thread_local char g_err[512] = "";
void snn_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* snn_last_error(void) { return g_err; }
extern "C" int snn_version(void) { return 100; }

int snn_make_layout(const snn_cfg* cfg, SnnLayout* y)
{
    if (!cfg) { snn_set_error("cfg is NULL"); return SNN_ERR_ARG; }
    if (snn_layout_init(y, cfg->input_size, cfg->hidden_size, cfg->num_layers, cfg->bidirectional,
                        cfg->nb_classes, cfg->pool, cfg->precision, cfg->cell) != 0) {
        snn_set_error("invalid cfg: D=%d H=%d L=%d C=%d pool=%d precision=%d cell=%d (need H %% 8 == 0, 1 <= L <= %d; GRU / RNN "
                      "cells in SNN_PREC_F32 only)", cfg->input_size, cfg->hidden_size, cfg->num_layers, cfg->nb_classes,
                      cfg->pool, cfg->precision, cfg->cell, SNN_MAX_LAYERS);
        return SNN_ERR_ARG;
    }
    return SNN_OK;
}

extern "C" int64_t snn_param_count(const snn_cfg* cfg)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    return rc ? rc : y.P;
}
extern "C" int64_t snn_kparam_count(const snn_cfg* cfg)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    return rc ? rc : y.PK;
}
extern "C" int64_t snn_x_step_floats(const snn_cfg* cfg)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    return y.prec != 0 ? (int64_t)y.XB[0] * y.NPC * 64 * 4 : (int64_t)y.Dp * SNN_TILE;
}
extern "C" int32_t snn_feat(const snn_cfg* cfg)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    return rc ? rc : y.F;
}

extern "C" int snn_plan_build(const snn_cfg* cfg, const int32_t* lengths, int32_t n, int32_t* order,
                              int32_t* len_sorted, int32_t* tile_T, int64_t* tile_off)
{
    SnnLayout y;
    int rc = snn_make_layout(cfg, &y);
    if (rc) return rc;
    if (!lengths || !order || !len_sorted || !tile_T || !tile_off || n < 1) {
        snn_set_error("snn_plan_build: null pointer or n_curves < 1");
        return SNN_ERR_ARG;
    }
    // counting sort by length, descending, stable
    int32_t maxlen = 0;
    for (int32_t i = 0; i < n; ++i) {
        if (lengths[i] < 1 || lengths[i] > SNN_MAX_STEPS) {
            snn_set_error("snn_plan_build: lengths[%d] = %d outside [1, %d]", i, lengths[i], SNN_MAX_STEPS);
            return SNN_ERR_ARG;
        }
        maxlen = std::max(maxlen, lengths[i]);
    }
    std::vector<int64_t> start((size_t)maxlen + 2, 0);
    for (int32_t i = 0; i < n; ++i) start[(size_t)(maxlen - lengths[i]) + 1]++;
    for (size_t b = 1; b < start.size(); ++b) start[b] += start[b - 1];
    const int32_t n_tiles = (int32_t)(((int64_t)n + SNN_TILE - 1) / SNN_TILE);
    for (int64_t i = n; i < (int64_t)n_tiles * SNN_TILE; ++i) { order[i] = -1; len_sorted[i] = 0; }
    for (int32_t i = 0; i < n; ++i) {
        const int64_t pos = start[(size_t)(maxlen - lengths[i])]++;
        order[pos] = i;
        len_sorted[pos] = lengths[i];
    }
    int64_t off = 0;
    for (int32_t t = 0; t < n_tiles; ++t) {
        tile_T[t] = len_sorted[(int64_t)t * SNN_TILE];
        tile_off[t] = off;
        off += (int64_t)tile_T[t] * y.Dp * SNN_TILE;
    }
    tile_off[n_tiles] = off;
    return SNN_OK;
}

