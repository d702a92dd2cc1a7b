// snn_host.h -- host-side helpers shared by the translation units of libsnn_hip.so
#pragma once
#include <stdarg.h>
#include "../../include/snn_hip.h"
#include "snn_layout.h"

void snn_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int snn_make_layout(const snn_cfg* cfg, SnnLayout* y);
// the Bayesian entry points exist for the LSTM cell only (bayesian_rnn.py:47-69 registers no other layer type)
static inline int snn_lstm_only(const SnnLayout& y, const char* fn)
{
    if (y.cell == 0) return SNN_OK;
    snn_set_error("%s: cell %d (1 GRU, 2 RNN) has no Bayes-by-Backprop kernels: inference-only here (bayesian_rnn.py:47-69: lstm only)", fn, y.cell);
    return SNN_ERR_UNSUPPORTED;
}
