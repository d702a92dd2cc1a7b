// snn_host.h -- host-side helpers shared by the translation units of libsnn_hip.so
#pragma once
#include <stdarg.h>
#include "../../include/snn_hip.h"
#include "snn_layout.h"

void snn_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int snn_make_layout(const snn_cfg* cfg, SnnLayout* y);
