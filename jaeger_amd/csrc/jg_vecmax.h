// element-wise maximum over the k groups of a per-window vector (jg_vecmax.hip; JG_OP_VECMAX)
#pragma once
#include "jg_common.h"

int jg_launch_vecmax(const float *in, int in_ld, int groups, int width, int64_t n_rows, float *out, int out_ld, int out_off,
                     hipStream_t s);
