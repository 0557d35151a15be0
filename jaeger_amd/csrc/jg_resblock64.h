// 64-channel fused residual block (jg_resblock64.hip): the launcher and planner hooks beside jg_resblock.hip's
#pragma once
#include "jg_common.h"

bool jg_resblock64_supports(int c, int k, int dil);
void jg_resblock64_tiling(int L, int k, int dil, int *nb, int *tile_out, int *tiles);
int jg_launch_resblock64(jg_engine *e, const JgResBlockArgs &a, hipStream_t s);
