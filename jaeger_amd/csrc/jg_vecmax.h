// small ops of the widened model families (jg_vecmax.hip): the element-wise maximum over the k groups of a per-window vector
// (JG_OP_VECMAX), the embedding lookup with sinusoidal position rows added (JG_OP_EMBED with a position table)
#pragma once
#include "jg_common.h"

int jg_launch_vecmax(const float *in, int in_ld, int groups, int width, int64_t n_rows, float *out, int out_ld, int out_off,
                     hipStream_t s);
// out[pos][c] = table[min(id, vocab - 1)][c] + (pe != nullptr ? pe[pos % L][c] : 0), mask[pos] = id != 0; ids one or two bytes wide
int jg_launch_embed_pos(const void *ids, int id_bytes, int64_t n_pos, int L, const float *table, int vocab, int c, const float *pe,
                        float *out, uint8_t *mask, hipStream_t s);
