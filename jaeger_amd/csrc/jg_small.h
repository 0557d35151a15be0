// Small-window fused network kernel (jg_small.hip): argument block and launchers.
#pragma once
#include "jg_common.h"

#define JG_SMALL_MAX_LAYERS 5     /* first conv + up to 4 k = 3 convs */
#define JG_SMALL_PARTW 36         /* floats per row of the partial-pool buffer (32 channels, count, pad) */

// every layer is  x = gelu(conv * s1 + t1 [+ shortcut])  optionally followed by  x = gelu(x * s2 + t2)  (tanh-GELU)
struct JgSmallLayer {
  int add;          // + saved shortcut in front of the first activation
  int aff2;         // a second per-channel affine + GELU (the norm behind a residual stack)
  int save;         // keep the layer's output as the shortcut of a later layer
  int tap;          // NMD tap behind the layer's last stage: slot (>= 1) of the row's partial block, 0 = none
};

struct JgSmallArgs {
  const uint8_t *ids;   // (rows, L) codon ids
  const float *lut;     // [k0][vocab + 1][32] first-layer table (row `vocab` = zeros: padding)
  const uint4 *wfrag;   // [n_conv][3 taps][2 chunks][hi|lo][64 lanes] 32x32x16 A-fragments (8 halfs each): one wave per row

  const float *epi;     // [layers][s1 | t1 | s2 | t2][32]
  float *part;          // [rows][n_slots][JG_SMALL_PARTW]: slot 0 pool, slots 1.. NMD taps
  int *overflow;
  long rows;
  int L, L0, pad0, vocab;
  int use_mask, pool_kind;
  int n_conv, k0;       // k = 3 convs behind the first layer; taps of the first layer
  int n_slots;          // 1 + number of NMD taps
  int dbg;              // experiment build only (JG_SMALL_DBG): ablation mask
  JgSmallLayer layer[JG_SMALL_MAX_LAYERS];
};

// per-model state of the fused path (built at model creation when the op program matches the family)
struct JgSmallNet {
  bool valid = false;
  int n_conv = 0, k0 = 0, pad_same0 = 0, use_mask = 0, pool_kind = 0;
  int n_slots = 1, n_taps = 0;
  int tap_part_slot[JG_SMALL_MAX_LAYERS + 1] = {};   // per tap: the program's NMD partial slot and the conv op that fills it
  int tap_conv_op[JG_SMALL_MAX_LAYERS + 1] = {};
  int first_op = 0, pool_op = -1;       // ops [first_op, pool_op) are replaced by the kernel, pool_op by the final reduce
  float *d_lut = nullptr, *d_epi = nullptr, *d_part = nullptr;
  uint4 *d_wfrag = nullptr;
  int64_t part_cap = 0;
  double flops_per_pos0 = 0.0, flops_per_pos = 0.0;   // algorithmic conv FLOPs per output position: first conv, all k = 3 convs
  JgSmallLayer layer[JG_SMALL_MAX_LAYERS];
};

int jg_small_lds_bytes(int n_conv, int k0, int vocab);
bool jg_small_supports(int n_conv, int k0, int vocab);
int jg_small_max_positions(void);
int jg_launch_small_net(jg_engine *e, const JgSmallArgs &a, int n_conv, int k0, hipStream_t s);
int jg_launch_small_pool_final(const float *part, int frames, int n_slots, int slot, int n_win, int kind,
                               const float *moving_mean, float eps, float *out, int out_ld, hipStream_t s);
