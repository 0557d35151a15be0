// instantiation set 11 of the split-f16 conv (jg_conv_f16_impl.h, bottom): k = 9, run-time output geometry
#define JG_CONV_PART 11
#include "jg_conv_f16_impl.h"
