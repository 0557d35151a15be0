// instantiation set 13 of the split-f16 conv (jg_conv_f16_impl.h, bottom): k = 9, run-time output geometry
#define JG_CONV_PART 13
#include "jg_conv_f16_impl.h"
