// instantiation set 10 of the split-f16 conv (jg_conv_f16_impl.h, bottom): k = 7, run-time output geometry
#define JG_CONV_PART 10
#include "jg_conv_f16_impl.h"
