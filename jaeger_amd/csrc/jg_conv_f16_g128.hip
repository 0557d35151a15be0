// instantiation set 7 of the split-f16 convolution kernel: 128-wide tiles with run-time output geometry (see jg_conv_f16_impl.h)
#define JG_CONV_PART 7
#include "jg_conv_f16_impl.h"
