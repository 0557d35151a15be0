// instantiation set 5 of the split-f16 convolution kernel: 64-channel convs (see jg_conv_f16_impl.h)
#define JG_CONV_PART 5
#include "jg_conv_f16_impl.h"
