// instantiation set 6 of the split-f16 convolution kernel: 32-channel convs (see jg_conv_f16_impl.h)
#define JG_CONV_PART 6
#include "jg_conv_f16_impl.h"
