// instantiation set 2 of the split-f16 convolution kernel (see jg_conv_f16_impl.h)
#define JG_CONV_PART 2
#include "jg_conv_f16_impl.h"
