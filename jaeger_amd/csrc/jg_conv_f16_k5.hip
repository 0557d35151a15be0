// instantiation set 1 of the split-f16 convolution kernel (see jg_conv_f16_impl.h)
#define JG_CONV_PART 1
#include "jg_conv_f16_impl.h"
