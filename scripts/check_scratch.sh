#!/bin/bash
# Register allocation of every split-f16 conv instantiation (hipcc -Rpass-analysis=kernel-resource-usage): prints the
# ones that use scratch memory.  Known and accepted: the NMD / stack-end patterns 27, 347, 376 (32 - 136 bytes per lane);
# anything in the hundreds of bytes is a spill in the main loop or the epilogue and costs 2 - 10x (round 2 met two).
cd "$(dirname "$0")/../jaeger_amd/csrc" || exit 1
for f in jg_conv_f16_k5 jg_conv_f16_k79 jg_conv_f16_flat jg_conv_f16_lut jg_conv_f16_n64 jg_conv_f16_n32 jg_conv_f16_g128 jg_small; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -I. -Wno-pass-failed \
      -Rpass-analysis=kernel-resource-usage -c -o /tmp/check_scratch.o $f.hip 2>&1 |
    grep -E "Function Name|VGPRs:|ScratchSize" | sed 's/.*remark: [^ ]* //; s/\[-Rpass.*//' | paste - - - |
    sed 's/Function Name: //; s/ScratchSize \[bytes\/lane\]/scratch/' | tr -s ' \t' ' ' |
    awk -v f=$f '{ if ($NF + 0 > 0) print f, $0 }'
done
