#!/bin/bash
# builds masic_amd/lib/ablate_N/libmasic_hip.so with -DF16K_ABLATE=N (timing experiments; results are garbage by design)
set -e
cd "$(dirname "$0")/../masic_amd/csrc"
OTHERS=$(ls build/*.o | grep -v "conv_f16k")
# MACRO=CONVA_ABLATE bash tools/ablate_f16k.sh 1 2 ...  ablates the first-layer kernel instead (libraries lib/ablate_a<N>)
MACRO=${MACRO:-F16K_ABLATE}
for n in "$@"; do
  if [ "$MACRO" = CONVA_ABLATE ]; then out=a$n; else out=$n; fi
  mkdir -p ../lib/ablate_$out
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -ffp-contract=off -D$MACRO=$n -c conv_f16k.hip -o build/conv_f16k_ab$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/conv_f16k_ab$n.o $OTHERS -o ../lib/ablate_$out/libmasic_hip.so
done
