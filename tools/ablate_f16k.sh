#!/bin/bash
# builds masic_amd/lib/ablate_N/libmasic_hip.so with -DF16K_ABLATE=N (timing experiments; results are garbage by design)
set -e
cd "$(dirname "$0")/../masic_amd/csrc"
OTHERS=$(ls build/*.o | grep -v "conv_f16k")
for n in "$@"; do
  mkdir -p ../lib/ablate_$n
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -ffp-contract=off -DF16K_ABLATE=$n -c conv_f16k.hip -o build/conv_f16k_ab$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/conv_f16k_ab$n.o $OTHERS -o ../lib/ablate_$n/libmasic_hip.so
done
