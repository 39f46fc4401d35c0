#!/bin/bash
# Build the stand-alone probes (gfx950 binaries next to their sources; git-ignored, they travel to the GPU box with gpurun):
#   bash tools/probes/build.sh [name ...]        (default: the ones tools/sessions/profiles_r04.sh runs)
cd "$(dirname "$0")" || exit 1
names=${*:-"xcd_sem_probe stream_tile_probe wgrad_probe valu_rate_probe"}
for n in $names; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result $n.hip -o $n && echo "built tools/probes/$n"
done
