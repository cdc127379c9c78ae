#!/bin/bash
# register / spill summary of the scan kernel instantiations (cross-compile, no GPU needed)
cd /root/repo/image-restoration-models_amd/csrc || exit 1
mkdir -p /tmp/sl
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -S --cuda-device-only -o /tmp/sl/mamba.s mamba.hip 2>&1 | grep -v "warning: argument" | head
grep -E "^\s+\.(sgpr|vgpr)_(count|spill_count)|\.name:.*scan_chunk" /tmp/sl/mamba.s | grep -A4 scan_chunk | grep -E "name|spill|count" | paste - - - - - | awk '{print $2, "sgpr", $4, "spill", $6, "vgpr", $8, "spill", $10}'
