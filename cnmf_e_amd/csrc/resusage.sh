#!/bin/bash
# usage: resusage.sh file.hip  -> compact per-kernel register/LDS/occupancy table
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/$(basename "$1").o -Rpass-analysis=kernel-resource-usage 2>&1 | \
sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | \
awk '/error|warning:/ {print} /Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /AGPRs:/ {ag=$NF} /Occupancy/ {o=$NF} /VGPRs Spill/ {sp=$NF} /ScratchSize/ {sc=$NF} /LDS Size/ {printf "%-60s vgpr=%s agpr=%s occ=%s spill=%s scratch=%s lds=%s\n", name, v, ag, o, sp, sc, $NF}'
