#!/bin/bash
# GPU visit: PMC evidence (tools/gpu_pmc.sh: 3 counter passes + kernel trace, single stream)
TAG=${1:-r02p}
bash tools/gpu_pmc.sh $TAG 2>&1 | cut -c1-220
find gpurun_out/$TAG -name "*.db" -size +20M -delete
find gpurun_out/$TAG -name "*.csv" -size +8M -delete
ls -la gpurun_out/$TAG | head
