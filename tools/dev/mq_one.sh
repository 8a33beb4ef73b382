#!/bin/bash
# usage: tools/dev/mq_one.sh L V M NB WPB [extra flags]  -> prints register counts + loop mix
HERE=$(cd "$(dirname "$0")" && pwd)
L=$1; V=$2; M=$3; NB=$4; WPB=$5; shift 5
D=/tmp/mq1_${L}_${V}_${M}_${NB}_${WPB}; mkdir -p $D; cd $D
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -DMQ_L=$L -DMQ_V=$V -DMQ_M=$M -DMQ_NB=$NB -DMQ_WPB=$WPB "$@" --save-temps -c $HERE/mq_one.hip -o mq_one.o 2>&1 | grep -v "^$" | head
grep -E "^\s+\.(vgpr_count|agpr_count|vgpr_spill_count|private_segment_fixed_size)" mq_one-hip-amdgcn-amd-amdhsa-gfx950.s | tr -s ' ' | tr '\n' ' '; echo
python3 $HERE/isa_blocks.py mq_one-hip-amdgcn-amd-amdhsa-gfx950.s
