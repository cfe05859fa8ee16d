#!/bin/bash
# Kernel trace of the YOLO forward (64 frames) + its launch-by-launch timeline; runs on the GPU box:
#   gpurun -- 'tools/wsdbg.sh [ENV=VALUE ...]'      e.g. tools/wsdbg.sh AVHOT_YOLO_NO_FUSE=1
mkdir -p gpurun_out/ws
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "$@"; do case "$kv" in *=*) export "$kv";; esac; done
rocprofv3 --kernel-trace -d gpurun_out/ws/dbg0 -o t -- python3 tools/ybench.py --batch 64 --reps 5 > gpurun_out/ws/dbg0.log 2>&1 || exit 1
python tools/rocpd_stats.py gpurun_out/ws/dbg0/t_results.db | awk -F'"' 'NR>1 {print $2, $3}' | sed 's/void (anonymous namespace):://; s/((anonymous namespace)::[^)]*)//' | head -30
python tools/ytimeline.py gpurun_out/ws/dbg0/t_results.db
