#!/bin/bash
# Phase-toggle timing of conv3x3_ws_kernel (AVHOT_WS_DBG: 1 skip MFMA loop, 2 skip epilogue, 4 skip patch loads); runs on the GPU box.
mkdir -p gpurun_out/ws
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in "$@"; do
  AVHOT_WS_DBG=$m rocprofv3 --kernel-trace -d gpurun_out/ws/dbg$m -o t -- python3 tools/ybench.py --batch 64 --reps 5 > gpurun_out/ws/dbg$m.log 2>&1 || exit 1
  echo "== dbg $m"; python tools/rocpd_stats.py gpurun_out/ws/dbg$m/t_results.db | grep conv3x3 | awk -F'"' '{print $2, $3}' | sed 's/void (anonymous namespace):://; s/((anonymous namespace)::ConvArgs, int, int, int, int)//'
done
