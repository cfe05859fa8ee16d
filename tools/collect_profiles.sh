#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel traces and counter passes behind the numbers bench.py / DESIGN.md quote.
# Counters in passes of their own (FETCH_SIZE and WRITE_SIZE do not fit one pass; never together with --stats/sys traces).
# Output: gpurun_out/<tag>/<name>/*_results.db (rocpd SQLite) -> tools/profiles_from_db.py turns them into profiles/*.csv / *.json
set -e
TAG=${1:-r04}
ONLY=${2:-all}          # "hot": only the config-4 / config-2 traces and the tracker counters (after a change to tracker / kf / step only);
                        # "yolo": only the detector's traces and MFMA counters + the config-3 trace (after a change to yolo.hip only)
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
trace() { name=$1; shift; rocprofv3 --kernel-trace -d $OUT/$name -o t -- "$@" > $OUT/$name.log 2>&1; echo "trace $name done"; }
pmc() { name=$1; ctr=$2; shift 2; rocprofv3 --kernel-trace --pmc $ctr -d $OUT/$name -o p -- "$@" > $OUT/$name.log 2>&1; echo "pmc $name done"; }
if [ "$ONLY" = yolo ]; then
  trace bench_config3 python3 bench.py --no-also --no-cpu-baseline --workload config3 --min-seconds 0.02
  trace yolo_b64 python3 tools/ybench.py --batch 64 --reps 5
  pmc yolo_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" python3 tools/ybench.py --batch 64 --reps 3
  ls $OUT; exit 0
fi
# the headline (config 4 as worded: one hot_step_kernel launch per time-step, up to four in flight), the same with serial launches,
# and the 256-frame-window throughput form
if [ "$ONLY" != rest ]; then
trace bench_config4 python3 bench.py --no-also --no-cpu-baseline --min-seconds 0.02
trace bench_config4_serial python3 bench.py --no-also --no-cpu-baseline --overlap 1 --min-seconds 0.02
trace bench_config4_window256 python3 bench.py --no-also --no-cpu-baseline --window 256 --min-seconds 0.02
trace bench_config2 python3 bench.py --no-also --no-cpu-baseline --workload config2 --steps 4 --warmup 1 --window 32768 --min-seconds 0.02
fi
# (counter passes run kernels one after the other in any order: the overlapped steps are launched one at a time, tools/stepsync.py)
pmc step_fetch FETCH_SIZE python3 tools/stepsync.py 300
pmc step_write WRITE_SIZE python3 tools/stepsync.py 300
pmc trk_fetch FETCH_SIZE python3 tools/kbench.py --streams 64 --window 256 --stages detect,track --reps 2
pmc trk_write WRITE_SIZE python3 tools/kbench.py --streams 64 --window 256 --stages detect,track --reps 2
if [ "$ONLY" = hot ]; then ls $OUT; exit 0; fi
trace bench_config3 python3 bench.py --no-also --no-cpu-baseline --workload config3 --min-seconds 0.02
trace lane_S64 python3 tools/lbench.py --reps 5
trace yolo_b64 python3 tools/ybench.py --batch 64 --reps 5
pmc lane_fetch FETCH_SIZE python3 tools/lbench.py --reps 2
pmc lane_write WRITE_SIZE python3 tools/lbench.py --reps 2
pmc lane_sq "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" python3 tools/lbench.py --reps 2
pmc plan_fetch FETCH_SIZE python3 tools/kbench.py --streams 64 --window 256 --stages plan --reps 2
pmc plan_write WRITE_SIZE python3 tools/kbench.py --streams 64 --window 256 --stages plan --reps 2
pmc yolo_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" python3 tools/ybench.py --batch 64 --reps 3
pmc yolo_fetch FETCH_SIZE python3 tools/ybench.py --batch 64 --reps 2
pmc yolo_write WRITE_SIZE python3 tools/ybench.py --batch 64 --reps 2
pmc yolo_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES" python3 tools/ybench.py --batch 64 --reps 2
trace yolo_b64_fp32 python3 tools/ybench.py --batch 64 --reps 3 --precision fp32
trace yolo_b1_frame python3 tools/yframe.py
ls $OUT
