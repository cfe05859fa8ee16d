#!/bin/bash
# bench.py under a counter pass (kernels run one at a time in the profiler's order): the overlapped loop must report the broken chain
# and the bench must fall back to serial launches and still print its line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/chkfb
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/chkfb -o p -- python3 bench.py --no-also --no-cpu-baseline --min-seconds 0.005 --steps 5 > gpurun_out/chkfb.json 2> gpurun_out/chkfb.err
echo "rc $?"
python3 -c "
import json; j=json.load(open('gpurun_out/chkfb.json')); print(j['value'], j['ms_per_step'], j['config'].get('overlapped_steps'), j['config'].get('overlap_fallback'))"
