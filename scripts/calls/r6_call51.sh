#!/bin/bash
# roofline object (rocprofv3 duration + PMC FETCH_SIZE traffic) of the Q8_0 9..128-column kernel on the two Q8_0 -np 32 lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for p in tinyllama-1.1b-q8_0 llama3-8b-q8_0; do
  timeout 900 python bench.py --preset $p --np 32 --prefill 128 --steps 64 --cpu-steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:60], d['value'], d['ms_per_step'], json.dumps(d['roofline']))" | cut -c1-900
done
