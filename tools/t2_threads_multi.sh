#!/bin/bash
# On the GPU box: 12-sequence throughput against the Tier-2 worker count per context.  usage: tools/t2_threads_multi.sh "1 2 4 8"
for t in $1; do
QSVC_T2_THREADS=$t python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-decode --no-iso 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('t2=$t', d['value'], d['ms_per_step'])"
done
