#!/bin/bash
# On the GPU box: single-sequence encode latency against the Tier-2 worker count.  usage: tools/t2_threads.sh "8 16 24 32 51"
for t in $1; do
QSVC_T2_THREADS=$t python3 bench.py --sessions 1 --steps 30 --warmup 3 --no-cpu-baseline --no-decode --no-iso 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('t2=$t', d['ms_per_step'], {k:v['ms_per_step'] for k,v in d.get('host_stages',{}).items()})"
done
