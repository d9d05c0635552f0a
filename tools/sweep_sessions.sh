#!/bin/bash
# usage: tools/sweep_sessions.sh "<S list>" [ENV=VAL ...]  -- one bench line per session count
for kv in "${@:2}"; do export "$kv"; done
for S in $1; do
  timeout -k 10 200 python3 bench.py --sessions $S 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('S=$S ${*:2}', d['value'], 'fps', d['ms_per_step'], 'ms/step single', d.get('single_sequence',{}).get('ms_per_encode'))" || exit 1
done
