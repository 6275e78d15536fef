#!/bin/bash
# the driver's form of the bench (--steps 20 --warmup 5: 1.3 ms timed) and the default form with 3 / 4 frames in flight, interleaved
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out/k20_by_flight.jsonl; : > $O
for rep in 1 2 3 4 5 6 7 8; do
  for F in 4 3; do
    timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --frames-in-flight $F --no-cpu-baseline --no-host-paths --no-c4 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'F': $F, 'K': 20, 'value': b['value'], 'from_idle': b['from_idle']['value']}))" >> $O
  done
done
for rep in 1 2 3; do
  for F in 4 3; do
    timeout -k 10 200 python3 bench.py --frames-in-flight $F --no-cpu-baseline --no-host-paths --no-c4 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'F': $F, 'K': 300, 'value': b['value'], 'from_idle': b['from_idle']['value'], 'rotating': b['rotating_inputs_panoramas_per_s']}))" >> $O
  done
done
cat $O
