R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "frame_streams or frame_slots or in_flight" 2>&1 | tail -3
for Q in 4 8; do
  echo -n "Q=$Q probed: "; GPU_MAX_HW_QUEUES=$Q timeout -k 10 120 python3 $R/tools/inflight_time.py 4 2000 2>/dev/null
  echo -n "Q=$Q torch:  "; PANO_TORCH_STREAMS=1 GPU_MAX_HW_QUEUES=$Q timeout -k 10 120 python3 $R/tools/inflight_time.py 4 2000 2>/dev/null
  echo -n "bench Q=$Q: "; GPU_MAX_HW_QUEUES=$Q timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-host-paths --no-c4 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.0f (%.1f us) from_idle %.0f rotating %.0f  %s' % (d['value'], 1e3*d['ms_per_step'], d['from_idle']['value'], d['rotating_inputs_panoramas_per_s'], d['config']['parallelism']))"
done
