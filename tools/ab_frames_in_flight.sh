for rep in 1 2; do
for f in 2 3 4; do
 python bench.py --frames-in-flight $f --no-cpu-baseline --no-host-paths --no-isolated-pass --no-c4 --steps 600 > gpurun_out/fif.json 2>gpurun_out/fif.err || exit 1
 python -c "
import json
d=json.loads(open('gpurun_out/fif.json').read().strip().splitlines()[-1]); print('F=$f', d['value'], round(1e3*d['ms_per_step'],1))"
done
PANO_GRAPH=1 python bench.py --no-cpu-baseline --no-host-paths --no-isolated-pass --no-c4 --steps 600 > gpurun_out/fif.json 2>gpurun_out/fif.err || exit 1
python -c "
import json
d=json.loads(open('gpurun_out/fif.json').read().strip().splitlines()[-1]); print('graph F=4', d['value'], round(1e3*d['ms_per_step'],1))"
done
