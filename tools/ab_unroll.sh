mkdir -p gpurun_out/r5
for r in 1 2; do for u in 1 2 4; do
python bench.py --steps 200 --warmup 20 --graph-unroll $u --no-cpu-baseline --no-pmc --no-aux > gpurun_out/r5/unroll_${u}_$r.json 2>/dev/null && python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('unroll', sys.argv[2], 'ms/step', d['ms_per_step'], 'iter/s', d['value'])" gpurun_out/r5/unroll_${u}_$r.json $u || exit 1
done; done
