for i in 1 2 3 4; do for t in 1 4; do
python bench.py --steps 400 --warmup 40 --placement-trials $t --no-cpu-baseline --no-pmc --no-aux 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('trials $t ms/step', d['ms_per_step'], d['config'].get('selfcheck'), d['config'].get('placement_trials_ms'))" || exit 1
done; done
