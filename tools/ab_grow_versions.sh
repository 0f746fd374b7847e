mkdir -p gpurun_out/r5
summ() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g=d['config']['growth_steps']
a=sorted(x['ms'] for x in g); b=sorted(x['ms_parts']['filter_attach_scale_init_new_mapping_call'] for x in g)
print(sys.argv[2], 'iter/s', d['value'], 'growth ms median', a[len(a)//2], 'min', a[0], 'max', a[-1], '| searches+attach median', b[len(b)//2], 'min', b[0], 'max', b[-1])
" $1 $2; }
for r in 1 2; do
  python .ab_old/bench.py --cfg 5 --steps 1000 --warmup 50 > gpurun_out/r5/ab_old_$r.json 2> gpurun_out/r5/ab_old_$r.err && summ gpurun_out/r5/ab_old_$r.json old &&
  python bench.py --cfg 5 --steps 1000 --warmup 50 > gpurun_out/r5/ab_new_$r.json 2> gpurun_out/r5/ab_new_$r.err && summ gpurun_out/r5/ab_new_$r.json new || exit 1
done
