#!/bin/bash
# A/B of DqoRastCtx.list_split on one box: tools/ab_split.sh "<cfg>:<shard or -> ..." -> gpurun_out/split/<cfg>_<shard>_<runs>.json
mkdir -p gpurun_out/split
values=${2:-"0 512 1024"}
for item in $1; do
  cfg=${item%%:*}; sh=${item##*:}
  for runs in $values; do
    extra=""; tag=full
    if [ "$sh" != "-" ]; then extra="--as-shard $sh"; tag=${sh/\//of}; fi
    timeout -k 10 300 python bench.py --cfg $cfg --growth-every 0 $extra --steps 60 --warmup 10 --no-cpu-baseline --no-pmc --no-aux --list-split $runs \
      > gpurun_out/split/c${cfg}_${tag}_${runs}.json 2> gpurun_out/split/c${cfg}_${tag}_${runs}.err || exit 1
  done
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/split/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    k = d["config"].get("kernel_us") or {}
    print(os.path.basename(f), d["ms_per_step"], {a: round(b, 1) for a, b in k.items()} if isinstance(k, dict) else "")
PY
