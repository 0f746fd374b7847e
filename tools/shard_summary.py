"""Summarise gpurun_out/shards/c<cfg>_<N>_<R>.json (tools/shard_times.sh): per N the slowest shard's time per iteration = the N-GPU
iteration time the strong-scaling job is bounded by (no data-path collective), and the speed-up over N = 1 that predicts."""
import glob, json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
for f in glob.glob(os.path.join(R, "gpurun_out/shards/c*_*_*.json")):
    m = re.search(r"c(\d+)_(\d+)_(\d+)\.json$", f)
    txt = open(f).read().strip()
    if not m or not txt:
        continue
    d = json.loads(txt.splitlines()[-1])
    c = d["config"]
    rows.setdefault(int(m.group(1)), {}).setdefault(int(m.group(2)), []).append(
        dict(shard=int(m.group(3)), ms=d["ms_per_step"], P_shard=c["P_shard"], P_visible=c["P_visible"], instances=c.get("N_instances"),
             tiles=c.get("num_tiles"), longest_list=c.get("max_tile_list"), objects=len(c["objects_of_rank0"]), selfcheck=c.get("selfcheck"),
             list_split=c.get("list_split"), kernel_us=c.get("kernel_us")))
out = {}
for cfg in sorted(rows):
    base = max(r["ms"] for r in rows[cfg].get(1, [dict(ms=float("nan"))]))
    out[f"cfg{cfg}"] = {}
    for n in sorted(rows[cfg]):
        sh = sorted(rows[cfg][n], key=lambda r: r["shard"])
        worst = max(r["ms"] for r in sh)
        out[f"cfg{cfg}"][f"N={n}"] = dict(slowest_shard_ms=worst, predicted_speedup=round(base / worst, 2), mean_shard_ms=round(sum(r["ms"] for r in sh) / len(sh), 4),
                                         shards=sh)
        slow = max(sh, key=lambda r: r["ms"])
        if slow.get("kernel_us"):
            print("   slowest shard", slow["shard"], "list_split", slow.get("list_split"), {k: round(v) for k, v in slow["kernel_us"].items()})
        print(f"cfg{cfg} N={n}: slowest shard {worst:.3f} ms, mean {sum(r['ms'] for r in sh) / len(sh):.3f} ms -> predicted speed-up {base / worst:.2f}x"
              f" ({len(sh)} of {n} shards measured)")
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "gpurun_out/shards/summary.json"), "w"), indent=1)
