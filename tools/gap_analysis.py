"""Launch gaps of the replayed iteration from a rocprofv3 kernel trace:
    cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt -o p --output-format csv -- python3 $REPO/bench.py --inner --steps 50 --warmup 10 --no-cpu-baseline
    python tools/gap_analysis.py /tmp/kt
Per kernel of the iteration: mean duration, mean idle time before it (start - previous kernel's end), over the replays of the timed loop."""
import csv, glob, os, sys
from collections import defaultdict

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
# iterations = runs that start with zero_words_kernel followed by preprocess_kernel ... gaussian_tail_kernel
names = [short(r[2]) for r in rows]
# the iteration = the launches from one preprocess_kernel to the next gaussian_tail_kernel (the most frequent such run: the fused path's)
from collections import Counter
runs = Counter()
for i, n in enumerate(names):
    if n == "preprocess_kernel":
        j = i
        while j < len(names) and j - i < 16 and names[j] != "gaussian_tail_kernel":
            j += 1
        if j < len(names) and names[j] == "gaussian_tail_kernel":
            runs[tuple(names[i:j + 1])] += 1
seq = list(runs.most_common(1)[0][0]) if runs else ["preprocess_kernel"]
print("iteration =", " -> ".join(seq), f"({runs.most_common(1)[0][1] if runs else 0} times in the trace)")
dur, gap, n_it, it_len = defaultdict(float), defaultdict(float), 0, 0.0
i = 0
prev_end = None
while i + len(seq) <= len(rows):
    if names[i:i + len(seq)] == seq:
        if prev_end is not None and rows[i][0] - prev_end < 50_000:  # back-to-back replays only
            for k in range(len(seq)):
                s, e, _ = rows[i + k]
                dur[seq[k]] += e - s
                gap[seq[k]] += s - (rows[i + k - 1][1] if k else prev_end)
            n_it += 1
            it_len += rows[i + len(seq) - 1][1] - prev_end
        prev_end = rows[i + len(seq) - 1][1]
        i += len(seq)
    else:
        i += 1
print(f"{n_it} back-to-back iterations, mean length {it_len / max(n_it, 1) / 1e3:.1f} us")
tg = td = 0.0
for k in seq:
    d, g = dur[k] / max(n_it, 1) / 1e3, gap[k] / max(n_it, 1) / 1e3
    td, tg = td + d, tg + g
    print(f"{k:26s} {d:8.1f} us   idle before {g:6.1f} us")
print(f"{'sum':26s} {td:8.1f} us   idle        {tg:6.1f} us")
