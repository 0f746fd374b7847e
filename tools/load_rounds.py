"""Serial load rounds per kernel: compiles every kernel file to gfx950 assembly and counts, per kernel, how often a group of
global loads (or returning atomics) is followed by an s_waitcnt vmcnt — each such alternation is one memory latency on the
wave's critical path ("Serial load rounds").      python tools/load_rounds.py"""
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dqo-map_amd", "csrc")
NO_SLP = {"rast_backward_blend.hip", "rast_forward_blend.hip"}  # as in the Makefile

for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
    with tempfile.NamedTemporaryFile(suffix=".s") as out:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-S",
               "--cuda-device-only", src, "-o", out.name]
        if os.path.basename(src) in NO_SLP:
            cmd.insert(1, "-fno-slp-vectorize")
        if subprocess.run(cmd, capture_output=True).returncode != 0:
            continue
        lines = open(out.name).read().split("\n")
    for i, l in enumerate(lines):
        if not (re.match(r"^_Z\w+:", l) and "kernel" in l):
            continue
        end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
        rounds, pending, loads = 0, False, 0
        for b in lines[i:end]:
            if re.search(r"\bglobal_load|\bbuffer_load|global_atomic.*sc0", b):
                pending, loads = True, loads + 1
            elif "s_waitcnt" in b and "vmcnt" in b and pending:
                rounds, pending = rounds + 1, False
        name = re.search(r"(\w+_kernel)", l)
        print("%-22s %-30s loads %3d  load->wait rounds %2d" % (os.path.basename(src), name.group(1)[-30:] if name else "?", loads, rounds))
