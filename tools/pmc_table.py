"""Per-kernel table from a tools/pmc_summary.py dump of the SQ counter pass (tools/pmc_sq.sh): python tools/pmc_table.py <file>"""
import sys

cur, rows = None, {}
for l in open(sys.argv[1]):
    if not l.startswith("   "):
        cur = l.strip()
        continue
    c = l.split()
    try:
        rows.setdefault(cur, {})[c[0].replace("SQ_", "")] = float(c[2])
    except (ValueError, IndexError):
        pass
for k, r in sorted(rows.items()):
    if k.endswith("_kernel") and "::" not in k and "WAVES" in r:
        vm = (r.get("INSTS_VMEM_RD", 0) + r.get("INSTS_VMEM_WR", 0)) / 1e6
        print("%-26s waves %7.0f VALU %7.2fM SALU %6.2fM LDS %5.2fM VMEM %5.2fM wait%% %4.0f" % (
            k, r["WAVES"], r["INSTS_VALU"] / 1e6, r["INSTS_SALU"] / 1e6, r.get("INSTS_LDS", 0) / 1e6, vm,
            100 * r["WAIT_INST_ANY"] / max(r["WAVE_CYCLES"], 1)))
