"""profiles/r01_hbm_traffic.json from the dumps of tools/pmc_hbm.sh and tools/pmc_sq.sh:
    python tools/pmc_to_json.py <pmc_hbm dump> <pmc_sq dump> > profiles/r01_hbm_traffic.json
read_bytes = RDREQ x 64 B, write_bytes = 64 B x WRREQ_64B + 32 B x the rest (MI355X_MICROARCH.md, HBM section)."""
import json
import sys


def parse(path):
    cur, rows = None, {}
    for l in open(path):
        if not l.startswith("   "):
            cur = l.strip()
            continue
        c = l.split()
        try:
            rows.setdefault(cur, {})[c[0]] = float(c[2])
        except (ValueError, IndexError):
            pass
    return rows


hbm, sq = parse(sys.argv[1]), parse(sys.argv[2])
sq2 = parse(sys.argv[3]) if len(sys.argv) > 3 else {}  # tools/pmc_sq2.sh: SQ_ACTIVE_INST_* (units of 4 cycles)
out = {"_comment": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum / TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -- "
                   "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline (tools/pmc_hbm.sh), mean per launch, cfg 3. "
                   "read_bytes = RDREQ x 64 B (the FETCH_SIZE convention of MI355X_MICROARCH.md; wide coalesced reads are 128-B requests "
                   "tallied at 64 B, so streaming kernels such as adam_kernel read 2x this figure; the 16-byte gathers of the blend "
                   "kernels are uncalibrated), write_bytes = 64 B x WRREQ_64B + 32 B x the rest. Infinity-Cache hits are included "
                   "(memory-side of L2).",
       "kernels": {}}
for k, r in sorted(hbm.items()):
    if not k.endswith("_kernel") or "::" in k or "TCC_EA0_RDREQ_sum" not in r:
        continue
    rd, rd32 = r.get("TCC_EA0_RDREQ_sum", 0), r.get("TCC_EA0_RDREQ_32B_sum", 0)
    wr, wr64 = r.get("TCC_EA0_WRREQ_sum", 0), r.get("TCC_EA0_WRREQ_64B_sum", 0)
    e = dict(rdreq=int(rd), rdreq_32B=int(rd32), wrreq=int(wr), wrreq_64B=int(wr64), read_bytes=int(rd * 64),
             write_bytes=int(wr64 * 64 + (wr - wr64) * 32))
    s = sq.get(k, {})
    if "SQ_INSTS_VALU" in s:
        e["valu_insts"] = int(s["SQ_INSTS_VALU"])
        e["salu_insts"] = int(s.get("SQ_INSTS_SALU", 0))
    s2 = sq2.get(k, {})
    if "SQ_ACTIVE_INST_VALU" in s2:
        e["valu_active_quadcycles"] = int(s2["SQ_ACTIVE_INST_VALU"])
        e["wave_quadcycles"] = int(s2.get("SQ_WAVE_CYCLES", 0))
        e["wait_any_quadcycles"] = int(s2.get("SQ_WAIT_ANY", 0))
    out["kernels"][k] = e
out["_comment_valu"] = ("valu_insts = SQ_INSTS_VALU per launch (tools/pmc_sq.sh, same command with --pmc SQ_*): wave-level VALU "
                        "instructions. tools/ubench_valu.hip measures what one costs a SIMD with >= 2 resident waves: 3.0-3.9 cycles "
                        "(at the nominal 2.4 GHz) for plain fp32 / integer ops, 4.3-5.4 for DPP, v_cmp and v_cndmask, 8.3 for "
                        "v_exp / v_rcp / v_permlane*_swap; bench.py prices the issue time at 4 cycles per instruction: "
                        "valu_insts x 4 / (1024 SIMDs x 2.4e9 Hz).  valu_active_quadcycles = SQ_ACTIVE_INST_VALU (tools/pmc_sq2.sh): "
                        "time the waves spend executing VALU instructions, in units of 4 cycles, summed over the chip; x 4 / (1024 SIMDs "
                        "x duration x 2.4e9 Hz) = the fraction of all SIMD cycles in which a VALU instruction executes")
print(json.dumps(out, indent=1))
