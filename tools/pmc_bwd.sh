#!/bin/bash
# SQ counters of the blend kernels of the current build (two --pmc passes; run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d /tmp/pmA -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-aux --no-selfcheck "$@" > /tmp/pmA.log 2>&1 || { tail -5 /tmp/pmA.log; exit 1; }
PMC_KEEP_TEMPLATE=1 python tools/pmc_summary.py /tmp/pmA blend_
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d /tmp/pmB -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-aux --no-selfcheck "$@" > /tmp/pmB.log 2>&1 || { tail -5 /tmp/pmB.log; exit 1; }
PMC_KEEP_TEMPLATE=1 python tools/pmc_summary.py /tmp/pmB blend_
