#!/bin/bash
# SQ counters of the backward blend kernel with the row walk on / off (DQO_BWD_ROWS), separate --pmc passes (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
for r in 1 0; do
  export DQO_BWD_ROWS=$r
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d /tmp/pmsq$r -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-aux --no-selfcheck "$@" > /tmp/pmsq$r.log 2>&1 || { tail -5 /tmp/pmsq$r.log; exit 1; }
  echo "== rows=$r pass 1"; PMC_KEEP_TEMPLATE=1 python tools/pmc_summary.py /tmp/pmsq$r _kernel | grep -A 12 "^blend_backward"
  timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d /tmp/pmsqb$r -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-aux --no-selfcheck "$@" > /tmp/pmsqb$r.log 2>&1 || { tail -5 /tmp/pmsqb$r.log; exit 1; }
  echo "== rows=$r pass 2"; PMC_KEEP_TEMPLATE=1 python tools/pmc_summary.py /tmp/pmsqb$r _kernel | grep -A 12 "^blend_backward"
done
