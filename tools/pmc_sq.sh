# SQ instruction counters of one short bench run, aggregated per kernel (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d /tmp/pmsq -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-aux --no-selfcheck --sustained 0 --window 0 --placement-trials 1 "$@" > /tmp/pmsq.log 2>&1
python tools/pmc_summary.py /tmp/pmsq _kernel
