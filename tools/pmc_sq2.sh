# second SQ counter set: where the wave cycles go (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY --kernel-trace -d /tmp/pmsq2 -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-aux --no-selfcheck --sustained 0 --window 0 --placement-trials 1 "$@" > /tmp/pmsq2.log 2>&1
echo "rc=$?"; tail -3 /tmp/pmsq2.log | cut -c1-300
python tools/pmc_summary.py /tmp/pmsq2 _kernel
