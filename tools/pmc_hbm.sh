# usage: tools/pmc_hbm.sh [extra bench.py flags]
# HBM-side traffic counters (MI355X_MICROARCH.md, HBM): one counter set per pass, per-kernel means via tools/pmc_summary.py
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmh$i -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-aux --no-selfcheck --sustained 0 "$@" > /tmp/pmh$i.log 2>&1
  echo "== set $i rc=$?"; tail -2 /tmp/pmh$i.log | cut -c1-200
  python tools/pmc_summary.py /tmp/pmh$i _kernel
done
