# usage: tools/traffic_calib.sh   — memory-side request counters on access patterns of known size (tools/ubench_traffic.hip)
R="${GRAFT_REPO_ROOT:-$PWD}"
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$R/tools/ubench_traffic.hip" -o /tmp/ubt || exit 1
/tmp/ubt
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rm -rf /tmp/ubt_p$i
  timeout -k 5 120 rocprofv3 --pmc $set --kernel-trace -d /tmp/ubt_p$i -o p --output-format csv -- /tmp/ubt > /tmp/ubt_p$i.log 2>&1
  echo "== set $i rc=$?"
  python "$R/tools/pmc_summary.py" /tmp/ubt_p$i
done
