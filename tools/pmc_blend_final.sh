# SQ counters of the two blend kernels as they run in the captured iteration (bucket mode, kept tile order, loss tap): two passes.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rm -rf /tmp/pmf$i
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmf$i -o p --output-format csv -- python3 bench.py --path fused --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-aux --no-selfcheck --inner > /tmp/pmf$i.log 2>&1
  echo "== set $i rc=$?"
  python tools/pmc_summary.py /tmp/pmf$i blend_
done
