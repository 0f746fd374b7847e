# SQ counters of the blend kernels (two passes), optional experiment switch in $1 (DQO_BWD_EXP).  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
export DQO_BWD_EXP=${1:-0} DQO_BWD_NB=${2:-7}
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rm -rf /tmp/pmb$i
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmb$i -o p --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-aux --no-selfcheck --sustained 0 --no-graph > /tmp/pmb$i.log 2>&1
  echo "== set $i rc=$? (EXP=$DQO_BWD_EXP NB=$DQO_BWD_NB)"
  python tools/pmc_summary.py /tmp/pmb$i blend_
done
