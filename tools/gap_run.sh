cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT}"
mkdir -p gpurun_out/r5
rm -rf /tmp/kt
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/kt -o p --output-format csv -- python3 bench.py --inner --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-aux > gpurun_out/r5/gap_run.log 2>&1 || { tail -5 gpurun_out/r5/gap_run.log; exit 1; }
python tools/gap_analysis.py /tmp/kt > gpurun_out/r5/gaps.txt 2>&1
cat gpurun_out/r5/gaps.txt
