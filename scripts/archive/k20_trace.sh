cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/k20t
python scripts/k20_trace.py > gpurun_out/k20t/plain.txt 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/k20t/trace -- python scripts/k20_trace.py > gpurun_out/k20t/traced.txt 2>&1 &&
python scripts/k20_trace.py --read gpurun_out/k20t/trace > gpurun_out/k20t/timeline.txt 2>&1; cat gpurun_out/k20t/plain.txt gpurun_out/k20t/traced.txt; tail -8 gpurun_out/k20t/timeline.txt
