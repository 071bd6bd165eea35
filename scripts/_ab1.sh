mkdir -p gpurun_out/w2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/w2/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/w2/pytest.log
VARIANTS="r02a cur w5:-DCR_WPE32=5" WORKLOADS="synth10m bunny4096 trex8192" scripts/ab_variants.sh > gpurun_out/w2/ab.log 2>&1; cat gpurun_out/w2/ab.log
