mkdir -p gpurun_out/r02c
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02c/pytest.log 2>&1; echo pytest rc=$?; tail -15 gpurun_out/r02c/pytest.log
