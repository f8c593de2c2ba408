mkdir -p gpurun_out/r02n
timeout -k 10 200 python3 exp/create_phases.py > gpurun_out/r02n/phases.log 2>&1 || { tail -20 gpurun_out/r02n/phases.log; exit 1; }
grep -v "^mirt_scene_create\|bvh:" gpurun_out/r02n/phases.log
timeout -k 10 900 python -u -m pytest tests/test_gpu_configs.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02n/t.log 2>&1 || { tail -30 gpurun_out/r02n/t.log; exit 1; }
tail -2 gpurun_out/r02n/t.log
