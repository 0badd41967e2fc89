set -e
mkdir -p gpurun_out/r2
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -k "full_size" -x -v --timeout 300 --timeout-method thread --durations=5 > gpurun_out/r2/t7.txt 2>&1 || { tail -40 gpurun_out/r2/t7.txt; exit 1; }
tail -12 gpurun_out/r2/t7.txt
