set -e
mkdir -p gpurun_out/r2
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/t14.txt 2>&1 || { tail -40 gpurun_out/r2/t14.txt; exit 1; }
tail -2 gpurun_out/r2/t14.txt
