set -e
mkdir -p gpurun_out/r2
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/t12.txt 2>&1 || { tail -40 gpurun_out/r2/t12.txt; exit 1; }
tail -2 gpurun_out/r2/t12.txt
timeout -k 10 200 python tools/kbench.py --tag c4 --soup 10000000 --depth 22 --steps 20
timeout -k 10 200 python tools/kbench.py --tag c4b2 --soup 10000000 --depth 22 --steps 10 --bounces 3
