set -e
mkdir -p gpurun_out/r2
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/t6.txt 2>&1 || { tail -40 gpurun_out/r2/t6.txt; exit 1; }
tail -2 gpurun_out/r2/t6.txt
for i in 1 2; do
  timeout -k 10 120 python tools/kbench.py --tag c3 --size 3840x2160 --bounces 4 --steps 20
  timeout -k 10 120 python tools/kbench.py --tag c3n --size 3840x2160 --bounces 4 --steps 20 --nested
  timeout -k 10 120 python tools/kbench.py --tag c2 --steps 200
done
