set -e
mkdir -p gpurun_out/r2
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/t3.txt 2>&1 || { tail -40 gpurun_out/r2/t3.txt; exit 1; }
tail -2 gpurun_out/r2/t3.txt
timeout -k 10 120 python tools/kbench.py --tag c3_wave4 --size 3840x2160 --bounces 4 --steps 20 
timeout -k 10 120 python tools/kbench.py --lib build/variants/libacr_hip_ww3.so --tag c3_wave3 --size 3840x2160 --bounces 4 --steps 20 
