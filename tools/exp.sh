set -e
mkdir -p gpurun_out/r2
timeout -k 10 500 python -u tools/near_first_stress.py > gpurun_out/r2/stress.txt 2>&1 || { tail -20 gpurun_out/r2/stress.txt; exit 1; }
tail -3 gpurun_out/r2/stress.txt
timeout -k 10 500 python -u tools/soak.py > gpurun_out/r2/soak.txt 2>&1 || { tail -20 gpurun_out/r2/soak.txt; exit 1; }
tail -4 gpurun_out/r2/soak.txt
