set -e
mkdir -p gpurun_out/r2
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py tests/test_gpu_parity.py -k "static or small_frame" -x -v --timeout 120 --timeout-method thread > gpurun_out/r2/t1.log 2>&1 || { tail -30 gpurun_out/r2/t1.log; exit 1; }
tail -5 gpurun_out/r2/t1.log
for v in default w4 w3 w2; do
  L=""; [ $v != default ] && L="--lib build/variants/libacr_hip_$v.so"
  timeout -k 10 120 python tools/kbench.py $L --tag c3_$v --size 3840x2160 --bounces 4 --steps 20 >> gpurun_out/r2/kb1.jsonl 2>> gpurun_out/r2/kb1.err
  timeout -k 10 120 python tools/kbench.py $L --tag c2_$v --steps 100 >> gpurun_out/r2/kb1.jsonl 2>> gpurun_out/r2/kb1.err
done
for v in default w4 w3 w2; do
  L=""; [ $v != default ] && L="--lib build/variants/libacr_hip_$v.so"
  timeout -k 10 200 python tools/kbench.py $L --tag c4_$v --soup 10000000 --depth 22 --steps 20 >> gpurun_out/r2/kb1.jsonl 2>> gpurun_out/r2/kb1.err
done
cat gpurun_out/r2/kb1.jsonl
