mkdir -p gpurun_out/r2
export ACR_BENCH_REHEARSAL=1
for i in 1 2 3; do
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 2953$i bench.py --gpus 4 --steps 30 --warmup 3 --workload c5 --no-also > gpurun_out/r2/rehy_$i.json 2> gpurun_out/r2/rehy_$i.err
echo "== run $i rc $?"; grep "check" gpurun_out/r2/rehy_$i.err
done
unset ACR_BENCH_REHEARSAL
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/t9.txt 2>&1; tail -3 gpurun_out/r2/t9.txt
