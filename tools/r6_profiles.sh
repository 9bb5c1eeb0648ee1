set -u
python bench.py > gpurun_out/r6_bench_n1.json 2> gpurun_out/r6_bench_n1.err
for N in 2 4; do
  FDAPDE_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus $N --steps 3 --warmup 1 > gpurun_out/r6_bench_n${N}_one_gpu_shared.json 2> gpurun_out/r6_bench_n${N}.err
  FDAPDE_BENCH_BACKEND=gloo timeout 1500 python bench.py --gpus $N --scaling weak --steps 2 --warmup 1 > gpurun_out/r6_bench_n${N}_weak_one_gpu_shared.json 2> gpurun_out/r6_bench_n${N}_weak.err
done
python tools/group_time.py 119 2 4 8 > gpurun_out/r6_group_time.txt 2>&1
wc -c gpurun_out/r6_bench_n*.json
tail -3 gpurun_out/r6_group_time.txt
