set -u
bash tools/profile_gpu.sh r5 > gpurun_out/prof_r5.log 2>&1
bash tools/profile_setup.sh r5_setup c3 pmc > gpurun_out/prof_r5_setup.log 2>&1
bash tools/profile_asm.sh r5_c5asm > gpurun_out/prof_r5_c5asm.log 2>&1
bash tools/profile_c5.sh r5 > gpurun_out/prof_r5_c5.log 2>&1
python bench.py > gpurun_out/r5_bench_n1.json 2> gpurun_out/r5_bench_n1.err
for N in 2 4 8; do
  FDAPDE_BENCH_BACKEND=gloo python bench.py --gpus $N --steps 3 --warmup 1 > gpurun_out/r5_bench_n${N}_one_gpu_shared.json 2> gpurun_out/r5_bench_n${N}.err
done
python tools/small_pde_time.py > gpurun_out/r5_small_pde.txt 2>&1
python tools/gmres_probe.py > gpurun_out/r5_gmres_probe.txt 2>&1
tail -3 gpurun_out/r5_small_pde.txt
wc -c gpurun_out/r5_bench_n*.json
