bash tools/profile_small.sh r02 > gpurun_out/prof_small_r02.log 2>&1
mkdir -p gpurun_out/summ_small
for d in gpurun_out/prof_small_r02/L*_F*/; do n=$(basename $d); cp $d/p_kernel_stats.csv gpurun_out/summ_small/small_${n}_kernel_stats.csv; cp gpurun_out/prof_small_r02/$n.txt gpurun_out/summ_small/small_${n}.txt; done
rm -rf gpurun_out/prof_small_r02
python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err
cat gpurun_out/summ_small/*.txt | grep -v amdgpu
