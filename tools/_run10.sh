bash tools/profile_round.sh r02 > gpurun_out/prof_r02.log 2>&1
python3 tools/make_profile_summary.py r02 > gpurun_out/prof_r02_summary.log 2>&1
bash tools/profile_bf16.sh r02 c3 > gpurun_out/prof16_r02.log 2>&1
bash tools/profile_small.sh r02 > gpurun_out/prof_small_r02.log 2>&1
mkdir -p gpurun_out/summ_r02
cp profiles/r02_* gpurun_out/summ_r02/ 2>/dev/null
cp gpurun_out/prof16_r02_c3/pmc.txt gpurun_out/summ_r02/bf16_c3_pmc.txt
cp gpurun_out/prof16_r02_c3/trace/p_kernel_stats.csv gpurun_out/summ_r02/bf16_c3_kernel_stats.csv
cp gpurun_out/prof16_r02_c3/bench_trace_run.json gpurun_out/summ_r02/bf16_c3_bench_trace_run.json
for d in gpurun_out/prof_small_r02/L*_F*/; do n=$(basename $d); cp $d/p_kernel_stats.csv gpurun_out/summ_r02/small_${n}_kernel_stats.csv; cp gpurun_out/prof_small_r02/$n.txt gpurun_out/summ_r02/small_${n}.txt; done
./build/coexec > gpurun_out/summ_r02/coexec.txt 2>&1
./build/clock > gpurun_out/summ_r02/clock.txt 2>&1
python3 tools/read_stamps.py > gpurun_out/summ_r02/stamps_fused.txt 2>&1
python3 tools/read_stamps16.py 9 512 > gpurun_out/summ_r02/stamps_k16.txt 2>&1
python3 tools/read_stamps_small.py 5 22 > gpurun_out/summ_r02/stamps_small.txt 2>&1
rm -rf gpurun_out/prof_r02 gpurun_out/prof16_r02_c3 gpurun_out/prof_small_r02
ls gpurun_out/summ_r02; tail -n 3 gpurun_out/prof_r02_summary.log
