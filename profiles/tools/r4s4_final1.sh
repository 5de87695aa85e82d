cd $GRAFT_REPO_ROOT
bash profiles/tools/profile_cxx_amr_gaps.sh r4s4_gaps2 > gpurun_out/r4s4_gaps2.txt 2>&1
bash profiles/tools/profile_cxx_amr_trace.sh r4s4_cxx_amr2 > gpurun_out/r4s4_cxx_amr2.txt 2>&1
python bench.py > gpurun_out/r4s4_bench_v6.json 2> gpurun_out/r4s4_bench_v6.err
head -40 gpurun_out/r4s4_gaps2.txt; tail -c 600 gpurun_out/r4s4_bench_v6.err
