# Final round-2 profiles: kernel trace (by-grid summary + rocprofv3's own stats CSV) and step-level PMC traffic
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/fin_prof && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin_prof -o g -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-bf16-line > gpurun_out/fin_prof.log 2>&1
echo rc=$?
find gpurun_out/fin_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_kernel_stats.csv \;
python tools/prof_agg.py $(find gpurun_out/fin_prof -name "*kernel_trace.csv" | head -1) 0 70 > gpurun_out/r02_kernel_by_grid.txt 2> gpurun_out/prof_agg.err || cat gpurun_out/prof_agg.err
rm -rf gpurun_out/fin_prof
rm -rf gpurun_out/bf_prof && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bf_prof -o g -- python3 bench.py --dtype bf16 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-bf16-line > gpurun_out/bf_prof.log 2>&1
echo rc=$?
python tools/prof_agg.py $(find gpurun_out/bf_prof -name "*kernel_trace.csv" | head -1) 0 50 > gpurun_out/r02_bf16_kernel_by_grid.txt 2>> gpurun_out/prof_agg.err
rm -rf gpurun_out/bf_prof
rm -rf gpurun_out/r02_pmc_f gpurun_out/r02_pmc_w
bash tools/r02_pmc_step.sh
