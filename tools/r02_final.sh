# Round-2 measurement set, all on ONE box so that the numbers agree with each other:
#   1. rocprofv3 --kernel-trace --stats of the bench command  -> profiles/r02_kernel_by_grid.txt (+ rocprofv3's own kernel_stats CSV)
#   2. the same for --dtype bf16                                -> profiles/r02_bf16_kernel_by_grid.txt
#   3. step-level PMC passes (eager step)                       -> profiles/r02_pmc/hbm_traffic.json
#   4. python bench.py (reads 1 and 3)                          -> profiles/r02_bench_n1.json
# Everything is also copied to gpurun_out/final/ (the box's profiles/ does not travel back).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
CMD="bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-bf16-line"
rm -rf gpurun_out/fin_prof && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin_prof -o g -- python3 $CMD > gpurun_out/final/fin_prof.log 2>&1 || exit 1
find gpurun_out/fin_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/final/r02_kernel_stats.csv \;
{ echo "# rocprofv3 --kernel-trace --stats of: python3 $CMD (15 steps in the trace: 2 eager + capture + 12 replays; divide calls by 15 for per-step counts; the 3221 copyBuffer rows are the host->device parameter copies of model construction)"; python tools/prof_agg.py $(find gpurun_out/fin_prof -name "*kernel_trace.csv" | head -1) 0 80; } > gpurun_out/final/r02_kernel_by_grid.txt
rm -rf gpurun_out/fin_prof
rm -rf gpurun_out/bf_prof && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bf_prof -o g -- python3 $CMD --dtype bf16 > gpurun_out/final/bf_prof.log 2>&1 || exit 1
{ echo "# rocprofv3 --kernel-trace --stats of: python3 $CMD --dtype bf16"; python tools/prof_agg.py $(find gpurun_out/bf_prof -name "*kernel_trace.csv" | head -1) 0 50; } > gpurun_out/final/r02_bf16_kernel_by_grid.txt
rm -rf gpurun_out/bf_prof
rm -rf gpurun_out/r02_pmc_f gpurun_out/r02_pmc_w
bash tools/r02_pmc_step.sh > gpurun_out/final/pmc.log 2>&1
python tools/pmc_step_json.py gpurun_out/r02_pmc_f/f_counter_collection.csv gpurun_out/r02_pmc_w/w_counter_collection.csv 5 gpurun_out/final/hbm_traffic.json > gpurun_out/final/pmc_summary.txt || exit 1
rm -rf gpurun_out/r02_pmc_f gpurun_out/r02_pmc_w
cp gpurun_out/final/r02_kernel_by_grid.txt gpurun_out/final/r02_bf16_kernel_by_grid.txt gpurun_out/final/r02_kernel_stats.csv profiles/
cp gpurun_out/final/hbm_traffic.json profiles/r02_pmc/hbm_traffic.json
python bench.py > gpurun_out/final/r02_bench_n1.json 2> gpurun_out/final/bench.err || exit 1
cat gpurun_out/final/r02_bench_n1.json | cut -c1-600
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
