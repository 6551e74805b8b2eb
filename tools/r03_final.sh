# Round-3 measurement set, all on ONE box so that the numbers agree with each other:
#   1. rocprofv3 --kernel-trace --stats of the bench command (fp32)   -> profiles/r03_kernel_by_grid.txt (+ rocprofv3's own kernel_stats CSV)
#   2. the same for --dtype bf16                                       -> profiles/r03_bf16_kernel_by_grid.txt
#   3. step-level PMC passes (eager step), fp32 and bf16                -> profiles/r03_pmc/hbm_traffic.json, hbm_traffic_bf16.json
#   4. SQ counters + in-kernel phase stamps of the dominant kernel      -> profiles/r03_pmc/wino2_sq_counters.txt, profiles/r03_wino2_stamps.txt
#   5. python bench.py (reads 1-3)                                      -> profiles/r03_bench_n1.json
# Everything is also copied to gpurun_out/final3/ (the box's profiles/ does not travel back).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/final3
mkdir -p $OUT profiles/r03_pmc
bash tools/r03_prof.sh f32 || exit 1
cp gpurun_out/r03/f32_kernel_by_grid.txt $OUT/r03_kernel_by_grid.txt; cp gpurun_out/r03/f32_kernel_stats.csv $OUT/r03_kernel_stats.csv
bash tools/r03_prof.sh bf16 --dtype bf16 || exit 1
cp gpurun_out/r03/bf16_kernel_by_grid.txt $OUT/r03_bf16_kernel_by_grid.txt
bash tools/r03_pmc_step.sh f32 > $OUT/pmc_f32.log 2>&1
python tools/pmc_step_json.py gpurun_out/r03_pmc_f32_f/f_counter_collection.csv gpurun_out/r03_pmc_f32_w/w_counter_collection.csv 5 $OUT/hbm_traffic.json > $OUT/pmc_summary_f32.txt || exit 1
bash tools/r03_pmc_step.sh bf16 --dtype bf16 > $OUT/pmc_bf16.log 2>&1
python tools/pmc_step_json.py gpurun_out/r03_pmc_bf16_f/f_counter_collection.csv gpurun_out/r03_pmc_bf16_w/w_counter_collection.csv 5 $OUT/hbm_traffic_bf16.json 57.96e6 > $OUT/pmc_summary_bf16.txt || exit 1
rm -rf gpurun_out/r03_pmc_f32_f gpurun_out/r03_pmc_f32_w gpurun_out/r03_pmc_bf16_f gpurun_out/r03_pmc_bf16_w
bash tools/wino_pmc.sh > $OUT/wino2_sq_counters_raw.txt 2>&1
HS=16 VARIANTS="stamps=-DLVAE_WINO_DBG=64" bash tools/wino_ab.sh > $OUT/r03_wino2_stamps.txt 2>&1
cp $OUT/r03_kernel_by_grid.txt $OUT/r03_bf16_kernel_by_grid.txt $OUT/r03_kernel_stats.csv profiles/
cp $OUT/hbm_traffic.json $OUT/hbm_traffic_bf16.json profiles/r03_pmc/
python bench.py > $OUT/r03_bench_n1.json 2> $OUT/bench.err || exit 1
cut -c1-700 $OUT/r03_bench_n1.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cat $OUT/pmc_summary_f32.txt | head -12; cat $OUT/pmc_summary_bf16.txt | head -8
