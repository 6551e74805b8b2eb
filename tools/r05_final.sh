# Round-5 measurement set, all on ONE box so that the numbers agree with each other:
#   1. rocprofv3 --kernel-trace --stats of the bench command (fp32)   -> r05_kernel_by_grid.txt, r05_last_step_by_grid.txt (one replayed step) (+ rocprofv3's own kernel_stats CSV)
#   2. the same for --dtype bf16                                       -> r05_bf16_kernel_by_grid.txt
#   3. step-level PMC passes (eager step), fp32 and bf16                -> r05_pmc/hbm_traffic.json, hbm_traffic_bf16.json
#   4. whole-step A/B of the deferred BatchNorm-1 apply on this box
#   5. python bench.py (reads 1-3)                                      -> r05_bench_n1.json ; smoke()
# Everything lands in gpurun_out/final5/ (the box's profiles/ does not travel back); copy the summaries to profiles/ afterwards.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/final5
mkdir -p $OUT profiles/r05_pmc gpurun_out/r05
bash tools/r05_prof.sh f32 || exit 1
cp gpurun_out/r05/f32_kernel_by_grid.txt $OUT/r05_kernel_by_grid.txt; cp gpurun_out/r05/f32_kernel_stats.csv $OUT/r05_kernel_stats.csv
bash tools/r05_prof.sh bf16 --dtype bf16 || exit 1
cp gpurun_out/r05/bf16_kernel_by_grid.txt $OUT/r05_bf16_kernel_by_grid.txt
cp gpurun_out/r05/f32_last_step_by_grid.txt $OUT/r05_last_step_by_grid.txt; cp gpurun_out/r05/bf16_last_step_by_grid.txt $OUT/r05_bf16_last_step_by_grid.txt
bash tools/r05_pmc_step.sh f32 > $OUT/pmc_f32.log 2>&1 || { echo "PMC passes (fp32) failed"; tail -5 $OUT/pmc_f32.log; exit 1; }
N=$(grep steps_in_run $OUT/pmc_f32.log | cut -d= -f2)
python tools/pmc_step_json.py gpurun_out/r05_pmc_f32_f/f_counter_collection.csv gpurun_out/r05_pmc_f32_w/w_counter_collection.csv $N $OUT/hbm_traffic.json > $OUT/pmc_summary_f32.txt || exit 1
bash tools/r05_pmc_step.sh bf16 --dtype bf16 > $OUT/pmc_bf16.log 2>&1 || { echo "PMC passes (bf16) failed"; tail -5 $OUT/pmc_bf16.log; exit 1; }
N=$(grep steps_in_run $OUT/pmc_bf16.log | cut -d= -f2)
python tools/pmc_step_json.py gpurun_out/r05_pmc_bf16_f/f_counter_collection.csv gpurun_out/r05_pmc_bf16_w/w_counter_collection.csv $N $OUT/hbm_traffic_bf16.json 57.96e6 > $OUT/pmc_summary_bf16.txt || exit 1
rm -rf gpurun_out/r05_pmc_f32_f gpurun_out/r05_pmc_f32_w gpurun_out/r05_pmc_bf16_f gpurun_out/r05_pmc_bf16_w
cp $OUT/r05_kernel_by_grid.txt $OUT/r05_bf16_kernel_by_grid.txt $OUT/r05_kernel_stats.csv profiles/
cp $OUT/hbm_traffic.json $OUT/hbm_traffic_bf16.json profiles/r05_pmc/
# round-5 whole-step A/B pairs on THIS box (python-side switches, product library): deferred BatchNorm-1 apply on / off
CASES="defer=LVAE_DEFER_APPLY=1;nodefer=LVAE_DEFER_APPLY=0" bash tools/rb_step_ab.sh --no-other-configs > $OUT/r05_defer_step_ab.txt 2>&1
python bench.py > $OUT/r05_bench_n1.json 2> $OUT/bench.err || exit 1
cut -c1-600 $OUT/r05_bench_n1.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cat $OUT/pmc_summary_f32.txt | head -12; cat $OUT/pmc_summary_bf16.txt | head -8; cat $OUT/r05_defer_step_ab.txt
