cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/bf_prof && rocprofv3 --kernel-trace -d gpurun_out/bf_prof -o g -- python3 bench.py --dtype bf16 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-bf16-line > gpurun_out/bf_prof.log 2>&1
echo rc=$?
db=$(find gpurun_out/bf_prof -name "*.db" | head -1)
python tools/db_agg.py $db "" 70 > gpurun_out/bf16_by_grid.txt
rm -rf gpurun_out/bf_prof
grep ms_per_step gpurun_out/bf_prof.log | cut -c1-200
