# rocprofv3 kernel trace of the bench step (fp32 or bf16) -> gpurun_out/r05/<tag>_kernel_by_grid.txt
# usage: bash tools/r05_prof.sh <tag> [extra bench args]
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
mkdir -p gpurun_out/r05
CMD="bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-bf16-line --no-other-configs $@"
rm -rf /tmp/prof_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o g -- python3 $CMD > gpurun_out/r05/${TAG}_prof.log 2>&1 || exit 1
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/${TAG}_kernel_stats.csv \;
{ echo "# rocprofv3 --kernel-trace --stats of: python3 $CMD (15 steps in the trace: 2 eager + capture + 12 replays; divide calls by 15 for per-step counts; the ~3.2 k copyBuffer rows are the host->device parameter copies of model construction)"; python tools/prof_agg.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1) 0 90; } > gpurun_out/r05/${TAG}_kernel_by_grid.txt
python tools/prof_agg.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1) 0 200 last-step > gpurun_out/r05/${TAG}_last_step_by_grid.txt
tail -2 gpurun_out/r05/${TAG}_prof.log
