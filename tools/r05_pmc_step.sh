# Step-level HBM traffic: rocprofv3 --pmc over an EAGER training step (--no-graph), FETCH_SIZE and WRITE_SIZE in separate passes
# (counters only beside --kernel-trace, as the pool requires). usage: bash tools/r05_pmc_step.sh <tag> [bench args, e.g. --dtype bf16]
#   -> gpurun_out/r05_pmc_<tag>_{f,w}/ ; exits non-zero when either pass fails; the number of steps in the run is echoed
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
STEPS=2; WARM=3
rm -rf $R/gpurun_out/r05_pmc_${TAG}_f $R/gpurun_out/r05_pmc_${TAG}_w
rc=0
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r05_pmc_${TAG}_f -o f -- python3 $R/bench.py --steps $STEPS --warmup $WARM --no-graph --no-cpu-baseline --no-roofline --no-bf16-line --no-other-configs "$@" > $R/gpurun_out/r05_pmc_${TAG}_f.log 2>&1 || rc=1
echo "fetch rc=$rc"
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r05_pmc_${TAG}_w -o w -- python3 $R/bench.py --steps $STEPS --warmup $WARM --no-graph --no-cpu-baseline --no-roofline --no-bf16-line --no-other-configs "$@" > $R/gpurun_out/r05_pmc_${TAG}_w.log 2>&1 || rc=1
echo "write rc=$rc"
echo "steps_in_run=$((STEPS + WARM))"
exit $rc
