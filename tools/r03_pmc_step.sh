# Step-level HBM traffic: rocprofv3 --pmc over an EAGER training step (--no-graph), FETCH_SIZE and WRITE_SIZE in separate passes.
# usage: bash tools/r03_pmc_step.sh <tag> [bench args, e.g. --dtype bf16]   -> gpurun_out/r03_pmc_<tag>_{f,w}/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rm -rf $R/gpurun_out/r03_pmc_${TAG}_f $R/gpurun_out/r03_pmc_${TAG}_w
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03_pmc_${TAG}_f -o f -- python3 $R/bench.py --steps 2 --warmup 3 --no-graph --no-cpu-baseline --no-roofline --no-bf16-line "$@" > $R/gpurun_out/r03_pmc_${TAG}_f.log 2>&1
echo "fetch rc=$?"
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03_pmc_${TAG}_w -o w -- python3 $R/bench.py --steps 2 --warmup 3 --no-graph --no-cpu-baseline --no-roofline --no-bf16-line "$@" > $R/gpurun_out/r03_pmc_${TAG}_w.log 2>&1
echo "write rc=$?"
