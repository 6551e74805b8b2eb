# Step-level HBM traffic: rocprofv3 --pmc over an EAGER training step (--no-graph), FETCH_SIZE and WRITE_SIZE in separate passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02_pmc_f -o f -- python3 $R/bench.py --steps 2 --warmup 3 --no-graph --no-cpu-baseline --no-roofline --no-bf16-line > $R/gpurun_out/r02_pmc_f.log 2>&1
echo "fetch rc=$?"
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02_pmc_w -o w -- python3 $R/bench.py --steps 2 --warmup 3 --no-graph --no-cpu-baseline --no-roofline --no-bf16-line > $R/gpurun_out/r02_pmc_w.log 2>&1
echo "write rc=$?"
ls -la $R/gpurun_out/r02_pmc_f $R/gpurun_out/r02_pmc_w | head -20
tail -3 $R/gpurun_out/r02_pmc_f.log
