set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "winograd or wgrad or affine or bn_ or grouped or gate" > gpurun_out/wt_tests.log 2>&1 && \
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-bf16-line > gpurun_out/wt_on.log 2>&1 && \
LVAE_GATE_FWD_WT=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-bf16-line > gpurun_out/wt_gate0.log 2>&1 && \
rm -rf gpurun_out/wt_prof && rocprofv3 --kernel-trace -d gpurun_out/wt_prof -o g -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-bf16-line > gpurun_out/wt_prof.log 2>&1
echo rc=$?
tail -3 gpurun_out/wt_tests.log
for f in wt_on wt_gate0; do grep -h ms_per_step gpurun_out/$f.log | python -c "
import sys, json
for l in sys.stdin:
    print('$f', json.loads(l)['ms_per_step'])"; done
db=$(find gpurun_out/wt_prof -name "*.db" | head -1)
python tools/db_agg.py $db "" 45 > gpurun_out/wt_by_grid.txt
rm -rf gpurun_out/wt_prof
