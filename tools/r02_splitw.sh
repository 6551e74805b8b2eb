set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "winograd or wgrad" > gpurun_out/splitw_tests.log 2>&1 && \
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-bf16-line > gpurun_out/splitw_on.log 2>&1 && \
LVAE_F32_SPLIT_WGRAD=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-bf16-line > gpurun_out/splitw_off.log 2>&1 && \
LVAE_F32_SPLIT_WGRAD_MIN_M=16384 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-bf16-line > gpurun_out/splitw_16k.log 2>&1
echo rc=$?
tail -3 gpurun_out/splitw_tests.log
grep -h ms_per_step gpurun_out/splitw_on.log gpurun_out/splitw_off.log gpurun_out/splitw_16k.log | python -c "
import sys, json
for l in sys.stdin:
    print(json.loads(l)['ms_per_step'])"
