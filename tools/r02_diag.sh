set -x
cd $GRAFT_REPO_ROOT
# 1) N>1 code path as one forced RCCL rank (overlap in one graph) vs fused single rank
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r02_bench_fused.log 2>&1 && \
LVAE_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r02_bench_forced_overlap.log 2>&1 && \
LVAE_FORCE_DIST=1 LVAE_DDP_MODE=split python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r02_bench_forced_split.log 2>&1 && \
# 2) the round-1 stall: 2 gloo ranks on one GPU, graphs, phases timed with host syncs
LVAE_DDP_MODE=split LVAE_ALLOW_GLOO_GRAPH=1 LVAE_STEP_TRACE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --steps 2 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r02_ddp2_trace.log 2>&1
echo rc=$?
grep -h "timed\|step-trace" gpurun_out/r02_bench_fused.log gpurun_out/r02_bench_forced_overlap.log gpurun_out/r02_bench_forced_split.log gpurun_out/r02_ddp2_trace.log
