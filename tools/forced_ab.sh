# One-rank rehearsal (LVAE_FORCE_DIST=1) of the multi-rank step on ONE box: what each form of the gradient exchange costs on top of the
# single-rank step (skip = process group initialised, no exchange). The out-of-place exchange puts real RCCL + copy kernels on the stream.
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $2 LVAE_FORCE_DIST=1 python bench.py --steps 20 --no-cpu-baseline --no-roofline --no-bf16-line --no-other-configs 2>/tmp/err_$1.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.2f ms' % d['ms_per_step'], d['config'].get('grad_exchange'))" || tail -3 /tmp/err_$1.log; }
run skip LVAE_SKIP_ALLREDUCE=1
run overlap LVAE_DDP_MODE=overlap
run split7 "LVAE_DDP_MODE=split LVAE_BUCKET_MB=8"
run onebucket_overlap "LVAE_DDP_MODE=overlap LVAE_BUCKET_MB=64"
run inplace_overlap "LVAE_DDP_MODE=overlap LVAE_FORCE_INPLACE=1"
run twobuckets_overlap "LVAE_DDP_MODE=overlap LVAE_BUCKET_MB=32"
run default_split Y=1
