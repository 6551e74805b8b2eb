# Whole-step A/B of the kernel-argument warm-up (lvae_common.h kernarg_warmup) on ONE box: the product library against a scratch build with
# -DLVAE_KERNARG_WARM=0 (the product .so is never touched).   bash tools/kernarg_ab.sh [bench.py arguments]
set -e
cd $GRAFT_REPO_ROOT
DBG=/tmp/lvae_kernarg_ab
rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
rm -f $DBG/pkg/csrc/*.o
make -C $DBG/pkg/csrc -j16 EXTRA=-DLVAE_KERNARG_WARM=0 > $DBG/build.log 2>&1 || { tail -20 $DBG/build.log; exit 1; }
for rep in 1 2; do
  for c in warm:ladder-vae-pytorch_amd/liblvae_hip.so nowarm:$DBG/pkg/liblvae_hip.so; do
    n=${c%%:*}; lib=${c#*:}
    echo -n "$n (rep $rep): "
    python tools/step_ab.py $lib --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-roofline --no-bf16-line --no-other-configs "$@" 2> $DBG/err_$n.log | python -c "import json,sys; print('%.3f ms/step' % json.loads(sys.stdin.readline())['ms_per_step'])" || tail -5 $DBG/err_$n.log
  done
done
