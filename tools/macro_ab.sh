# Whole-step A/B of one compile-time switch on ONE box: the product library against a scratch build with -D<MACRO> (the product .so is
# never touched).   bash tools/macro_ab.sh LVAE_W2_NO_UTOUCH [bench.py arguments]       (MACRO may carry a value: NAME=0)
set -e
cd $GRAFT_REPO_ROOT
MACRO=$1; shift
DBG=/tmp/lvae_macro_ab
rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
rm -f $DBG/pkg/csrc/*.o
make -C $DBG/pkg/csrc -j16 EXTRA=-D$MACRO > $DBG/build.log 2>&1 || { tail -20 $DBG/build.log; exit 1; }
for rep in 1 2; do
  for c in product:ladder-vae-pytorch_amd/liblvae_hip.so $MACRO:$DBG/pkg/liblvae_hip.so; do
    n=${c%%:*}; lib=${c#*:}
    echo -n "$n (rep $rep): "
    python tools/step_ab.py $lib --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-roofline --no-bf16-line --no-other-configs "$@" 2> $DBG/err.log | python -c "import json,sys; print('%.3f ms/step' % json.loads(sys.stdin.readline())['ms_per_step'])" || tail -5 $DBG/err.log
  done
done
