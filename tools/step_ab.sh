# Whole-step A/B of library tuning switches on ONE box. The switches exist only in a -DLVAE_TUNING_ENV build, made here in a scratch
# copy of csrc/ (the product library reads no environment variable and is never touched).
#   CASES="name=ENV=val ENV=val;name=..." bash tools/step_ab.sh [bench.py arguments]
set -e
cd $GRAFT_REPO_ROOT
DBG=/tmp/lvae_step_ab
CASES="${CASES:-fold=LVAE_DISABLE_WINO_FOLD=0;nofold=LVAE_DISABLE_WINO_FOLD=1}"
rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
rm -f $DBG/pkg/csrc/*.o
make -C $DBG/pkg/csrc -j8 EXTRA=-DLVAE_TUNING_ENV > $DBG/build.log 2>&1 || { tail -20 $DBG/build.log; exit 1; }
IFS=';' read -ra CS <<< "$CASES"
for rep in 1 2; do
  for c in "${CS[@]}"; do
    n=${c%%=*}; e=${c#*=}
    echo -n "$n (rep $rep): "
    env $e python tools/step_ab.py $DBG/pkg/liblvae_hip.so --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-roofline --no-bf16-line "$@" 2> $DBG/err_$n.log | python -c "import json,sys; print('%.3f ms/step' % json.loads(sys.stdin.readline())['ms_per_step'])" || tail -5 $DBG/err_$n.log
  done
done
