# Whole-step A/B of source files on ONE box: the product library against a scratch build in which every file found in tools/ab_prev/ (the
# previous versions, put there by hand for the experiment, e.g. `git show HEAD:<path> > tools/ab_prev/<file>`) replaces its namesake in csrc/.
#   bash tools/file_ab.sh [bench args]
set -e
cd $GRAFT_REPO_ROOT
DBG=/tmp/lvae_file_ab
rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
cp tools/ab_prev/* $DBG/pkg/csrc/
rm -f $DBG/pkg/csrc/*.o
make -C $DBG/pkg/csrc -j16 > $DBG/build.log 2>&1 || { tail -20 $DBG/build.log; exit 1; }
for rep in 1 2; do
  for c in new:ladder-vae-pytorch_amd/liblvae_hip.so prev:$DBG/pkg/liblvae_hip.so; do
    n=${c%%:*}; lib=${c#*:}
    echo -n "$n (rep $rep): "
    python tools/step_ab.py $lib --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-roofline --no-bf16-line --no-other-configs "$@" 2> $DBG/err.log | python -c "import json,sys; print('%.3f ms/step' % json.loads(sys.stdin.readline())['ms_per_step'])" || tail -5 $DBG/err.log
  done
done
