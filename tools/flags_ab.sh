# Whole-step A/B of compiler flags on ONE box: the product library against scratch builds of the same sources in which the files of a
# case are compiled with EXTRA flags (a case without files: every file).
#   EXTRA="-fno-slp-vectorize" CASES="all=;wino=conv3x3_wino,resblock_img" bash tools/flags_ab.sh [bench args]
set -e
cd $GRAFT_REPO_ROOT
DBG=/tmp/lvae_flags_ab
CASES="${CASES:-all=}"
IFS=';' read -ra CS <<< "$CASES"
LIBS="product:ladder-vae-pytorch_amd/liblvae_hip.so"
for c in "${CS[@]}"; do
  n=${c%%=*}; files=${c#*=}
  rm -rf $DBG/$n && mkdir -p $DBG/$n/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/$n/pkg/csrc && cp -r include $DBG/$n/include
  rm -f $DBG/$n/pkg/csrc/*.o
  if [ -z "$files" ]; then
    make -C $DBG/$n/pkg/csrc -j16 EXTRA="$EXTRA" > $DBG/$n/build.log 2>&1 || { tail -20 $DBG/$n/build.log; exit 1; }
  else
    args=(); IFS=',' read -ra FS <<< "$files"; for f in "${FS[@]}"; do args+=("FLAGS_$f=$EXTRA"); done
    make -C $DBG/$n/pkg/csrc -j16 "${args[@]}" > $DBG/$n/build.log 2>&1 || { tail -20 $DBG/$n/build.log; exit 1; }
  fi
  LIBS="$LIBS $n:$DBG/$n/pkg/liblvae_hip.so"
done
for rep in 1 2; do
  for c in $LIBS; do
    n=${c%%:*}; lib=${c#*:}
    echo -n "$n (rep $rep): "
    python tools/step_ab.py $lib --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-roofline --no-bf16-line --no-other-configs "$@" 2> $DBG/err.log | python -c "import json,sys; print('%.3f ms/step' % json.loads(sys.stdin.readline())['ms_per_step'])" || tail -5 $DBG/err.log
  done
done
