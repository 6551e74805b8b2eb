# Whole-step A/B of the fused residual-block launches on ONE box (python-side thresholds, product library):
#   bash tools/rb_step_ab.sh [bench.py arguments]
cd $GRAFT_REPO_ROOT
CASES="${CASES:-old=LVAE_RB_FWD_MIN_HW=0 LVAE_RB_BWD_MIN_HW=0;fused=LVAE_RB_FWD_MIN_HW=16 LVAE_RB_BWD_MIN_HW=1}"
IFS=';' read -ra CS <<< "$CASES"
for rep in 1 2; do
  for c in "${CS[@]}"; do
    n=${c%%=*}; e=${c#*=}
    echo -n "$n (rep $rep): "
    env $e python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-roofline --no-bf16-line "$@" 2> /tmp/rb_ab_err_$n.log | python -c "import json,sys; print('%.3f ms/step' % json.loads(sys.stdin.readline())['ms_per_step'])" || tail -5 /tmp/rb_ab_err_$n.log
  done
done
