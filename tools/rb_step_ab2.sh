# Whole-step A/B of the conv2 + gate fusion at the >= 16x16 levels on ONE box
cd $GRAFT_REPO_ROOT
CASES="${CASES:-sep=LVAE_RB_GATE_LARGE=0;fusedgate=LVAE_RB_GATE_LARGE=1}" bash tools/rb_step_ab.sh "$@"
