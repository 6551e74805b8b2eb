# Phase-skip builds of conv3x3_wino.hip (-DLVAE_PHASE_DEBUG) on the GPU box, built in a scratch copy (the product .so is never touched).
# bits: 1 no halo loads, 2 no transform/split/MFMA, 4 no output stores, 8 U from one hot KB, 16 no epilogue, 32 no split VALU
set -e
cd $GRAFT_REPO_ROOT
DBG=/tmp/lvae_phase_build
rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
rm -f $DBG/pkg/csrc/*.o
make -C $DBG/pkg/csrc -j16 EXTRA=-DLVAE_PHASE_DEBUG > $DBG/build.log 2>&1 || { tail -20 $DBG/build.log; exit 1; }
for H in 16 8; do for dbg in 0 1 2 4 8 16 32 3 6 20 22 23 31 34; do LVAE_WINO_DEBUG=$dbg python tools/wino_phase.py $H $DBG/pkg/liblvae_hip.so 2>&1 | grep debug || true; done; done
