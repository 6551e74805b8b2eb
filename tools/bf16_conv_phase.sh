# Phase costs of the bf16 3x3 convolution of the bf16 step (tools/bf16_conv_phase.py) with a -DLVAE_PHASE_DEBUG scratch build.
set -e
cd $GRAFT_REPO_ROOT
DBG=/tmp/lvae_phase_build
rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
rm -f $DBG/pkg/csrc/*.o
make -C $DBG/pkg/csrc -j16 EXTRA=-DLVAE_PHASE_DEBUG > $DBG/build.log 2>&1 || { tail -20 $DBG/build.log; exit 1; }
for dbg in 0 1 2 4 8 3 7 15; do LVAE_BF16_DEBUG=$dbg python tools/bf16_conv_phase.py $DBG/pkg/liblvae_hip.so 2>&1 | grep debug || true; done
