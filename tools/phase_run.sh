# Phase-skip builds of conv3x3_bf16.hip (-DLVAE_PHASE_DEBUG) on the GPU box. The debug library and its objects are built in a
# scratch copy of csrc/, so neither the product liblvae_hip.so nor the in-tree objects are ever replaced.
set -e
cd $GRAFT_REPO_ROOT
DBG=/tmp/lvae_phase_build
rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
rm -f $DBG/pkg/csrc/*.o
# the Makefile writes ../liblvae_hip.so relative to csrc/, i.e. $DBG/pkg/liblvae_hip.so; headers are found through ../../include
make -C $DBG/pkg/csrc -j8 EXTRA=-DLVAE_PHASE_DEBUG > $DBG/build.log 2>&1
for prec in f32 bf16; do for dbg in 0 1 2 4 8 3 6 7 15; do LVAE_BF16_DEBUG=$dbg python tools/phase_bench.py 16 $prec $DBG/pkg/liblvae_hip.so 2>&1 | grep debug || true; done; done
