set -e
cd $GRAFT_REPO_ROOT
cp ladder-vae-pytorch_amd/liblvae_hip.so /tmp/lib_ok.so
touch ladder-vae-pytorch_amd/csrc/conv3x3_bf16.hip
make -C ladder-vae-pytorch_amd/csrc EXTRA=-DLVAE_PHASE_DEBUG > /dev/null 2>&1
for prec in f32 bf16; do for dbg in 0 1 2 4 8 3 6 7 15; do LVAE_BF16_DEBUG=$dbg python tools/phase_bench.py 16 $prec 2>&1 | grep debug; done; done
cp /tmp/lib_ok.so ladder-vae-pytorch_amd/liblvae_hip.so
