# Phase-skip builds of conv3x3_wino.hip on the GPU box: one library per compile-time mask (-DLVAE_WINO_DBG=mask), built in a scratch
# copy (the product .so is never touched). bits: 1 no halo loads, 2 no transform/split/MFMA, 4 no output stores, 8 U from one hot KB,
# 16 no epilogue, 32 no split VALU
set -e
cd $GRAFT_REPO_ROOT
DBG=/tmp/lvae_phase_build
MASKS="${MASKS:-0 1 4 8 16 32 64}"
rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
cd $DBG/pkg/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off"
OBJS=$(ls *.o | grep -v conv3x3_wino.o | tr '\n' ' ')
for m in $MASKS; do ( /opt/rocm/bin/hipcc $FLAGS -DLVAE_WINO_DBG=$m -c conv3x3_wino.hip -o wino_$m.o && /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o $DBG/lib_$m.so $OBJS wino_$m.o ) & done
wait
cd $GRAFT_REPO_ROOT
for H in ${HS:-16}; do for m in $MASKS; do LVAE_WINO_DEBUG=$m python tools/wino_phase.py $H $DBG/lib_$m.so 2>&1 | grep debug || true; done; done
if [ -f $DBG/lib_64.so ]; then for H in ${HS:-16}; do python tools/wino_stamps.py $H $DBG/lib_64.so 2>&1 | grep -v Warning; done; fi
