# Phase stamps of the fused residual-block kernels (resblock_img.hip) on the GPU box: a -DLVAE_RB_DBG library built in a scratch copy
# (the product .so is never touched), then tools/rb_stamps.py per level.
set -e
cd $GRAFT_REPO_ROOT
DBG=/tmp/lvae_rb_build
rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
cd $DBG/pkg/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off"
OBJS=$(ls *.o | grep -v resblock_img.o | tr '\n' ' ')
/opt/rocm/bin/hipcc $FLAGS -DLVAE_RB_DBG $RB_EXTRA -c resblock_img.hip -o rb_dbg.o && /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o $DBG/lib_rb.so $OBJS rb_dbg.o
cd $GRAFT_REPO_ROOT
for H in ${HS:-8 4 2}; do python tools/rb_stamps.py $H $DBG/lib_rb.so 2>&1 | grep -v Warning; done
if [ -n "$INSTEP" ]; then python tools/rb_stamps_instep.py $DBG/lib_rb.so 2>&1 | grep -v "Warning\|amdgpu"; fi
