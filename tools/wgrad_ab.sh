# A/B builds of conv3x3_wgrad_wino.hip on the GPU box: one scratch library per flag set in VARIANTS ("name=flags;name=flags"), timed with
# tools/wgrad_phase.py. The product .so is never touched.
set -e
cd $GRAFT_REPO_ROOT
DBG=/tmp/lvae_wgab_build
VARIANTS="${VARIANTS:-base=-DLVAE_WGW_DBG=0;noload=-DLVAE_WGW_DBG=1;nomfma=-DLVAE_WGW_DBG=2;nostore=-DLVAE_WGW_DBG=4;notf=-DLVAE_WGW_DBG=8;noepi=-DLVAE_WGW_DBG=16;loop=-DLVAE_WGW_DBG=17;mfmaonly=-DLVAE_WGW_DBG=25}"
rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
cd $DBG/pkg/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off"
OBJS=$(ls *.o | grep -v conv3x3_wgrad_wino.o | tr '\n' ' ')
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do n=${v%%=*}; f=${v#*=}; ( /opt/rocm/bin/hipcc $FLAGS $f -c conv3x3_wgrad_wino.hip -o wgw_$n.o && /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o $DBG/lib_$n.so $OBJS wgw_$n.o ) & done
wait
cd $GRAFT_REPO_ROOT
for H in ${HS:-16}; do for v in "${VS[@]}"; do n=${v%%=*}; echo -n "$n: "; python tools/wgrad_phase.py $H $DBG/lib_$n.so 2>&1 | grep debug || true; done; done
