# A/B builds of conv_wgrad_img.hip on the GPU box: one scratch library per flag set in VARIANTS ("name=flags;name=flags"), timed with
# tools/wgi_phase.py. The product .so is never touched.   HS="8" NS="1024" bash tools/wgi_ab.sh
set -e
cd $GRAFT_REPO_ROOT
DBG=/tmp/lvae_wgi_build
VARIANTS="${VARIANTS:-base=-DLVAE_WGI_DBG=0;noload=-DLVAE_WGI_DBG=1;nomfma=-DLVAE_WGI_DBG=2;nostage=-DLVAE_WGI_DBG=4;nofrag=-DLVAE_WGI_DBG=8;nostore=-DLVAE_WGI_DBG=16;mfmaonly=-DLVAE_WGI_DBG=29;fragonly=-DLVAE_WGI_DBG=23}"
rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
cd $DBG/pkg/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DLVAE_TUNING_ENV"
/opt/rocm/bin/hipcc $FLAGS -c conv_wgrad.hip -o conv_wgrad.o
OBJS=$(ls *.o | grep -v conv_wgrad_img.o | grep -v "^wgi_" | tr '\n' ' ')
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do n=${v%%=*}; f=${v#*=}; ( /opt/rocm/bin/hipcc $FLAGS $f -c conv_wgrad_img.hip -o wgi_$n.o && /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o $DBG/lib_$n.so $OBJS wgi_$n.o -ldl ) & done
wait
cd $GRAFT_REPO_ROOT
for H in ${HS:-8}; do for N in ${NS:-1024}; do for T in ${TPWS:-4}; do for v in "${VS[@]}"; do n=${v%%=*}; echo -n "$n: "; LVAE_WGRAD_IMG_TPW=$T python tools/wgi_phase.py $H $N $DBG/lib_$n.so 2>&1 | grep debug || true; done; done; done; done
