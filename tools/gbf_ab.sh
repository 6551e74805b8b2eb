# A/B builds of conv1x1_gate_bwd_fused.hip on the GPU box: one scratch library per flag set in VARIANTS ("name=flags;name=flags"), timed with
# tools/gbf_phase.py (plain and with the deferred apply). The product .so is never touched.
set -e
cd $GRAFT_REPO_ROOT
DBG=/tmp/lvae_gbf_build
VARIANTS="${VARIANTS:-base=-DLVAE_GBF_DBG=0;noload=-DLVAE_GBF_DBG=1;nomfma=-DLVAE_GBF_DBG=2;nostore=-DLVAE_GBF_DBG=4;nogate=-DLVAE_GBF_DBG=8;nomem=-DLVAE_GBF_DBG=5;onlymem=-DLVAE_GBF_DBG=10}"
rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
cd $DBG/pkg/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off"
OBJS=$(ls *.o | grep -v conv1x1_gate_bwd_fused.o | tr '\n' ' ')
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do n=${v%%=*}; f=${v#*=}; ( /opt/rocm/bin/hipcc $FLAGS $f -c conv1x1_gate_bwd_fused.hip -o gbf_$n.o && /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o $DBG/lib_$n.so $OBJS gbf_$n.o -ldl ) & done
wait
cd $GRAFT_REPO_ROOT
for H in ${HS:-16}; do for v in "${VS[@]}"; do n=${v%%=*}; echo -n "$n: "; python tools/gbf_phase.py $H $DBG/lib_$n.so 2>&1 | grep debug || true; echo -n "$n: "; python tools/gbf_phase.py $H $DBG/lib_$n.so ap 2>&1 | grep debug || true; done; done
