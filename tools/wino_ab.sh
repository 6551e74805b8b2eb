# A/B builds of conv3x3_wino.hip on the GPU box: one scratch library per flag set in VARIANTS ("name=flags;name=flags"), timed with
# tools/wino_phase.py. The product .so is never touched.
set -e
cd $GRAFT_REPO_ROOT
DBG=/tmp/lvae_ab_build
VARIANTS="${VARIANTS:-ring3=-DLVAE_W2_RING=3;ring2=-DLVAE_W2_RING=2;old=-DLVAE_TUNING_ENV}"
rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
cd $DBG/pkg/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off"
OBJS=$(ls *.o | grep -v conv3x3_wino.o | tr '\n' ' ')
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do n=${v%%=*}; f=${v#*=}; ( /opt/rocm/bin/hipcc $FLAGS $f -c conv3x3_wino.hip -o wino_$n.o && /opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o $DBG/lib_$n.so $OBJS wino_$n.o ) & done
wait
cd $GRAFT_REPO_ROOT
for H in ${HS:-16 32}; do for v in "${VS[@]}"; do n=${v%%=*}; echo -n "$n: "; LVAE_DISABLE_WINO2=$(case $n in old*) echo 1;; *) echo 0;; esac) python tools/wino_phase.py $H $DBG/lib_$n.so 2>&1 | grep debug || true; done; done
for v in "${VS[@]}"; do n=${v%%=*}; f=${v#*=}; case "$f" in *LVAE_WINO_DBG=64*) for H in ${HS:-16}; do LVAE_STAMPS_4WAVES=$(case $n in q*) echo 1;; *) echo 0;; esac) LVAE_DISABLE_WINO2=$(case $n in old*) echo 1;; *) echo 0;; esac) python tools/wino_stamps.py $H $DBG/lib_$n.so 2>&1 | grep -v Warn; done;; esac; done
