# Phase costs of the half-slab bf16 weight gradient: scratch builds with -DLVAE_BFH_DBG=<mask> (1 no k-steps, 2 no staging, 4 no global loads,
# 8 no slab stores, 16 no input transform) and the old form (-DLVAE_BF16_WGRAD_HALF=0), timed by tools/bfq_bench.py on ONE box.
set -e
cd $GRAFT_REPO_ROOT
echo "product:"; python tools/bfq_bench.py 2>&1 | grep us
for m in "$@"; do
  DBG=/tmp/lvae_bfq_$m
  rm -rf $DBG && mkdir -p $DBG/pkg && cp -r ladder-vae-pytorch_amd/csrc $DBG/pkg/csrc && cp -r include $DBG/include
  rm -f $DBG/pkg/csrc/*.o
  make -C $DBG/pkg/csrc -j16 EXTRA=-D$m > $DBG/build.log 2>&1 || { tail -20 $DBG/build.log; exit 1; }
  echo "$m:"; python tools/bfq_bench.py $DBG/pkg/liblvae_hip.so 2>&1 | grep us
done
