# full front-end (bench.py full_frontend, 512 frames per step) for library variants and steps in flight:
#   bash tools/ff_variant_sweep.sh "<lib or -> ..." "<inflight> ..."
for lib in $1; do for n in $2; do
  if [ "$lib" = "-" ]; then unset DRFE_LIB; else export DRFE_LIB=$PWD/build/libdrfe_$lib.so; fi
  echo "lib ${lib} inflight $n"; DRFE_FF_INFLIGHT=$n python tools/full_frontend_sweep.py 512 2>&1 | grep -v amdgpu | cut -c1-420
done; done
