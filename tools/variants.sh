# Build variants of libsccd_hip.so with different -D switches for A/B measurements on the GPU box:
#   bash tools/variants.sh name1="-DNW_PICK_MIN=8" name2="-DNW_OCC=2" ...   ->  scalable-ccd_amd/sccd/variants/libsccd_<name>.so
# Select one at run time with SCCD_LIB=<path> (sccd/__init__.py).  Only narrow.hip is rebuilt per variant.
set -e
cd "$(dirname "$0")/.."
CS=scalable-ccd_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-inline-asm"
mkdir -p scalable-ccd_amd/sccd/variants /tmp/variants
for spec in "$@"; do
  name="${spec%%=*}"; defs="${spec#*=}"
  ( /opt/rocm/bin/hipcc $FL $defs -c $CS/narrow.hip -o /tmp/variants/narrow_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o scalable-ccd_amd/sccd/variants/libsccd_$name.so \
      $CS/api.o $CS/boxes.o $CS/scan.o $CS/sort.o $CS/sweep.o /tmp/variants/narrow_$name.o && echo built $name ) &
done
wait
