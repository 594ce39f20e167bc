# Build variants of libsccd_hip.so with different -D switches for A/B measurements on the GPU box:
#   bash tools/variants.sh name1="-DNW_PICK_MIN=8" name2="-DNW_OCC=2" ...   ->  scalable-ccd_amd/sccd/variants/libsccd_<name>.so
# Select one at run time with SCCD_LIB=<path> (sccd/__init__.py).  Only ONE source file is rebuilt per variant:
# narrow.hip, or the one named by VARIANT_SRC (sort, sweep, boxes, api, build, drivers, scan); run `make` first for the other objects.
set -e
cd "$(dirname "$0")/.."
CS=scalable-ccd_amd/csrc
SRC=${VARIANT_SRC:-narrow}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-inline-asm"
mkdir -p scalable-ccd_amd/sccd/variants /tmp/variants
OTHERS="$CS/ti_census.o"
for f in api build drivers boxes scan sort sweep narrow; do [ $f = $SRC ] || OTHERS="$OTHERS $CS/$f.o"; done
for spec in "$@"; do
  name="${spec%%=*}"; defs="${spec#*=}"
  ( /opt/rocm/bin/hipcc $FL $defs -c $CS/$SRC.hip -o /tmp/variants/${SRC}_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o scalable-ccd_amd/sccd/variants/libsccd_$name.so \
      $OTHERS /tmp/variants/${SRC}_$name.o && echo built $name ) &
done
wait
