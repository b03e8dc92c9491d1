#!/bin/bash
# Build a VARIANT of libcartnet_hip.so next to the product library, for same-box A/B runs and diagnostic (stamped)
# builds:   tools/build_variant.sh NAME "EXTRA HIPCC FLAGS" file1.hip [file2.hip ...]
# compiles the named translation units of cartnet_amd/csrc with the extra flags into /tmp objects and links them with
# the product build's other objects into cartnet_amd/libcartnet_hip_NAME.so (travels to the GPU box; select it with
# CARTNET_LIB=cartnet_amd/libcartnet_hip_NAME.so).  The product objects and library are not touched.
set -e
NAME=$1; EXTRA=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/cartnet_amd/csrc
python -c "from cartnet_amd import build; build.build(verbose=False)"
TMP=$(mktemp -d)
OBJS=""
for o in $CSRC/*.o; do
  b=$(basename $o .o)
  use=$o
  for f in "$@"; do
    if [ "$(basename $f .hip)" == "$b" ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc -Wno-unused-function -Wno-inline-asm $EXTRA -c $CSRC/$b.hip -o $TMP/$b.o &
      use=$TMP/$b.o
    fi
  done
  OBJS="$OBJS $use"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/cartnet_amd/libcartnet_hip_$NAME.so $OBJS
rm -rf $TMP
echo $ROOT/cartnet_amd/libcartnet_hip_$NAME.so
