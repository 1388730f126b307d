#!/bin/bash
# A kernel-experiment copy of the library for same-box A/B runs (VD_LIB=tools/_timing/<name>.so): the named sources are compiled
# with the extra flags, everything else comes from the product build's object cache (csrc/.obj; run _lib.build() first).
#   tools/build_variant.sh <name> "<src.hip ...>" [extra hipcc flags...]      e.g.  tools/build_variant.sh abl1 conv_wino_r64.hip -DVD_R64_ABL=1
set -e
NAME=$1; SRCS=$2; shift; shift
ROOT=$(cd $(dirname $0)/.. && pwd); OUT=$ROOT/tools/_timing; mkdir -p $OUT/$NAME
CS=$ROOT/video-diffusion_amd/csrc
OBJS=""
for o in $CS/.obj/*.o; do
  b=$(basename $o .o); skip=0
  for s in $SRCS; do [ "$b" = "$(basename $s .hip)" ] && skip=1; done
  [ $skip -eq 0 ] && OBJS="$OBJS $o"
done
for s in $SRCS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c $CS/$s -o $OUT/$NAME/$(basename $s .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $OBJS $OUT/$NAME/*.o -o $OUT/$NAME.so
rm -rf $OUT/$NAME
echo built $OUT/$NAME.so
