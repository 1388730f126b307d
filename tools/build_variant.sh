#!/bin/bash
# Build a kernel-experiment copy of the library for same-box A/B runs (VD_LIB=tools/_timing/<name>.so):
#   tools/build_variant.sh <name> [git-rev|WORK] [extra hipcc flags...]
set -e
NAME=$1; REV=${2:-WORK}; shift; shift || true
ROOT=$(cd $(dirname $0)/.. && pwd); OUT=$ROOT/tools/_timing; mkdir -p $OUT
SRC=$ROOT/video-diffusion_amd/csrc; INC=$ROOT/include
if [ "$REV" != "WORK" ]; then
  TMP=$(mktemp -d); git -C $ROOT archive $REV video-diffusion_amd/csrc include | tar -x -C $TMP
  SRC=$TMP/video-diffusion_amd/csrc
fi
ls $SRC/*.hip | xargs -P 8 -I{} sh -c '/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value '"$*"' -c {} -o {}.o'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $SRC/*.hip.o -o $OUT/$NAME.so
rm -f $SRC/*.hip.o
echo built $OUT/$NAME.so
