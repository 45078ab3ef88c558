#!/bin/bash
# usage: build_variant.sh <name> <source.hip> "<extra hipcc flags>"  -> anime_recommendations_amd/libanirec_<name>.so
# (one source recompiled with the flags, linked with the objects of the last regular build; load it with
#  ANIREC_LIB_PATH for a same-box A/B)
set -e
cd "$(dirname "$0")/.."
P=anime_recommendations_amd
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall $3 -c $P/csrc/$2 -o /tmp/variant_$1.o
OBJS=$(ls $P/_obj/*.o | grep -v "/${2%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libanirec_$1.so $OBJS /tmp/variant_$1.o
echo $P/libanirec_$1.so
