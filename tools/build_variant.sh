#!/bin/bash
# tools/build_variant.sh NAME "extra CXXFLAGS"  ->  tools/variants/NAME.so  (A/B builds for tools/variant_bench.py)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/dig_variant_$NAME
rm -rf $B && mkdir -p $B/pkg/csrc $B/include $ROOT/tools/variants
cp $ROOT/digdriver_amd/csrc/*.hip $ROOT/digdriver_amd/csrc/*.hpp $ROOT/digdriver_amd/csrc/Makefile $B/pkg/csrc/
cp $ROOT/include/dig_hip.h $B/include/
make -s -C $B/pkg/csrc -j8 OUT=$ROOT/tools/variants/$NAME.so EXTRA="$*"
ls -la $ROOT/tools/variants/$NAME.so
