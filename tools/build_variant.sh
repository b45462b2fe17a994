#!/bin/bash
# tools/build_variant.sh NAME "extra CXXFLAGS"  ->  digdriver_amd/lib/variants/NAME.so  (A/B builds for tools/variant_bench.py)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/dig_variant_$NAME
rm -rf $B && mkdir -p $B/pkg/csrc $B/include $ROOT/digdriver_amd/lib/variants
cp $ROOT/digdriver_amd/csrc/*.hip $ROOT/digdriver_amd/csrc/*.hpp $ROOT/digdriver_amd/csrc/Makefile $B/pkg/csrc/
cp $ROOT/include/dig_hip.h $B/include/
make -s -C $B/pkg/csrc -j8 OUT=$ROOT/digdriver_amd/lib/variants/$NAME.so EXTRA="$*"
ls -la $ROOT/digdriver_amd/lib/variants/$NAME.so
