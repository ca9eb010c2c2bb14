#!/bin/bash
# Builds tuning variants of the library (every object rebuilt with the same -D flags) into
# build_variants/lib_<name>.so, then restores the default build.  usage: tools/build_variants.sh name:flags ...
set -e
cd "$(dirname "$0")/../vk3dgaussiansplatting_amd/csrc"
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  make clean >/dev/null
  make -j8 EXTRA="$flags" LIBNAME=../../build_variants/lib_$name.so ../../build_variants/lib_$name.so 2>&1 | grep -E "error|warning" || true
done
make clean >/dev/null
make -j8 2>&1 | grep -E "error|warning" || true
ls -la ../../build_variants/
