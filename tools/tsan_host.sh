#!/bin/bash
# Builds libboslam_hip's HOST code with -fsanitize=thread (device code as usual) into $1 (default /tmp/bh_tsan) and runs
# tests/tsan/tsan_host.cpp from two threads.  Exit code 0 and no "WARNING: ThreadSanitizer" line = clean.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=${1:-/tmp/bh_tsan}
mkdir -p $OUT
cd $ROOT/boslam_amd/csrc
for f in *.hip; do
  b=${f%.hip}
  fl="-ffp-contract=off"; [ $b = ba ] && fl="-ffp-contract=fast"
  if [ ! -f $OUT/$b.o ] || [ $f -nt $OUT/$b.o ] || [ bh_internal.h -nt $OUT/$b.o ]; then
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 $fl -Wno-option-ignored -Xarch_host -fsanitize=thread -c $f -o $OUT/$b.o
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wno-option-ignored -fsanitize=thread -o $OUT/libboslam_tsan.so $OUT/*.o
/opt/rocm/lib/llvm/bin/clang++ -x c++ -O1 -g -std=c++17 -fsanitize=thread -I$ROOT/include $ROOT/tests/tsan/tsan_host.cpp -o $OUT/tsan_host -L$OUT -lboslam_tsan -Wl,-rpath,$OUT -lpthread
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66" $OUT/tsan_host
