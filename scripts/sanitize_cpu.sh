#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over everything the CPU suite executes natively: the host module
# (eva_amd/host/*.h: IR, compiler passes, executors' host logic, wire / SEAL object formats, the CKKS host encoder) and the
# oracle (oracle/*.c).  GPU sanitizers are not available on this pool, so this is the sanitizer coverage there is.
# A copy of the tree is built with the instrumented flags (the in-tree binaries stay as they are) and
# `pytest -m "not gpu"` runs under it:   bash scripts/sanitize_cpu.sh [out file]
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/profiles/r06_sanitizer_cpu.txt}
W=$(mktemp -d /tmp/eva_san.XXXXXX)
trap 'rm -rf "$W"' EXIT
(cd "$R" && tar --exclude=.git --exclude=gpurun_out --exclude=__pycache__ --exclude='_eva*.so' --exclude=libeva_oracle.so -cf - .) | tar -xf - -C "$W"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
cd "$W"
make -C oracle -s clean
make -C oracle -s CFLAGS="$SAN -march=x86-64-v3 -ffp-contract=off -fPIC -std=gnu11" || exit 1
EVA_HOST_CXXFLAGS="$SAN" python -c "
import importlib.util
spec = importlib.util.spec_from_file_location('b', 'eva_amd/buildlib.py'); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.build_host(force=True, verbose=False)" || exit 1
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
{
  echo "# CPU sanitizer pass (scripts/sanitize_cpu.sh): host module + oracle built with"
  echo "#   $SAN"
  echo "# tree $(cd "$R" && git rev-parse --short HEAD 2>/dev/null), $(gcc --version | head -1)"
  echo "# instrumented: $(ldd eva_amd/_eva*.so | grep -c 'libasan\|libubsan') sanitizer runtimes linked by the host module, $(ldd oracle/libeva_oracle.so | grep -c 'libasan\|libubsan') by the oracle"
  LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    timeout 3000 python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -40
} > "$OUT" 2>&1
tail -5 "$OUT"
