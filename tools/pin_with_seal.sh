#!/bin/bash
# tools/pin_with_seal.sh — ONE command that pins this repository's oracle (and with it every "bit-exact" claim of the
# HIP path) to Microsoft SEAL's own bits, on any host that has SEAL >= 3.6 installed — the reference's own requirement
# (/root/reference/CMakeLists.txt:24 `find_package(SEAL 3.6 REQUIRED)`, README.md:28-30 pins v3.6.4).
#
#   bash tools/pin_with_seal.sh [WORK_DIR]          (SEAL_DIR / CMAKE_PREFIX_PATH point cmake at SEAL when it is not
#                                                    in a default location)
#   bash tools/pin_with_seal.sh --dry-run [WORK_DIR] (no SEAL needed: exports the vectors, type-checks the checker
#                                                    against the declarations in tools/seal_api_stub, writes a report
#                                                    that says "dry run"; what tests/test_seal_parity.py runs)
#
# Steps: (1) export the vector sets with tests/golden/export_seal_vectors.py — the committed N = 1024 golden set, a
# fresh set at N = 8192 with one of EVA's 20-bit output primes (60,20,60,60) and a fresh set at the metric configuration
# N = 65536, 10 data limbs + special prime (PIN_CONFIGS overrides the list: "N:bits N:bits ..."; "golden" = the committed
# set); (2) build tools/seal_parity.cpp against find_package(SEAL 3.6) (tools/CMakeLists.txt); (3) run its sections 1-7 on
# every set — primes, psi, NTT, every Evaluator call of seal_executor.h:114-243, the hoisting / fused shortcuts, encode,
# decrypt, decode, SEAL's object format — and --time-triple on the last set; (4) write profiles/seal_pin.json
# ({"sections": ..., "all_identical": ..., "seal_triples_per_s": ...}), which bench.py reports as cpu_baseline.kind =
# "reference" and DESIGN.md section 2 cites.  Exit status 0 only when every section of every set is identical.
set -u
DRY=0
if [ "${1:-}" = "--dry-run" ]; then DRY=1; shift; fi
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WORK=${1:-$ROOT/build/seal_pin}
OUT=${PIN_JSON:-$ROOT/profiles/seal_pin.json}
CONFIGS=${PIN_CONFIGS:-"golden 8192:60,20,60,60 65536:60,60,60,60,60,60,60,60,60,60,60"}
PY=${PYTHON:-python3}
mkdir -p "$WORK"
echo "[pin] vector sets: $CONFIGS"
DIRS=()
for cfg in $CONFIGS; do
  if [ "$cfg" = "golden" ]; then
    d=$WORK/vec_golden
    $PY "$ROOT/tests/golden/export_seal_vectors.py" "$d" > "$WORK/export_golden.log" 2>&1 || { echo "[pin] export failed: $cfg"; tail -5 "$WORK/export_golden.log"; exit 2; }
  else
    n=${cfg%%:*}; bits=${cfg#*:}
    d=$WORK/vec_$n
    $PY "$ROOT/tests/golden/export_seal_vectors.py" "$d" "$n" "$bits" > "$WORK/export_$n.log" 2>&1 || { echo "[pin] export failed: $cfg"; tail -5 "$WORK/export_$n.log"; exit 2; }
  fi
  DIRS+=("$d")
done
if [ $DRY -eq 1 ]; then
  echo "[pin] dry run: type-checking tools/seal_parity.cpp against tools/seal_api_stub (declarations only, pins nothing)"
  g++ -std=c++17 -fsyntax-only -Wall -Wextra -I "$ROOT/tools/seal_api_stub" "$ROOT/tools/seal_parity.cpp" > "$WORK/syntax.log" 2>&1 || { cat "$WORK/syntax.log"; exit 3; }
  $PY "$ROOT/tools/seal_pin_report.py" --dry-run --out "$OUT" "${DIRS[@]}"
  exit $?
fi
echo "[pin] building tools/seal_parity against find_package(SEAL 3.6)"
cmake -S "$ROOT/tools" -B "$WORK/build" -DCMAKE_BUILD_TYPE=Release > "$WORK/cmake.log" 2>&1 || { echo "[pin] SEAL absent or cmake failed:"; tail -5 "$WORK/cmake.log"; exit 4; }
cmake --build "$WORK/build" -j 8 > "$WORK/build.log" 2>&1 || { echo "[pin] build failed:"; tail -20 "$WORK/build.log"; exit 5; }
rc=0
last=$((${#DIRS[@]} - 1))
for i in "${!DIRS[@]}"; do
  d=${DIRS[$i]}
  flag=""; [ "$i" -eq "$last" ] && flag="--time-triple"
  echo "[pin] seal_parity $d $flag"
  "$WORK/build/seal_parity" "$d" $flag > "$d/seal_parity.log" 2>&1 || rc=1
  tail -2 "$d/seal_parity.log"
done
$PY "$ROOT/tools/seal_pin_report.py" --out "$OUT" --cmake-log "$WORK/cmake.log" "${DIRS[@]}" || rc=1
echo "[pin] wrote $OUT (exit $rc)"
exit $rc
