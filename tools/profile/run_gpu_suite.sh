#!/bin/bash
# The -m gpu suite, one pytest process per test file (a device fault in one file cannot hide the others), summary lines into $1.
# usage (GPU box): bash tools/profile/run_gpu_suite.sh gpurun_out/suite.txt [extra pytest args]
out=${1:-gpurun_out/suite.txt}; shift
mkdir -p "$(dirname "$out")"
: > "$out"
echo "AOD_CONV_PREC=${AOD_CONV_PREC:-<unset: library default bf16x3>}" >> "$out"
t0=$(date +%s)
for f in tests/test_gpu_*.py; do
  log=$(mktemp)
  timeout 1500 python -m pytest "$f" -m gpu -q --timeout 900 -p no:cacheprovider "$@" > "$log" 2>&1
  rc=$?
  echo "== $f rc=$rc: $(grep -E '(passed|failed|error|no tests ran)' "$log" | tail -1)" >> "$out"
  grep -E '^(FAILED|ERROR)' "$log" >> "$out"
  if [ $rc -ne 0 ]; then grep -E 'Error|error|assert' "$log" | head -12 >> "$out"; fi
  rm -f "$log"
done
echo "total $(( $(date +%s) - t0 )) s" >> "$out"
cat "$out"
