#!/bin/bash
# Same-box A/B of the host pipeline's settings on verify_batch from typed objects (tools/e2e_probe.py):
# one line per scheme and setting into $OUT.  Usage: tools/ab_e2e.sh OUT "VAR=VAL VAR=VAL" "VAR=VAL" ...
# ("-" = the defaults).  Stops at the first failing run (no GPU step after a failed one).
set -e -o pipefail
OUT=$1
shift
mkdir -p "$(dirname "$OUT")"
for setting in "$@"; do
  if [ "$setting" = "-" ]; then setting=""; fi
  echo "# setting: ${setting:-defaults}" >> "$OUT"
  # shellcheck disable=SC2086
  env $setting timeout -k 10 420 python tools/e2e_probe.py >> "$OUT" 2>&1
done
