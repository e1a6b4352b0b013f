#!/bin/bash
# Same-box A/B of two library builds (see ab_libs.sh) on BASELINE configs[3] at full size: scripts/c3_time.py, two alternations.
set -uo pipefail
cd "$(dirname "$0")/.."
L=matching-pursuit_amd/lib
cp $L/libmpcore.so /tmp/new.so; cp $L/libmpcore_base.so /tmp/base.so
for i in 1 2; do
  cp /tmp/base.so $L/libmpcore.so; echo -n "base  "; python3 scripts/c3_time.py 2>&1 | grep "default"
  cp /tmp/new.so $L/libmpcore.so; echo -n "new   "; python3 scripts/c3_time.py 2>&1 | grep "default"
done
cp /tmp/new.so $L/libmpcore.so
