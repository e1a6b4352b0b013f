#!/bin/bash
# Same-box A/B of two builds of the library: bench.py's headline (no variants, no CPU leg) alternately on
# matching-pursuit_amd/lib/libmpcore.so (new) and libmpcore_base.so (a build of an older commit put beside it by hand), three rounds.
#   usage (on the GPU box): bash scripts/ab_libs.sh [bench arguments]
set -uo pipefail
cd "$(dirname "$0")/.."
L=matching-pursuit_amd/lib
cp $L/libmpcore.so /tmp/new.so; cp $L/libmpcore_base.so /tmp/base.so
one() { python3 bench.py --steps 30 --no-variants --no-cpu --no-configs3 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-5s %.0f seg-it/s  %.4f ms  launch %.4f ms' % ('$TAG', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
for i in 1 2 3; do
  cp /tmp/base.so $L/libmpcore.so; TAG=base; one "$@"
  cp /tmp/new.so $L/libmpcore.so; TAG=new; one "$@"
done
cp /tmp/new.so $L/libmpcore.so
