set -uo pipefail
L=matching-pursuit_amd/lib
cp $L/libmpcore.so /tmp/new.so; cp $L/libmpcore_base.so /tmp/base.so
for i in 1 2; do
  cp /tmp/base.so $L/libmpcore.so; echo base; python3 scripts/c3_time.py 2>&1 | grep "default"
  cp /tmp/new.so $L/libmpcore.so; echo new; python3 scripts/c3_time.py 2>&1 | grep "default"
done
cp /tmp/new.so $L/libmpcore.so
