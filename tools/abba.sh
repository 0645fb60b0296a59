#!/bin/bash
# Same-box A/B in B A A B order: tools/abba.sh OUT.log "variant args" ["baseline args"]   (each run: tools/ab_runtime.py --steps 100)
# The first process on a fresh box runs ~0.05 ms/step faster than the ones after it ("next 2"), so a variant is compared on the
# mean of its first and last place against the baseline's two middle places.
out=$1; var=$2; base=${3:-}
for a in "$var" "$base" "$base" "$var"; do
  timeout -k 10 240 python tools/ab_runtime.py $a --steps 100 2>/dev/null | grep "ms/step" >> $out || exit 1
done
