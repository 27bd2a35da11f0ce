#!/bin/bash
# the N-rank shares of a C3 frame under several environment settings: tools/share_probe.sh "RT_OVERLAP=3" "RT_SHADE_GRID=4" ...
for e in "" "$@"; do
  echo "== $e"; env $e timeout -k 10 200 python tools/scaling_probe.py ${WL:-c3} 2>/dev/null | grep "N=" || exit 1
done
