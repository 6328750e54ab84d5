#!/bin/bash
# timing-only ablations of the 256x256 conv tile (results are WRONG by construction for 1/2/4)
export TMPDIR=/tmp
for v in 0 1 2 3; do
  if [ $v = 0 ]; then unset DL_CONV_ABLATE; else export DL_CONV_ABLATE=$v; fi
  echo "=== ablate=$v"
  timeout 300 python tools/microbench.py 2>/dev/null | grep "bf16" | grep -E "res3x3" | cut -c1-200
done
