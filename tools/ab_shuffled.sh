#!/bin/bash
# Same-box A/B of library builds on the BASELINE shapes (shuffled Gaussian data): configs[2], configs[4], 128-query panels.
# usage: tools/ab_shuffled.sh "head tree" [rounds=3]
cd "$(dirname "$0")/.."
names=${1:-"head tree"}; rounds=${2:-3}
for rep in $(seq 1 $rounds); do
  for which in $names; do
    if [ $which = tree ]; then unset SVS_AMD_LIB; else export SVS_AMD_LIB=$PWD/tools/libsvs_amd_$which.so; fi
    a=$(timeout -k 10 200 python tools/call_breakdown.py 1000000 1536 f16 1024 2>&1 | grep -a "round 1 upload=chunked pull prefix=n/64" | sed -e 's/.*call \([0-9.]*\) ms.*dominant kernel \([0-9.]*\) .*/call \1 kernel \2/')
    b=$(timeout -k 10 200 python tools/call_breakdown.py 10000000 3072 fp8 256 2>&1 | grep -a "round 1 upload=staged" | sed -e 's/.*call \([0-9.]*\) ms.*dominant kernel \([0-9.]*\) .*/call \1 kernel \2/')
    c=$(timeout -k 10 200 python tools/call_breakdown.py 1000000 1536 f16 128 2>&1 | grep -a "round 1 upload=chunked pull prefix=n/64" | sed -e 's/.*call \([0-9.]*\) ms.*dominant kernel \([0-9.]*\) .*/call \1 kernel \2/')
    d=$(timeout -k 10 200 python tools/call_breakdown.py 1000000 1536 fp8 128 2>&1 | grep -a "round 1 upload=staged" | sed -e 's/.*call \([0-9.]*\) ms.*dominant kernel \([0-9.]*\) .*/call \1 kernel \2/')
    echo "round $rep  $which: configs[2] $a | configs[4] $b | f16 x 128 $c | fp8 x 128 $d"
  done
done
