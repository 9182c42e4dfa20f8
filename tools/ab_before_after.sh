#!/bin/bash
# Same-box A/B of several builds of the library (SVS_AMD_LIB; tools/libsvs_amd_<name>.so, "tree" = the tree's own build).
# usage: tools/ab_before_after.sh "before park1536 aggr1024 tree" [rounds=3]
cd "$(dirname "$0")/.."
names=${1:-"before tree"}; rounds=${2:-3}
for rep in $(seq 1 $rounds); do
  for which in $names; do
    if [ $which = tree ]; then unset SVS_AMD_LIB; else export SVS_AMD_LIB=$PWD/tools/libsvs_amd_$which.so; fi
    a=$(timeout -k 10 200 python tools/call_breakdown.py 1000000 1536 f16 1024 2>&1 | grep -a "round 1 upload=chunked pull prefix=n/64" | sed -e 's/.*call \([0-9.]*\) ms.*dominant kernel \([0-9.]*\) .*/call \1 kernel \2/')
    b=$(timeout -k 10 200 python tools/clustered_corpus_time.py 1000000 1536 f16 1024 256 2>&1 | grep -a "spread rows" | sed -e 's/.*: \([0-9.]*\) ms per call.*score stage \([0-9.]*\) ms.*/call \1 score \2/')
    c=$(timeout -k 10 200 python tools/clustered_corpus_time.py 1000000 1536 f16 1024 32 2>&1 | grep -a "spread rows" | sed -e 's/.*: \([0-9.]*\) ms per call.*score stage \([0-9.]*\) ms.*/call \1 score \2/')
    d=$(timeout -k 10 200 python tools/clustered_corpus_time.py 1000000 1536 f16 128 32 2>&1 | grep -a "spread rows" | sed -e 's/.*: \([0-9.]*\) ms per call.*score stage \([0-9.]*\) ms.*/call \1 score \2/')
    echo "round $rep  $which: configs[2] shuffled $a | 256 topics $b | 32 topics $c | 128 queries, 32 topics $d"
  done
done
