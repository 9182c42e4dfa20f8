#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/prof_<tag>_* (run on the GPU box through
# gpurun, from the repo root); tools/summarize_prof.py / summarize_cfg.py then condense it into
# profiles/.  Counters are collected in their own passes, with --kernel-trace only.
#   usage: tools/profile_round.sh r1 [a|b|c|q|ab]   (a: configs[1], batch-16;  c: configs[2];  b: configs[4];  q: 128-query panels)
set -e -o pipefail
tag=${1:-r1}
part=${2:-ab}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
# identity of the kernel sources these passes measure (content hash of svs_amd/csrc + the header: svs_amd/buildinfo.py):
# written on the GPU box at profile time (it has no .git), read by the summarisers, checked by bench.py before it
# quotes any traffic figure from profiles/
python3 $R/svs_amd/buildinfo.py > $O/prof_${tag}_csrc_sha.txt
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; rm -rf "$O/prof_${tag}_$name"; rocprofv3 "$@"; echo "[profile_round] $name done" >&2; }
if [[ $part == *a* ]]; then
# configs[1]: the bench line itself, then HBM bytes of its kernels
run trace --kernel-trace --stats --output-format csv -d $O/prof_${tag}_trace -- python3 $R/bench.py > $O/prof_${tag}_trace.json
run fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_${tag}_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 60 --batch "" --concurrent 0 --configs "" --kb 0 > /dev/null
run write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_${tag}_write -- python3 $R/bench.py --no-cpu-baseline --steps 60 --batch "" --concurrent 0 --configs "" --kb 0 > /dev/null
# batched f32, 16 queries per call
run b16_trace --kernel-trace --stats --output-format csv -d $O/prof_${tag}_b16_trace -- python3 $R/tools/prof_batch.py 1000000 1536 f32 16 20 > /dev/null
run b16_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_${tag}_b16_fetch -- python3 $R/tools/prof_batch.py 1000000 1536 f32 16 6 > /dev/null
fi
if [[ $part == *c* || $part == *a* ]]; then
# configs[2]: f16, 1024 queries per call
run cfg2_trace --kernel-trace --stats --output-format csv -d $O/prof_${tag}_cfg2_trace -- python3 $R/tools/prof_batch.py 1000000 1536 f16 1024 24 > /dev/null
run cfg2_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_${tag}_cfg2_fetch -- python3 $R/tools/prof_batch.py 1000000 1536 f16 1024 4 > /dev/null
run cfg2_pmc --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $O/prof_${tag}_cfg2_pmc -- python3 $R/tools/prof_batch.py 1000000 1536 f16 1024 3 > /dev/null
fi
if [[ $part == *b* ]]; then
# configs[4]: fp8 10M x 3072, 256 queries per call
run cfg4_trace --kernel-trace --stats --output-format csv -d $O/prof_${tag}_cfg4_trace -- python3 $R/tools/prof_batch.py 10000000 3072 fp8 256 12 > /dev/null
run cfg4_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_${tag}_cfg4_fetch -- python3 $R/tools/prof_batch.py 10000000 3072 fp8 256 3 > /dev/null
run cfg4_pmc --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $O/prof_${tag}_cfg4_pmc -- python3 $R/tools/prof_batch.py 10000000 3072 fp8 256 2 > /dev/null
fi

if [[ $part == *q* ]]; then
# panels of 128 queries over f16 / fp8 (gemm_phased_kernel<.., QT = 128>)
for dt in f16 fp8; do
run q128_${dt}_trace --kernel-trace --stats --output-format csv -d $O/prof_${tag}_q128_${dt}_trace -- python3 $R/tools/prof_batch.py 1000000 1536 $dt 128 24 > /dev/null
run q128_${dt}_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_${tag}_q128_${dt}_fetch -- python3 $R/tools/prof_batch.py 1000000 1536 $dt 128 4 > /dev/null
done
echo "[profile_round] q passes done"
fi
echo "[profile_round] all passes done"
