#!/bin/bash
# Round profiles, part $1 (1, 2 or 3), run on the GPU box from the repo root; results under gpurun_out/final/.
set -u
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
if [ "$1" = 1 ]; then
  timeout -k 10 500 python bench.py > $O/r02_bench.json 2> $O/bench.err
  tail -c 400 $O/r02_bench.json
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_main -- python3 bench.py --no-cpu-baseline --no-pipeline --c4-steps 0 > $O/r02_bench_under_rocprof.json 2> $O/rocprof.err
  cp "$(find /tmp/prof_main -name '*kernel_stats.csv' | head -1)" $O/r02_graph_kernel_stats.csv
elif [ "$1" = 2 ]; then
  for spec in "egnn_equihnns 1024 pcqm" "mhnns 256 qm9" "mhnn 256 qm9" "mhnnm 256 qm9" "egnn_equihnn 256 qm9" "egnn_equihnnm 256 qm9"; do
    set -- $spec
    timeout -k 10 300 python bench.py --method $1 --batch $2 --flavour $3 --steps 20 --warmup 5 --no-pipeline --c4-steps 0 --no-cpu-baseline > $O/r02_bench_$1_$2.json 2>> $O/bench.err
    echo "$1 $2: $(cut -c1-160 $O/r02_bench_$1_$2.json | tail -1)"
  done
else
  timeout -k 10 500 python bench.py --method equiformer_equihnns --batch 128 --steps 20 --warmup 5 --no-pipeline --c4-steps 0 > $O/r02_bench_equiformer_equihnns_128.json 2>> $O/bench.err
  cut -c1-200 $O/r02_bench_equiformer_equihnns_128.json | tail -1
  timeout -k 10 600 python bench.py --method faformer_equihnns --batch 512 --flavour pcqm --steps 10 --warmup 3 --no-pipeline --c4-steps 0 --cpu-batch 16 --cpu-seconds 8 > $O/r02_bench_faformer_equihnns_512.json 2>> $O/bench.err
  cut -c1-200 $O/r02_bench_faformer_equihnns_512.json | tail -1
fi
