#!/bin/bash
# Round profiles, part $1 (tests, 1, 2, 3 or pmc), run on the GPU box from the repo root; results under
# gpurun_out/final/, named for round $2 (default r03).  Copy what should be judged into profiles/.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh 1'
set -u
R=${2:-r06}
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
if [ "$1" = tests ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/${R}_pytest_gpu.log 2>&1
  rc=$?
  tail -4 $O/${R}_pytest_gpu.log
  exit $rc
elif [ "$1" = 1 ]; then
  # the headline line, then the same command under the kernel trace (stats CSV + the order of one replayed step)
  timeout -k 10 500 python bench.py > $O/${R}_bench.json 2> $O/bench.err || exit 1
  tail -c 600 $O/${R}_bench.json
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_main -- python3 bench.py --no-cpu-baseline --no-pipeline --no-other-configs --c4-steps 0 > $O/${R}_bench_with_roofline_probes_under_rocprof.json 2> $O/rocprof.err || exit 1
  cp "$(find /tmp/prof_main -name '*kernel_stats.csv' | head -1)" $O/${R}_kernel_stats_with_roofline_probes.csv
  # kernel order of one replayed step: a trace of the step alone (no roofline probes after it)
  # and per-kernel averages of the training step ALONE (95 replays + 2 set-up steps; the CSV above also holds the
  # roofline probes, which launch the same scatter kernels at the saturating size)
  rm -rf /tmp/prof_order
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_order -- python3 bench.py --no-cpu-baseline --no-roofline --no-pipeline --no-collective-probe --no-other-configs --c4-steps 0 > $O/${R}_bench_under_rocprof.json 2>> $O/rocprof.err || exit 1
  cp "$(find /tmp/prof_order -name '*kernel_stats.csv' | head -1)" $O/${R}_graph_kernel_stats.csv
  python3 tools/step_order.py /tmp/prof_order $O/${R}_step_order.txt --anchor k_embed_fwd
elif [ "$1" = 1b ]; then
  rm -rf /tmp/prof_order
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_order -- python3 bench.py --no-cpu-baseline --no-roofline --no-pipeline --no-collective-probe --no-other-configs --c4-steps 0 > $O/${R}_bench_under_rocprof.json 2>> $O/rocprof.err || exit 1
  cp "$(find /tmp/prof_order -name '*kernel_stats.csv' | head -1)" $O/${R}_graph_kernel_stats.csv
  python3 tools/step_order.py /tmp/prof_order $O/${R}_step_order.txt --anchor k_embed_fwd
elif [ "$1" = 2 ]; then
  for spec in "egnn_equihnns 1024 pcqm" "mhnns 256 qm9" "mhnn 256 qm9" "mhnnm 256 qm9" "egnn_equihnn 256 qm9" "egnn_equihnnm 256 qm9"; do
    set -- $spec
    timeout -k 10 300 python bench.py --method $1 --batch $2 --flavour $3 --steps 20 --warmup 5 --no-pipeline --no-other-configs --c4-steps 0 --cpu-seconds 8 > $O/${R}_bench_$1_$2.json 2>> $O/bench.err || exit 1
    echo "$1 $2: $(cut -c1-160 $O/${R}_bench_$1_$2.json | tail -1)"
  done
elif [ "$1" = 3 ]; then
  timeout -k 10 500 python bench.py --method equiformer_equihnns --batch 128 --steps 20 --warmup 5 --no-pipeline --no-other-configs --c4-steps 0 --cpu-batch 16 --cpu-seconds 8 > $O/${R}_bench_equiformer_equihnns_128.json 2>> $O/bench.err || exit 1
  cut -c1-200 $O/${R}_bench_equiformer_equihnns_128.json | tail -1
  timeout -k 10 600 python bench.py --method faformer_equihnns --batch 512 --flavour pcqm --steps 10 --warmup 3 --no-pipeline --no-other-configs --c4-steps 0 --cpu-batch 16 --cpu-seconds 8 > $O/${R}_bench_faformer_equihnns_512.json 2>> $O/bench.err || exit 1
  cut -c1-200 $O/${R}_bench_faformer_equihnns_512.json | tail -1
  for spec in "equiformer_equihnns 128 qm9" "faformer_equihnns 512 pcqm"; do
    set -- $spec
    rm -rf /tmp/prof_$1
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -- python3 bench.py --method $1 --batch $2 --flavour $3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-pipeline --c4-steps 0 > $O/${R}_bench_$1_$2_under_rocprof.json 2>> $O/rocprof.err || exit 1
    cp "$(find /tmp/prof_$1 -name '*kernel_stats.csv' | head -1)" $O/${R}_kernel_stats_$1_$2.csv
    # kernel order of one replayed step (launch count, ATen share)
    rm -rf /tmp/order_$1
    timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/order_$1 -- python3 bench.py --method $1 --batch $2 --flavour $3 --steps 6 --warmup 3 --blocks 1 --no-cpu-baseline --no-roofline --no-pipeline --no-collective-probe --c4-steps 0 > /dev/null 2>> $O/rocprof.err || exit 1
    python3 tools/step_order.py /tmp/order_$1 $O/${R}_step_order_$1_$2.txt
    tail -1 $O/${R}_step_order_$1_$2.txt
  done
elif [ "$1" = pmc ]; then
  # HBM traffic of the scatter kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md), over the
  # roofline part of the bench alone; counters only with --kernel-trace
  timeout -k 10 300 python bench.py --only-roofline --timeline-replays 20 > $O/${R}_roofline_line.json 2>> $O/bench.err || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 bench.py --only-roofline --timeline-replays 20 > $O/pmc_$c.log 2>&1 || exit 1
  done
  python3 tools/pmc_to_json.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/${R}_roofline_line.json $O/${R}_pmc_scatter_workload.json --method egnn_equihnns --batch 256 --flavour qm9
  cut -c1-600 $O/${R}_pmc_scatter_workload.json
fi
