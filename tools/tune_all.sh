set -e
export TMPDIR=/tmp/eqh_tune; mkdir -p $TMPDIR gpurun_out/tuned
for spec in "egnn_equihnns 256 qm9" "egnn_equihnns 1024 pcqm" "mhnns 256 qm9" "mhnn 256 qm9" "mhnnm 256 qm9" "egnn_equihnn 256 qm9" "egnn_equihnnm 256 qm9" "equiformer_equihnns 128 qm9" "faformer_equihnns 512 pcqm"; do
  set -- $spec
  timeout -k 10 400 python bench.py --method $1 --batch $2 --flavour $3 --steps 5 --warmup 3 --no-roofline --no-cpu-baseline --no-pipeline --c4-steps 0 > gpurun_out/tuned/$1_$2.json 2> gpurun_out/tuned/$1_$2.log
  echo "$1 $2 done: $(wc -l < $TMPDIR/eqh_tunableop_0.csv) lines"
done
cp $TMPDIR/eqh_tunableop_0.csv gpurun_out/tuned/all_0.csv
