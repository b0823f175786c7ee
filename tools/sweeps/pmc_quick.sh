# the two HBM-traffic PMC passes of one train step alone (what tools/collect_profiles.sh does among much else) -> gpurun_out/pmc_quick.json
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_quick
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 --profile-steps 1 > /dev/null 2> $OUT/f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 --profile-steps 1 > /dev/null 2> $OUT/w.err
cd $ROOT
python tools/pmc_traffic.py $(ls $OUT/f/*/*_counter_collection.csv | head -1) $(ls $OUT/w/*/*_counter_collection.csv | head -1) gpurun_out/pmc_quick.json
rm -rf $OUT
