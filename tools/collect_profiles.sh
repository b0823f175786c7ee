#!/bin/bash
# Round evidence on the GPU box (run from the repo root through gpurun): rocprofv3 kernel statistics, the two PMC passes for
# HBM traffic, the bench lines.  Outputs under gpurun_out/profiles_rNN/ -- copy what is to be judged into profiles/.
#   tools/collect_profiles.sh r03
set -e
R=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${R}_bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 --profile-steps 1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 --profile-steps 1 > /dev/null 2> $OUT/pmc_write.err
# forward-only evidence (the pass north_star quotes): kernel statistics of the eval forward, fused and layer by layer, and the MFMA-busy counters
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_eval -- python $ROOT/bench.py --workload yolov8_eval --steps 20 --warmup 5 > $OUT/${R}_bench_yolov8_eval_under_rocprof.json 2> $OUT/stats_eval.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_eval -- python $ROOT/bench.py --workload yolov8_eval --steps 3 --warmup 2 > /dev/null 2> $OUT/pmc_mfma_eval.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_train -- python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 --profile-steps 1 > /dev/null 2> $OUT/pmc_mfma_train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cn -- python $ROOT/bench.py --workload centernet --steps 5 --warmup 2 > $OUT/${R}_bench_centernet_under_rocprof.json 2> $OUT/stats_cn.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dl -- python $ROOT/bench.py --workload deeplab_train --steps 5 --warmup 2 > $OUT/${R}_bench_deeplab_train_under_rocprof.json 2> $OUT/stats_dl.err
for wl in centernet_train ssd_train yolov7_train; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -- python $ROOT/bench.py --workload $wl --steps 5 --warmup 2 > $OUT/${R}_bench_${wl}_under_rocprof.json 2> $OUT/stats_$wl.err
done
for wl in ssd yolov7 deeplab; do   # inference tails (decode / NMS / resize kernels beside the convolutions)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_inf_$wl -- python $ROOT/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/stats_inf_$wl.err
done
cd $ROOT
for wl in ssd yolov7 deeplab; do cp $(ls $OUT/stats_inf_$wl/*/*_kernel_stats.csv | head -1) $OUT/${R}_bench_${wl}_kernel_stats.csv; done
for wl in centernet_train ssd_train yolov7_train; do cp $(ls $OUT/stats_$wl/*/*_kernel_stats.csv | head -1) $OUT/${R}_bench_${wl}_kernel_stats.csv; done
cp $(ls $OUT/stats/*/*_kernel_stats.csv | head -1) $OUT/${R}_bench_kernel_stats.csv
cp $(ls $OUT/stats_eval/*/*_kernel_stats.csv | head -1) $OUT/${R}_bench_yolov8_eval_kernel_stats.csv
python tools/pmc_mfma.py $(ls $OUT/pmc_mfma_eval/*/*_counter_collection.csv | head -1) $OUT/${R}_mfma_busy_eval.json
python tools/pmc_mfma.py $(ls $OUT/pmc_mfma_train/*/*_counter_collection.csv | head -1) $OUT/${R}_mfma_busy.json
cp $OUT/${R}_mfma_busy.json profiles/${R}_mfma_busy.json   # bench.py quotes it when its lib_sha256 is the running library's
cp $(ls $OUT/stats_cn/*/*_kernel_stats.csv | head -1) $OUT/${R}_bench_centernet_kernel_stats.csv
cp $(ls $OUT/stats_dl/*/*_kernel_stats.csv | head -1) $OUT/${R}_bench_deeplab_train_kernel_stats.csv
python tools/pmc_traffic.py $(ls $OUT/pmc_fetch/*/*_counter_collection.csv | head -1) $(ls $OUT/pmc_write/*/*_counter_collection.csv | head -1) $OUT/${R}_conv_traffic.json
cp $OUT/${R}_conv_traffic.json profiles/${R}_conv_traffic.json   # bench.py quotes it when its lib_sha256 is the running library's
python bench.py --steps 30 --warmup 5 > $OUT/${R}_bench.json 2> $OUT/bench.err
python bench.py --workload yolov8_eval --steps 50 --warmup 5 > $OUT/${R}_bench_yolov8_eval.json 2>> $OUT/bench.err
CVX_LIB=build/libcvx_tuning.so python bench.py --workload yolov8_eval --fusion 1 --steps 50 --warmup 5 > $OUT/${R}_bench_yolov8_eval_fused.json 2>> $OUT/bench.err
python tools/op_profile.py 5 yolov8_eval > $OUT/${R}_op_profile_yolov8_eval.txt 2>> $OUT/bench.err
python bench.py --workload centernet --steps 10 --warmup 2 > $OUT/${R}_bench_centernet.json 2>> $OUT/bench.err
python bench.py --model s --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${R}_bench_yolov8s.json 2>> $OUT/bench.err
python bench.py --workload deeplab --steps 10 --warmup 2 > $OUT/${R}_bench_deeplab.json 2>> $OUT/bench.err
python bench.py --workload deeplab_train --steps 20 --warmup 3 > $OUT/${R}_bench_deeplab_train.json 2>> $OUT/bench.err
python bench.py --workload centernet_train --steps 20 --warmup 3 > $OUT/${R}_bench_centernet_train.json 2>> $OUT/bench.err
python bench.py --workload yolov7_train --steps 20 --warmup 3 > $OUT/${R}_bench_yolov7_train.json 2>> $OUT/bench.err
python bench.py --workload ssd_train --steps 20 --warmup 3 > $OUT/${R}_bench_ssd_train.json 2>> $OUT/bench.err
python bench.py --workload yolov7 --steps 10 --warmup 2 > $OUT/${R}_bench_yolov7.json 2>> $OUT/bench.err
python bench.py --workload ssd --steps 10 --warmup 2 > $OUT/${R}_bench_ssd.json 2>> $OUT/bench.err
python bench.py --workload yolov7 --nms-load 256 --steps 10 --warmup 2 > $OUT/${R}_bench_yolov7_nms256.json 2>> $OUT/bench.err
python bench.py --workload ssd --nms-load 256 --steps 10 --warmup 2 > $OUT/${R}_bench_ssd_nms256.json 2>> $OUT/bench.err
# data-parallel step with a 1-rank RCCL group (the exchange path of bench.py --gpus N, both flavours) and the launch-stream probes
CVX_FORCE_DIST=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>> $OUT/bench.err | grep '"metric"' > $OUT/${R}_bench_dp_1rank_torch.json
CVX_FORCE_DIST=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --exchange c 2>> $OUT/bench.err | grep '"metric"' > $OUT/${R}_bench_dp_1rank_c.json
for m in default pool raw; do python tools/stream_probe.py $m 2>/dev/null | grep "launch stream" >> $OUT/${R}_stream_probe.txt; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/slb tools/micro/stream_launch_bench.hip 2>> $OUT/bench.err && /tmp/slb >> $OUT/${R}_stream_probe.txt
python tools/op_profile.py 5 > $OUT/${R}_op_profile.txt 2>> $OUT/bench.err
python tools/op_profile.py 3 deeplab > $OUT/${R}_op_profile_deeplab_train.txt 2>> $OUT/bench.err
python tools/op_profile.py 3 ssd > $OUT/${R}_op_profile_ssd_train.txt 2>> $OUT/bench.err
python tools/op_profile.py 3 yolo7 > $OUT/${R}_op_profile_yolov7_train.txt 2>> $OUT/bench.err
python tools/op_profile.py 3 centernet > $OUT/${R}_op_profile_centernet_train.txt 2>> $OUT/bench.err
rm -rf $OUT/stats_eval $OUT/pmc_mfma_eval $OUT/pmc_mfma_train $OUT/stats_inf_ssd $OUT/stats_inf_yolov7 $OUT/stats_inf_deeplab $OUT/stats $OUT/stats_cn $OUT/stats_dl $OUT/stats_centernet_train $OUT/stats_ssd_train $OUT/stats_yolov7_train $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
