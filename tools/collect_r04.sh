#!/bin/bash
# Turn gpurun_out/r04final/ (tools/gpu_r04_final.sh) into the committed summaries under profiles/.
cd /root/repo
O=gpurun_out/r04final
for n in default train resnet_h_train16; do
  cp $O/bench_${n}_kernel_stats.csv profiles/r04_bench_${n}_kernel_stats.csv
  cp $O/bench_${n}_conv_dispatches.csv profiles/r04_bench_${n}_conv_dispatches.csv
done
cp $O/pmc_traffic.json profiles/r04_pmc_traffic.json
cp $O/pmc_traffic_resnet_train.json profiles/r04_pmc_traffic_resnet_train.json
cp $O/pmc_mfma.json profiles/r04_pmc_mfma.json
for n in default train resnet_h_train16 resnet_h_train16_three_launch_bn resnet_h_train128 resnet_h_train128_three_launch_bn resnet_f_b32 resnet_h_b128 vgg_f_b32; do tail -1 $O/bench_$n.log > profiles/r04_bench_${n}_line.json; done
grep -h '^{"metric' $O/rocprof_default.log > profiles/r04_bench_default_under_rocprof_line.json
grep -E "passed|failed" $O/pytest_gpu.log | tail -1 > profiles/r04_pytest_gpu_tail.txt
for f in $O/layer_profile_*.txt; do cp $f profiles/r04_$(basename $f); done
cp $O/microbench_wino4_b128.txt profiles/r04_microbench_wino4_b128.txt
cp $O/ab_wino4_pinning.txt profiles/r04_ab_wino4_pinning.txt
cp $O/microbench_conv1x1_b16.txt profiles/r04_microbench_conv1x1_b16.txt
cp $O/l2_hit_wino4_pinning.txt profiles/r04_l2_hit_wino4_pinning.txt
tail -1 $O/rehearsal_2ranks_selflaunch.log > profiles/r04_rehearsal_2ranks_gloo_selflaunch_line.json
tail -1 $O/rehearsal_single_process_train.log > profiles/r04_rehearsal_single_process_4replicas_train_line.json
tail -3 $O/smoke.log > profiles/r04_smoke_tail.txt
