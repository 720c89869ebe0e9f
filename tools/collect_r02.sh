#!/bin/bash
# Turn gpurun_out/r02final/ (tools/gpu_r02_final.sh) into the committed summaries under profiles/.
cd /root/repo
O=gpurun_out/r02final
for n in default train resnet_h_train16; do
  cp $O/bench_${n}_kernel_stats.csv profiles/r02_bench_${n}_kernel_stats.csv
  cp $O/bench_${n}_conv_dispatches.csv profiles/r02_bench_${n}_conv_dispatches.csv
done
cp $O/pmc_traffic.json profiles/r02_pmc_traffic.json
cp $O/pmc_traffic_resnet_train.json profiles/r02_pmc_traffic_resnet_train.json
cp $O/pmc_mfma.json profiles/r02_pmc_mfma.json
for n in default train resnet_h_train16 resnet_h_train128 resnet_f_b32 resnet_h_b128 vgg_f_b32 vgg_f_train32; do tail -1 $O/bench_$n.log > profiles/r02_bench_${n}_line.json; done
grep -h '^{"metric' $O/rocprof_default.log > profiles/r02_bench_default_under_rocprof_line.json
tail -3 $O/pytest_gpu.log > profiles/r02_pytest_gpu_tail.txt
for f in $O/layer_profile_*.txt; do cp $f profiles/r02_$(basename $f); done
cp $O/microbench_wino_b128.txt profiles/r02_microbench_wino_b128.txt
cp $O/microbench_wgrad_wino_b128.txt profiles/r02_microbench_wgrad_wino_b128.txt
for b in 16 128; do cp $O/microbench_conv1x1_b$b.txt profiles/r02_microbench_conv1x1_b$b.txt; done
tail -1 $O/rehearsal_2ranks_infer.log > profiles/r02_rehearsal_2ranks_gloo_infer_line.json
tail -1 $O/rehearsal_2ranks_train.log > profiles/r02_rehearsal_2ranks_gloo_train_line.json
tail -1 $O/rehearsal_single_process_train.log > profiles/r02_rehearsal_single_process_2replicas_train_line.json
tail -3 $O/smoke.log > profiles/r02_smoke_tail.txt
