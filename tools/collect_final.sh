#!/bin/bash
# Turn gpurun_out/final/ (tools/gpu_final.sh) into the committed summaries under profiles/.
cd /root/repo
O=gpurun_out/final
python tools/prof_summary.py $O/prof_default/dflt_results.db profiles/r01_bench_default > /dev/null
python tools/prof_summary.py $O/prof_train/train_results.db profiles/r01_bench_train > /dev/null
python tools/prof_summary.py $O/prof_rtrain/rtrain_results.db profiles/r01_bench_resnet_h_train16 > /dev/null
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db profiles/r01_pmc_traffic.json > /dev/null
for n in default train resnet_h_train16 resnet_h_train128 resnet_f_b32 resnet_h_b128 vgg_f_b32 vgg_f_train32; do tail -1 $O/bench_$n.log > profiles/r01_bench_${n}_line.json; done
grep -h '^{"metric' $O/rocprof_default.log > profiles/r01_bench_default_under_rocprof_line.json
tail -3 $O/pytest_gpu.log > profiles/r01_pytest_gpu_tail.txt
