#!/bin/bash
# copy the judged summaries of one scripts/gpu_r05_final.sh run (gpurun_out/TAG) into profiles/r05_*
R=gpurun_out/${1:?tag}
set -e
cp $R/attn_pmc_summary.txt profiles/r05_attn_long_pmc_summary.txt
for p in bench:r05_bench_line bench_long:r05_bench_long_line bench_train:r05_bench_train_line; do
  grep '^{' $R/${p%%:*}.json | tail -1 > profiles/${p##*:}.json
done
cp $R/hbm_kernels.txt profiles/r05_hbm_kernels.txt
cp $R/hbm_traffic.json profiles/r05_hbm_traffic.json
cp $R/stats_long/long_kernel_stats.csv profiles/r05_long_1000step_kernel_stats.csv
cp $R/mega_dataset_shapes_ab.txt profiles/r05_mega_dataset_shapes_ab.txt
cp $R/stats/bench_kernel_stats.csv profiles/r05_mega_kernel_stats.csv
cp $R/parity_errors.txt profiles/r05_parity_errors.txt
cp $R/stats_train/train_ecg_kernel_stats.csv profiles/r05_train_ecg_kernel_stats.csv
cp $R/stats_train/train_nasdaq_kernel_stats.csv profiles/r05_train_nasdaq_kernel_stats.csv
cp $R/serial_ecg/s_kernel_stats.csv profiles/r05_train_ecg_serial_kernel_stats.csv
cp $R/serial_nasdaq/s_kernel_stats.csv profiles/r05_train_nasdaq_serial_kernel_stats.csv
git status --short profiles | head -20
