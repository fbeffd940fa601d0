#!/bin/bash
# copy the judged summaries of one scripts/gpu_r06_final.sh run (gpurun_out/TAG) into profiles/<ROUND>_*   usage: sync_evidence.sh TAG [ROUND=r06]
R=gpurun_out/${1:?tag}; P=${2:-r06}
set -e
cp $R/attn_lds_pmc_summary.txt profiles/${P}_attn_long_lds_pmc_summary.txt
for p in bench:bench_line bench_long:bench_long_line bench_train:bench_train_line; do
  grep '^{' $R/${p%%:*}.json | tail -1 > profiles/${P}_${p##*:}.json
done
cp $R/hbm_kernels.txt profiles/${P}_hbm_kernels.txt
cp $R/hbm_traffic.json profiles/${P}_hbm_traffic.json
cp $R/stats_long/long_kernel_stats.csv profiles/${P}_long_1000step_kernel_stats.csv
cp $R/mega_dataset_shapes_ab.txt profiles/${P}_mega_dataset_shapes_ab.txt
cp $R/stats/bench_kernel_stats.csv profiles/${P}_mega_kernel_stats.csv
cp $R/parity_errors.txt profiles/${P}_parity_errors.txt
cp $R/stats_train/train_ecg_kernel_stats.csv profiles/${P}_train_ecg_kernel_stats.csv
cp $R/stats_train/train_nasdaq_kernel_stats.csv profiles/${P}_train_nasdaq_kernel_stats.csv
cp $R/serial_ecg/s_kernel_stats.csv profiles/${P}_train_ecg_serial_kernel_stats.csv
cp $R/serial_nasdaq/s_kernel_stats.csv profiles/${P}_train_nasdaq_serial_kernel_stats.csv
grep -v "^$" $R/ab.txt | cut -c1-220 > profiles/${P}_train_persist_final_ab.txt
[ -f $R/phase_clocks.txt ] && cp $R/phase_clocks.txt profiles/${P}_train_fwd_layers_phase_clocks_final.txt || true      # (only when the -DFD_TRP_PROF variant library was built for the run)
cp $R/gpu_suite.txt profiles/${P}_gpu_suite.txt
git status --short profiles | head -30
