# Round-4 profile set, part `$1` (a | b), on the GPU box; everything lands under gpurun_out/r04final/
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04final
mkdir -p $OUT && cd $GRAFT_REPO_ROOT
if [ "$1" = "a" ]; then
  timeout -k 10 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  bash tools/prof_bench.sh > $OUT/prof_bench.log 2>&1
  cp gpurun_out/prof_bench/kernel_stats.csv $OUT/bench_b256_kernel_stats.csv
  cp gpurun_out/prof_bench/bench_under_rocprof.json $OUT/bench_b256_under_rocprof.json
  bash tools/leaf_prof.sh r4final > $OUT/leaf_prof.log 2>&1
  cp gpurun_out/leaf_prof/r4final_summary.txt $OUT/leaf_stage_kernel_stats.txt
  for v in base cs_first own extra extra_cs_first extra_own; do timeout -k 10 120 python3 tools/h2d_probe.py $v 2>/dev/null | grep variant; done > $OUT/h2d_probe.txt
else
  bash tools/pmc_final.sh > $OUT/pmc_final.log 2>&1
  python3 tools/pmc_summarize.py 32 1080 1920 33.253 r04_pmc_counters.json > $OUT/pmc_summarize.log 2>&1
  cp profiles/r04_pmc_counters.json $OUT/
  bash tools/pmc_cnn.sh r4f43 > $OUT/pmc_cnn.log 2>&1
  cp gpurun_out/pmc_cnn/r4f43_summary.txt $OUT/cnn_f43_trace_pmc.txt
  bash tools/pmc_lds_ubench.sh > $OUT/ubench_lds.log 2>&1
  cp gpurun_out/lds_ubench/summary_b.txt $OUT/ubench_lds_conflicts.txt
fi
ls -la $OUT | tail -20
