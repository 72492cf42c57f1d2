# rocprofv3 --kernel-trace --stats of the headline bench command (only 256-frame launches in the file)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bench
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 bench.py --steps 10 --warmup 3 --cpu-frames 0 --node-steps 0 --train-steps 0 --dense-steps 0 --h2d-steps 0 --config-steps 0 --pipelined 0 > $OUT/bench_under_rocprof.json 2> $OUT/err.log
cp $(ls -t $OUT/t/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
tail -1 $OUT/bench_under_rocprof.json | cut -c1-600
head -14 $OUT/kernel_stats.csv | cut -c1-200
