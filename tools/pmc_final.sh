set -e
mkdir -p gpurun_out/pmc && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 --batch 32 --node-steps 0 --train-steps 0 --dense-steps 0 --h2d-steps 0 --config-steps 0 --pipelined 0"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc/sq -- $CMD > gpurun_out/pmc/sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/fetch -- $CMD > gpurun_out/pmc/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc/write -- $CMD > gpurun_out/pmc/write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc/tcc -- $CMD > gpurun_out/pmc/tcc.log 2>&1
find gpurun_out/pmc -name "*counter_collection.csv" | head
