# Round-4 measurement bundle (run on the GPU box via gpurun; results land in gpurun_out/r04prof/, the summaries that are
# committed are copied to profiles/ by hand).  Every profiled command is bench.py itself (python3 directly after `--`).
#   1. plain bench (the JSON line + the per-launch-group table), incl. the window512 leg
#   2. rocprofv3 --kernel-trace --stats of the same command  -> kernel_stats.csv, per-(kernel, grid) table
#   3. PMC passes (kernel-trace only, separate runs): FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04prof
mkdir -p $O
ARGS="--steps 10 --warmup 3"
timeout -k 10 500 python3 bench.py $ARGS --dump-kernels $O/bench_launch_groups.tsv > $O/bench.json 2> $O/bench.err
echo "bench done"; tail -c 300 $O/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o seunet -- python3 bench.py $ARGS --no-cpu-baseline --no-secondary --no-kernel-timing > $O/trace.log 2>&1
python3 scripts/trace_by_grid.py $O/trace/seunet_kernel_trace.csv conv_march_kernel wgrad_march_kernel conv_igemm_kernel wgrad_kernel conv_stream_kernel wgrad_stream_kernel > $O/conv_by_grid.txt
cp $O/trace/seunet_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || true
echo "trace done"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  T=$(echo $C | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$T -o seunet -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing > $O/pmc_$T.log 2>&1
  echo "pmc $T done"
done
python3 scripts/pmc_summary.py $O > $O/pmc_summary.txt
tail -25 $O/pmc_summary.txt
