# rocprofv3 kernel-trace summary of the default bench workload (run on the GPU box via gpurun)
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
timeout -k 10 600 python bench.py --no-cpu-baseline --dump-kernels gpurun_out/kernels.tsv > gpurun_out/bench_plain.log 2>&1
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o seunet -- python3 bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_prof.log 2>&1
# the conv / wgrad kernels are one instantiation for many layers: per-(kernel, grid) averages identify the layer
python3 scripts/trace_by_grid.py gpurun_out/prof/seunet_kernel_trace.csv > gpurun_out/prof/seunet_conv_by_grid.txt
ls -R gpurun_out/prof | head -20
