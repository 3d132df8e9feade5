# quick check of HBM read traffic after a kernel change: one FETCH_SIZE pass over a 4-step bench (see profile_r02.sh)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02q
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_FETCH_SIZE -o seunet -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing > $O/pmc_FETCH_SIZE.log 2>&1
python3 scripts/pmc_summary.py $O > $O/pmc_summary.txt
tail -3 $O/pmc_summary.txt
