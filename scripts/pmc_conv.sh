# PMC passes on the conv microbenchmark (separate --pmc runs, kernel-trace only).  usage: pmc_conv.sh <case> <fwd|dgrad|wgrad>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CASE=${1:-dc5}; export WHICH=${2:-fwd}; export REPS=2
mkdir -p gpurun_out/pmc
for i in 1 2 3; do
  case $i in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS";;
    2) C="SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES";;
    3) C="TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE";;
  esac
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc -o ${CASE}_${WHICH}_p$i -- python3 scripts/bench_conv.py $CASE > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/${CASE}_${WHICH}_p*_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if ('conv_igemm' in k or 'wgrad_kernel' in k):
            agg[(k[:60], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in sorted(agg.items()):
        print(k, c, "%.4g" % (sum(v) / len(v)), len(v))
PY
