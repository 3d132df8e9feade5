# PMC stall breakdown of single conv launches (scripts/bench_conv.py <layer>); $1 = layer, results in gpurun_out/pmc_conv_<layer>/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=${1:-dc5}
O=gpurun_out/pmc_conv_$L
mkdir -p $O
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  REPS=3 timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -o x -- python3 scripts/bench_conv.py $L > $O/p$i.log 2>&1 || { echo "pass $i ($C) failed"; tail -3 $O/p$i.log; }
  echo "pass $i done"
done
python3 - "$O" <<'PY'
import csv, sys, glob, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/x_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "march" in k or "wgrad_kernel" in k or "igemm" in k or "stream" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k[:110])
    for c in sorted(d):
        v = d[c]
        print("   %-28s %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
