# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only) of the dc5 conv launches at the bench shape.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
export REPS=2
for W in fwd dgrad wgrad; do
  export WHICH=$W
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc -o traffic_${W}_f -- python3 scripts/bench_conv.py dc5 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc -o traffic_${W}_w -- python3 scripts/bench_conv.py dc5 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for W, tag in (("fwd", "conv_fwd:dc5"), ("dgrad", "dgrad:dc5"), ("wgrad", "wgrad:dc5")):
    rec = {}
    for kind, cname in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(f"gpurun_out/pmc/traffic_{W}_{kind}_counter_collection.csv")):
            k = r["Kernel_Name"]
            main = ("wgrad_kernel" in k) if W == "wgrad" else ("conv_igemm_kernel" in k and "27" in k)
            if main and r["Counter_Name"] == cname:
                vals[k].append(float(r["Counter_Value"]))
        # the timed launches are the last REPS+1 dispatches of the main kernel
        best = max(vals.items(), key=lambda kv: sum(kv[1]))[1] if vals else []
        rec[cname + "_KB"] = sum(best) / max(len(best), 1)
    fetch = rec["FETCH_SIZE_KB"] * 1024 * 2      # gfx950: FETCH_SIZE reports half of a wide coalesced stream (MI355X_MICROARCH.md, HBM)
    write = rec["WRITE_SIZE_KB"] * 1024
    rec["hbm_bytes_per_launch"] = fetch + write
    out[tag] = rec
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on scripts/bench_conv.py dc5, B=4 128^3 bf16; FETCH_SIZE x2 (gfx950 correction)",
           "kernels": out}, open("gpurun_out/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
