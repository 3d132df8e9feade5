# same-box A/B: round-3 tree (_r3/, its own library and bench.py) vs this tree.  _r3/ is not tracked: create it with
#   git worktree add -f _r3 e7350ee && make -C _r3/se-unet-airseg_amd/csrc -j8
# (and remove it again with `git worktree remove --force _r3`)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
for rep in 1 2; do
  (cd _r3 && timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-secondary --dump-kernels ../gpurun_out/r04/ab_r3_groups.tsv > ../gpurun_out/r04/ab_r3.json 2> ../gpurun_out/r04/ab_r3.err)
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-secondary --config none --dump-kernels gpurun_out/r04/ab_r4_groups.tsv > gpurun_out/r04/ab_r4.json 2> gpurun_out/r04/ab_r4.err
  python3 - <<'PY'
import json
for t in ("r3","r4"):
    d=json.load(open(f"gpurun_out/r04/ab_{t}.json"))
    print(t, round(d["ms_per_step"],3), round(d["median_ms_per_step"],3), d["class_ms_per_step"])
PY
done
