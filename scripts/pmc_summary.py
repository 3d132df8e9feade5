#!/usr/bin/env python3
"""Summarise the PMC passes of scripts/profile_r02.sh: per (kernel, grid) averages of FETCH_SIZE / WRITE_SIZE (HBM bytes per
launch; FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md 'HBM') and of the SQ counters; writes <dir>/traffic.json with the
launch groups of the dominant layer (dc5) identified by kernel template + grid size.
usage: scripts/pmc_summary.py gpurun_out/r02"""
import collections, csv, glob, json, os, sys
d = sys.argv[1]
def load(sub):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"][:90], int(r["Grid_Size"]), int(r["Workgroup_Size"]))
            rows[key][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return rows
fetch, write, sq = load("pmc_FETCH_SIZE"), load("pmc_WRITE_SIZE"), load("pmc_SQ_VALU_MFMA_BUSY_CYCLES")
STEPS = 4        # the profiled command runs --warmup 1 --steps 3
def avg(v):
    v = [x[1] if isinstance(x, tuple) else x for x in v]
    return sum(v) / max(len(v), 1)
def first_of_each_step(v):
    """Layers that share a template instantiation AND a grid (the 27-tap weight-gradient kernel runs every layer on the
    same persistent grid) are told apart by launch order: the dispatches of one key repeat with the step, and dc5's weight
    gradient is the first of them in every backward pass (dc6 runs on the streaming kernel)."""
    v = sorted(v)
    per = max(len(v) // STEPS, 1)
    return [x for i, x in enumerate(v) if i % per == 0]
print("%-90s %10s %5s %12s %12s %12s" % ("kernel", "workitems", "n", "fetch MB x2", "write MB", "HBM MB"))
table = {}
for key in sorted(set(fetch) | set(write), key=lambda k: -(avg(fetch[k].get("FETCH_SIZE", [0])))):
    f = avg(fetch[key].get("FETCH_SIZE", [0])) * 1024 * 2      # FETCH_SIZE is reported in KB; x2: gfx950 tallies 128-B requests at 64 B
    w = avg(write[key].get("WRITE_SIZE", [0])) * 1024
    table[key] = (f, w)
    if f + w > 20e6:
        print("%-90s %10d %5d %12.1f %12.1f %12.1f" % (key[0], key[1], len(fetch[key].get("FETCH_SIZE", [])), f / 1e6, w / 1e6, (f + w) / 1e6))
print()
print("%-90s %10s %12s %12s %10s %10s" % ("kernel", "workitems", "MFMA busy", "CU busy", "mfma/busy", "wait/wave"))
for key, c in sorted(sq.items(), key=lambda kv: -avg(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))):
    m, b = avg(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), avg(c.get("SQ_BUSY_CU_CYCLES", [0]))
    wv, wt = avg(c.get("SQ_WAVE_CYCLES", [0])), avg(c.get("SQ_WAIT_ANY", [0]))
    if m > 0:
        print("%-90s %10d %12.3e %12.3e %10.3f %10.3f" % (key[0], key[1], m, b, m / b if b else 0, wt / wv if wv else 0))
# the dc5 launch groups at the bench shape (4 x 128^3): tiled conv kernel, 27 taps (", 27, 1>" / "Li27ELi1E" in the name),
# forward = 4096 tiles x 1 column block x 4 samples, data gradient = 2 column blocks; 256 threads per workgroup
def is27(name):
    n = name.replace(" ", "")
    return ",27,1>" in n or "Li27ELi1E" in n
groups = {}
for key, (f, w) in table.items():
    name, items, wg = key
    if "conv_igemm_kernel" in name and is27(name) and items == 4096 * 2 * 4 * 256: groups["dgrad:dc5"] = (f, w)
    if "conv_igemm_kernel" in name and is27(name) and items == 4096 * 1 * 4 * 256: groups["conv_fwd:dc5"] = (f, w)
# round 3: dc5 runs on the marching kernel (csrc/conv_march.hip), 256 workgroups of 256 threads at the bench shape.
# conv_march_kernel<T, KS, NGW, RYW, DIL, MODE>: the data gradient of dc5 is the only <.., 1, 4, 8, 1, 1> launch of a step; its
# forward <.., 2, 2, 4, 1, 0> shares instantiation and grid with dc4's forward and is the LAST such launch of every forward pass.
def march_is(name, sig):
    n = name.replace(" ", "")
    # (rocprofv3 prints some instantiations demangled with the first integer argument swallowed: "<bool _Accum, int, E, 4, 8, 1, 1>")
    # (round 4: a trailing `bool BUF` template argument -- ",true>" / "Lb1E" -- follows MODE)
    tails = (">", ",true>", ",false>")
    return "conv_march_kernel" in n and (any(",".join(map(str, sig)) + t in n for t in tails) or "".join("Li%dE" % v for v in sig) in n or
                                         (len(sig) == 5 and any("E," + ",".join(map(str, sig[1:])) + t in n for t in tails)))
def last_of_each_step(v):
    v = sorted(v)
    per = max(len(v) // STEPS, 1)
    return [x for i, x in enumerate(v) if i % per == per - 1]
for key in fetch:
    name, items, wg = key
    for tag, sig, pick in (("dgrad:dc5", (1, 4, 8, 1, 1), None), ("conv_fwd:dc5", (2, 2, 4, 1, 0), last_of_each_step)):
        if march_is(name, sig) and items == 256 * 256:
            fv, wv = fetch[key].get("FETCH_SIZE", []), write.get(key, {}).get("WRITE_SIZE", [])
            if pick:
                fv, wv = pick(fv), pick(wv)
            groups[tag] = (avg(fv) * 1024 * 2, avg(wv) * 1024)
            c = sq.get(key, {})
            mv, bv = c.get("SQ_VALU_MFMA_BUSY_CYCLES", []), c.get("SQ_BUSY_CU_CYCLES", [])
            if pick:
                mv, bv = pick(mv), pick(bv)
            if bv:
                print("%-90s %10s %12.3e %12.3e %10.3f   (%s)" % (key[0], "dc5", avg(mv), avg(bv), avg(mv) / avg(bv), tag))
for key in fetch:
    name, items, wg = key
    # the weight gradient of dc5: the tiled kernel (round 2; 512 persistent workgroups) or the marching kernel
    # wgrad_march_kernel<T, 4, 2, 1> (round 3; 256 workgroups, shared with dc4) -- in both cases the first launch of a step
    tiled = "wgrad_kernel" in name and "stream" not in name and ("Li27ELi1E" in name or ", 27, 1" in name) and items == 256 * 2 * 256
    n_ = name.replace(" ", "")
    march = "wgrad_march_kernel" in n_ and ("Li4ELi2ELi1E" in n_ or ",4,2,1>" in n_) and items == 256 * 256
    if tiled and any("wgrad_march_kernel" in k[0] for k in fetch):
        continue                         # (round 3: some other, coarser layer owns the tiled kernel's first launch now)
    if tiled or march:
        f = avg(first_of_each_step(fetch[key].get("FETCH_SIZE", []))) * 1024 * 2
        w = avg(first_of_each_step(write[key].get("WRITE_SIZE", []))) * 1024
        groups["wgrad:dc5"] = (f, w)
        c = sq.get(key, {})
        m, b = avg(first_of_each_step(c.get("SQ_VALU_MFMA_BUSY_CYCLES", []))), avg(first_of_each_step(c.get("SQ_BUSY_CU_CYCLES", [])))
        if b:
            print("%-90s %10s %12.3e %12.3e %10.3f   (first launch of each step = wgrad:dc5)" % (key[0], "dc5", m, b, m / b))
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over `python3 bench.py --steps 3 --warmup 1`; "
                   "bytes per launch, FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section)",
           "kernels": {k: {"fetch_bytes": v[0], "write_bytes": v[1], "hbm_bytes_per_launch": v[0] + v[1]} for k, v in groups.items()}},
          open(os.path.join(d, "traffic.json"), "w"), indent=1)
print("\ntraffic.json:", json.dumps({k: round((v[0] + v[1]) / 1e6, 1) for k, v in groups.items()}), "MB per launch")
