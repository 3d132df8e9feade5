#!/usr/bin/env python3
"""bench.py -- SE-UNet fwd+bwd voxels/s on MI355X (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python bench.py --gpus 8 ...          # without WORLD_SIZE in the environment: starts the 8 ranks itself (below)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one synthetic batch per GPU (BASELINE.json configs[1]:
bf16, 4 x 2 x 128^3 patches per GPU): SE_UNet forward (HIP) -> stage-1 loss (sigmoid + Dice on both heads,
train.py:594-599, HIP) -> backward (HIP) -> one flat-bucket RCCL all-reduce of the gradients (N > 1) ->
AdamW step (train.py:569,603).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

`value` is the conservative whole step (optimizer and all-reduce included).  The same line also carries
  * `fwd_bwd_only`: SURVEY.md 8(d)'s metric as stated (forward + loss + backward, no optimizer), its own timed loop;
  * `hbm_model_frac` / `mfma_frac`: the step against the 8(d) byte model at 8 TB/s and the FLOP count at the dense MFMA peak;
  * `roofline`: the dominant launch group (chosen from a full per-launch table taken during warm-up), timed with HIP
    events on the launch stream inside the timed region;
  * `parity_mode` (N=1): ms/step of the fp32 activation mode, the one that meets north_star's 1e-3 tolerance;
  * `fp16_mode` (N=1): ms/step with fp16 activation storage (same bytes / MFMA rate as bf16, 8 x smaller gradient error);
  * `cpu_baseline` (N=1): the CPU oracle (oracle/, torch fp32, the reference's op sequence), 1 warm-up + 3 timed steps.

When `--gpus N > 1` and no torchrun environment is present, the parent process -- before anything touches the GPU --
starts `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a CHILD process, lets the
ranks print, and exits with the child's status (never an exec of a process that initialised the GPU).
"""
import argparse
import gc
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0

# (name, taps, cin, cout, level) at width 1; 'in' = in_channel.  Mirrors csrc/net.cpp's kOps.
CONVS = [("ec1", 27, "in", 8, 0), ("ec2", 27, 8, 16, 0), ("ec3", 27, 16, 32, 0), ("ec33", 1, 56, 32, 0), ("x33", 1, "in", 32, 0),
         ("ec4", 27, 32, 32, 1), ("ec5", 27, 32, 32, 1), ("ec6", 27, 32, 64, 1), ("ec63", 1, 128, 64, 1), ("x63", 1, "in", 64, 1),
         ("ec7", 27, 64, 64, 2), ("ec8", 27, 64, 64, 2), ("ec9", 27, 64, 64, 2), ("ec93", 1, 192, 64, 2), ("x93", 1, "in", 64, 2),
         ("ec10", 27, 64, 64, 3), ("ec11", 27, 64, 64, 3), ("ec12", 27, 64, 64, 3), ("ec123", 1, 192, 64, 3),
         ("dc1", 27, 128, 64, 2), ("dc2", 27, 64, 64, 2), ("dc22", 1, 128, 64, 2), ("dc3", 27, 128, 64, 1), ("dc4", 27, 64, 32, 1),
         ("dc42", 1, 96, 32, 1), ("dc5", 27, 64, 32, 0), ("dc6", 27, 32, 16, 0)]
# resampling ops of the graph: name -> (channels at width 1, level of the LOW-resolution side)
RESAMPLE = {"pool0": (32, 1), "pool0x": (8, 1), "pool1": (64, 2), "pool1x": (8, 2), "pool2": (64, 3),
            "up0": (64, 3), "up1": (64, 2), "up2": (32, 1)}
# forward bytes per voxel of the "two-pass fused" model and fwd+bwd FLOP per voxel (in_channel 2)
MODEL_FWD_BYTES_PER_VOXEL_2B = {1: 2098.0, 2: 4133.0}     # 2-byte activations; x2 for fp32
MODEL_FLOP_PER_VOXEL = {1: 901.0e3, 2: 3594.0e3}


def conv_table(in_channel, width, batch, size):
    t = {}
    for name, taps, cin, cout, lvl in CONVS:
        ci = in_channel if cin == "in" else cin * width
        vox = batch * (size >> lvl) ** 3
        t[name] = {"flops": 2.0 * taps * ci * cout * width * vox, "cin": ci, "cout": cout * width, "vox": vox, "taps": taps}
    return t


def algorithmic_work(tag, table, esz, batch, size, width):
    """(flops, bytes) of one launch of a launch group, from SURVEY.md 8(d)'s per-layer figures: every tensor the op must
    read or write, once.  Level maps / logits are f32 (4 B per voxel of their level)."""
    kind, _, name = tag.partition(":")
    v0 = batch * size ** 3
    if kind in ("conv_fwd", "dgrad", "wgrad") and name in table:
        c = table[name]
        return c["flops"], (c["cin"] + c["cout"]) * c["vox"] * esz
    if kind in ("epi_fwd", "epi_bwd", "in_bwd", "cat_fwd", "cat_bwd") and name in table:
        c = table[name]
        tensors = {"epi_fwd": 2, "epi_bwd": 2, "in_bwd": 3, "cat_fwd": 2, "cat_bwd": 2}[kind]   # tensors of cout channels moved
        extra = 8 if c["taps"] == 27 else 0       # gated blocks also read-modify-write their f32 level map (gradient: read)
        return 0.0, (tensors * c["cout"] * esz + extra) * c["vox"]
    if kind in ("pool_fwd", "pool_bwd", "up_fwd", "up_bwd") and name in RESAMPLE:
        ch, lvl = RESAMPLE[name]
        ch = ch if ch == 8 else ch * width
        low = batch * (size >> lvl) ** 3
        moved = 9 * low * ch * esz                     # the low-resolution tensor + the 8x larger high-resolution one
        if kind == "pool_bwd":
            moved += 8 * low * ch * esz                # the forward input is read again (first-max recompute)
        return 0.0, moved
    if kind in ("head_fwd", "head_bwd"):              # both heads: level maps (1 + 1/8 + 1/64 (+ 1/512)) + logits, f32
        return 0.0, (2 * 4 + 4 * (2 + 2 / 8 + 2 / 64 + 1 / 512)) * v0
    if kind == "pack_input":
        return 0.0, (2 * 4 + 8 * esz) * v0
    return 0.0, 0.0


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def host_cores():
    """Threads for the CPU baseline: the cores this process may run on, bounded by the cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except Exception:
        pass
    return int(os.environ.get("SEUNET_CPU_THREADS", n))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="patches per GPU (BASELINE configs[1]: 4)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--in-channel", type=int, default=2)
    ap.add_argument("--width", type=int, default=1)
    ap.add_argument("--train-mode", action="store_true", help="model.train(): DropLayer active (the reference's training callers)")
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--torch-optimizer", action="store_true", help="torch.optim.AdamW instead of the fused seunet AdamW (SURVEY 8(f1))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the fwd_bwd_only and parity_mode legs")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not record HIP events inside the timed region")
    ap.add_argument("--cpu-size", type=int, default=128)
    ap.add_argument("--dump-kernels", default="", help="write the full per-launch-group timing table (TSV) here")
    ap.add_argument("--config", default="window512", choices=["", "none", "window512"],
                    help="window512 (default at N=1): also time BASELINE configs[3] (one 512^3 volume, 128^3 windows, stride 64, 343 "
                         "windows, forward only; ~3 s) and report it under `window512`; none: skip it")
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 outside torchrun: start the N ranks as a child process tree.  Nothing in this process has touched
    the GPU (no torch.cuda call, torch not even imported)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import seunet_amd as A
    from seunet_amd import _lib, ddp
    import torch.distributed as dist
    import ctypes as C

    # RCCL ("nccl") is the backend for real runs; SEUNET_DIST_BACKEND=gloo lets two ranks share one GPU for rehearsal
    backend = os.environ.get("SEUNET_DIST_BACKEND", "nccl")
    local = ddp.init_from_env(backend)
    if dist.is_initialized() and dist.get_backend() != "nccl" and "SEUNET_DIST_BACKEND" not in os.environ:
        raise SystemExit(f"bench.py: the multi-GPU exchange must run over RCCL (torch backend 'nccl'), got {dist.get_backend()!r}; "
                         "set SEUNET_DIST_BACKEND=gloo only to rehearse the launcher on one GPU")
    local = local % max(torch.cuda.device_count(), 1)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but torch.distributed reports world_size {world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = _lib.load()
    S, B = args.size, args.batch

    def make_model(dtype):
        torch.manual_seed(0)
        m = A.SE_UNet(in_channel=args.in_channel, n_classes=1, width_mult=args.width, act_dtype=dtype).to(dev)
        m.train(args.train_mode)   # default eval(): DropLayer off (parity configuration); the kernels are the same
        ddp.broadcast_parameters(m)
        return m

    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    x = torch.rand((B, args.in_channel, S, S, S), generator=g, device=dev)
    label = (torch.rand((B, 1, S, S, S), generator=g, device=dev) < 0.03).float()
    group = True if world > 1 else None

    # data parallel: the exchange is overlapped with the backward pass (decoder bucket on a side stream behind an event the
    # library records, the rest after the backward; ddp.GradSync).  SEUNET_DDP_SERIAL=1: one all-reduce after the backward.
    overlap = world > 1 and not os.environ.get("SEUNET_DDP_SERIAL")

    def make_step(model, opt, ar_events=None):
        if overlap:
            model.grad_sync = ddp.GradSync(timing=ar_events is not None)

        def step():
            if opt is not None:
                opt.zero_grad(set_to_none=True)
            else:
                for p in model.parameters():
                    p.grad = None
            pe, pd = model(x)
            loss = A.fused_stage_loss(1, pe, pd, label, group=group)
            loss.backward()
            if world > 1 and not overlap:
                if ar_events is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                ddp.allreduce_gradients(model.parameters())
                if ar_events is not None:
                    e1.record()
                    ar_events.append((e0, e1))
            if opt is not None:
                opt.step()
            return loss
        return step

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, k):
        """k steps between two fences: (wall seconds, per-step milliseconds from events on the launch stream)."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        for e in ev:          # first use of an event allocates its completion signal: do that outside the timed region
            e.record()
        fence()
        gc.collect()
        gc.disable()
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(k):
            loss = step()
            ev[i + 1].record()
        fence()
        dt = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt)
        return dt, [ev[i].elapsed_time(ev[i + 1]) for i in range(k)], loss

    def make_opt(model):
        if args.no_optimizer:
            return None
        return torch.optim.AdamW(model.parameters(), lr=1e-4) if args.torch_optimizer else A.AdamW(model.parameters(), lr=1e-4)

    model = make_model(args.dtype)
    opt = make_opt(model)
    ar_events = []
    step = make_step(model, opt, ar_events)

    # ---- warm-up: one plain step (lazy initialisation), two fully marked steps (an event per launch group) to find the
    # dominant launch group, then the rest of the warm-up plain again, so the timed region does not start in the wake of
    # ~700 freshly created events
    buf = C.create_string_buffer(1 << 16)
    table_steps = min(2, max(args.warmup - 1, 0)) if not args.no_kernel_timing else 0
    first = min(1, args.warmup - table_steps)
    for _ in range(first):
        step()
    report = ""
    if table_steps:
        fence()
        lib.seunet_prof_enable(1)
        for _ in range(table_steps):
            step()
        fence()
        lib.seunet_prof_report(buf, len(buf))
        lib.seunet_prof_enable(0)
        report = buf.value.decode()
    for _ in range(args.warmup - table_steps - first):
        step()
    table = conv_table(args.in_channel, args.width, B, S)
    esz = 4 if args.dtype == "fp32" else 2
    rows = []
    for line in report.strip().splitlines():
        tag, ms, cnt = line.split("\t")
        if tag != "outside":
            rows.append((tag, float(ms), int(cnt)))
    rows.sort(key=lambda r: -r[1])
    # dominant launch group = the single kernel launch (one per step and direction) with the largest total time; groups of
    # many small launches ("stats") are reported in `kernels` but are not one kernel
    single = [r for r in rows if ":" in r[0] and algorithmic_work(r[0], table, esz, B, S, args.width)[1] > 0]
    dom_tag = single[0][0] if single else None
    if world > 1:      # every rank must mark the same group (the marks cost stream time)
        obj = [dom_tag]
        dist.broadcast_object_list(obj, src=0)
        dom_tag = obj[0]

    # ---- the timed region: EXACTLY --steps steps, only the dominant launch group bracketed by events (two events per
    # step; a fully marked step is ~5 % slower)
    ar_events.clear()
    if overlap:
        model.grad_sync.elapsed_ms()       # (drop the warm-up's marks)
    if dom_tag and not args.no_kernel_timing:
        lib.seunet_prof_enable_filtered(dom_tag.encode())
    dt, per_step_ms, loss = timed(step, args.steps)
    dom_report = ""
    if dom_tag and not args.no_kernel_timing:
        lib.seunet_prof_report(buf, len(buf))
        lib.seunet_prof_enable(0)
        dom_report = buf.value.decode()
    voxels_step = world * B * S ** 3
    value = voxels_step * args.steps / dt
    ms_step = 1e3 * dt / args.steps
    ar_ms = model.grad_sync.elapsed_ms() if overlap else [a.elapsed_time(b) for a, b in ar_events]
    if world > 1 and len(ar_ms) != args.steps:
        # a data-parallel step without its gradient exchange is a different (and wrong) workload: never report it
        raise SystemExit(f"bench.py: world_size {world} but {len(ar_ms)} gradient exchanges were timed in {args.steps} steps")
    if rank == 0:
        print("per-step ms: " + " ".join(f"{v:.2f}" for v in per_step_ms), file=sys.stderr)

    out = {
        "metric": "voxels/sec SE-UNet fwd+bwd, 128^3 patch", "value": value, "unit": "voxels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"SE-UNet base x{args.width} ({args.in_channel}-ch input), {B}x{S}^3 patches per GPU, "
                               f"fwd + stage-1 Dice loss + bwd" + ("" if args.no_optimizer else " + AdamW step")
                               + ((" + gradient all-reduce over %s (%s)" % ("RCCL" if dist.get_backend() == "nccl" else dist.get_backend(),
                                                                              "decoder bucket overlapped with the encoder's backward" if overlap
                                                                              else "one flat bucket after the backward")) if world > 1 else ""),
                   "global_batch": B * world, "patch": S, "parallelism": f"dp{world}", "world_size": world,
                   "dist_backend": (dist.get_backend() if dist.is_initialized() else None),
                   "droplayer": "train" if args.train_mode else "eval", "final_loss": float(loss.detach())},
        "median_ms_per_step": statistics.median(per_step_ms), "max_ms_per_step": max(per_step_ms),
    }
    if ar_ms:
        out["config"]["grad_allreduce_ms"] = {"median": statistics.median(ar_ms), "max": max(ar_ms), "exposed_only": bool(overlap),
                                              "elements": sum(p.numel() for n, p in model.named_parameters() if not n.startswith("dc62."))}
    if args.in_channel == 2 and args.width in MODEL_FLOP_PER_VOXEL:
        per_gpu_vox = B * S ** 3
        model_bytes = 3.0 * MODEL_FWD_BYTES_PER_VOXEL_2B[args.width] * (esz / 2) * per_gpu_vox
        model_flops = MODEL_FLOP_PER_VOXEL[args.width] * per_gpu_vox
        out["step_model"] = {"bytes": model_bytes, "flops": model_flops, "source": "SURVEY.md 8(d): 3 x two-pass-fused forward bytes; live-conv FLOPs"}
        out["hbm_model_frac"] = model_bytes / (ms_step * 1e-3) / (PEAK_HBM_GBS * 1e9)
        out["mfma_frac"] = model_flops / (ms_step * 1e-3) / (PEAK_TFLOPS[args.dtype] * 1e12)
        out["model_tflops"] = model_flops / (ms_step * 1e-3) / 1e12

    def group_record(tag, total_ms, cnt, per_steps, step_ms):
        fl, by = algorithmic_work(tag, table, esz, B, S, args.width)
        avg = total_ms / cnt
        rec = {"kernel": tag, "avg_ms": avg, "launches_per_step": cnt / per_steps, "share_of_step": total_ms / per_steps / step_ms,
               "tflops": fl / (avg * 1e-3) / 1e12 if fl else None, "gbs": by / (avg * 1e-3) / 1e9 if by else None}
        rec["mfma_frac"] = rec["tflops"] / PEAK_TFLOPS[args.dtype] if fl else None
        rec["hbm_frac"] = rec["gbs"] / PEAK_HBM_GBS if by else None
        return rec, fl, by

    if rank == 0 and rows:
        lib_ms = sum(ms for _, ms, _ in rows) / table_steps
        if args.dump_kernels:
            with open(args.dump_kernels, "w") as f:
                f.write("kernel\ttotal_ms_per_step\tlaunches_per_step\tavg_ms\tTFLOP/s\tGB/s\tfrac_mfma\tfrac_hbm\n")
                for tag, ms, cnt in rows:
                    rec, _, _ = group_record(tag, ms, cnt, table_steps, ms_step)
                    f.write("%s\t%.4f\t%g\t%.4f\t%s\t%s\t%s\t%s\n" % (
                        tag, ms / table_steps, cnt / table_steps, rec["avg_ms"],
                        "%.1f" % rec["tflops"] if rec["tflops"] else "-", "%.1f" % rec["gbs"] if rec["gbs"] else "-",
                        "%.3f" % rec["mfma_frac"] if rec["mfma_frac"] else "-", "%.3f" % rec["hbm_frac"] if rec["hbm_frac"] else "-"))
        out["kernels"] = [group_record(tag, ms, cnt, table_steps, ms_step)[0] for tag, ms, cnt in rows[:10]]
        out["library_ms_per_step"] = lib_ms    # (from the fully marked warm-up steps)
        # per kernel class: share of the marked step
        classes = {}
        for tag, ms, cnt in rows:
            k = tag.partition(":")[0]
            classes[k] = classes.get(k, 0.0) + ms / table_steps
        out["class_ms_per_step"] = {k: round(v, 4) for k, v in sorted(classes.items(), key=lambda kv: -kv[1])}
    if rank == 0 and dom_report:
        for line in dom_report.strip().splitlines():
            tag, ms, cnt = line.split("\t")
            if tag != dom_tag:
                continue
            rec, fl, by = group_record(tag, float(ms), int(cnt), args.steps, ms_step)
            mfma_bound = fl > 0 and (fl / (PEAK_TFLOPS[args.dtype] * 1e12)) >= (by / (PEAK_HBM_GBS * 1e9))
            out["roofline"] = {"kernel": tag, "bound": "mfma" if mfma_bound else "hbm",
                               "achieved": rec["tflops"] if mfma_bound else rec["gbs"],
                               "peak": PEAK_TFLOPS[args.dtype] if mfma_bound else PEAK_HBM_GBS,
                               "unit": "TFLOP/s" if mfma_bound else "GB/s",
                               "frac": rec["mfma_frac"] if mfma_bound else rec["hbm_frac"], "traffic": None,
                               "avg_launch_ms": rec["avg_ms"], "share_of_step": rec["share_of_step"],
                               "algorithmic_flops_per_launch": fl, "algorithmic_bytes_per_launch": by,
                               "selection": "launch group with the largest total time in a fully marked warm-up step; timed here "
                                            "with HIP events on the launch stream inside the timed region"}
            # HBM traffic of that kernel from rocprofv3 PMC passes over THIS command (scripts/pmc_traffic.sh ->
            # profiles/r03_traffic.json: FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE x2 per the gfx950 note
            # in MI355X_MICROARCH.md)
            tpath = os.path.join(ROOT, "profiles", "r04_traffic.json")
            if not os.path.exists(tpath):
                tpath = os.path.join(ROOT, "profiles", "r03_traffic.json")
            if os.path.exists(tpath) and B == 4 and S == 128 and args.dtype == "bf16" and args.width == 1:
                t = json.load(open(tpath)).get("kernels", {}).get(tag)
                if t:
                    out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = os.path.relpath(tpath, ROOT) + " (rocprofv3 --pmc over bench.py, per launch)"

    # ---- SURVEY 8(d)'s metric as stated: forward + loss + backward (+ the gradient all-reduce for N > 1), no optimizer
    if not args.no_secondary and not args.no_optimizer:
        step2 = make_step(model, None)
        step2()
        dt2, per2, _ = timed(step2, args.steps)
        out["fwd_bwd_only"] = {"value": voxels_step * args.steps / dt2, "unit": "voxels/s", "ms_per_step": 1e3 * dt2 / args.steps,
                               "median_ms_per_step": statistics.median(per2), "steps": args.steps,
                               "what": "forward + stage-1 loss + backward" + (" + gradient all-reduce" if world > 1 else "") + ", no optimizer step (SURVEY.md 8(d))"}

    # ---- the 1e-3-parity mode (fp32 activations, f64 InstanceNorm sums): a driver-timed number for it (N=1 only)
    if not args.no_secondary and world == 1 and args.dtype != "fp32":
        del model, opt, step
        torch.cuda.empty_cache()
        m32 = make_model("fp32")
        s32 = make_step(m32, make_opt(m32))
        for _ in range(2):
            s32()
        k32 = max(3, min(5, args.steps))
        dt3, per3, _ = timed(s32, k32)
        out["parity_mode"] = {"dtype": "fp32", "ms_per_step": 1e3 * dt3 / k32, "median_ms_per_step": statistics.median(per3),
                              "value": voxels_step * k32 / dt3, "unit": "voxels/s", "steps": k32,
                              "what": "same step with fp32 activation storage and fp32 MFMA: the mode the 1e-3 parity tests run in"}
        del m32, s32
        torch.cuda.empty_cache()
        # ---- BASELINE configs[3]: sliding-window inference over one 512^3 volume (prediction.py:78-109)
        if args.config == "window512":
            from seunet_amd.sliding_window import auto_batch
            mw = make_model(args.dtype).eval()
            vol = torch.rand((1, args.in_channel, 512, 512, 512), device=dev)
            A.sliding_window_predict(mw, vol[:, :, :192, :128, :128], batch=4, return_tensor=True)     # warm-up (4 windows)
            times = []
            for _ in range(2):
                fence()
                t0 = time.perf_counter()
                res = A.sliding_window_predict(mw, vol, return_tensor=True)
                fence()
                times.append(time.perf_counter() - t0)
            # 343 windows x the forward byte model (1.51 TB in 16-bit storage, 3.02 TB in fp32) against 8 TB/s, and
            # the forward FLOPs (216.3 TFLOP) against the dense MFMA peak
            w_bytes = 343 * MODEL_FWD_BYTES_PER_VOXEL_2B.get(args.width, 2098.0) * (esz / 2) * 128 ** 3
            w_flops = 343 * (MODEL_FLOP_PER_VOXEL.get(args.width, 901.0e3) / 3.0) * 128 ** 3
            out["window512"] = {"seconds": min(times), "runs": times, "windows": 343, "batch": auto_batch(mw, dev),
                                "output_voxels_per_s": 512 ** 3 / min(times), "window_voxels_per_s": 343 * 128 ** 3 / min(times),
                                "dtype": args.dtype, "finite": bool(torch.isfinite(res).all()),
                                "roofline": {"bound": "hbm", "achieved": w_bytes / min(times) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                             "frac": w_bytes / min(times) / 1e9 / PEAK_HBM_GBS, "model_bytes": w_bytes,
                                             "mfma_frac": w_flops / min(times) / (PEAK_TFLOPS[args.dtype] * 1e12),
                                             "what": "whole loop against SURVEY.md 8(d)'s forward byte model x 343 windows (traffic: null, "
                                                     "a model figure, not a PMC reading)", "traffic": None},
                                "what": "one 512^3 two-channel volume resident in HBM -> overlap-averaged sigmoid volume (float64) in HBM; "
                                        "decoder-head-only forward per window batch, device-side gather / accumulate / divide"}
            del mw, vol, res
            torch.cuda.empty_cache()
        # ---- fp16 activation storage (BASELINE configs[4]'s dtype): the same bytes and MFMA rate as bf16 with three more
        # mantissa bits -- 8 x smaller gradient error against float64 (profiles/r03_lowprec_attribution_32.md)
        if args.dtype == "bf16":
            m16 = make_model("fp16")
            s16 = make_step(m16, make_opt(m16))
            for _ in range(2):
                s16()
            dt4, per4, _ = timed(s16, k32)
            out["fp16_mode"] = {"dtype": "fp16", "ms_per_step": 1e3 * dt4 / k32, "median_ms_per_step": statistics.median(per4),
                                "value": voxels_step * k32 / dt4, "unit": "voxels/s", "steps": k32,
                                "what": "same step with fp16 activation storage (static loss scale 65536): large-tensor gradient error "
                                        "against float64 with the same discrete choices 6e-3 (bf16: 5e-2)"}
            del m16, s16
            torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import seunet_oracle as orc
        cores = host_cores()
        torch.set_num_threads(cores)
        o = orc.build_oracle(args.in_channel, 1, 1, seed=0)
        cs = args.cpu_size
        b = orc.synthetic_batch(1, (cs, cs, cs), args.in_channel, seed=0)
        times = []
        for i in range(4):                     # 1 warm-up (thread pool, allocator, oneDNN primitives) + 3 timed
            for p in o.parameters():
                p.grad = None
            t1 = time.perf_counter()
            pe, pd = o(b["image"])
            orc.stage_loss(1, pe, pd, b["label"]).backward()
            if i:
                times.append(time.perf_counter() - t1)
        best, med = min(times), statistics.median(times)
        out["cpu_baseline"] = {"value": cs ** 3 / best, "median_value": cs ** 3 / med, "unit": "voxels/s", "cores": torch.get_num_threads(),
                               "kind": "port", "cpu": cpu_model(),
                               "sample": f"1 warm-up + 3 timed steps (fwd + stage-1 loss + bwd) of 1x{args.in_channel}x{cs}^3 fp32 on the CPU "
                                         f"oracle (oracle/seunet_oracle.py, torch {torch.__version__}); best {best:.1f} s, median {med:.1f} s"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
