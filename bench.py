#!/usr/bin/env python3
"""bench.py -- SE-UNet fwd+bwd voxels/s on MI355X (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one synthetic batch per GPU (BASELINE.json configs[1]:
bf16, 4 x 2 x 128^3 patches per GPU): SE_UNet forward (HIP) -> stage-1 loss (sigmoid + Dice on both heads,
train.py:594-599, HIP) -> backward (HIP) -> one flat-bucket RCCL all-reduce of the gradients (N > 1) ->
AdamW step (train.py:569,603).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON
line.  `roofline` is computed for the kernel class with the largest share of the timed region from HIP-event
timings taken inside the timed region on the launch stream (native recorder, seunet_prof_*), and
`cpu_baseline` times the CPU oracle (oracle/, torch fp32, the reference's op sequence) on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0

# (name, taps, cin, cout, level, dilation) at width 1; 'in' = in_channel.  Mirrors csrc/net.cpp's kOps.
CONVS = [("ec1", 27, "in", 8, 0), ("ec2", 27, 8, 16, 0), ("ec3", 27, 16, 32, 0), ("ec33", 1, 56, 32, 0), ("x33", 1, "in", 32, 0),
         ("ec4", 27, 32, 32, 1), ("ec5", 27, 32, 32, 1), ("ec6", 27, 32, 64, 1), ("ec63", 1, 128, 64, 1), ("x63", 1, "in", 64, 1),
         ("ec7", 27, 64, 64, 2), ("ec8", 27, 64, 64, 2), ("ec9", 27, 64, 64, 2), ("ec93", 1, 192, 64, 2), ("x93", 1, "in", 64, 2),
         ("ec10", 27, 64, 64, 3), ("ec11", 27, 64, 64, 3), ("ec12", 27, 64, 64, 3), ("ec123", 1, 192, 64, 3),
         ("dc1", 27, 128, 64, 2), ("dc2", 27, 64, 64, 2), ("dc22", 1, 128, 64, 2), ("dc3", 27, 128, 64, 1), ("dc4", 27, 64, 32, 1),
         ("dc42", 1, 96, 32, 1), ("dc5", 27, 64, 32, 0), ("dc6", 27, 32, 16, 0)]


def conv_table(in_channel, width, batch, size):
    t = {}
    for name, taps, cin, cout, lvl in CONVS:
        ci = in_channel if cin == "in" else cin * width
        vox = batch * (size >> lvl) ** 3
        t[name] = {"flops": 2.0 * taps * ci * cout * width * vox, "cin": ci, "cout": cout * width, "vox": vox}
    return t


def algorithmic_work(tag, table, esz):
    """(flops, bytes) of one launch group, from SURVEY.md 8(d)'s per-layer figures."""
    kind, _, name = tag.partition(":")
    if kind in ("conv_fwd", "dgrad", "wgrad") and name in table:
        c = table[name]
        return c["flops"], (c["cin"] + c["cout"]) * c["vox"] * esz
    if kind in ("epi_fwd", "epi_bwd", "in_bwd", "cat_fwd", "cat_bwd") and name in table:
        c = table[name]
        per = {"epi_fwd": 2, "epi_bwd": 2, "in_bwd": 3, "cat_fwd": 2, "cat_bwd": 2}[kind]   # tensors of cout channels moved
        return 0.0, per * c["cout"] * c["vox"] * esz
    return 0.0, 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="patches per GPU (BASELINE configs[1]: 4)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--in-channel", type=int, default=2)
    ap.add_argument("--width", type=int, default=1)
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--torch-optimizer", action="store_true", help="torch.optim.AdamW instead of the fused seunet AdamW (SURVEY 8(f1))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not record HIP events inside the timed region")
    ap.add_argument("--cpu-size", type=int, default=128)
    ap.add_argument("--dump-kernels", default="", help="write the full per-launch-group timing table (TSV) here")
    args = ap.parse_args()

    import seunet_amd as A
    from seunet_amd import _lib, ddp
    import torch.distributed as dist

    # RCCL ("nccl") is the backend for real runs; SEUNET_DIST_BACKEND=gloo lets two ranks share one GPU for rehearsal
    local = ddp.init_from_env(os.environ.get("SEUNET_DIST_BACKEND", "nccl"))
    local = local % max(torch.cuda.device_count(), 1)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE is {world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = _lib.load()

    torch.manual_seed(0)
    model = A.SE_UNet(in_channel=args.in_channel, n_classes=1, width_mult=args.width, act_dtype=args.dtype).to(dev)
    model.eval()      # DropLayer off (parity configuration); everything else is identical in train()
    ddp.broadcast_parameters(model)
    opt = None if args.no_optimizer else (torch.optim.AdamW(model.parameters(), lr=1e-4) if args.torch_optimizer else A.AdamW(model.parameters(), lr=1e-4))
    group = True if world > 1 else None

    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    S, B = args.size, args.batch
    x = torch.rand((B, args.in_channel, S, S, S), generator=g, device=dev)
    label = (torch.rand((B, 1, S, S, S), generator=g, device=dev) < 0.03).float()

    def step():
        if opt is not None:
            opt.zero_grad(set_to_none=True)
        else:
            for p in model.parameters():
                p.grad = None
        pe, pd = model(x)
        loss = A.fused_stage_loss(1, pe, pd, label, group=group)
        loss.backward()
        if world > 1:
            ddp.allreduce_gradients(model.parameters())
        if opt is not None:
            opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # Per-launch timing with HIP events on the library's stream.  In the TIMED region only the launches of the largest
    # layer (":dc5": conv forward, data gradient, weight gradient -- the dominant-kernel candidates the roofline is
    # quoted on) are bracketed: an event per launch group costs ~1.5 us of stream time, ~5 % of the step when all ~350
    # groups are marked.  The full per-kernel table comes from two extra, untimed steps afterwards.
    import ctypes as C
    if not args.no_kernel_timing:
        lib.seunet_prof_enable_filtered(b":dc5")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    report, dom_report, table_steps = "", "", 2
    if not args.no_kernel_timing:
        buf = C.create_string_buffer(1 << 16)
        lib.seunet_prof_report(buf, len(buf))
        dom_report = buf.value.decode()
        lib.seunet_prof_enable(1)
        for _ in range(table_steps):
            step()
        fence()
        lib.seunet_prof_report(buf, len(buf))
        lib.seunet_prof_enable(0)
        report = buf.value.decode()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    voxels = world * B * S ** 3 * args.steps
    value = voxels / dt

    out = {
        "metric": "voxels/sec SE-UNet fwd+bwd, 128^3 patch", "value": value, "unit": "voxels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"SE-UNet base x{args.width} ({args.in_channel}-ch input), {B}x{S}^3 patches per GPU, "
                               f"fwd + stage-1 Dice loss + bwd" + ("" if args.no_optimizer else " + AdamW step")
                               + (" + flat-bucket RCCL all-reduce" if world > 1 else ""),
                   "global_batch": B * world, "patch": S, "parallelism": f"dp{world}", "final_loss": float(loss.detach())},
    }

    if rank == 0 and report:
        table = conv_table(args.in_channel, args.width, B, S)
        esz = 2 if args.dtype == "bf16" else 4
        rows = []
        for line in report.strip().splitlines():
            tag, ms, cnt = line.split("\t")
            rows.append((tag, float(ms), int(cnt)))
        timed = {}      # launch groups bracketed inside the timed region: tag -> (total ms, launches)
        for line in dom_report.strip().splitlines():
            tag, ms, cnt = line.split("\t")
            if tag != "(untimed)":
                timed[tag] = (float(ms), int(cnt))
        lib_ms = sum(ms for tag, ms, _ in rows if tag != "outside")
        if args.dump_kernels:
            with open(args.dump_kernels, "w") as f:
                f.write("kernel\ttotal_ms\tlaunches\tavg_ms\tTFLOP/s\tGB/s\n")
                for tag, ms, cnt in sorted(rows, key=lambda r: -r[1]):
                    fl, by = algorithmic_work(tag, table, esz)
                    avg = ms / cnt
                    f.write(f"{tag}\t{ms:.3f}\t{cnt}\t{avg:.4f}\t{fl / (avg * 1e-3) / 1e12:.1f}\t{by / (avg * 1e-3) / 1e9:.1f}\n")
        rows.sort(key=lambda r: -r[1])
        kernels = []
        for tag, ms, cnt in rows[:8]:
            fl, by = algorithmic_work(tag, table, esz)
            avg = ms / cnt
            kernels.append({"kernel": tag, "avg_ms": avg, "launches": cnt, "share_of_step": avg * (cnt / table_steps) / (1e3 * dt / args.steps),
                            "tflops": fl / (avg * 1e-3) / 1e12 if fl else None, "gbs": by / (avg * 1e-3) / 1e9 if by else None})
        # dominant kernel: the per-layer launch group (one launch per step) with the largest total time; the aggregated
        # classes of many small launches ("stats", "up_bwd", ...) are listed in "kernels" but are not one kernel
        dom = next((k for k in kernels if ":" in k["kernel"]), None) or next((k for k in kernels if k["kernel"] != "outside"), None)
        if timed:   # the roofline is quoted on the launch group with the largest total time INSIDE the timed region
            tag, (ms, cnt) = max(timed.items(), key=lambda kv: kv[1][0])
            fl, by = algorithmic_work(tag, table, esz)
            avg = ms / cnt
            dom = {"kernel": tag, "avg_ms": avg, "launches": cnt, "share_of_step": ms / (1e3 * dt),
                   "tflops": fl / (avg * 1e-3) / 1e12 if fl else None, "gbs": by / (avg * 1e-3) / 1e9 if by else None}
        if dom is not None:
            if dom["tflops"]:
                out["roofline"] = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom["tflops"], "peak": PEAK_TFLOPS[args.dtype],
                                   "unit": "TFLOP/s", "frac": dom["tflops"] / PEAK_TFLOPS[args.dtype], "traffic": None,
                                   "avg_launch_ms": dom["avg_ms"], "share_of_step": dom["share_of_step"]}
            else:
                out["roofline"] = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": (dom["gbs"] or 0.0) / PEAK_HBM_GBS, "traffic": None, "avg_launch_ms": dom["avg_ms"],
                                   "share_of_step": dom["share_of_step"]}
        # HBM traffic of the dominant kernel from rocprofv3 PMC passes (scripts/pmc_traffic.sh -> profiles/r01_traffic.json:
        # FETCH_SIZE and WRITE_SIZE collected in separate runs, FETCH_SIZE x2 per the gfx950 note in MI355X_MICROARCH.md)
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if dom is not None and "roofline" in out and os.path.exists(tpath) and B == 4 and S == 128 and args.dtype == "bf16":
            rec = json.load(open(tpath))["kernels"].get(dom["kernel"])
            if rec:
                out["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
                out["roofline"]["traffic_unit"] = "bytes per launch (PMC); algorithmic bytes per launch: %d" % int(
                    algorithmic_work(dom["kernel"], table, esz)[1])
        out["kernels"] = kernels
        out["library_ms_per_step"] = lib_ms / table_steps   # (from the fully marked, untimed steps)
        flops_step = 3.0 * sum(c["flops"] for c in table.values())   # fwd + dgrad + wgrad (upper bound: ec1/x* have no dgrad)
        out["model_tflops"] = flops_step / (dt / args.steps) / 1e12

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import seunet_oracle as orc
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        cores = int(os.environ.get("SEUNET_CPU_THREADS", min(cores, 16)))   # a 1-GPU box owns a 16-core share of the host
        torch.set_num_threads(cores)
        o = orc.build_oracle(args.in_channel, 1, 1, seed=0)
        cs = args.cpu_size
        b = orc.synthetic_batch(1, (cs, cs, cs), args.in_channel, seed=0)
        t1 = time.perf_counter()
        pe, pd = o(b["image"])
        orc.stage_loss(1, pe, pd, b["label"]).backward()
        cpu_dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": cs ** 3 / cpu_dt, "unit": "voxels/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"1 step (fwd + stage-1 loss + bwd) of 1x{args.in_channel}x{cs}^3 fp32 on the CPU oracle "
                                         f"(oracle/seunet_oracle.py, torch {torch.__version__}), {cpu_dt:.1f} s"}
    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
