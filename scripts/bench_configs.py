"""Timing of the non-headline BASELINE.json configs on one GPU (for DESIGN.md; bench.py stays on configs[1])."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import seunet_amd as A

def sync_time(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

torch.manual_seed(0)
# configs[3]: sliding-window whole-volume inference, 512^3, stride 64, 343 windows
m = A.SE_UNet(2, 1, act_dtype="bf16").cuda().eval()
x = torch.rand(1, 2, 512, 512, 512, device="cuda")
A.sliding_window_predict(m, x[:, :, :128, :128, :256], 128, 64, batch=1, return_tensor=True)      # warm-up
for batch in (16, 4, 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = A.sliding_window_predict(m, x, 128, 64, batch=batch, return_tensor=True)     # first call at this batch size: includes the
    torch.cuda.synchronize(); dt_first = time.perf_counter() - t0                         # allocator growing to the new workspace
    del out
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = A.sliding_window_predict(m, x, 128, 64, batch=batch, return_tensor=True)     # result stays on the device (what the
    torch.cuda.synchronize(); dt = time.perf_counter() - t0                               # GPU post-processing takes next)
    ok = bool(float(out.max()) < 1.0001)
    t1 = time.perf_counter()
    host = out.cpu().numpy()                                                             # the reference's API returns numpy
    dt2 = time.perf_counter() - t1
    print("configs[3] 512^3 stride 64 (343 windows, batch %d): %.3f s  -> %.1f M output voxels/s, %.1f M window-voxels/s, finite=%s; "
          "first call at this batch size %.3f s; + %.2f s for the 1-GB float64 D2H copy when the caller wants numpy"
          % (batch, dt, 512 ** 3 / dt / 1e6, 343 * 128 ** 3 / dt / 1e6, ok, dt_first, dt2), flush=True)
    del host
del x, out
# forward-only window throughput
xw = torch.rand(4, 2, 128, 128, 128, device="cuda")
with torch.no_grad():
    t = sync_time(lambda: m(xw), 5)
print("forward only 4x128^3 bf16: %.2f ms -> %.1f M voxels/s" % (t * 1e3, 4 * 128 ** 3 / t / 1e6), flush=True)
del m
# configs[4] on one GPU: 2x width, 160^3, fp16 activation storage (static loss scale 65536) and the bf16 mode beside it
for B, dt_ in ((1, "fp16"), (2, "fp16"), (2, "bf16")):
    m2 = A.SE_UNet(2, 1, width_mult=2, act_dtype=dt_).cuda().eval()
    x2 = torch.rand(B, 2, 160, 160, 160, device="cuda")
    lab = (torch.rand(B, 1, 160, 160, 160, device="cuda") < 0.03).float()
    def step():
        for p in m2.parameters():
            p.grad = None
        e, d = m2(x2)
        A.fused_stage_loss(1, e, d, lab).backward()
    t = sync_time(step, 3)
    ok = all(torch.isfinite(p.grad).all() for n, p in m2.named_parameters() if not n.startswith("dc62."))
    print("configs[4]: width x2, %dx160^3 %s fwd+bwd: %.1f ms -> %.1f M voxels/s (finite grads: %s, peak mem %.1f GB)"
          % (B, dt_, t * 1e3, B * 160 ** 3 / t / 1e6, bool(ok), torch.cuda.max_memory_allocated() / 1e9), flush=True)
    del m2, x2, lab

# SURVEY 8(f2)/(f4) on a 512^3 probability volume: double-threshold sweep, largest 26-connected component, hole fill
g = torch.Generator(device="cuda").manual_seed(5)
prob = torch.rand((512, 512, 512), generator=g, device="cuda", dtype=torch.float64)
prob = torch.nn.functional.avg_pool3d(prob[None, None].float(), 5, 1, 2)[0, 0].double()     # blobs instead of salt and pepper
for name, fn in (("double_threshold_iteration (h 0.5 / l 0.4)", lambda: A.double_threshold_iteration(prob, 0.5, 0.4)),
                 ("largest_component (26-conn)", lambda: A.largest_component((prob > 0.5).to(torch.uint8))),
                 ("maximum_3d (largest 26-conn + 2-D hole fill)", lambda: A.maximum_3d((prob > 0.5).to(torch.uint8)))):
    try:
        t = sync_time(fn, 3)
        print("512^3 %s: %.1f ms" % (name, t * 1e3), flush=True)
    except Exception as e:        # (argument conventions differ between the wrappers; the tests are the reference for them)
        print("512^3 %s: skipped (%s)" % (name, str(e)[:80]), flush=True)
