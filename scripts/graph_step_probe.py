"""Probe: is the eager training step host-bound anywhere?  (a) eager fwd+loss+bwd, (b) the same with a 2 ms host sleep before
backward (absorbed when the host runs ahead), (c) the same captured in a HIP graph via torch.cuda.graph and replayed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import seunet_amd as A

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = A.SE_UNet(in_channel=2, n_classes=1, act_dtype="bf16").to(dev).eval()
x = torch.rand((4, 2, 128, 128, 128), device=dev)
label = (torch.rand((4, 1, 128, 128, 128), device=dev) < 0.03).float()

def step(sleep=0.0):
    for p in m.parameters():
        p.grad = None
    pe, pd = m(x)
    loss = A.fused_stage_loss(1, pe, pd, label)
    if sleep:
        time.sleep(sleep)
    loss.backward()
    return loss

def timeit(fn, k=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3

for _ in range(3):
    step()
print("eager            %.3f ms/step" % timeit(step), flush=True)
print("eager            %.3f ms/step" % timeit(step), flush=True)
print("eager + 2ms nap  %.3f ms/step" % timeit(lambda: step(0.002)), flush=True)
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        loss = step()
    torch.cuda.synchronize()
    print("captured; loss", float(loss), flush=True)
    print("graph replay     %.3f ms/step" % timeit(g.replay), flush=True)
    print("graph replay     %.3f ms/step" % timeit(g.replay), flush=True)
    print("loss after replay", float(loss))
except Exception as e:
    print("capture failed:", repr(e)[:1500])
