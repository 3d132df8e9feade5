"""Microbenchmark of single conv launches (fwd / dgrad / wgrad) on the heavy SE-UNet layer shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import seunet_amd
from seunet_amd import ops as S, _lib

B = int(os.environ.get("B", 4))
only = sys.argv[1] if len(sys.argv) > 1 else ""
reps = int(os.environ.get("REPS", 5))
# name, splits, cout, dil, size
CASES = [("dc5", [32, 32], 32, 1, 128), ("dc3", [64, 64], 64, 1, 64), ("ec3", [16], 32, 2, 128), ("dc6", [32], 16, 1, 128),
         ("ec2", [8], 16, 1, 128), ("ec6", [32], 64, 2, 64), ("dc4", [64], 32, 1, 64), ("ec4", [32], 32, 1, 64), ("ec5", [32], 32, 2, 64),
         ("ec8", [64], 64, 2, 32), ("dc2", [64], 64, 1, 32), ("dc1", [128, 128], 64, 1, 32), ("ec11", [128], 128, 2, 16),
         # 1x1x1 layers (name ends in "_1"): the x33 / ec33 shortcut convolutions
         ("x33_1", [8], 32, 1, 128), ("ec33_1", [32], 32, 1, 128), ("ec63_1", [64, 32, 32], 64, 1, 64), ("ec93_1", [64, 64, 64], 128, 1, 32),
         ("dc42_1", [32, 32], 32, 1, 64), ("dc22_1", [64, 64], 64, 1, 32)]
dt = torch.bfloat16
for name, split, cout, dil, size in CASES:
    if only and only != name:
        continue
    if not only and name.endswith("_1") and not os.environ.get("ONE"):
        continue
    K = 1 if name.endswith("_1") else 3
    TAPS = K ** 3
    cin = sum(split)
    srcs = [torch.randn((B, size, size, size, c), device="cuda", dtype=dt) for c in split]
    w = torch.randn((cout, cin, K, K, K), device="cuda") * 0.05
    dy = torch.randn((B, size, size, size, cout), device="cuda", dtype=dt)
    flops = 2.0 * TAPS * cin * cout * B * size ** 3
    def timeit(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    lib = _lib.load()
    code = _lib.BF16
    wp = S.pack_weights(w, code, False)
    wpd = S.pack_weights(w, code, True)
    dims = _lib.Dims(B, size, size, size)
    out = torch.empty((B, size, size, size, cout), device="cuda", dtype=dt)
    slots = lib.seunet_conv_stats_slots(0, TAPS, dil, dims)
    stats = torch.zeros((B, slots, cout, 2), dtype=torch.float64, device="cuda")
    def fwd():
        _lib.check(lib.seunet_conv3d_fwd(code, 0, TAPS, dil, len(srcs), _lib.ptr_array(srcs), _lib.int_array(split), cin, wp.data_ptr(), 0,
                                         None, 1, _lib.ptr_array([out]), _lib.int_array([cout]), _lib.int_array([0]), stats.data_ptr(), dims, _lib.stream_ptr()))
    gs = [torch.empty_like(s) for s in srcs]
    def dgrad():
        _lib.check(lib.seunet_conv3d_fwd(code, 0, TAPS, dil, 1, _lib.ptr_array([dy]), _lib.int_array([cout]), cout, wpd.data_ptr(), 1,
                                         None, len(gs), _lib.ptr_array(gs), _lib.int_array(split), _lib.int_array([0] * len(gs)), None, dims, _lib.stream_ptr()))
    nb = lib.seunet_conv3d_wgrad_workspace_bytes(TAPS, cin, cout)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    dw = torch.empty((cout, cin, K, K, K), device="cuda")
    def wgrad(impl=3):
        _lib.check(lib.seunet_conv3d_wgrad(code, impl, TAPS, dil, len(srcs), _lib.ptr_array(srcs), _lib.int_array(split), cin, dy.data_ptr(), cout,
                                           dw.data_ptr(), ws.data_ptr(), nb, dims, _lib.stream_ptr()))
    # the marching kernel (csrc/conv_march.hip) on the same operands, where it serves the shape
    march = TAPS == 27 and bool(lib.seunet_conv3d_march_supported(code, dil, len(srcs), _lib.int_array(split), 1, _lib.int_array([cout])))
    march_d = TAPS == 27 and bool(lib.seunet_conv3d_march_supported(code, dil, 1, _lib.int_array([cout]), len(split), _lib.int_array(split)))
    if march:
        wm = torch.empty(lib.seunet_conv3d_march_wpack_bytes(cin, cout), dtype=torch.uint8, device="cuda")
        _lib.check(lib.seunet_conv3d_march_pack(code, w.data_ptr(), cin, cout, 0, cin, cout, wm.data_ptr(), _lib.stream_ptr()))
        mslots = lib.seunet_conv3d_march_slots(dil, cin, cout, dims)
        mstats = torch.zeros((B, mslots, cout, 2), dtype=torch.float64, device="cuda")
        bias = torch.zeros(cout, device="cuda")
        out_m = torch.empty_like(out)
    if march_d:
        wmd = torch.empty(lib.seunet_conv3d_march_wpack_bytes(cout, cin), dtype=torch.uint8, device="cuda")
        _lib.check(lib.seunet_conv3d_march_pack(code, w.data_ptr(), cin, cout, 1, cout, cin, wmd.data_ptr(), _lib.stream_ptr()))
        gs_m = [torch.empty_like(s) for s in srcs]
    def mfwd():
        _lib.check(lib.seunet_conv3d_march(code, dil, len(srcs), _lib.ptr_array(srcs), _lib.int_array(split), wm.data_ptr(), bias.data_ptr(),
                                           1, _lib.ptr_array([out_m]), _lib.int_array([cout]), _lib.int_array([0]), mstats.data_ptr(), dims, _lib.stream_ptr()))
    def mdgrad():
        _lib.check(lib.seunet_conv3d_march(code, dil, 1, _lib.ptr_array([dy]), _lib.int_array([cout]), wmd.data_ptr(), None,
                                           len(gs_m), _lib.ptr_array(gs_m), _lib.int_array(split), _lib.int_array([0] * len(gs_m)), None, dims, _lib.stream_ptr()))
    which = os.environ.get("WHICH", "fwd,dgrad,wgrad").split(",")
    if os.environ.get("SEUNET_STAMP"):
        import ctypes as C
        dbg = torch.zeros(12 * 4 * 400000, dtype=torch.int64, device="cuda")
        lib.seunet_debug_set_buffer.argtypes = [C.c_void_p]
        lib.seunet_debug_set_buffer(dbg.data_ptr())
        names = ["first-prefetch", "barrier1", "regwait+ldswrite", "barrier2", "prefetch-issue", "mfma-block", "bias+stats", "barrier-after-K", "index-plan", "xwave-stats", "stage-writes", "stage-reads+stores"]
        if os.environ.get("SEUNET_STAMP") == "stream":     # (a -DSEUNET_STAMP build of conv_stream.hip)
            names = ["prologue", "plane-wait", "barrier", "frag+dma+mfma-issue", "mfma-drain", "epilogue+stores"] + [""] * 6
        for nm, fn in (("fwd", fwd), ("dgrad", dgrad)):
            if nm not in which or os.environ.get("SEUNET_STAMP") == "stream":
                continue
            fn(); torch.cuda.synchronize(); dbg.zero_(); fn(); torch.cuda.synchronize()
            rec = dbg.view(-1, 12).double()
            used = rec.sum(1) > 0
            v = rec[used].sum(0).cpu(); nw = int(used.sum()); tot = float(v.sum())
            print("  stamps %s %s: " % (name, nm) + "  ".join("%s %.1f%%" % (names[i], 100 * float(v[i]) / tot) for i in range(12)) + "  (waves %d, cycles/wave %.0f)" % (nw, tot / nw), flush=True)
        if "wgrad" in which and os.environ.get("SEUNET_STAMP") != "stream":
            wn = ["prologue", "barrier1", "dma-issue", "dma-land+barrier2", "mfma-rows", "slab-store"]
            wgrad(); torch.cuda.synchronize(); dbg.zero_(); wgrad(); torch.cuda.synchronize()
            rec = dbg.view(-1, 12).double()
            used = rec.sum(1) > 0
            v = rec[used].sum(0).cpu(); nw = int(used.sum()); tot = float(v.sum())
            print("  stamps %s wgrad: " % name + "  ".join("%s %.1f%%" % (wn[i], 100 * float(v[i]) / tot) for i in range(6)) + "  (waves %d, cycles/wave %.0f)" % (nw, tot / nw), flush=True)
    res = []
    for nm, fn in (("fwd", fwd), ("dgrad", dgrad), ("wgrad", wgrad)):
        if nm in which:
            ms = timeit(fn)
            res.append("%s %.3f ms %.0f TF/s" % (nm, ms, flops / ms / 1e9))
    if march and "fwd" in which:
        fwd(); mfwd(); torch.cuda.synchronize()
        err = float((out_m.float() - out.float()).abs().max()) / float(out.float().abs().max())
        ms = timeit(mfwd)
        res.append("MARCH fwd %.3f ms %.0f TF/s (rel diff %.1e)" % (ms, flops / ms / 1e9, err))
    if march_d and "dgrad" in which:
        dgrad(); mdgrad(); torch.cuda.synchronize()
        err = max(float((a.float() - b.float()).abs().max()) / float(b.float().abs().max()) for a, b in zip(gs_m, gs))
        ms = timeit(mdgrad)
        res.append("MARCH dgrad %.3f ms %.0f TF/s (rel diff %.1e)" % (ms, flops / ms / 1e9, err))
    if TAPS == 1 and "wgrad" in which and all(c in (32, 64) for c in split) and cin in (64, 128, 192):
        wgrad(3); ref_dw = dw.clone(); wgrad(2); torch.cuda.synchronize()
        err = float((dw - ref_dw).abs().max()) / float(ref_dw.abs().max())
        ms = timeit(lambda: wgrad(2))
        res.append("1x1 wgrad %.3f ms (rel diff %.1e)" % (ms, err))
    if TAPS == 27 and cin % 32 == 0 and cout % 32 == 0 and len(set(split)) == 1 and len(split) <= 2 and "wgrad" in which:
        wgrad(3); ref_dw = dw.clone(); wgrad(2); torch.cuda.synchronize()
        err = float((dw - ref_dw).abs().max()) / float(ref_dw.abs().max())
        ms = timeit(lambda: wgrad(2))
        res.append("MARCH wgrad %.3f ms %.0f TF/s (rel diff %.1e)" % (ms, flops / ms / 1e9, err))
    # the streaming kernels (csrc/conv_stream.hip, wgrad_stream.hip) on the same operands, where they serve the shape
    if TAPS == 27 and len(split) == 1 and lib.seunet_conv3d_stream_supported(code, dil, cin, cout) and "fwd" in which:
        ws_ = torch.empty(lib.seunet_conv3d_stream_wpack_bytes(cin), dtype=torch.uint8, device="cuda")
        _lib.check(lib.seunet_conv3d_stream_pack(code, w.data_ptr(), cin, cout, 0, cin, cout, ws_.data_ptr(), _lib.stream_ptr()))
        sslots = lib.seunet_conv3d_stream_slots(dil, dims)
        sstats = torch.zeros((B, sslots, cout, 2), dtype=torch.float64, device="cuda")
        out_s = torch.empty_like(out)
        sb = torch.zeros(cout, device="cuda")
        def sfwd():
            _lib.check(lib.seunet_conv3d_stream(code, dil, srcs[0].data_ptr(), cin, ws_.data_ptr(), sb.data_ptr(), out_s.data_ptr(), cout, 0,
                                                sstats.data_ptr(), dims, _lib.stream_ptr()))
        ms = timeit(sfwd)
        if os.environ.get("SEUNET_STAMP") == "stream":
            sfwd(); torch.cuda.synchronize(); dbg.zero_(); sfwd(); torch.cuda.synchronize()
            rec = dbg.view(-1, 12).double(); used = rec.sum(1) > 0
            v = rec[used].sum(0).cpu(); nw = int(used.sum()); tot = float(v.sum())
            sn = ["first-plane-wait+barrier", "plane-wait", "barrier", "frag+dma+mfma-issue", "mfma-drain", "epilogue-rest", "stats", "pack+store-issue", "pro:plan", "pro:weights+geometry", "pro:dma-issue+weight-wait", ""]
            print("  stamps %s STREAM fwd: " % name + "  ".join("%s %.1f%%" % (sn[i], 100 * float(v[i]) / tot) for i in range(11)) + "  (waves %d, cycles/wave %.0f)" % (nw, tot / nw), flush=True)
        res.append("STREAM fwd %.3f ms %.0f TF/s" % (ms, flops / ms / 1e9))
    if TAPS == 27 and len(split) == 1 and lib.seunet_conv3d_stream_supported(code, dil, cout, cin) and "dgrad" in which:
        wd_ = torch.empty(lib.seunet_conv3d_stream_wpack_bytes(cout), dtype=torch.uint8, device="cuda")
        _lib.check(lib.seunet_conv3d_stream_pack(code, w.data_ptr(), cin, cout, 1, cout, cin, wd_.data_ptr(), _lib.stream_ptr()))
        g_s = torch.zeros_like(srcs[0])
        def sdgrad():
            _lib.check(lib.seunet_conv3d_stream(code, dil, dy.data_ptr(), cout, wd_.data_ptr(), None, g_s.data_ptr(), cin, 1,
                                                None, dims, _lib.stream_ptr()))
        ms = timeit(sdgrad)
        if os.environ.get("SEUNET_STAMP") == "stream":
            sdgrad(); torch.cuda.synchronize(); dbg.zero_(); sdgrad(); torch.cuda.synchronize()
            rec = dbg.view(-1, 12).double(); used = rec.sum(1) > 0
            v = rec[used].sum(0).cpu(); nw = int(used.sum()); tot = float(v.sum())
            sn = ["first-plane-wait+barrier", "plane-wait", "barrier", "frag+dma+mfma-issue", "mfma-drain", "epilogue-rest", "stats", "pack+store-issue", "pro:plan", "pro:weights+geometry", "pro:dma-issue+weight-wait", ""]
            print("  stamps %s STREAM dgrad: " % name + "  ".join("%s %.1f%%" % (sn[i], 100 * float(v[i]) / tot) for i in range(11)) + "  (waves %d, cycles/wave %.0f)" % (nw, tot / nw), flush=True)
        res.append("STREAM dgrad+= %.3f ms %.0f TF/s" % (ms, flops / ms / 1e9))
    print("%-4s %s->%d d%d @%d^3 B%d: " % (name, split, cout, dil, size, B) + " | ".join(res), flush=True)
