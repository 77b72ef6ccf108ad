"""probe: the 256 -> 64 conv1 forward of layer1.1 launched on the tensors where the executor keeps them (its arena), against the same launch on
freshly allocated tensors — is it the placement that makes the launch 130 us in the step and 100 us in the harness?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sota_imagenet_amd import native, ops
from sota_imagenet_amd.models import resnet50
from sota_imagenet_amd.synth import synthetic_batch
from sota_imagenet_amd.losses import CrossEntropyLoss

N, S = 256, 224
m = resnet50(dtype="bf16").cuda(); m.train()
data, target = synthetic_batch(N, S, seed=0, index=0, device="cuda")
loss = CrossEntropyLoss(smoothing=0.1).cuda()(m(data), target); loss.backward(); torch.cuda.synchronize()
L = native.lib()
def dptr(name):
    p, dt, nd, sh = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int(), (ctypes.c_int * 4)()
    native.check(L.mi355_resnet50_debug_tensor(m._ctx(N, S, S), name.encode(), ctypes.byref(p), ctypes.byref(dt), ctypes.byref(nd), sh))
    return p.value, [sh[i] for i in range(nd.value)]
w = (torch.randn(64, 1, 1, 256, device="cuda") * 0.05).bfloat16()
part = torch.empty(4096 * 2 * 64, dtype=torch.float32, device="cuda")
nblk = ctypes.c_int(0)
def launch(px, py):
    native.check(L.mi355_conv2d_fwd_stats(native.BF16, ctypes.c_void_p(px), native.ptr(w), ctypes.c_void_p(py), native.ptr(part), part.numel() * 4, ctypes.byref(nblk),
                                          N, 56, 56, 256, 64, 1, 1, 1, 0, native.cur_stream()))
def t(fn, n=30):
    ts = []
    for i in range(n + 3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if i >= 3: ts.append(a.elapsed_time(b) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
xa, sx = dptr("layer1.0.out"); ya, sy = dptr("layer1.1.conv1.y")
print("arena: x at %#x %s, y at %#x %s, y - x = %d MB" % (xa, sx, ya, sy, (ya - xa) >> 20))
xf = torch.randn(N, 56, 56, 256, device="cuda").bfloat16(); yf = torch.empty(N, 56, 56, 64, device="cuda", dtype=torch.bfloat16)
print("fresh: x at %#x, y at %#x" % (xf.data_ptr(), yf.data_ptr()))
print("in the arena        %.1f us  (%s)" % (t(lambda: launch(xa, ya)), ops.last_conv_kernel()))
print("fresh tensors       %.1f us" % t(lambda: launch(xf.data_ptr(), yf.data_ptr())))
print("arena x, fresh y    %.1f us" % t(lambda: launch(xa, yf.data_ptr())))
print("fresh x, arena y    %.1f us" % t(lambda: launch(xf.data_ptr(), ya)))
xf.copy_(m.debug_tensor((N, S, S), "layer1.0.out"))
print("fresh x with the arena's DATA, fresh y  %.1f us" % t(lambda: launch(xf.data_ptr(), yf.data_ptr())))
