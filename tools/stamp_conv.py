"""Cycle stamps of workgroup 0 inside the igemm main loop (library built with -DMI355_STAMP): per k-step
   t0 loop top | t1 after the counted vmcnt wait | t2 after the barrier | t3 after the last MFMA group was issued | t4 after L_advance."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from sota_imagenet_amd import ops, native
L = native.lib()
dt = torch.bfloat16
import numpy as np
for (N,H,Cin,Cout,K,s) in [(256,14,256,256,3,1),(256,7,512,512,3,1),(256,14,1024,256,1,1)]:
    x = torch.randn(N,H,H,Cin, device='cuda').to(dt); w = (torch.randn(Cout,K,K,Cin, device='cuda')*0.05).to(dt)
    for _ in range(2): y = ops.conv2d_fwd(x,w,s,K//2)
    torch.cuda.synchronize(); L.mi355_debug_stamps_clear()
    y = ops.conv2d_fwd(x,w,s,K//2); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8*4096))(); L.mi355_debug_stamps(buf, 8*4096)
    a = np.array(buf[:], dtype=np.uint64).reshape(4, 1024, 8).astype(np.int64)
    for wv in range(4):
        st = a[wv]; n = int((st[:,0] > 0).sum())
        if n < 4: continue
        st = st[1:n-1]
        d = lambda i,j: float(np.mean(st[:,j]-st[:,i]))
        step = float(np.mean(np.diff(a[wv][:n,0])))
        print(f"conv {Cin}->{Cout} k{K} wave {wv}: k-steps {n}, cycles/step {step:7.0f} | vmcnt wait {d(0,1):6.0f} | barrier {d(1,2):6.0f} | body(issue frags+pieces+mfma) {d(2,3):6.0f} | advance {d(3,4):5.0f}")
