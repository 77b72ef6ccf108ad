import sys, time, torch
sys.path.insert(0, '.')
from sota_imagenet_amd import ops
dt = torch.bfloat16
for (N,H,Cin,Cout,K,s) in [(256,14,256,256,3,1),(256,28,128,128,3,1),(256,56,64,64,3,1),(256,14,1024,256,1,1),(256,14,256,1024,1,1),(256,56,64,256,1,1)]:
    x = torch.randn(N,H,H,Cin, device='cuda').to(dt); w = (torch.randn(Cout,K,K,Cin, device='cuda')*0.05).to(dt)
    for _ in range(3): y = ops.conv2d_fwd(x,w,s,K//2)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): y = ops.conv2d_fwd(x,w,s,K//2)
    torch.cuda.synchronize(); t=(time.perf_counter()-t0)/20
    fl = 2.0*N*(H//s)**2*Cout*Cin*K*K
    print(f"conv {Cin}->{Cout} k{K} H{H}: {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s")
