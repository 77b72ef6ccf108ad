"""Per-layer conv forward timing through the per-op C-ABI (MI355_IGEMM_BIG=0/1 forces the 128- / 256-wide tile)."""
import sys, time, torch
sys.path.insert(0, '.')
from sota_imagenet_amd import ops
dt = torch.bfloat16
check = len(sys.argv) > 1 and sys.argv[1] == "check"
for (N,H,Cin,Cout,K,s) in [(256,14,256,256,3,1),(256,7,512,512,3,1),(256,14,1024,256,1,1),(256,14,256,1024,1,1),(256,7,2048,512,1,1),
                           (256,7,512,2048,1,1),(256,28,128,512,1,1),(256,56,64,256,1,1),(256,28,512,1024,1,2),(256,14,1024,2048,1,2)]:
    x = torch.randn(N,H,H,Cin, device='cuda').to(dt); w = (torch.randn(Cout,K,K,Cin, device='cuda')*0.05).to(dt)
    for _ in range(3): y = ops.conv2d_fwd(x,w,s,K//2)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): y = ops.conv2d_fwd(x,w,s,K//2)
    torch.cuda.synchronize(); t=(time.perf_counter()-t0)/20
    fl = 2.0*N*(H//s)**2*Cout*Cin*K*K
    msg = ""
    if check:
        ref = torch.nn.functional.conv2d(x.float().permute(0,3,1,2), w.float().permute(0,3,1,2), stride=s, padding=K//2).permute(0,2,3,1)
        err = (y.float()-ref).norm()/ref.norm()
        msg = f" rel err {err.item():.2e}"
    print(f"conv {Cin}->{Cout} k{K} s{s} H{H}: {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s{msg}")
