import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from visualdet3d_amd import hip_ops as ops
torch.manual_seed(0)
def run(B,H,W,Cin,Cout,k,dtype, ident=False):
    x = torch.randn(B,Cin,H,W)
    if ident:
        w = torch.zeros(Cout,Cin,k,k); 
        for i in range(min(Cin,Cout)): w[i,i,k//2,k//2]=1
    else:
        w = torch.randn(Cout,Cin,k,k)*0.1
    y = F.conv2d(x,w,None,1,k//2)
    pc = ops.pack_conv(w.cuda(), None, None, dtype, 1, k//2, 1)
    out = ops.conv2d(x.permute(0,2,3,1).contiguous().cuda().to(dtype), pc)
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0,3,1,2)
    err = (got-y).abs().max().item()/y.abs().max().item()
    print('B%d %dx%d Cin%d Cout%d k%d %s ident=%s err=%.3e'%(B,H,W,Cin,Cout,k,dtype,ident,err))
    if err>1e-2:
        d = (got-y).abs()
        bad = (d>1e-2*y.abs().max()).nonzero()
        print('  nbad', len(bad), 'of', d.numel(), 'first', bad[:5].tolist())
        print('  got', got[0,:8,0,0].tolist()); print('  ref', y[0,:8,0,0].tolist())
        print('  got pix', got[0,0,0,:8].tolist()); print('  ref pix', y[0,0,0,:8].tolist())
for dt in (torch.float32, torch.bfloat16):
    run(1,8,32,64,64,1,dt,True)
    run(1,8,32,64,64,1,dt,False)
    run(1,8,32,64,128,1,dt,False)
    run(1,8,32,64,32,1,dt,False)
    run(1,8,32,64,64,3,dt,True)
    run(1,8,32,64,64,3,dt,False)
