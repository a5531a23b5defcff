import sys, torch
sys.path.insert(0,'.')
from visualdet3d_amd import hip_ops as ops
x = torch.randn(8,24,80,1152,device='cuda').to(torch.bfloat16)
w = torch.randn(1152,1152,3,3,device='cuda')*0.01
pc = ops.pack_conv(w,None,None,torch.bfloat16,1,1,1)
ops.conv2d(x,pc); torch.cuda.synchronize()
