import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daydreamer_amd import hipops
ops = hipops.HipOps('cuda:0')
M, N, K = (int(x) for x in sys.argv[1:4])
A = torch.randn(M, K, device='cuda'); B = torch.randn(K, N, device='cuda'); C = torch.empty(M, N, device='cuda')
for _ in range(5): ops.gemm(A, B, C)
torch.cuda.synchronize()
