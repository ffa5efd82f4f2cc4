import sys, time
sys.path.insert(0, '.')
import torch, bench
from vqnerf_release_amd.decomp.nerfactor import train_nfr
dev = torch.device('cuda:0')
for graph in (False, True):
    model, tr, step = bench.decomp_train_setup(dev, 0, 1, graph=graph)
    for _ in range(6): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print(f'2048 points graph={graph}: {(time.perf_counter()-t0)/20*1e3:.3f} ms/step')
    del model, tr, step
