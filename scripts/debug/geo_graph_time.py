import sys, time
sys.path.insert(0, '.')
import torch, bench
dev = torch.device('cuda:0')
for graph in (False, True):
    runner, step = bench.geo_train_setup(dev, 0, 2560, graph=graph)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    # host-only enqueue time of one step (eager): how long until the call returns
    t1 = time.perf_counter(); step(); t2 = time.perf_counter(); torch.cuda.synchronize()
    print(f'graph={graph}: {dt*1e3:.2f} ms/step; host enqueue of one step {1e3*(t2-t1):.2f} ms; captured={runner._cap is not None}')
    # with the x3 coarse passes
    runner.renderer.matrix_mode = 'x3'
    if not graph:
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); print(f'   eager, coarse passes on the x3 SDF kernel: {(time.perf_counter()-t0)/10*1e3:.2f} ms/step')
    del runner, step
