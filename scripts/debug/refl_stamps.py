"""In-kernel phase breakdown of vqn_refl_train_bwd_x3 from the diagnostic build (-DVQN_RT_STAMPS):
   VQN_LIB=vqnerf_release_amd/lib/libvqnerf_hip_rtstamps.so python scripts/debug/refl_stamps.py [points]"""
import ctypes, os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from tests.decomp_util import make_config
from vqnerf_release_amd import _C
from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
from vqnerf_release_amd.decomp.refl_train import ReflStackEngine
assert 'stamps' in _C.LIB_PATH
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
m = get_model_class('vq_nfr')(make_config()); m.build_nets(device='cuda', seed=1).to('cuda')
names = ['0 unit set-up / unattributed', '1 delta_2 (loads, D2 store, barrier)', '2 delta_1: rank-c VALU + tile-format I/O + commit',
         '3 GEMM entry: next_stream, first LDS operands, vmcnt drain', '4 init (epilogue-operand loads issued, rank-c, stash)',
         '5 K loop (incl. waiting for init loads / weights)', '6 epilogue (act\', tile-format store, split)', '7 commit / barriers', '8 -', '9 total']
lib = _C.lib()
for which in ('A', 'B'):
    if which == 'A':
        eng = ReflStackEngine([m.net['fine_enc'], m.net['bottleneck']], m.embedder['xyz'].n_freqs, [m.net[n] for n in ('diff_main', 'spec_main', 'rough_main')], 256, 'cuda')
        x = torch.nn.functional.normalize(torch.randn(N, 3, device='cuda'), dim=-1)
    else:
        eng = ReflStackEngine(None, 0, [m.net[n] for n in ('diff_vq', 'spec_vq', 'rough_vq')], 256, 'cuda')
        x = torch.nn.functional.normalize(torch.rand(N, 256, device='cuda'), dim=-1)
    ps = [p.detach() for p in eng.params()]
    with torch.no_grad():
        S, z, outs = eng.forward(x, ps)
        go = [torch.randn_like(o) for o in outs]
        gz = torch.randn(N, 256, device='cuda') if which == 'A' else None
        eng.backward(S, gz, go); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 128)()
        lib.vqn_debug_read_rt_stamps(buf, 1)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(); eng.backward(S, gz, go); ev1.record(); torch.cuda.synchronize()
        lib.vqn_debug_read_rt_stamps(buf, 1)
    a = np.array([int(t) for t in buf], np.float64).reshape(8, 16)
    print(f'stack {which}: N = {N}, backward incl. contractions {ev0.elapsed_time(ev1):.3f} ms; ticks per workgroup {a[0, 9] / max(a[0, 10], 1):.3e} over {int(a[0, 10])} workgroups; % of each wave time per phase:')
    print('    ' + ' ' * 72 + ''.join(f'  w{w}  ' for w in range(8)))
    for i in range(8):
        print(f'    {names[i]:72s}' + ''.join(f'{100.0 * a[w, i] / max(a[w, 9], 1):5.1f} ' for w in range(8)))
