#!/usr/bin/env python
"""Build-container only: time the oracle's faithful geo restatement next to the REAL reference renderer on identical
inputs (CPU, same thread count), so that the `cpu_baseline` bench.py reports on the GPU box (the oracle; the reference
source does not travel) can be read as reference-equivalent.  Prints the ratio recorded in BASELINE.md section 4."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import geo as og  # noqa: E402
from oracle.gen_golden_geo import import_reference, build_ref  # noqa: E402


def main(B=512, reps=3):
    torch.set_num_threads(os.cpu_count())
    R, Fd = import_reference()
    cfg = og.FULL_CFG
    p_sdf, p_col = og.make_sdf_params(cfg, 0), og.make_color_params(cfg, 1)
    sdf, col, var, ren = build_ref(Fd, R, cfg, p_sdf, p_col, 0.3)
    o, d, near, far = [torch.tensor(a) for a in og.make_rays(B, 2)]
    ts_p, tc_p = og.to_torch(p_sdf), og.to_torch(p_col)

    def ref():
        return ren.render(o, d, near, far, 2.0, perturb_overwrite=0, background_rgb=torch.ones(1, 3), cos_anneal_ratio=1.0)

    def ora():
        return og.render(ts_p, tc_p, torch.tensor(0.3), cfg, o, d, near, far, 2.0, background_rgb=torch.ones(1, 3), cos_anneal_ratio=1.0)

    out = {}
    for name, fn in (('reference', ref), ('oracle', ora)):
        fn()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            best = min(best, time.perf_counter() - t0)
        out[name] = best
        print(f'{name:10s} B={B}: {best:.2f} s  {B / best:.1f} rays/s')
    print(f'oracle / reference time ratio: {out["oracle"] / out["reference"]:.3f}  (threads: {torch.get_num_threads()})')


if __name__ == '__main__':
    main()
