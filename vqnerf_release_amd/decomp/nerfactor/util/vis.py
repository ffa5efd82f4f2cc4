"""Output path of the reflectance models: what `Model.vis_batch` writes per view
(decomp/nerfvq_nfr3/nerfactor/models/vq_nfr.py:988-1134, `_vis_embed` :1140-1152, `util/light.py:30-50`), restated for
device-resident result dicts with an asynchronous writer.

Same files per view directory as the reference: `<key>.png` for every image-like entry (rgb / diff / normal composited over
the white or black background with the thresholded ground-truth alpha; albedo / spec / rough / ks / basecolor as they are,
plus `<key>.npy`), `<key>_<probe>.png` per relighting probe, `embed_map.png`, `<key>.npy` for xyz (and z / embed under
`full_vis_path`), `metadata.json` (`id`, and `psnr` of the written 8-bit gt_rgb / pred_rgb when ground truth exists); once per
run `pred_light.png`, `np_light.npy` (and `vq_embed.npy` under `full_vis_path`).

What is different on purpose: nothing is written on the calling thread.  The tensors of a view leave the device through
pinned buffers on a side stream, and PNG / NPY encoding runs on worker threads (`AsyncWriter`), so a 16-probe relighting
pass does not stall the kernels behind ~20 PNG encodes per view.  `writer.flush()` joins.

Third-party arithmetic restated (xiuminglib, not in the reference tree): `xm.io.img.write_arr(arr, path, clip=True)` clips to
[0, 1] and casts `arr * 255` to uint8 (truncation); `xm.metric.PSNR('uint8')` = 10 log10(1 / mean((a/255 - b/255)^2)).
"""
import json
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

_EMBED_COLOURS = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [255, 0, 255], [0, 255, 255],
                           [128, 0, 0], [0, 128, 0], [0, 0, 128], [128, 128, 0], [128, 0, 128], [0, 128, 128],
                           [255, 128, 128], [128, 255, 128], [128, 128, 255], [255, 255, 128], [255, 128, 255],
                           [128, 255, 255]], np.uint8)           # vq_nfr.py:1141-1146 (handed to cv2.imwrite, i.e. B, G, R)


def to_uint8(arr01):
    return (np.clip(arr01, 0.0, 1.0) * 255.0).astype(np.uint8)


def alpha_blend(a, alpha, b):
    """a * alpha + b * (1 - alpha), alpha [H,W] broadcast over channels (util/img.py:78-97)."""
    if a.ndim == 3 and alpha.ndim == 2:
        alpha = alpha[:, :, None]
    return a * alpha + b * (1.0 - alpha)


def psnr_uint8(a, b):
    mse = np.mean((a.astype(np.float64) / 255.0 - b.astype(np.float64) / 255.0) ** 2)
    return float('inf') if mse == 0 else float(10.0 * np.log10(1.0 / mse))


def write_png(path, img_uint8):
    from PIL import Image
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    Image.fromarray(img_uint8).save(path, compress_level=3)


def embed_map(embed):
    """Code index image (0 = background, 1..18 = codes) -> the colour map of `_vis_embed`, RGB order of the written file."""
    out = np.zeros(embed.shape + (3,), np.uint8)
    e = np.rint(embed).astype(np.int64)
    for i in range(1, 19):
        out[e == i] = _EMBED_COLOURS[i - 1][::-1]
    return out


class AsyncWriter:
    """Device -> pinned host copies on a side stream + a pool of encoder threads.  submit() returns at once; flush() waits
    for everything submitted so far and re-raises the first worker error."""

    def __init__(self, n_threads=8):
        self.pool = ThreadPoolExecutor(max_workers=n_threads)       # leaf jobs: one PNG / NPY file each
        self.views = ThreadPoolExecutor(max_workers=2)              # per-view coordinators (they wait on leaf jobs: own pool)
        self.futures = []
        self.lock = threading.Lock()
        self.stream = None

    def fetch(self, tensors):
        """{k: device tensor} -> (host dict of pinned tensors, event or None); the copies are asynchronous."""
        dev = next((v.device for v in tensors.values() if torch.is_tensor(v) and v.is_cuda), None)
        if dev is None:
            return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in tensors.items()}, None
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        host = {}
        with torch.cuda.stream(self.stream):
            for k, v in tensors.items():
                if torch.is_tensor(v) and v.is_cuda:
                    v = v.detach()
                    v.record_stream(self.stream)
                    buf = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                    buf.copy_(v, non_blocking=True)
                    host[k] = buf
                else:
                    host[k] = v.detach() if torch.is_tensor(v) else v
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return host, ev

    def submit(self, fn, *args):
        f = self.views.submit(fn, *args)
        with self.lock:
            self.futures.append(f)
        return f

    def flush(self):
        with self.lock:
            fs, self.futures = self.futures, []
        for f in fs:
            f.result()


_default_writer = None


def default_writer():
    global _default_writer
    if _default_writer is None:
        _default_writer = AsyncWriter()
    return _default_writer


def _shape_views(host, hw):
    """Rows back to images (vq_nfr.py:1021-1034)."""
    out = {}
    for k, v in host.items():
        if v is None:
            continue
        a = v.numpy() if torch.is_tensor(v) else np.asarray(v)
        a = a.astype(np.float32, copy=False) if a.dtype.kind == 'f' else a.astype(np.float32)
        if k in ('pred_rgb_olat', 'pred_rgb_probes'):
            a = a.reshape(hw + (a.shape[1], 3))
        elif k.endswith(('rgb', 'albedo', 'normal', 'diff', 'spec', 'xyz', 'basecolor')):
            a = a.reshape(hw + (3,))
        elif k.endswith(('occu', 'depth', 'disp', 'alpha', 'rough', 'embed', 'ks')):
            a = a.reshape(hw)
        elif k.endswith(('z',)):
            a = a.reshape(hw + (a.shape[1],))
        else:
            raise NotImplementedError(k)
        out[k] = a
    return out


def _write_view(host, event, hw, id_, outdir, mode, white_bg, probe_names, olat_names, olat_first_n, alpha_thres, simp, full_vis_path,
                pool_submit):
    if event is not None:
        event.synchronize()
    d = _shape_views(host, hw)
    os.makedirs(outdir, exist_ok=True)
    alpha = d['gt_alpha'].copy()
    alpha[alpha < alpha_thres] = 0                               # stricter compositing (vq_nfr.py:1039-1042)
    full_vis = full_vis_path is not None
    jobs, written = [], {}

    def png(key, arr01):
        img = to_uint8(arr01)
        written[key] = img
        jobs.append(pool_submit(write_png, os.path.join(outdir, key + '.png'), img))

    def over_bg(v):
        return alpha_blend(v, alpha, np.ones_like(v) if white_bg else np.zeros_like(v))

    for k, v in d.items():
        if k in ('pred_rgb_olat', 'pred_rgb_probes'):
            names = olat_names if k == 'pred_rgb_olat' else probe_names
            for i, lname in enumerate(names[:v.shape[2]]):
                if k == 'pred_rgb_olat' and i >= olat_first_n:   # top half of the light grid only (:1049-1057)
                    break
                png(k + '_' + lname, over_bg(v[:, :, i, :]))
        elif k.endswith(('rgb', 'diff')):
            png(k, over_bg(v))
        elif k.endswith(('albedo', 'spec', 'rough', 'ks', 'basecolor')):
            jobs.append(pool_submit(np.save, os.path.join(outdir, k + '.npy'), v))
            png(k, v)
        elif k.endswith(('embed',)):
            if full_vis:
                jobs.append(pool_submit(np.save, os.path.join(full_vis_path, k + '.npy'), v))
            jobs.append(pool_submit(write_png, os.path.join(outdir, 'embed_map.png'), embed_map(v)))
        elif k.endswith(('z',)) and full_vis:                     # (catches '...xyz' too when full_vis is on, as the reference does)
            jobs.append(pool_submit(np.save, os.path.join(full_vis_path, k + '.npy'), v))
        elif k.endswith(('xyz',)):
            jobs.append(pool_submit(np.save, os.path.join(outdir, k + '.npy'), v))
        elif k.endswith('normal'):
            png(k, over_bg((v + 1.0) / 2.0))
        elif k.endswith(('z',)):
            pass                                                 # latent codes are only dumped under full_vis
        elif mode != 'render' and not simp:
            png(k, v)
    meta = {'id': id_}
    if not simp:
        if mode not in ('test', 'render') and 'gt_rgb' in written and 'pred_rgb' in written:
            meta['psnr'] = psnr_uint8(written['gt_rgb'], written['pred_rgb'])
        with open(os.path.join(outdir, 'metadata.json'), 'w') as f:
            json.dump(meta, f)
    for j in jobs:
        j.result()


def vis_light_uint8(light, h=None):
    """[h0,w0,3] probe -> uint8 image, optionally resized to height h with antialiased bilinear filtering (light.py:30-50)."""
    t = torch.as_tensor(light, dtype=torch.float32).detach().cpu()
    if h is not None and h != t.shape[0]:
        w = int(round(t.shape[1] * h / t.shape[0]))
        t = torch.nn.functional.interpolate(t.permute(2, 0, 1)[None], size=(h, w), mode='bilinear', align_corners=False,
                                            antialias=True)[0].permute(1, 2, 0)
    return to_uint8(t.numpy())


def vis_batch(model, data_dict, outdir, mode='train', light_vis_h=256, alpha_thres=0.8, simp=False, full_vis_path=None,
              writer=None):
    """Model.vis_batch (vq_nfr.py:988-1134).  data_dict: the `to_vis` dict of `call` / `fast_render` / `vis_mat` (device
    tensors with N = H*W rows, plus 'hw' [N,2] and 'id').  Returns the AsyncWriter the files were queued on."""
    model._validate_mode(mode)
    writer = writer or default_writer()
    if mode == 'vali':
        root = os.path.dirname(outdir.rstrip('/'))
        light_png = os.path.join(root, 'pred_light.png')
        if not os.path.exists(light_png):                        # the same for every view: once
            os.makedirs(root or '.', exist_ok=True)
            light = model.light.detach().cpu()
            write_png(light_png, vis_light_uint8(light, light_vis_h))
            np.save(os.path.join(root, 'np_light.npy'), light.numpy())
        if full_vis_path is not None and hasattr(model, 'get_codebook'):
            os.makedirs(full_vis_path, exist_ok=True)
            np.save(os.path.join(full_vis_path, 'vq_embed.npy'), model.get_codebook().detach().cpu().numpy())
    if mode == 'train':                                          # randomly sampled rays do not form an image
        return writer
    data_dict = dict(data_dict)
    hw_t = data_dict.pop('hw')
    hw = tuple(int(x) for x in (hw_t[0] if hw_t.ndim == 2 else hw_t).tolist())
    id_ = data_dict.pop('id')
    if isinstance(id_, (list, tuple)):
        id_ = id_[0]
    if isinstance(id_, bytes):
        id_ = id_.decode()
    host, ev = writer.fetch({k: v for k, v in data_dict.items() if v is not None})
    probe_names = list(getattr(model, 'novel_probes', {}) or {})
    olat_names = list(getattr(model, 'novel_olat', {}) or {})
    light_res = getattr(model, 'light_res', (16, 32))
    if full_vis_path is not None:
        os.makedirs(full_vis_path, exist_ok=True)
    writer.submit(_write_view, host, ev, hw, str(id_), outdir, mode, bool(model.white_bg), probe_names, olat_names,
                  int(np.prod(light_res)) // 2, alpha_thres, simp, full_vis_path, writer.pool.submit)
    return writer
