"""INI IO: mirror of decomp/nerfvq_nfr3/nerfactor/util/io.py:51-61 (single [DEFAULT] section, configparser)."""
from configparser import ConfigParser


def read_config(path):
    config = ConfigParser()
    with open(path, 'r') as h:
        config.read_file(h)
    return config


def write_config(config, path):
    with open(path, 'w') as h:
        config.write(h)


def config_from_dict(d):
    config = ConfigParser()
    for k, v in d.items():
        config.set('DEFAULT', str(k), str(v))
    return config


# ---- Radiance .hdr (RGBE) light probes --------------------------------------------------------------------------
# The reference reads its relighting probes with xiuminglib's `xm.io.hdr.read` (OpenCV's decoder; third party, not in the
# tree): float32 [H, W, 3] in R, G, B order, value = mantissa * 2^(exponent - 136), zero exponent = black.

def read_hdr(path):
    import numpy as np
    with open(path, 'rb') as f:
        data = f.read()
    pos = data.find(b'\n\n')
    if not data.startswith(b'#?') or pos < 0:
        raise ValueError(f'{path}: not a Radiance HDR file')
    header = data[:pos].decode('ascii', 'replace')
    if 'FORMAT=32-bit_rle_rgbe' not in header:
        raise ValueError(f'{path}: only FORMAT=32-bit_rle_rgbe is supported')
    end = data.index(b'\n', pos + 2)
    res = data[pos + 2:end].decode('ascii').split()
    if len(res) != 4 or res[0] != '-Y' or res[2] != '+X':
        raise ValueError(f'{path}: unsupported orientation {" ".join(res)!r} (expected -Y H +X W)')
    H, W = int(res[1]), int(res[3])
    buf = np.frombuffer(data, np.uint8, offset=end + 1)
    rgbe = np.empty((H, W, 4), np.uint8)
    p = 0
    for y in range(H):
        if 8 <= W <= 0x7fff and p + 4 <= buf.size and buf[p] == 2 and buf[p + 1] == 2 and ((int(buf[p + 2]) << 8) | int(buf[p + 3])) == W:
            p += 4                                               # adaptive run-length encoding, one channel after the other
            for c in range(4):
                x = 0
                while x < W:
                    n = int(buf[p]); p += 1
                    if n > 128:
                        n -= 128
                        rgbe[y, x:x + n, c] = buf[p]; p += 1
                    else:
                        rgbe[y, x:x + n, c] = buf[p:p + n]; p += n
                    if n == 0 or x + n > W:
                        raise ValueError(f'{path}: corrupt scanline {y}')
                    x += n
        else:                                                    # flat pixels
            rgbe[y] = buf[p:p + 4 * W].reshape(W, 4); p += 4 * W
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e == 0, 0.0, np.ldexp(1.0, e - 136)).astype(np.float32)
    return rgbe[..., :3].astype(np.float32) * scale[..., None]


def write_hdr(path, arr):
    """float [H, W, 3] (R, G, B) -> flat (uncompressed) RGBE file; the inverse of read_hdr up to the 8-bit mantissas."""
    import numpy as np
    a = np.asarray(arr, np.float32)
    m = a.max(-1)
    ex = np.zeros(m.shape, np.int32)
    nz = m > 1e-32
    ex[nz] = np.floor(np.log2(m[nz])).astype(np.int32) + 1       # m = f * 2^ex with f in [0.5, 1)
    scale = np.where(nz, np.ldexp(256.0, -ex), 0.0)
    rgbe = np.zeros(a.shape[:2] + (4,), np.uint8)
    rgbe[..., :3] = np.clip(a * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(nz, ex + 128, 0).astype(np.uint8)
    with open(path, 'wb') as f:
        f.write(b'#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n' + f'-Y {a.shape[0]} +X {a.shape[1]}\n'.encode('ascii') + rgbe.tobytes())
