"""Config helpers: mirror of decomp/nerfvq_nfr3/nerfactor/util/config.py:15-26 plus the `--config_override`
grammar of trainvali.py / train_nfr.py ('k=v,k=v'; values may contain ';' but not ',')."""


def config2dict(config):
    out = {}
    for k, v in config.items('DEFAULT'):
        assert k not in out, 'Duplicate flags not allowed'
        out[k] = v
    return out


def get_config_ini(ckpt_path):
    return '/'.join(ckpt_path.split('/')[:-2]) + '.ini'


def apply_override(config, override):
    """`override` = 'key=value,key=value' (train_nfr.py:56-64)."""
    if not override:
        return config
    for kv in override.split(','):
        if not kv:
            continue
        k, v = kv.split('=', 1)
        config.set('DEFAULT', k, v)
    return config
