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
