"""Config surface of the geo half: a small parser for the HOCON subset the reference's `confs/*.conf` use
(geo/NeuS-ours2/confs/nerf.conf etc., read there with pyhocon's `ConfigFactory.parse_string`,
nerf_runner.py:27-33): nested `name { ... }` blocks, `key = value` / `key: value`, numbers, True/False, bare or
quoted strings, `[a, b, ...]` lists (possibly multi-line), optional trailing commas, `#` and `//` comments.
The returned tree offers the pyhocon calls the runners make: `conf['a.b']`, `get_int/float/bool/string/list`,
`**conf['model.sdf_network']`."""
import re


class ConfigTree(dict):
    def _walk(self, key):
        node = self
        for part in key.split('.'):
            if not isinstance(node, dict) or part not in node:
                raise KeyError(key)
            node = node[part]
        return node

    def __getitem__(self, key):
        if isinstance(key, str) and '.' in key and not dict.__contains__(self, key):
            return self._walk(key)
        return dict.__getitem__(self, key)

    def __setitem__(self, key, value):
        if isinstance(key, str) and '.' in key:
            head, tail = key.rsplit('.', 1)
            dict.__setitem__(self._walk(head), tail, value)
        else:
            dict.__setitem__(self, key, value)

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def _typed(self, key, conv, default):
        try:
            return conv(self[key])
        except KeyError:
            if default is _MISSING:
                raise
            return default

    def get_int(self, key, default=None):
        return self._typed(key, int, _MISSING if default is None else default)

    def get_float(self, key, default=None):
        return self._typed(key, float, _MISSING if default is None else default)

    def get_bool(self, key, default=None):
        def conv(v):
            if isinstance(v, str):
                return v.strip().lower() in ('true', 'yes', 'on', '1')
            return bool(v)
        return self._typed(key, conv, _MISSING if default is None else default)

    def get_string(self, key, default=None):
        return self._typed(key, str, _MISSING if default is None else default)

    def get_list(self, key, default=None):
        return self._typed(key, list, _MISSING if default is None else default)


_MISSING = object()
_TOKEN = re.compile(r'''\s*(?:(?P<comment>(?:\#|//)[^\n]*)|(?P<punct>[{}\[\],=:])|"(?P<qstr>(?:[^"\\]|\\.)*)"|(?P<bare>[^\s{}\[\],=:#"]+))''')


def _scalar(tok):
    low = tok.lower()
    if low in ('true', 'yes', 'on'):
        return True
    if low in ('false', 'no', 'off'):
        return False
    if low in ('null', 'none'):
        return None
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok


def _tokens(text):
    pos, out = 0, []
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == '':
                break
            raise ValueError(f'conf syntax error near {text[pos:pos + 30]!r}')
        pos = m.end()
        if m.group('comment') is not None:
            continue
        if m.group('punct'):
            out.append(('p', m.group('punct')))
        elif m.group('qstr') is not None:
            out.append(('s', m.group('qstr')))
        else:
            out.append(('b', m.group('bare')))
    return out


def _parse_value(toks, i):
    kind, tok = toks[i]
    if kind == 'p' and tok == '{':
        return _parse_block(toks, i + 1, closing=True)
    if kind == 'p' and tok == '[':
        items, i = [], i + 1
        while not (toks[i][0] == 'p' and toks[i][1] == ']'):
            if toks[i] == ('p', ','):
                i += 1
                continue
            v, i = _parse_value(toks, i)
            items.append(v)
        return items, i + 1
    if kind == 's':
        return tok, i + 1
    if kind == 'b':
        return _scalar(tok), i + 1
    raise ValueError(f'unexpected token {tok!r}')


def _parse_block(toks, i, closing):
    tree = ConfigTree()
    while i < len(toks):
        kind, tok = toks[i]
        if kind == 'p' and tok == '}':
            if not closing:
                raise ValueError("unbalanced '}'")
            return tree, i + 1
        if kind == 'p' and tok == ',':
            i += 1
            continue
        if kind == 'p':
            raise ValueError(f'unexpected {tok!r}')
        key = tok
        i += 1
        if i < len(toks) and toks[i][0] == 'p' and toks[i][1] in '=:':
            i += 1
        value, i = _parse_value(toks, i)
        node = tree
        parts = key.split('.')
        for part in parts[:-1]:
            node = node.setdefault(part, ConfigTree())
        if isinstance(value, dict) and isinstance(node.get(parts[-1]), dict):
            node[parts[-1]].update(value)
        else:
            dict.__setitem__(node, parts[-1], value)
    if closing:
        raise ValueError("missing '}'")
    return tree, i


def parse_string(text):
    tree, _ = _parse_block(_tokens(text), 0, closing=False)
    return tree


def parse_file(path, case=None):
    with open(path) as f:
        text = f.read()
    if case is not None:
        text = text.replace('CASE_NAME', case)          # nerf_runner.py:27-31
    return parse_string(text)
