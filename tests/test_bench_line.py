"""CPU: the ONE stdout line of bench.py (driver contract).  Round 4's line had grown to 22 KB of prose and the driver could not capture it
(BENCH_r04.json: parsed = null); the line is now built by bench.compact_line from the full report and must stay under 8 KB, carry the
contract keys + `roofline` + `cpu_baseline`, and round-trip through JSON.  The canned report is round 4's own full result
(profiles/r04_bench.json, 22 KB)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
            'data', 'config', 'roofline', 'cpu_baseline')


def _canned():
    with open(os.path.join(ROOT, 'profiles', 'r04_bench.json')) as f:
        return json.load(f)


def test_line_is_short_complete_and_round_trips():
    full = _canned()
    assert len(json.dumps(full)) > 20000                         # the report that did not parse
    text = bench.compact_line(full, detail_path='gpurun_out/bench_detail.json')
    assert '\n' not in text and len(text.encode()) < 8192
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line['value'] == float(f"{full['value']:.7g}") and line['unit'] == 'rays/s' and line['dtype'] == 'f32'
    assert set(line['config']) == {'workload', 'rays_per_step_per_gpu', 'parallelism'} and 'model' not in line['config']
    rf = line['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'kernel'):
        assert k in rf, k
    assert abs(rf['frac'] - full['roofline']['frac']) < 1e-5 and abs(rf['achieved'] / rf['peak'] - rf['frac']) < 1e-4
    cb = line['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cb, k
    assert cb['kind'] == 'port' and cb['cores'] == 16 and len(cb['sample']) <= 120
    # prose is gone from every level; numbers of the secondary legs stay
    def walk(o):
        if isinstance(o, dict):
            for k, v in o.items():
                assert k not in ('note', 'traffic_note', 'frac_note', 'peak_note', 'windows_ms_per_step'), k
                walk(v)
        else:
            assert not isinstance(o, list)
            assert not (isinstance(o, str) and len(o) > 200), o
    walk(line)
    assert line['extra']['decomp_train_graph']['ms_per_step'] == float(f"{full['extra']['decomp_train_graph']['ms_per_step']:.5g}")
    assert line['extra']['dp_train']['n_ranks_seen'] == 1
    assert line['detail'] == 'gpurun_out/bench_detail.json'


def test_extra_is_truncated_never_the_headline():
    full = _canned()
    for i in range(400):                                         # an `extra` that would blow any limit
        full['extra'][f'leg_{i}'] = {'ms_per_step': 1.0 + i, 'frac': 0.5, 'inner': {'a': 1.25, 'b': 2.5}}
    text = bench.compact_line(full)
    assert len(text.encode()) < 8192
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert 'leg_0' in line['extra_dropped'] and line['roofline']['frac'] > 0.7 and line['cpu_baseline']['value'] > 300
    # a tiny limit: extra goes completely, the contract keys stay
    text = bench.compact_line(_canned(), limit=2300)
    line = json.loads(text)
    assert len(text) <= 2300 and 'extra' not in line
    for k in CONTRACT:
        assert k in line, k


def test_non_finite_numbers_do_not_break_the_line():
    full = _canned()
    full['extra']['geo_train']['rays_per_s'] = float('nan')
    full['psnr_vs_oracle_db'] = float('inf')
    line = json.loads(bench.compact_line(full))                  # strict JSON: no NaN / Infinity tokens
    assert 'NaN' not in json.dumps(line) and line['psnr_vs_oracle_db'] is None


def test_emit_writes_sidecar_and_prints_one_line(tmp_path, capsys, monkeypatch):
    side = tmp_path / 'detail.json'
    monkeypatch.setenv('VQN_BENCH_DETAIL', str(side))
    bench.emit(_canned())
    out = capsys.readouterr().out
    assert out.count('\n') == 1 and len(out) < 8192
    with open(side) as f:
        detail = json.load(f)
    assert 'traffic_note' in detail['roofline'] and 'note' in detail['extra']['geo_train']      # the prose lives here
    assert json.loads(out)['detail'] == str(side)


def test_self_launch_starts_ranks_as_a_child_before_any_gpu_call(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start `python -m torch.distributed.run ... bench.py --gpus 2` as a CHILD and
    leave with its exit code.  Here (no GPU) each rank dies on bench.py's `needs an MI355X` assertion: the parent must relay a non-zero
    code, and must itself never have asked for the GPU (it would die on the same assertion before launching anything)."""
    env = dict(os.environ, VQN_BENCH_BACKEND='gloo', VQN_BENCH_DETAIL=str(tmp_path / 'd.json'))
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-extras'],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert 'launching 2 ranks as a child torch.distributed.run' in r.stderr, r.stderr[-2000:]
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
        assert r.stderr.count('bench.py needs an MI355X') >= 2, r.stderr[-3000:]      # BOTH ranks were started and got as far as the GPU check


def test_world_size_mismatch_is_still_an_error():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr
