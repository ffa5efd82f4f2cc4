"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer runs of the host-side native code (GPU sanitizers are not
available on the pool, so this covers what runs on the host): the C pack planner of the library and the oracle's strict-order VQ."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ['-g', '-O1', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-fno-omit-frame-pointer']


def _run(cmd, exe, tmp_path):
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    if r.returncode != 0 and ('asan' in r.stderr.lower() or 'sanitize' in r.stderr.lower()) and 'cannot find' in r.stderr:
        pytest.skip('sanitizer runtime not installed: ' + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return r.stdout


@pytest.mark.skipif(shutil.which('g++') is None, reason='g++ not found')
def test_pack_planner_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / 'pack_plan_asan')
    out = _run(['g++', '-std=c++17'] + SAN + ['-I', 'include', '-I', 'vqnerf_release_amd/csrc', 'tests/native/pack_plan_asan.cpp',
                                               'vqnerf_release_amd/csrc/neus_pack_plan.cpp', 'vqnerf_release_amd/csrc/chain_pack_plan.cpp',
                                               'vqnerf_release_amd/csrc/error.cpp', '-o', exe],
               exe, tmp_path)
    assert 'shapes ok' in out


@pytest.mark.skipif(shutil.which('gcc') is None, reason='gcc not found')
def test_oracle_vq_strict_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / 'vq_strict_asan')
    out = _run(['gcc'] + SAN + ['-ffp-contract=off', '-mfma', 'tests/native/vq_strict_asan.c', 'oracle/vq_strict.c', '-lm', '-o', exe],
               exe, tmp_path)
    assert 'runs ok' in out
