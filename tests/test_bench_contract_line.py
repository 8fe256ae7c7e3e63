"""bench.py's output contract (BENCH_r05.json had
`parsed: null` because the one stdout line had grown to ~20 KB).

The LAST stdout line is the contract object: <= 4 096 bytes, strict JSON (no
bare NaN / Infinity), the contract keys + a trimmed `roofline` +
`cpu_baseline`; it is the only stdout line that starts with `{`.  Everything
else travels on `#bench-extra` / `#bench-detail` lines before it and in
bench_extras.json.  Checked on records of real runs (profiles/), on the
`--workload lntm` and N > 1 shapes of the line, with non-finite values and
with more extras than fit; and the N > 1 guard that prints the headline when
the sharded extra stalls or the launcher sends SIGTERM.
"""
import io
import json
import os
import signal
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                             # noqa: E402

RECORD = os.path.join(ROOT, 'profiles', 'r05x_bench_driver_flags.json')
CONTRACT_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup',
                 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                 'dtype', 'data', 'config')


def _no_constants(name):
    raise AssertionError('non-strict JSON constant %s in the line' % name)


def _parse(text):
    return json.loads(text, parse_constant=_no_constants)


def _record():
    with open(RECORD) as f:
        return json.load(f)


def _emit(out, tmp_path):
    stream = io.StringIO()
    path = str(tmp_path / 'bench_extras.json')
    bench._emitted = False               # (emit prints once per process)
    text = bench.emit(out, stream=stream, extras_file=path)
    bench._emitted = False
    lines = stream.getvalue().splitlines()
    assert lines[-1] == text
    assert [l for l in lines if l.startswith('{')] == [text]
    assert len(text.encode()) <= bench.CONTRACT_MAX_BYTES == 4096
    with open(path) as f:
        full = _parse(f.read())
    return _parse(text), lines, full


def test_headline_line_of_a_recorded_run(tmp_path):
    out = _record()
    assert len(json.dumps(out)) > 15000          # the line that did not parse
    line, lines, full = _emit(out, tmp_path)
    for k in CONTRACT_KEYS:
        assert k in line, k
    assert line['value'] == out['value'] and line['n_gpus'] == 1
    assert line['config']['workload'].startswith('configs[1]')
    roof = line['roofline']
    assert roof['bound'] == 'hbm' and roof['peak'] == 8000.0
    assert roof['frac'] == pytest.approx(roof['achieved'] / roof['peak'])
    assert roof['achieved'] == pytest.approx(
        roof['algorithmic_bytes_per_launch'] / (roof['kernel_ms'] * 1e-3) / 1e9)
    assert roof['traffic'] == out['roofline']['traffic']
    assert 'other_generator' not in roof and 'kernel_timing' not in roof
    cpu = line['cpu_baseline']
    assert cpu['kind'] == 'port' and cpu['cores'] >= 1 and cpu['value'] > 0
    assert 'sample' in cpu
    # one short entry per extra configuration; the full ones are elsewhere
    assert len(line['extras']) == len(out['extra_configs'])
    assert all(len(json.dumps(e)) < 260 for e in line['extras'])
    assert full['extra_configs'] == out['extra_configs']
    detail = [l for l in lines if l.startswith(bench.DETAIL_PREFIX)]
    assert len(detail) == 1
    d = _parse(detail[0][len(bench.DETAIL_PREFIX):])
    assert d['mass_adaptation_modes'] == out['mass_adaptation_modes']
    assert d['roofline']['other_generator'] == out['roofline']['other_generator']


def test_recorded_reference_timing_is_present_and_loud_when_missing(capsys):
    ref = bench.cpu_reference_recorded()
    assert ref['kind'] == 'reference' and ref['value'] > 0
    assert 'build container' in ref['measured_on']
    c1 = bench._recorded('cpu_reference_over_shim_config1.json')
    assert c1['value'] > 0
    gone = bench._recorded('no_such_record.json')
    assert 'unreadable' in gone['error']
    assert 'no_such_record.json' in capsys.readouterr().err


def test_non_finite_values_become_null(tmp_path):
    out = _record()
    out['roofline']['traffic'] = float('inf')
    out['mean_acceptance'] = float('nan')
    out['extra_configs'][1]['ms_per_step'] = float('-inf')
    import numpy as np
    out['step_size'] = np.float32(0.125)
    line, lines, full = _emit(out, tmp_path)
    assert line['roofline']['traffic'] is None
    assert line['mean_acceptance'] is None and line['step_size'] == 0.125
    assert line['extras'][1]['ms_per_step'] is None
    for l in lines:                       # every line is strict, not only the last
        body = l[l.index('{'):]
        _parse(body)


def test_lntm_workload_line(tmp_path):
    r = dict(_record()['extra_configs'][2])
    r.setdefault('rccl_ranks', 0)
    out = bench.lntm_line_record(r, 1, 3, 1, 'no collective')
    line, _, _ = _emit(out, tmp_path)
    for k in CONTRACT_KEYS:
        assert k in line, k
    assert line['plan'] == 'mixture_multinomial'
    assert line['roofline']['bound'] == 'mfma' and line['roofline']['frac'] > 0
    assert line['roofline']['peak'] == 157.3
    assert 'method' not in line['ess'] and line['ess']['ess_per_sec'] > 0
    assert len(line['config']['workload']) <= 200


def test_line_at_eight_ranks(tmp_path):
    out = _record()
    for k in ('cpu_baseline', 'cpu_baseline_torch_all_threads',
              'cpu_baseline_numpy_1core', 'cpu_reference_over_shim'):
        out.pop(k)
    out.update(n_gpus=8, rccl_ranks=8, collective='RCCL over xGMI',
               allreduce_latency_us={'min_over_ranks': 11.2,
                                     'max_over_ranks': 13.9, 'what': 'x' * 200},
               strong_scaling={'n_chains_total': 65536, 'chains_per_gpu': 8192,
                               'ms_per_step': 0.031, 'value': 2.1e10,
                               'kernel': 'k' * 60, 'adaptation': 'on'})
    sharded = dict(out['extra_configs'][2], n_gpus=8, id='configs[4] sharded')
    out['extra_configs'] = [sharded]
    line, _, _ = _emit(out, tmp_path)
    assert line['n_gpus'] == 8 and line['rccl_ranks'] == 8
    assert 'cpu_baseline' not in line
    assert line['strong_scaling']['chains_per_gpu'] == 8192
    assert line['allreduce_latency_us'] == {'min_over_ranks': 11.2,
                                            'max_over_ranks': 13.9}
    (e,) = line['extras']
    assert e['id'] == 'configs[4] sharded' and e['n_gpus'] == 8
    assert e['frac'] > 0 and e['ms_per_step'] > 0


def test_too_many_extras_shed_optional_keys_not_the_contract(tmp_path):
    out = _record()
    out['extra_configs'] = out['extra_configs'] * 8
    out['config']['workload'] = 'w' * 3000
    line, _, full = _emit(out, tmp_path)
    assert 'extras' not in line
    for k in CONTRACT_KEYS + ('roofline', 'cpu_baseline'):
        assert k in line
    assert len(full['extra_configs']) == len(out['extra_configs'])


def test_emit_prints_once(tmp_path):
    out = _record()
    stream = io.StringIO()
    bench._emitted = False
    assert bench.emit(out, stream=stream, extras_file=None) is not None
    assert bench.emit(out, stream=stream, extras_file=None) is None
    assert len([l for l in stream.getvalue().splitlines()
                if l.startswith('{')]) == 1
    bench._emitted = False


GUARDED = r'''
import ctypes, json, os, sys, time
sys.path.insert(0, %(root)r)
import bench
bench.EXTRAS_FILE = %(extras)r
with open(%(record)r) as f:
    out = json.load(f)
out['n_gpus'] = 2
out['extra_configs'] = []
mode = sys.argv[1]
disarm = bench.guard_headline(out, int(sys.argv[2]),
                              1 if mode == 'timeout' else 300)
print('#armed', flush=True)
if mode == 'finish':
    disarm()
    out['extra_configs'] = [{'id': 'configs[4] sharded', 'ms_per_step': 1.0}]
    bench.emit(out, extras_file=bench.EXTRAS_FILE)
    sys.exit(0)
# the "sharded extra": a foreign call that holds the main thread (no bytecode
# boundary for a Python-level signal handler to run at); sleep(3) returns
# early when a signal arrives, a HIP call would not
libc = ctypes.CDLL(None)
for _ in range(4):
    libc.sleep(60)
print('#never', flush=True)
'''


def _guarded(tmp_path, mode, rank=0):
    script = tmp_path / 'guarded.py'
    script.write_text(GUARDED % dict(root=ROOT, record=RECORD,
                                     extras=str(tmp_path / 'extras.json')))
    return subprocess.Popen([sys.executable, str(script), mode, str(rank)],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            universal_newlines=True)


def _contract_of(stdout):
    lines = stdout.splitlines()
    json_lines = [l for l in lines if l.startswith('{')]
    assert len(json_lines) == 1 and lines[-1] == json_lines[0], lines[-3:]
    return _parse(json_lines[0])


def test_guard_prints_the_headline_on_sigterm(tmp_path):
    p = _guarded(tmp_path, 'sigterm')
    assert p.stdout.readline().strip() == '#armed'
    time.sleep(0.3)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 0, err[-800:]
    assert '#never' not in out
    line = _contract_of(out)
    assert line['n_gpus'] == 2 and line['roofline']['frac'] > 0
    assert 'SIGTERM' in line['extras'][0]['error']


def test_guard_prints_the_headline_on_timeout(tmp_path):
    p = _guarded(tmp_path, 'timeout')
    out, err = p.communicate(timeout=60)
    assert p.returncode == 0, err[-800:]
    line = _contract_of(out)
    assert 'did not finish' in line['extras'][0]['error']


def test_guard_on_other_ranks_exits_quietly(tmp_path):
    p = _guarded(tmp_path, 'sigterm', rank=1)
    assert p.stdout.readline().strip() == '#armed'
    time.sleep(0.3)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 0 and '{' not in out


def test_disarmed_guard_leaves_the_normal_path_alone(tmp_path):
    p = _guarded(tmp_path, 'finish')
    out, err = p.communicate(timeout=60)
    assert p.returncode == 0, err[-800:]
    line = _contract_of(out)
    assert line['extras'] == [{'id': 'configs[4] sharded', 'ms_per_step': 1.0}]


def test_foreign_writes_to_stdout_end_up_on_stderr(tmp_path):
    """gloo's C++ layer prints "[Gloo] Rank 0 is connected to ..." on file
    descriptor 1 of every rank: bench.py keeps a private duplicate of the
    descriptor for its own lines and points fd 1 at stderr."""
    script = tmp_path / 'claim.py'
    script.write_text(r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import bench
with open(%(record)r) as f:
    out = json.load(f)
bench.claim_stdout()
os.write(1, b'[Gloo] Rank 0 is connected to 7 peer ranks.\n')
print('a library greets the user')
bench.emit_extra({'id': 'configs[0]', 'ms_per_step': 1.0})
os.write(1, b'more chatter\n')
bench.emit(out, extras_file=None)
os.write(1, b'and after the line\n')
''' % dict(root=ROOT, record=RECORD))
    p = subprocess.run([sys.executable, str(script)], capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-800:]
    lines = p.stdout.splitlines()
    assert [l[:13] for l in lines[:2]] == ['#bench-extra ', '#bench-detail']
    assert len(lines) == 3 and lines[2].startswith('{"metric"')
    for noise in ('[Gloo] Rank 0', 'a library greets', 'more chatter',
                  'and after the line'):
        assert noise in p.stderr and noise not in p.stdout
