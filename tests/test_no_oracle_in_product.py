"""The oracle is test infrastructure: nothing under zhusuan_amd/ may import,
call or execute it, and the product has no CPU fallback path."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_never_touches_oracle():
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, 'zhusuan_amd')):
        for f in files:
            if not f.endswith(('.py', '.hip', '.h', '.cpp')):
                continue
            src = open(os.path.join(dp, f)).read()
            if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M) or \
                    'oracle/' in src.replace('oracle/philox.py', '').replace(
                        'oracle/__init__', ''):
                bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_bench_uses_oracle_only_in_cpu_baseline():
    src = open(os.path.join(ROOT, 'bench.py')).read()
    uses = [m.start() for m in re.finditer(r'from oracle', src)]
    # the NumPy, torch-CPU and C + OpenMP ports: all inside the CPU-baseline
    # functions only, which sit in front of everything that runs the product
    assert len(uses) == 3
    start = src.index('def cpu_baseline')
    end = src.index('def _tuned_start')
    assert all(start < u < end for u in uses)
    for fn in ('cpu_baseline_numpy', 'cpu_baseline_torch',
               'cpu_baseline_parallel'):
        body = src[src.index('def ' + fn):]
        body = body[:body.index('\ndef ', 1)]
        assert 'from oracle' in body, fn
