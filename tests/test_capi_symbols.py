"""The C-ABI library loads (no GPU needed) and exports exactly what
include/zshmc.h declares; the ctypes table in zhusuan_amd/_capi.py mirrors the
header argument for argument."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'zshmc.h')


def _header_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'^\s*#.*$', '', src, flags=re.M)
    src = re.sub(r'typedef struct \w+ \{.*?\} \w+;', '', src, flags=re.S)
    out = {}
    for m in re.finditer(
            r'([A-Za-z_][\w\s\*]*?)\b(zshmc_\w+)\s*\(([^;{]*?)\)\s*;', src):
        args = m.group(3).strip()
        n = 0 if args in ('', 'void') else len(args.split(','))
        out[m.group(2)] = (m.group(1).strip(), n, args)
    return out


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as g
    g.build()
    from zhusuan_amd import _capi
    return _capi.load()


def test_adapt_link_layout_matches_header():
    """The ctypes mirror of zshmc_adapt_link has the header's fields in the
    header's order (and therefore its layout: both follow the C ABI)."""
    from zhusuan_amd import _capi
    src = open(HEADER).read()
    body = re.search(r'typedef struct zshmc_adapt_link \{(.*?)\} zshmc_adapt_link;',
                     src, re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        parts = [p.strip().lstrip('*') for p in decl.split(',')]
        names.append(parts[0].split()[-1].lstrip('*'))
        names.extend(parts[1:])
    assert names == [f[0] for f in _capi.AdaptLink._fields_], names
    assert ctypes.sizeof(_capi.AdaptLink) == 3 * 8 + 8 + 3 * 4 + 6 * 4 + 4 + 2 * 8


def test_header_declares_functions():
    fns = _header_functions()
    assert 'zshmc_hmc_diag_normal_step' in fns
    assert len(fns) >= 20


def test_every_declared_symbol_is_exported(lib):
    for name in _header_functions():
        assert hasattr(lib, name), name
        assert isinstance(getattr(lib, name), ctypes._CFuncPtr)


def test_ctypes_table_matches_header(lib):
    from zhusuan_amd import _capi
    fns = _header_functions()
    assert set(fns) == set(_capi.PROTOTYPES), (
        set(fns) ^ set(_capi.PROTOTYPES))
    for name, (ret, nargs, args) in fns.items():
        restype, argtypes = _capi.PROTOTYPES[name]
        assert len(argtypes) == nargs, (name, len(argtypes), nargs)
        # pointer / scalar kinds agree position by position
        for decl, ct in zip([a.strip() for a in args.split(',')] if nargs
                            else [], argtypes):
            is_ptr = '*' in decl
            ct_ptr = ct is ctypes.c_void_p or issubclass(ct, ctypes._Pointer)
            assert is_ptr == ct_ptr, (name, decl, ct)
            if 'zshmc_adapt_link' in decl:
                assert ct is ctypes.POINTER(_capi.AdaptLink), (name, decl)
            if not is_ptr:
                kinds = {'float': ctypes.c_float, 'int64_t': ctypes.c_int64,
                         'uint64_t': ctypes.c_uint64,
                         'uint32_t': ctypes.c_uint32, 'int': ctypes.c_int}
                base = decl.split()[0]
                assert kinds[base] is ct, (name, decl, ct)


def test_version_and_limits_callable_without_gpu(lib):
    assert lib.zshmc_version() == 600
    assert lib.zshmc_fused_max_n_data() == 2048
    assert lib.zshmc_last_error() is not None


def test_likelihood_plans(lib):
    """zshmc_likelihood_plan: the kernel width rows of n columns are padded to
    and the chains a workgroup takes -- what the host asks instead of keeping
    a table (zhusuan_amd/_ops.py)."""
    from zhusuan_amd import _ops
    got = [_ops.likelihood_plan(n) for n in
           (1, 64, 65, 129, 192, 193, 256, 257, 320, 321, 449, 512, 513, 576,
            577, 784, 896, 897, 1024)]
    assert got == [(64, 64), (64, 64), (128, 64), (192, 64), (192, 64),
                   (256, 64), (256, 64), (320, 64), (320, 64), (384, 64),
                   (512, 64), (512, 64), (576, 64), (576, 64), (640, 64),
                   (832, 64), (896, 64), (1024, 32), (1024, 32)]
    # a Categorical's classes must sit inside one wave's chain block
    assert _ops.likelihood_plan(300, 16) == (320, 64)
    assert _ops.likelihood_plan(300, 32) == (512, 32)
    assert _ops.likelihood_plan(200, 32) == (256, 64)
    assert _ops.likelihood_plan(600, 32) == (1024, 32)
    assert _ops.likelihood_plan(784, 16) == (832, 64)
    assert lib.zshmc_likelihood_plan(_ops.MAX_LIKELIHOOD_WIDTH + 1, 0, None,
                                     None) != 0
    assert lib.zshmc_likelihood_plan(0, 0, None, None) != 0


def test_bad_arguments_are_rejected_before_any_launch(lib):
    """Argument validation happens on the host, so it works without a GPU."""
    from zhusuan_amd import _capi
    rc = lib.zshmc_hmc_diag_normal_step(
        None, None, None, None, 0.1, 4, 4, 0, 1, 0, 0, 1, None, None,
        None, None, None, None, None, None)
    assert rc == 1
    assert 'null' in _capi.last_error()
    # a pending step-size update without a state block to apply it to
    link = _capi.AdaptLink(pending=_capi.PEND_ADAPT, n_chains_global=4)
    rc = lib.zshmc_hmc_diag_normal_step(
        ctypes.c_void_p(16), None, ctypes.c_void_p(16), None, 0.1, 4, 4, 0, 1,
        0, 0, 1, None, None, None, None, None, None, ctypes.byref(link), None)
    assert rc == 1 and 'step-size update needs' in _capi.last_error()
    rc = lib.zshmc_comm_all_reduce_sum(None, None, 0, None)
    assert rc == 1
    rc = lib.zshmc_state_set(ctypes.c_void_p(8), 99, 0.0, None)
    assert rc == 1 and 'out of range' in _capi.last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from zhusuan_amd import _capi
    monkeypatch.setattr(_capi, '_lib', None)
    monkeypatch.setattr(_capi, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_capi.LibraryMissing, match='no CPU fallback'):
        _capi.load()


def test_header_is_plain_c_and_cpp(tmp_path):
    """include/zshmc.h is the boundary: it must compile as C99 and as C++
    on its own (no HIP, no torch types in the signatures)."""
    import shutil
    import subprocess
    src = tmp_path / 'use_header.c'
    src.write_text('#include "zshmc.h"\n'
                   'int use(void) { zshmc_adapt_link l; l.pending = '
                   'ZSHMC_PEND_NONE; return (int)sizeof(l) + '
                   'ZSHMC_STATE_WORDS; }\n')
    inc = os.path.join(ROOT, 'include')
    for cc, std in (('gcc', '-std=c99'), ('g++', '-std=c++11')):
        if shutil.which(cc) is None:
            pytest.skip(cc + ' not available')
        args = [cc, std, '-Wall', '-Werror', '-pedantic', '-fsyntax-only',
                '-I' + inc]
        args += ['-x', 'c++' if cc == 'g++' else 'c', str(src)]
        r = subprocess.run(args, capture_output=True, text=True)
        assert r.returncode == 0, (cc, r.stderr[-1500:])


def test_model_plan_layout_matches_header(tmp_path):
    """The ctypes mirror of zshmc_model_plan against the C compiler's layout
    of the header's struct: size and the offset of every field."""
    import shutil
    import subprocess
    from zhusuan_amd import _capi
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    names = [f[0] for f in _capi.ModelPlan._fields_]
    src = tmp_path / 'layout.c'
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "zshmc.h"\n'
        'int main(void) {\n  printf("%zu\\n", sizeof(zshmc_model_plan));\n' +
        ''.join('  printf("%%zu\\n", offsetof(zshmc_model_plan, %s));\n' % n
                for n in names) + '  return 0;\n}\n')
    exe = tmp_path / 'layout'
    r = subprocess.run(['gcc', '-std=c99', '-I' + os.path.join(ROOT, 'include'),
                        str(src), '-o', str(exe)], capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    out = subprocess.run([str(exe)], capture_output=True, text=True).stdout
    vals = [int(v) for v in out.split()]
    assert vals[0] == ctypes.sizeof(_capi.ModelPlan)
    for n, off in zip(names, vals[1:]):
        assert getattr(_capi.ModelPlan, n).offset == off, n
