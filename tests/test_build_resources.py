"""The LDS-DMA ring kernel (zhusuan_amd/csrc/hmc_fused_ring.hip) guards its
ring slots with hand-counted `s_waitcnt vmcnt(N)`.  That ledger is only valid
if the compiler adds no VMEM of its own inside the kernel: no scratch spills
(spill code is scratch_load/scratch_store = VMEM) for ANY instantiation.
hipcc cross-compiles for gfx950 without a GPU, so this runs on CPU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'zhusuan_amd', 'csrc', 'hmc_fused_ring.hip')


N_RING_KERNELS = 2 * (6 * 4 + 3 + 2) + 2 * 6 * 4   # + the COLSTATS variants


def _hipcc():
    for c in ('/opt/rocm/bin/hipcc', shutil.which('hipcc')):
        if c and os.path.exists(c):
            return c
    pytest.skip('hipcc not available')


@pytest.fixture(scope='module')
def ring_build(tmp_path_factory):
    out = tmp_path_factory.mktemp('ringasm')
    import __graft_entry__ as ge
    cmd = [_hipcc()] + ge.HIPCC_FLAGS + [
        '-c', SRC, '-save-temps', '-Rpass-analysis=kernel-resource-usage',
        '-o', str(out / 'ring.o')]
    p = subprocess.run(cmd, cwd=str(out), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, universal_newlines=True)
    assert p.returncode == 0, p.stdout[-4000:]
    asm = [f for f in os.listdir(str(out)) if f.endswith('gfx950.s')]
    assert asm, os.listdir(str(out))
    return p.stdout, open(os.path.join(str(out), asm[0])).read()


def _kernels(remarks):
    cur, table = None, {}
    for line in remarks.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = m.group(1)
            table[cur] = {}
            continue
        m = re.search(r'    ([A-Za-z \[\]/]+): (\d+)', line)
        if m and cur:
            table[cur][m.group(1).strip()] = int(m.group(2))
    return table


def test_ring_kernels_have_no_spills(ring_build):
    remarks, _ = ring_build
    table = {k: v for k, v in _kernels(remarks).items()
             if 'hmc_diag_normal_ring_kernel' in k}
    # NCH = 1..8 x {mass, no mass} x {mean tile, zero mean}, minus <8, mass>
    # and <7, mass, mean tile> (register-prefetch path: they do not fit 256
    # VGPRs), each with the per-chain scalars staged in LDS or not
    assert len(table) == N_RING_KERNELS, sorted(table)
    for name, row in table.items():
        assert row['VGPRs Spill'] == 0, (name, row)
        # (SGPR spills live in VGPR lanes -- v_writelane -- not in memory)
        assert row['ScratchSize [bytes/lane]'] == 0, (name, row)


def test_ring_trip_loop_has_only_hand_counted_vmem(ring_build):
    """Inside each ring kernel every global_load_lds / row store / HMCInfo
    store must come from an inline-asm statement (between ASMSTART/ASMEND);
    compiler-issued VMEM is allowed only in the prologue (parameter loads)
    and the epilogue (atomics)."""
    _, asm = ring_build
    n_kernels = 0
    for m in re.finditer(r'^(_ZN5zshmc27hmc_diag_normal_ring_kernel\w+):[^\n]*\n(.*?)s_endpgm',
                         asm, re.S | re.M):
        n_kernels += 1
        body = m.group(2)
        # (buffer_wbl2 / buffer_inv are the cache write-back / invalidate of the
        # epilogue's release / acquire fences -- link_retire -- not memory
        # operations on the vmcnt ledger)
        assert 'scratch_' not in body, m.group(1)
        assert not re.search(r'buffer_(load|store|atomic)', body), m.group(1)
        in_asm = False
        first_dma = last_store = None
        stray = []
        for i, line in enumerate(body.splitlines()):
            if '#ASMSTART' in line:
                in_asm = True
            elif '#ASMEND' in line:
                in_asm = False
            op = line.strip().split(' ')[0].split('\t')[0]
            if op.startswith('global_load_lds'):
                assert in_asm, (m.group(1), line)
                first_dma = i if first_dma is None else first_dma
            if op.startswith('global_store') and in_asm:
                last_store = i
            if op.startswith(('global_', 'flat_')) and not in_asm:
                stray.append((i, op))
        assert first_dma is not None and last_store is not None
        # compiler VMEM may only sit before the first DMA or after the last
        # hand-written store (epilogue atomics)
        for i, op in stray:
            assert i < first_dma or i > last_store, (m.group(1), i, op)
    assert n_kernels == N_RING_KERNELS


# ---------------------------------------------------------------------------
# The two-GEMM likelihood kernel (csrc/linear_bernoulli.hip) writes its tile
# loop as asm statements in issue order.  What this guards (hipcc pinned by the
# image, but the properties are the contract the hand-placed waits rely on):
# no instantiation spills (spill code is VMEM the waits do not count); the
# occupancy each width is designed for; every MFMA of a kernel comes out of an
# asm statement (the compiler schedules none of its own); and the hot
# instantiation -- D = 256, gradient only -- moves no value between AGPRs and
# VGPRs inside the tile loop.
LB_SRC = os.path.join(ROOT, 'zhusuan_amd', 'csrc', 'linear_bernoulli.hip')


@pytest.fixture(scope='module')
def lb_build(tmp_path_factory):
    out = tmp_path_factory.mktemp('lbasm')
    import __graft_entry__ as ge
    cmd = [_hipcc()] + ge.HIPCC_FLAGS + [
        '-c', LB_SRC, '-save-temps', '-Rpass-analysis=kernel-resource-usage',
        '-o', str(out / 'lb.o')]
    p = subprocess.run(cmd, cwd=str(out), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, universal_newlines=True)
    assert p.returncode == 0, p.stdout[-4000:]
    asm = [f for f in os.listdir(str(out)) if f.endswith('gfx950.s')]
    assert asm, os.listdir(str(out))
    return p.stdout, open(os.path.join(str(out), asm[0])).read()


def test_likelihood_kernels_have_no_spills_and_keep_their_occupancy(lb_build):
    remarks, _ = lb_build
    table = {k: v for k, v in _kernels(remarks).items()
             if 'linear_bernoulli_kernelILi' in k}
    # 4 widths x 3 element-wise stages x {ll+grad, grad only, ll only}
    assert len(table) == 36, sorted(table)
    for name, row in table.items():
        assert row['VGPRs Spill'] == 0, (name, row)
        assert row['ScratchSize [bytes/lane]'] == 0, (name, row)
        width = int(re.search(r'kernelILi(\d+)E', name).group(1))
        want = {256: 1, 192: 1, 128: 2, 64: 3}[width]
        assert row['Occupancy [waves/SIMD]'] >= want, (name, row)


def _issue_slots_between_mfmas(body):
    """Issue slots (instructions; `s_nop n` counts n + 1) between consecutive
    MFMAs of a kernel body, and from the loop header to the first MFMA."""
    lines = [l.split(';')[0].strip() for l in body.splitlines()]
    lines = [l for l in lines if l and not l.startswith('.') and
             not l.endswith(':')]

    def cost(l):
        m = re.match(r's_nop (\d+)', l)
        return 1 + int(m.group(1)) if m else 1
    gaps, run, seen = [], 0, False
    for l in lines:
        if l.startswith('v_mfma'):
            if seen:
                gaps.append(run)
            seen, run = True, 0
        elif seen:
            run += cost(l)
    return gaps


def test_likelihood_tile_loop_is_hand_ordered(lb_build):
    _, asm = lb_build
    n = 0
    for m in re.finditer(r'^(_ZN5zshmc23linear_bernoulli_kernelILi(\d+)ELb(\d)'
                         r'ELi(\d)ELb(\d)E\w+):[^\n]*\n(.*?)s_endpgm',
                         asm, re.S | re.M):
        n += 1
        name, width, grad, op, ll, body = m.groups()
        in_asm, mfma_out, mfma_in = False, 0, 0
        for line in body.splitlines():
            if '#ASMSTART' in line:
                in_asm = True
            elif '#ASMEND' in line:
                in_asm = False
            if line.strip().startswith('v_mfma'):
                if in_asm:
                    mfma_in += 1
                else:
                    mfma_out += 1
        assert mfma_out == 0, name
        # one copy of the tile: D/2 MFMAs of phase 1 (+ D/2 of phase 3)
        assert mfma_in == int(width) // 2 * (2 if grad == '1' else 1), (
            name, mfma_in)
        if (width, grad, op, ll) == ('256', '1', '0', '0'):
            lines = body.splitlines()
            head = next(i for i, l in enumerate(lines) if 'Loop Header' in l)
            first = next(i for i in range(head, len(lines))
                         if lines[i].strip().startswith('v_mfma'))
            last = max(i for i, l in enumerate(lines)
                       if l.strip().startswith('v_mfma'))
            moved = [l for l in lines[first:last] if 'v_accvgpr' in l]
            assert not moved, moved[:4]
    assert n == 36


# The 16-chain-block kernel (csrc/linear_bernoulli_mid.hip, widths 320 .. 896):
# the same contract -- no scratch, every MFMA from an asm statement.
MID_SRC = os.path.join(ROOT, 'zhusuan_amd', 'csrc', 'linear_bernoulli_mid.hip')


@pytest.fixture(scope='module')
def mid_build(tmp_path_factory):
    out = tmp_path_factory.mktemp('midasm')
    import __graft_entry__ as ge
    cmd = [_hipcc()] + ge.HIPCC_FLAGS + [
        '-c', MID_SRC, '-save-temps', '-Rpass-analysis=kernel-resource-usage',
        '-o', str(out / 'mid.o')]
    p = subprocess.run(cmd, cwd=str(out), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, universal_newlines=True)
    assert p.returncode == 0, p.stdout[-4000:]
    asm = [f for f in os.listdir(str(out)) if f.endswith('gfx950.s')]
    assert asm, os.listdir(str(out))
    return p.stdout, open(os.path.join(str(out), asm[0])).read()


def test_mid_likelihood_kernels_have_no_spills_and_hand_ordered_mfmas(mid_build):
    remarks, asm = mid_build
    table = {k: v for k, v in _kernels(remarks).items()
             if 'linear_bernoulli_mid_kernelILi' in k}
    # 10 widths x 3 element-wise stages x {ll+grad, grad only, ll only}
    assert len(table) == 90, len(table)
    for name, row in table.items():
        assert row['VGPRs Spill'] == 0, (name, row)
        assert row['ScratchSize [bytes/lane]'] == 0, (name, row)
    n = 0
    for m in re.finditer(r'^(_ZN5zshmc27linear_bernoulli_mid_kernelILi(\d+)ELb(\d)'
                         r'ELi(\d)ELb(\d)E\w+):[^\n]*\n(.*?)s_endpgm',
                         asm, re.S | re.M):
        n += 1
        name, width, grad, op, ll, body = m.groups()
        in_asm, mfma_out, mfma_in = False, 0, 0
        for line in body.splitlines():
            if '#ASMSTART' in line:
                in_asm = True
            elif '#ASMEND' in line:
                in_asm = False
            if line.strip().startswith('v_mfma'):
                if in_asm:
                    mfma_in += 1
                else:
                    mfma_out += 1
        assert mfma_out == 0, name
        # D/4 MFMAs of phase 1 (+ D/4 of phase 3), one copy of the tile
        assert mfma_in == int(width) // 4 * (2 if grad == '1' else 1), (
            name, mfma_in)
    assert n == 90


def test_likelihood_steps_fit_their_issue_slots(lb_build, mid_build):
    """DESIGN 3.3, rules (a) and (b): a wave issues one instruction per ~4
    clocks and a 32 x 32 x 2 MFMA holds the pipe for 64 (16 x 16 x 4: 32), so a
    gap between two MFMAs takes ~15 (~7) other instructions for free.  In the
    gradient-only instantiations no gap of GEMM 1 exceeds that (the steps'
    LDS reads, DMA instructions and address arithmetic are spread), and GEMM 2's
    only long gaps are the element-wise slots, the tile barrier and the three
    parts of the next tile's state."""
    _, asm = lb_build
    for width in (64, 128, 192, 256):
        m = re.search(r'^_ZN5zshmc23linear_bernoulli_kernelILi%dELb1ELi0ELb0E'
                      r'\w+:[^\n]*\n(.*?)s_endpgm' % width, asm, re.S | re.M)
        gaps = _issue_slots_between_mfmas(m.group(1))
        n1 = width // 2            # MFMAs of GEMM 1; gaps[n1 - 1] is the drain
        gemm1, gemm2 = gaps[:n1 - 1], gaps[n1:2 * n1 - 1]
        assert max(gemm1) <= 18, (width, gemm1)
        assert sorted(gemm1)[-2] <= 16 and sum(gemm1) / len(gemm1) <= 9, (
            width, gemm1)
        # GEMM 2: three element-wise slots, the tile barrier, three state parts
        assert sum(g > 12 for g in gemm2) <= 7, (width, gemm2)
    _, asm = mid_build
    for width in (320, 512, 896):
        m = re.search(r'^_ZN5zshmc27linear_bernoulli_mid_kernelILi%dELb1ELi0ELb0E'
                      r'\w+:[^\n]*\n(.*?)s_endpgm' % width, asm, re.S | re.M)
        gaps = _issue_slots_between_mfmas(m.group(1))
        n1 = width // 4
        gemm1, gemm2 = gaps[:n1 - 1], gaps[n1:2 * n1 - 1]
        assert max(gemm1) <= 10, (width, gemm1)
        assert sum(g > 8 for g in gemm2) <= 5, (width, gemm2)


def _tile_loop_top(body):
    """Instructions from the tile loop's header (the target of the backward
    branch behind the last MFMA) to the first MFMA."""
    lines = body.splitlines()
    mf = [i for i, l in enumerate(lines) if l.strip().startswith('v_mfma')]
    labels = {l.split(':')[0].strip(): i for i, l in enumerate(lines)
              if re.match(r'^\.LBB\d+_\d+:', l)}
    targets = []
    for l in lines[mf[-1]:mf[-1] + 40]:
        m = re.match(r'\s*(s_c?branch\w*)\s+(\.LBB\d+_\d+)', l)
        # (branches into the loop's own top: a few hundred lines at most in
        # front of its first MFMA; the scan may run into later blocks whose
        # branches go elsewhere)
        if m and mf[0] - 400 < labels.get(m.group(2), 1 << 30) < mf[0]:
            targets.append(labels[m.group(2)])
            if m.group(1) == 's_branch':     # the back edge itself
                break
    assert targets
    header = min(targets)
    out = [l.split(';')[0].strip() for l in lines[header:mf[0]]]
    return [l for l in out if l and not l.startswith('.') and
            not l.endswith(':')]


def test_nothing_heavy_sits_between_two_tiles(lb_build, mid_build):
    """DESIGN 3.3, rule (b): a tile's state is advanced by additions inside
    the tile before.  Between the last MFMA of a tile and the first of the
    next there is no 64-bit multiply and no 64-bit compare on the vector unit
    (what rebuilding the state from the tile index costs), and the whole top
    of the loop -- the ragged-tile path included -- stays short."""
    cases = [(lb_build[1], '_ZN5zshmc23linear_bernoulli_kernelILi%dELb1ELi%dELb0E', w, op)
             for w in (64, 128, 256) for op in (0, 1)]
    cases += [(mid_build[1], '_ZN5zshmc27linear_bernoulli_mid_kernelILi%dELb1ELi%dELb0E',
               w, op) for w in (320, 896) for op in (0, 1)]
    for asm, pat, width, op in cases:
        m = re.search(r'^' + pat % (width, op) + r'\w+:[^\n]*\n(.*?)s_endpgm',
                      asm, re.S | re.M)
        top = _tile_loop_top(m.group(1))
        heavy = [l for l in top if re.match(
            r'(s_mul_hi_u32|v_mul_hi_u32|v_cmp_\w+_[iu]64|v_mad_u64_u32)', l)]
        assert not heavy, (width, op, heavy)
        assert len(top) <= 90, (width, op, len(top))


# -- the bf16x3 likelihood kernel (csrc/b3_kernel.h) ---------------------------
# two translation units: Bernoulli / multinomial, and the Categorical family
# (one kernel per class stride)
def _b3_compile(out, name):
    import __graft_entry__ as ge
    src = os.path.join(ROOT, 'zhusuan_amd', 'csrc', name + '.hip')
    cmd = [_hipcc()] + ge.HIPCC_FLAGS + [
        '-c', src, '-save-temps', '-Rpass-analysis=kernel-resource-usage',
        '-o', str(out / (name + '.o'))]
    return subprocess.Popen(cmd, cwd=str(out), stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, universal_newlines=True)


@pytest.fixture(scope='module')
def b3_build(tmp_path_factory):
    out = tmp_path_factory.mktemp('b3asm')
    procs = [_b3_compile(out, n) for n in ('linear_bf16x3',
                                           'linear_bf16x3_cat')]
    remarks = ''
    for p in procs:
        text = p.communicate()[0]
        assert p.returncode == 0, text[-4000:]
        remarks += text
    asm = ''.join(open(os.path.join(str(out), f)).read()
                  for f in sorted(os.listdir(str(out)))
                  if f.endswith('gfx950.s'))
    return remarks, asm


def _b3_bodies(asm):
    """name -> text of each linear_b3_kernel<D, OP, LL, NACC, GL, PK, SP>."""
    out = {}
    for m in re.finditer(r'^(_ZN5zshmc16linear_b3_kernelILi\d+ELi\dELb[01]E'
                         r'Li\dELi\dELb[01]ELb[01]EEE\w+):[^\n]*\n(.*?)s_endpgm', asm,
                         re.S | re.M):
        out[m.group(1)] = m.group(2)
    return out


def test_bf16x3_kernels_keep_their_registers_and_occupancy(b3_build):
    """Two waves per SIMD (two workgroups per CU) at <= 128 columns, one
    above; whatever hipcc parks in scratch is parked AROUND the tile loop
    (epilogue addresses), never inside it: the loop's DMA waits are
    `s_waitcnt vmcnt(0)` and a scratch access would sit in the same queue."""
    remarks, asm = b3_build
    table = {k: v for k, v in _kernels(remarks).items()
             if 'linear_b3_kernelILi' in k}
    # 4 widths x {ll+grad, grad only} x (Bernoulli, multinomial, the
    # Categorical at class strides 1 .. 32) + the packed-rows multinomial
    # (PK: 64 / 128 / 192 columns; 48 KB of counts in the LDS: one workgroup
    # per CU, so one wave per SIMD is its register budget -- and no spills)
    # + the own-vocabulary multinomial (SP: all four widths, the occupancy
    # of the dense form)
    assert len(table) == 8 * (2 + 6) + 6 + 8, sorted(table)
    for name, row in table.items():
        width = int(re.search(r'kernelILi(\d+)E', name).group(1))
        packed = re.search(r'ELb1ELb0EEE', name) is not None
        if packed:
            assert re.search(r'kernelILi\d+ELi1ELb', name) and width <= 192, \
                name                                        # OP 1 only
            assert row['VGPRs Spill'] == 0, (name, row)
        assert row['Occupancy [waves/SIMD]'] >= (
            2 if width <= 128 and not packed else 1), (name, row)
        op, ll = re.search(r'kernelILi\d+ELi(\d)ELb([01])E', name).groups()
        grad_only = ll == '0'
        if width >= 192 or (width <= 128 and grad_only and op == '0'):
            assert row['VGPRs Spill'] == 0, (name, row)
        assert row['VGPRs Spill'] <= 12, (name, row)
    bodies = _b3_bodies(asm)
    assert len(bodies) == 78
    for name, body in bodies.items():
        # the tile loop: the backward branch with the most MFMAs in its body
        loops = []
        for m in re.finditer(r's_cbranch_\w+ (\.LBB\d+_\d+)\n', body):
            at = body.find('\n' + m.group(1) + ':')
            if 0 <= at < m.start():                      # a backward branch
                loops.append(body[at:m.start()])
        loop = max(loops, key=lambda t: t.count('v_mfma'))
        width = int(re.search(r'kernelILi(\d+)E', name).group(1))
        # GEMM 1: 6 terms x D/16 k-steps; GEMM 2: 6 terms x 2 x D/32 blocks
        assert loop.count('v_mfma_f32_32x32x16_bf16') == 12 * (width // 16), \
            (name, loop.count('v_mfma'))
        assert 'scratch_' not in loop, name
        assert loop.count('s_barrier') == 1, name          # ONE per tile
        # no operand shuffling between the register files inside the loop
        assert 'v_accvgpr_read' not in loop and 'v_accvgpr_write' not in loop \
            and 'v_accvgpr_mov' not in loop, name
        # the transposing read feeds GEMM 2: two per plane and step
        assert loop.count('ds_read_b64_tr_b16') == 6 * 2 * (width // 32), name
