#!/usr/bin/env python
"""Static view of an MFMA kernel's issue stream (no GPU): for every pair of
consecutive MFMAs of a kernel, how many other instructions sit between them
(`s_nop n` counted as n + 1), and what sits between the last MFMA of the tile
loop and the first of the next iteration.

A wave issues one instruction per ~4 clocks; a 32 x 32 x 2 fp32 MFMA holds the
pipe for 64 clocks, a 16 x 16 x 4 one for 32: ~15 / ~7 other instructions fit
in a gap for free and not one more (DESIGN 3.3 rule (a)); whatever is between
two tiles' MFMAs is exposed in full (rule (b)).  The two changes of round 4's
last session (profiles/archive/r04y_*) came from reading exactly these two lists.

    python tools/mfma_gaps.py zhusuan_amd/csrc/linear_bernoulli.hip 'linear_bernoulli_kernelILi128ELb1ELi0ELb0E'
    python tools/mfma_gaps.py file.s REGEX      # an existing hipcc -S output

Branches inside a gap make the static count an upper bound (all paths)."""
import os
import re
import subprocess
import sys
import tempfile

FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast',
         '-S', '--cuda-device-only']


def assembly(path):
    if path.endswith('.s'):
        return open(path).read()
    out = os.path.join(tempfile.mkdtemp(), 'k.s')
    hipcc = '/opt/rocm/bin/hipcc'
    subprocess.run([hipcc] + FLAGS + ['-o', out, path], check=True,
                   stderr=subprocess.DEVNULL)
    return open(out).read()


def instructions(lines):
    out = [l.split(';')[0].strip() for l in lines]
    return [l for l in out if l and not l.startswith('.') and
            not l.endswith(':')]


def cost(instr):
    m = re.match(r's_nop (\d+)', instr)
    return 1 + int(m.group(1)) if m else 1


def main():
    asm, pat = assembly(sys.argv[1]), sys.argv[2]
    for m in re.finditer(r'^(\w*%s\w*):[^\n]*\n(.*?)s_endpgm' % pat, asm,
                         re.S | re.M):
        name, lines = m.group(1), m.group(2).splitlines()
        mf = [i for i, l in enumerate(lines) if l.strip().startswith('v_mfma')]
        if not mf:
            continue
        gaps = [sum(cost(x) for x in instructions(lines[a + 1:b]))
                for a, b in zip(mf[:-1], mf[1:])]
        print(name)
        print('  %d MFMAs; issue slots between consecutive ones:' % len(mf))
        print('  ' + ' '.join(str(g) for g in gaps))
        labels = {l.split(':')[0].strip(): i for i, l in enumerate(lines)
                  if re.match(r'^\.LBB\d+_\d+:', l)}
        targets = []
        for l in lines[mf[-1]:mf[-1] + 40]:
            b = re.match(r'\s*(s_c?branch\w*)\s+(\.LBB\d+_\d+)', l)
            if b and mf[0] - 400 < labels.get(b.group(2), 1 << 30) < mf[0]:
                targets.append(labels[b.group(2)])
                if b.group(1) == 's_branch':
                    break
        if targets:
            top = instructions(lines[min(targets):mf[0]])
            print('  loop top (header .. first MFMA, all paths): %d '
                  'instructions' % len(top))
            heavy = [t for t in top if re.match(
                r'(s_mul_hi|v_mul_hi|v_cmp_\w+_[iu]64|v_mad_u64)', t)]
            if heavy:
                print('  64-bit arithmetic there: ' + '; '.join(heavy[:6]))


if __name__ == '__main__':
    main()
