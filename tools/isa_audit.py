#!/usr/bin/env python3
"""Audits the gfx950 assembly of the HIP library (hipcc -S / -save-temps output).

Two checks, both about things the compiler does not guard for hand-written instructions:

1. Hazards around inline assembly.  On gfx940/gfx950 a VALU instruction that reads the result of a
   transcendental instruction (v_rcp/v_rsq/v_sqrt/v_exp/v_log/v_sin/v_cos) needs one instruction
   between the two; the hardware does not interlock.  The compiler inserts the s_nop for its own
   instructions but does not look inside `asm` blocks, so a hand-written v_min/v_max that consumes a
   reciprocal straight away reads a stale register.  Reported per kernel with the line of the .s file.

2. Register spills per kernel (.vgpr_spill_count / .sgpr_spill_count / scratch bytes of the metadata)
   against the budget recorded in tools/spill_budget.json: a kernel that spills more than its audited
   figure fails the build (`make audit`), so a compiler or source change cannot move spill traffic into
   a hot loop unnoticed.  `--write-budget` records the current figures.

Usage: isa_audit.py file.s [--budget tools/spill_budget.json] [--write-budget] [--list]
"""
import json
import re
import subprocess
import sys

TRANS = re.compile(r'^v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag|_clamp|_legacy)?_(f32|f16|f64)\S*\s+(v\d+|v\[\d+:\d+\])')


def demangle(names):
    try:
        out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True, check=True).stdout.split('\n')
    except Exception:
        return list(names)
    short = []
    for d in out[:len(names)]:
        d = d.replace('(anonymous namespace)::', '').replace('void ', '')
        short.append(d.split('(')[0])
    return short


def registers(operand):
    """v3 -> {3}; v[4:7] -> {4,5,6,7}; anything else -> empty."""
    m = re.fullmatch(r'v(\d+)', operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def instruction_stream(lines):
    """(line number, text, inside asm block, function) for every instruction."""
    out = []
    inasm = False
    func = None
    for i, raw in enumerate(lines):
        t = raw.strip()
        if re.match(r'^_Z\S+:', raw):
            func = raw.split(':')[0]
        if t.startswith(';;#ASMSTART'):
            inasm = True
            continue
        if t.startswith(';;#ASMEND'):
            inasm = False
            continue
        if not t or t[0] in ';.' or t.endswith(':'):
            continue
        out.append((i + 1, t.split(';')[0].strip(), inasm, func))
    return out


def trans_hazards(stream):
    hits = []
    for k in range(1, len(stream)):
        ln, text, inasm, func = stream[k]
        pln, ptext, pinasm, pfunc = stream[k - 1]
        if pfunc != func or not (inasm or pinasm):
            continue          # both compiler-made: the hazard recogniser has dealt with the pair
        m = TRANS.match(ptext)
        if not m or not text.startswith('v_') or TRANS.match(text):
            continue
        written = registers(m.group(4))
        ops = re.split(r'[ ,]+', text)[1:]
        read = set()
        for op in ops[1:]:
            read |= registers(op)
        if written & read:
            hits.append((func, pln, ptext, text))
    return hits


def kernel_resources(text):
    i = text.find('.amdgpu_metadata')
    if i < 0:
        return {}
    res = {}
    for k in re.split(r'\n  - \.agpr_count', text[i:])[1:]:
        def g(key):
            m = re.search(r'\.' + key + r':\s+(\S+)', k)
            return m.group(1) if m else '0'
        res[g('name')] = {'vgpr': int(g('vgpr_count')), 'sgpr': int(g('sgpr_count')), 'vgpr_spill': int(g('vgpr_spill_count')),
                          'sgpr_spill': int(g('sgpr_spill_count')), 'scratch': int(g('private_segment_fixed_size')),
                          'lds': int(g('group_segment_fixed_size'))}
    return res


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    path = argv[1]
    budget_path = argv[argv.index('--budget') + 1] if '--budget' in argv else None
    text = open(path).read()
    stream = instruction_stream(text.split('\n'))
    hits = trans_hazards(stream)
    res = kernel_resources(text)
    names = list(res)
    short = dict(zip(names, demangle(names)))
    rc = 0
    if hits:
        rc = 1
        fn = demangle([h[0] for h in hits])
        for h, f in zip(hits, fn):
            print(f'HAZARD {f}: {path}:{h[1]}: `{h[2]}` is read by the next instruction `{h[3]}` (hand-written; no wait state)')
    if '--list' in argv:
        for n in names:
            r = res[n]
            print(f"{short[n]:62s} vgpr {r['vgpr']:3d} sgpr {r['sgpr']:3d} vgpr_spill {r['vgpr_spill']:3d} sgpr_spill {r['sgpr_spill']:3d} scratch {r['scratch']:4d} lds {r['lds']}")
    if '--write-budget' in argv and budget_path:
        data = {short[n]: {'vgpr_spill': res[n]['vgpr_spill'], 'sgpr_spill': res[n]['sgpr_spill'], 'scratch': res[n]['scratch']}
                for n in names if res[n]['vgpr_spill'] or res[n]['sgpr_spill'] or res[n]['scratch']}
        json.dump(data, open(budget_path, 'w'), indent=1, sort_keys=True)
        print(f'wrote {budget_path}: {len(data)} kernels with spills')
    elif budget_path:
        budget = json.load(open(budget_path))
        for n in names:
            b = budget.get(short[n], {'vgpr_spill': 0, 'sgpr_spill': 0, 'scratch': 0})
            r = res[n]
            over = [k for k in ('vgpr_spill', 'sgpr_spill', 'scratch') if r[k] > b[k]]
            if over:
                rc = 1
                print(f"SPILL {short[n]}: " + ', '.join(f'{k} {r[k]} > audited {b[k]}' for k in over))
    print(f'isa_audit: {len(names)} kernels, {len(hits)} hazards' + ('' if rc == 0 else ' -- FAILED'))
    return rc


if __name__ == '__main__':
    sys.exit(main(sys.argv))
