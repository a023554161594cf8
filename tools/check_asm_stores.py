#!/usr/bin/env python3
"""Build check: the recorder's hand-written stores (dfn_mlp.h DFN_GSTORE: `global_store_* voff, data, s[base]` in inline asm).
hipcc's hazard recogniser does not look into inline asm, so the two hazards of the gfx940 family that concern these stores are
checked on the ISA instead:
  * VALU writes an SGPR (v_readfirstlane, v_readlane, v_cmp ... s[n:m]) -> a VMEM instruction reads it: 5 wait states;
  * a store of more than 8 bytes -> a VALU write of its data registers: 2 wait states (the asm carries `s_nop 1` itself).
   python tools/check_asm_stores.py build/*-gfx950.s      (exit 1 on a finding)"""
import re, sys

def wait_states(ins):
    m = re.match(r's_nop (\d+)', ins)
    return int(m.group(1)) + 1 if m else 1

def main(paths):
    bad = 0
    for path in paths:
        lines = [l.strip() for l in open(path).read().split('\n')]
        n_asm = 0
        in_asm = False
        for i, l in enumerate(lines):
            if l.startswith(';;#ASMSTART'): in_asm = True; continue
            if l.startswith(';;#ASMEND'): in_asm = False; continue
            m = re.match(r'global_store_dword(x2|x3|x4)? v\d+, v\[?[\d:]+\]?, s\[(\d+):(\d+)\]', l) if in_asm else None
            if not m: continue
            n_asm += 1
            base = {int(m.group(2)), int(m.group(3))}
            # (1) who wrote the base, within 5 wait states?
            ws, j = 0, i - 1
            pending = set(base)
            while j >= 0 and ws < 5 and pending:
                t = lines[j]; j -= 1
                if not t or t[0] in ';.' or t.endswith(':'): continue
                ws += wait_states(t)
                dst = t.split(None, 1)[1].split(',')[0] if ' ' in t else ''
                regs = set()
                for r in re.finditer(r's\[(\d+):(\d+)\]', dst): regs.update(range(int(r.group(1)), int(r.group(2)) + 1))
                for r in re.finditer(r'\bs(\d+)\b', dst): regs.add(int(r.group(1)))
                hit = regs & pending
                if hit:
                    if t.startswith('v_'):
                        print(f"{path}:{i + 1}: {l}\n    base written by a VALU instruction {ws} wait state(s) earlier: {t}")
                        bad += 1
                    pending -= hit          # an SALU write: no hazard for that register
            # (2) the wide form carries its wait states
            if m.group(1) in ('x3', 'x4'):
                nxt = lines[i + 1] if i + 1 < len(lines) else ''
                if not re.match(r's_nop [1-9]', nxt):
                    print(f"{path}:{i + 1}: {l}\n    a store of more than 8 bytes without `s_nop 1` behind it")
                    bad += 1
        print(f"{path.split('/')[-1]}: {n_asm} hand-written stores, {'OK' if not bad else 'FINDINGS'}")
    return 1 if bad else 0

if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
