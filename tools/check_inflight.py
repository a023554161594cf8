#!/usr/bin/env python3
"""Developer tool: static check of the inline-asm fragment fetch (DFN_ASM_FETCH).  The asm `ds_read_b128` writes its
destination registers asynchronously; the compiler believes they are defined at the asm statement.  Between such a read
and the `s_waitcnt lgkmcnt(N)` that retires it, NO other instruction may read or write those registers (a register-
allocator copy or a spill of an in-flight destination would pick up stale data).  This script scans the ISA of every
kernel that contains asm fragment reads and reports such instructions.

  hipcc ... --cuda-device-only -S -o render.s dfn_render.hip -DDFN_ASM_FETCH=1 ; python tools/check_inflight.py render.s"""
import re, sys

def regs_of(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out

def scan(lines, start, end, state, report, tag):
    """state: list of (is_asm_read, regs) in issue order (all LGKM ops)."""
    in_asm = False
    for i in range(start, end):
        l = lines[i].strip()
        if l.startswith(';;#ASMSTART'):
            in_asm = True; continue
        if l.startswith(';;#ASMEND'):
            in_asm = False; continue
        if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
            continue
        op = l.split()[0]
        body = l.split(';')[0]
        if op == 's_waitcnt':
            m = re.search(r'lgkmcnt\((\d+)\)', body)
            if m:
                n = int(m.group(1))
                while len(state) > n:
                    state.pop(0)
            continue
        live = set().union(*[r for a, r in state if a]) if state else set()
        if op.startswith('ds_') or op.startswith('s_load') or op.startswith('s_memtime') or op.startswith('s_memrealtime'):
            used = regs_of(body.split(None, 1)[1] if ' ' in body else '')
            if op.startswith('ds_read') and in_asm:
                dst = regs_of(body.split(',')[0])
                if dst & live:
                    report.append((tag, i, 'asm read overwrites an in-flight destination', l))
                if (used - dst) & live:
                    report.append((tag, i, 'LDS op uses an in-flight register', l))
                state.append((True, dst))
            else:
                if used & live:
                    report.append((tag, i, 'LDS/SMEM op touches an in-flight register', l))
                state.append((False, set()))
            continue
        if live and (regs_of(body) & live):
            report.append((tag, i, 'instruction touches an in-flight asm destination', l))
    return state

def main(path):
    text = open(path).read().split('\n')
    # kernels = from "name:" of a .globl function to s_endpgm
    kernels, cur = [], None
    for i, l in enumerate(text):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            cur = [m.group(1), i, None]
        if cur and 's_endpgm' in l:
            cur[2] = i; kernels.append(cur); cur = None
    total = 0
    for name, a, b in kernels:
        if not any('ds_read_b128' in text[i] and text[i - 1].strip().startswith(';;#ASMSTART') for i in range(a + 1, b)):
            continue
        report = []
        labels = {m.group(1): i for i in range(a, b) for m in [re.match(r'^(\.LBB\d+_\d+):', text[i])] if m}
        # pass 1: linear, remembering the state at every backward branch
        state, states_at = [], {}
        in_asm = False
        i = a
        # run linearly in chunks between branches to capture states
        last = a
        for i in range(a, b):
            l = text[i].strip()
            m = re.match(r'^s_cbranch\w*\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)', l)
            if m:
                state = scan(text, last, i, state, report, name)
                last = i
                tgt = m.group(1) or m.group(2)
                if tgt in labels and labels[tgt] < i:
                    states_at[(labels[tgt], i)] = list(state)
        scan(text, last, b, state, report, name)
        # pass 2: loop-carried in-flight registers: re-run each loop body from the state at its back edge
        for (t, br), st in states_at.items():
            scan(text, t, br, list(st), report, name + ' (loop-carried)')
        uniq = sorted(set(report))
        n_reads = sum(1 for i in range(a + 1, b) if 'ds_read_b128' in text[i] and text[i - 1].strip().startswith(';;#ASMSTART'))
        print(f"{name}: {n_reads} asm fragment reads, {len(uniq)} hazard(s)")
        for tag, i, why, l in uniq[:12]:
            print(f"   line {i}: {why}: {l}")
        total += len(uniq)
    return total

if __name__ == '__main__':
    sys.exit(1 if main(sys.argv[1]) else 0)
