#!/usr/bin/env python3
"""Build-time check: the 16-bit inference render / decoder kernels must not touch scratch memory.

A stack object or a spilled VGPR in render_kernel<tier, *, TRAIN=false> turns into scratch_load / scratch_store traffic
on every MLP pass (round 1: the Stream struct indexed by a runtime field bit cost the two-field kernel 3.3 GB of HBM
traffic per frame).  Reads the kernel metadata hipcc leaves in the device ISA (--save-temps) and fails if any of those
kernels has a private segment or a `scratch_` instruction in its body.

usage: check_scratch.py <file.s> [...]"""
import re
import sys


def kernels(text):
    """name -> (private_segment_fixed_size, vgpr_spill_count, body)"""
    out = {}
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)",
                         text):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    for name, (priv, spill) in meta.items():
        i = text.find("\n" + name + ":")
        j = text.find(".Lfunc_end", i) if i >= 0 else -1
        out[name] = (priv, spill, text[i:j] if i >= 0 else "")
    return out


def main(paths):
    bad = 0
    for p in paths:
        for name, (priv, spill, body) in kernels(open(p).read()).items():
            # the last template argument is TRAIN / REC (recorder on): those are training kernels, not checked here
            infer = ("render_kernel" in name and name.endswith("ELb0EEEvNS_10RenderArgsE")) or \
                    ("decoder_kernel" in name and name.endswith("ELb0EEEvNS_11DecoderArgsE"))
            if not infer:
                continue
            n_scr = len(re.findall(r"^\s+scratch_", body, re.M))
            ok = priv == 0 and spill == 0 and n_scr == 0
            print(f"{name}: private segment {priv} B, {spill} spilled VGPRs, {n_scr} scratch instructions"
                  + ("" if ok else "   <-- FAIL"))
            bad += not ok
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
