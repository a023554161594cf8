#!/usr/bin/env python3
"""Developer tool (GPU box): the matrix pipe's power ceiling as a function of the instruction mix around the MFMAs
(dfn_debug_mfma_chain: 8 waves per workgroup, one workgroup per compute unit, the tier's 32x32x16 MFMA on the renderer's operand
statistics) - what a register blocking with fewer LDS fragment reads per MFMA would buy the headline kernel.
   python tools/power_mix_sweep.py [f16|bf16]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import torch
from dfanerf import engine, synth
from dfanerf._lib import check, lib

tier = sys.argv[1] if len(sys.argv) > 1 else "f16"
dev = torch.device("cuda")
flat = engine.flatten_state(synth.synth_all_states(0)["decoder"], dev)
pk = engine.PackedDecoder(flat, tier)
frags = pk.packed[0][:32768].contiguous()
g = torch.Generator(device=dev).manual_seed(5)
act = torch.randn(32768, device=dev, generator=g).abs() * (torch.rand(32768, device=dev, generator=g) < 0.5)
b = act.to(torch.float16 if tier == "f16" else torch.bfloat16).contiguous()
cus = torch.cuda.get_device_properties(dev).multi_processor_count
blocks, iters = 4 * cus, 12000
out = torch.empty(blocks * 512, dtype=torch.float32, device=dev)
clk = torch.zeros(2, dtype=torch.int64, device=dev)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
names = {(0, 0): "bare chain", (2, 4): "renderer mix: 1 fragment read + 2 epilogue instructions per MFMA", (1, 4): "half the fragment reads",
         (0, 4): "no fragment reads", (2, 0): "no epilogue instructions", (1, 2): "half of both",
         (2, 2): "half the epilogue instructions (one per two values: conversion and ReLU in one)"}
for rnd in range(2):
    for (l, v), name in names.items():
        best = 0.0
        for rep in range(3):
            n = 500 if rep == 0 else iters
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.dfn_debug_mfma_chain(engine.TIERS[tier], l, v, C.c_void_p(frags.data_ptr()), C.c_void_p(b.data_ptr()), n, blocks,
                                           C.c_void_p(out.data_ptr()), C.c_void_p(clk.data_ptr()), st), "dfn_debug_mfma_chain")
            e1.record()
            torch.cuda.synchronize()
            if rep:
                c = clk.cpu().numpy()
                tf = blocks * 8 * n * 32 * 32768.0 / (e0.elapsed_time(e1) * 1e-3) / 1e12
                if tf > best:
                    best, ghz = tf, float(c[0]) / float(c[1]) * 0.1
        print(f"round {rnd}  lds2={l} valu2={v}  {best:7.1f} TFLOP/s = {best / 2500:.3f} of peak at {ghz:.2f} GHz   ({name})")
