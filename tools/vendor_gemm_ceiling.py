#!/usr/bin/env python3
"""Developer tool (GPU box): calibrate the power ceiling of the matrix pipe with a VENDOR kernel.
torch.matmul (hipBLASLt / rocBLAS) 8192^3 in f16 and bf16 on three operand sets - zeros (no toggling), post-ReLU-like (half
zeros, |N(0,1)| otherwise: what an MLP's activations look like) and uniform random in [-1, 1) - TFLOP/s against the 2.5 PFLOP/s
dense peak, the shader clock sampled from rocm-smi while the GEMMs run, next to the library's own bare MFMA chain
(dfn_debug_mfma_chain) on the same operands' statistics.  If the vendor GEMM on random data exceeded 0.75 of peak, the renderer's
0.57 would not be a power story."""
import os, re, subprocess, sys, threading, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import torch
from dfanerf._lib import lib, check
from dfanerf.engine import TIERS

dev = torch.device("cuda")
N = 8192
PEAK = 2500.0


def sclk_sampler(stop, samples):
    while not stop.is_set():
        try:
            o = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            m = re.search(r"sclk clock level.*?\((\d+)Mhz\)", o)
            if m:
                samples.append(int(m.group(1)))
        except Exception:
            pass
        time.sleep(0.2)


def operands(kind, dtype, shape, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    if kind == "zeros":
        return torch.zeros(shape, dtype=dtype, device=dev)
    if kind == "relu":
        return (torch.randn(shape, device=dev, generator=g).abs() * (torch.rand(shape, device=dev, generator=g) < 0.5)).to(dtype)
    return (torch.rand(shape, device=dev, generator=g) * 2 - 1).to(dtype)


print(f"torch {torch.__version__}, {torch.cuda.get_device_name(0)}, GEMM {N}^3")
for dtype, tname in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
    for kind in ("zeros", "relu", "random"):
        a, b = operands(kind, dtype, (N, N), 1), operands(kind if kind != "relu" else "random", dtype, (N, N), 2)
        for _ in range(3):
            torch.matmul(a, b)
        torch.cuda.synchronize()
        stop, samples = threading.Event(), []
        th = threading.Thread(target=sclk_sampler, args=(stop, samples))
        th.start()
        n = 1200                     # ~1 s per configuration: the clock settles, rocm-smi gets several samples
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            torch.matmul(a, b)
        e1.record()
        torch.cuda.synchronize()
        stop.set(); th.join()
        ms = e0.elapsed_time(e1) / n
        tf = 2.0 * N ** 3 / (ms * 1e-3) / 1e12
        clk = f"{sum(samples) / len(samples):.0f} MHz (rocm-smi, {len(samples)} samples)" if samples else "n/a"
        print(f"vendor GEMM {tname:5s} A {kind:6s}: {ms:7.3f} ms  {tf:7.1f} TFLOP/s  frac {tf / PEAK:.3f}   sclk {clk}", flush=True)
    # the library's bare chain on the same statistics (A = uniform random fragments, B = post-ReLU-like / random / zeros)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    blocks, iters = 4 * cus, 12000
    out = torch.empty(blocks * 512, dtype=torch.float32, device=dev)
    clk = torch.zeros(2, dtype=torch.int64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for kind in ("zeros", "relu", "random"):
        fr = operands("random" if kind != "zeros" else "zeros", dtype, (16384,), 3).contiguous()
        bb = operands(kind, dtype, (32768,), 4).contiguous()
        for (l, v) in ((0, 0), (2, 4)):
            best = 0.0
            for rep in range(3):
                n = 500 if rep == 0 else iters
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib.dfn_debug_mfma_chain(TIERS[tname], l, v, C.c_void_p(fr.data_ptr()), C.c_void_p(bb.data_ptr()), n, blocks,
                                               C.c_void_p(out.data_ptr()), C.c_void_p(clk.data_ptr()), st), "chain")
                e1.record(); torch.cuda.synchronize()
                if rep:
                    best = max(best, blocks * 8 * n * 32 * 32768.0 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
            c = clk.cpu().numpy()
            print(f"own chain   {tname:5s} B {kind:6s} LDS {l / 2:.0f} KiB/MFMA VALU {v / 2:.0f}/MFMA: {best:7.1f} TFLOP/s  frac {best / PEAK:.3f}   "
                  f"clock {float(c[0]) / float(c[1]) * 0.1:.2f} GHz (in-kernel)", flush=True)
