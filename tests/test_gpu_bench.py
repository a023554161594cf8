"""bench.py as the driver runs it: one JSON line with the contract's keys - and the multi-rank forms on a one-GPU box
(DFN_BENCH_ONE_GPU=1: every rank on GPU 0 over gloo; functional, not a measurement): a plain `python bench.py --gpus 2`
launches its own ranks (the driver's N = 1 command shape with N > 1), the torchrun form keeps working."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
FAST = ["--steps", "3", "--warmup", "1", "--sustain-seconds", "0", "--no-extra", "--no-cpu-baseline"]


def _line(cmd, env=None, timeout=900):
    r = subprocess.run(cmd, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_and_the_parity_binding():
    out = _line([sys.executable, BENCH] + FAST)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["dtype"] == "f16" and "workload" in out["config"]
    assert 0.0 < out["roofline"]["frac"] < 1.0 and out["roofline"]["bound"] == "mfma"
    # 64 rays of the last timed frame against the oracle, in the timed tier (f16: the 49.4 dB clause of LABNOTES.md 3)
    pc = out["parity_check"]
    assert "error" not in pc, pc
    assert pc["rays"] == 64 and pc["psnr_db"] >= 49.4 and pc["max_abs_rgb"] < 2e-2, pc
    # the power ceiling measured in-process after the timed loops (dfn_debug_mfma_chain on the renderer's operand statistics):
    # the bare MFMA chain sustains more than the renderer's instruction mix, which the timed kernel cannot beat by much
    ce = out["roofline"]["power_ceiling"]
    assert "error" not in ce, ce
    assert 0.3 < ce["renderer_mix"]["frac_of_peak"] < ce["bare_chain"]["frac_of_peak"] < 1.0, ce
    assert 0.5 < ce["frac_of_renderer_mix"] < 1.15 and 0.0 < ce["frac_of_ceiling"] < 1.0, ce
    assert out["roofline"]["traffic"] is not None


@pytest.mark.parametrize("workload", ["c2", "c5", "c4"])
def test_plain_python_launches_its_own_ranks(workload):
    """`python bench.py --gpus 2` with no torchrun environment: two ranks, async double-buffered gather (c2), one gather
    per 8-frame batch (c5), the gradient bucket (c4)."""
    env = {"DFN_BENCH_ONE_GPU": "1"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    out = _line([sys.executable, BENCH, "--gpus", "2", "--workload", workload] + FAST, env=env)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"] == "gloo"
    assert len(out["per_rank"]["ms_per_step"]) == 2 and all(v > 0 for v in out["per_rank"]["ms_per_step"])
    if workload != "c4":
        assert out["gather"]["async"] and out["gather"]["ms_alone"] > 0
        # the last timed frame as it was gathered (two render streams, two buffer sets) = the frame rendered whole by one rank
        assert out["gather_check"].get("identical") is True, out["gather_check"]
        assert out["config"]["frames_per_step"] == (8 if workload == "c5" else 1)
        assert out["scaling"] == "strong"
    else:
        assert out["scaling"] == "weak"


def test_torchrun_form_still_works():
    from conftest import free_port
    out = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                 "127.0.0.1", "--master-port", str(free_port()), BENCH, "--gpus", "2", "--workload", "c3"] + FAST,
                env={"DFN_BENCH_ONE_GPU": "1"})
    assert out["n_gpus"] == 2 and out["config"]["fields"] == 2 and len(out["per_rank"]["render_kernel_ms"]) == 2
    assert out["gather_check"].get("identical") is True, out["gather_check"]


def test_training_step_in_the_multi_rank_schedule_over_rccl():
    """DFN_BENCH_RCCL_WORLD1: a process group of ONE rank on the real RCCL backend (RCCL refuses two ranks on one device) and
    the multi-rank training schedule - the gradient bucket reduced IN PLACE through RCCL's own stream, eight hardware queues,
    the optimizer streams ordered behind the collective: what every rank of an N > 1 run does per step, minus the exchange.
    The loss after 40 optimisation steps equals the single-rank schedule's (same seeds; averaging over one rank is the
    identity, and the weight gradients are bit-reproducible)."""
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    args = [sys.executable, BENCH, "--workload", "c4", "--steps", "30", "--warmup", "10", "--sustain-seconds", "0",
            "--no-extra", "--no-cpu-baseline"]
    single = _line(args, env={"DFN_BENCH_PRINT_LOSS": "1"})
    multi = _line(args, env={"DFN_BENCH_PRINT_LOSS": "1", "DFN_BENCH_RCCL_WORLD1": "1"})
    assert multi["backend"].startswith("nccl") and single["backend"] is None
    assert multi["ms_per_step"] > 0 and multi["n_gpus"] == 1
    assert multi["last_loss"] == single["last_loss"], (multi["last_loss"], single["last_loss"])


@pytest.mark.parametrize("workload", ["c2", "c5"])
def test_inference_in_the_multi_rank_schedule_over_rccl(workload):
    """the N > 1 INFERENCE schedule through real RCCL on one GPU (DFN_BENCH_RCCL_WORLD1: a one-rank "nccl" process
    group): two render streams, the front end one frame ahead (FramePrefetcher), the asynchronous double-buffered
    all_gather_into_tensor on RCCL's own stream, eight hardware queues - what a rank of an 8-GPU run does per frame, minus the
    wire.  The gathered frame equals the frame rendered on the plain single-stream path bit for bit."""
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    out = _line([sys.executable, BENCH, "--workload", workload, "--steps", "6", "--warmup", "2", "--sustain-seconds", "0",
                 "--no-extra", "--no-cpu-baseline"], env={"DFN_BENCH_RCCL_WORLD1": "1"})
    assert out["backend"].startswith("nccl") and out["n_gpus"] == 1 and out["rccl_ranks"] == 1
    assert out["gather"]["async"] and out["gather"]["collective"] == "all_gather_into_tensor" and out["gather"]["ms_alone"] > 0
    assert out["gather"]["render_streams"] == 2
    assert out["gather_check"].get("identical") is True, out["gather_check"]
    assert out["per_rank"]["ms_per_step"][0] > 0 and 0.0 < out["roofline"]["whole_job"]["frac"] < 1.0
    # two launches co-run on the two render streams: the per-launch figure is the union of the launch intervals over the
    # launches (what the kernel occupied the GPU for), so the per-launch fraction stays next to the wall-clock one; the
    # plain event pairs, which count the shared time twice, are reported beside it
    rf = out["roofline"]
    assert rf["render_streams"] == 2 and rf["kernel_ms"] <= rf["kernel_ms_event_pairs"] * 1.001
    assert rf["kernel_ms"] <= out["ms_per_step"] / out["config"]["frames_per_step"] * 1.02, rf
    assert rf["whole_job"]["frac"] <= rf["frac"] * 1.02 and rf["frac"] <= rf["whole_job"]["frac"] * 1.15, rf
    pc = out["parity_check"]                                   # and the timed configuration still matches the oracle
    assert "error" not in pc and pc["psnr_db"] >= 49.4, pc


def test_workload_all_two_ranks_and_fail_fast_without_devices():
    """`--workload all` at N > 1: ONE call yields the headline (c2) plus c3, c5, c4, c4s under the process group, each with its
    own error capture; and `--gpus 2` on a box with one GPU (no DFN_BENCH_ONE_GPU) fails fast: rc 2, a clear message."""
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    out = _line([sys.executable, BENCH, "--gpus", "2", "--workload", "all", "--steps", "2", "--warmup", "1",
                 "--sustain-seconds", "0", "--no-cpu-baseline"], env={"DFN_BENCH_ONE_GPU": "1"}, timeout=1500)
    assert out["n_gpus"] == 2 and out["config"]["fields"] == 1 and out["gather_check"].get("identical") is True
    ow = out["other_workloads"]
    assert set(ow) == {"c3", "c5", "c4", "c4s"}, sorted(ow)
    for k, v in ow.items():
        assert "error" not in v, (k, v)
        assert v["n_gpus"] == 2 and v["ms_per_step"] > 0 and len(v["per_rank"]["ms_per_step"]) == 2
    assert ow["c3"]["gather_check"]["identical"] and ow["c5"]["gather_check"]["identical"]
    assert ow["c4"]["scaling"] == "weak" and ow["c4s"]["scaling"] == "strong"
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + FAST, capture_output=True, text=True, timeout=300, cwd=ROOT,
                           env={k: v for k, v in os.environ.items() if k != "DFN_BENCH_ONE_GPU"})
        assert r.returncode == 2 and "HIP device" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_workload_all_eight_ranks_on_one_gpu():
    """`bench.py --gpus 8 --workload all` with all eight ranks on GPU 0 over gloo - the command the driver's
    8-GPU lease runs, executed before that lease is the first time eight ranks, the 25,309-ray short shard and 8 x 3 collectives
    per training step ever run.  Bitwise gather checks for c2 / c3 / c5, per-rank fields for 8 ranks, c4 weak and c4s strong."""
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    out = _line([sys.executable, BENCH, "--gpus", "8", "--workload", "all", "--steps", "2", "--warmup", "1",
                 "--sustain-seconds", "0", "--no-cpu-baseline"], env={"DFN_BENCH_ONE_GPU": "1", "OMP_NUM_THREADS": "4"},
                timeout=2400)
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["backend"] == "gloo"
    assert out["gather_check"].get("identical") is True, out["gather_check"]
    assert out["roofline"]["rays_per_launch"] == 25313 and len(out["per_rank"]["ms_per_step"]) == 8
    assert "25313 rays per rank" in out["config"]["parallelism"]
    ow = out["other_workloads"]
    assert set(ow) == {"c3", "c5", "c4", "c4s"}, sorted(ow)
    for k, v in ow.items():
        assert "error" not in v, (k, v)
        assert v["n_gpus"] == 8 and v["ms_per_step"] > 0 and len(v["per_rank"]["ms_per_step"]) == 8 and \
            all(x > 0 for x in v["per_rank"]["ms_per_step"]), (k, v["per_rank"])
    assert ow["c3"]["gather_check"]["identical"] and ow["c5"]["gather_check"]["identical"]
    assert ow["c4"]["scaling"] == "weak" and ow["c4s"]["scaling"] == "strong"
