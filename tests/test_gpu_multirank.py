"""The multi-rank code paths on hardware, on a box with ONE GPU: two and eight processes on GPU 0 over gloo (tests/multirank_worker.py).
What only an 8-GPU node can show - RCCL over xGMI, the scaling curve - is the driver's to measure (LABNOTES.md section 6)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_on_one_gpu_render_and_train(world):
    """world = 8: the partition the 8-GPU node runs - 25,313 rays x 7 + the 25,309-ray short shard, 8 x 3
    collectives per training step - executed for the first time before the driver's 8-GPU lease does."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    from conftest import free_port
    for port in (free_port(), free_port()):      # one retry: a free port can be taken between the probe and the rendezvous
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                            "--master-addr", "127.0.0.1", "--master-port", str(port),
                            os.path.join(HERE, "multirank_worker.py")], env=env, capture_output=True, text=True, timeout=1200)
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("MULTIRANK_OK")]
    assert line, r.stdout[-2000:]
    # the gathered images equal the frame rendered by one rank (float and uint8), the replicas are bit-identical after three
    # data-parallel steps that started from different parameters and saw different frames, and the steps did train
    assert line[0] == "MULTIRANK_OK image=True replicas_identical=True trained=True finite=True", line[0]
