"""Structure manifest of a training checkpoint (the 13-key dict of run_nerf_com_trainExpLater.py:1101-1115) - test
infrastructure shared by tests/golden/make_golden.py (which builds the checkpoint from the REFERENCE's modules and
torch.optim.Adam) and tests/test_gpu_driver.py (which builds it from this repo's save_checkpoint): keys and their order,
nesting, tensor shapes and dtypes, the optimizers' param_groups fields and which parameters carry Adam state."""
import torch


def _leaf(v):
    if isinstance(v, torch.Tensor):
        return {"tensor": list(v.shape), "dtype": str(v.dtype).replace("torch.", "")}
    if isinstance(v, bool):
        return {"bool": v}
    if isinstance(v, int):
        return {"int": None}                  # value not compared (step counters, global_step)
    if isinstance(v, float):
        return {"float": v}
    if isinstance(v, (tuple, list)) and all(isinstance(x, (int, float)) for x in v):
        return {"seq": [float(x) for x in v]}
    if v is None:
        return {"none": None}
    return {"type": type(v).__name__}


def checkpoint_manifest(ck):
    out = {"keys": list(ck.keys()), "entries": {}}
    for k, v in ck.items():
        if k.startswith("network_"):
            out["entries"][k] = [[name, list(p.shape), str(p.dtype).replace("torch.", "")] for name, p in v.items()]
        elif k.startswith("optimizer_"):
            groups = []
            for g in v["param_groups"]:
                # lr is a value, not structure; fused / foreach / capturable / differentiable select torch's implementation
                # of the same update (this repo: fused=True with the step as one HIP launch): presence only
                impl = ("fused", "foreach", "capturable", "differentiable")
                groups.append({name: (val if name == "params" else ({"impl": None} if name in impl else _leaf(val)))
                               for name, val in sorted(g.items()) if name not in ("lr", "initial_lr")})
            state = {str(i): {name: _leaf(val) for name, val in sorted(s.items())} for i, s in sorted(v["state"].items())}
            out["entries"][k] = {"top": sorted(v.keys()), "param_groups": groups, "state": state}
        else:
            out["entries"][k] = _leaf(v)
    return out
