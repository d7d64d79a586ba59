"""Worker of tests/test_switches.py: one training-mode forward + backward of the Berlin network (N = 40000 points, dense
enough for the on-chip walks at r = 32 / 64 and the deferred levels) under whatever T2H_* switches the parent put into the
environment, then two more pipelined steps through the Trainer.  Prints one JSON line: the heights and every parameter
gradient as float64 checksums + the tensors themselves saved to the path given (for the parent's comparison with the default
run), and the library-fallback count."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from detinit import det_init_, synth_cloud          # noqa: E402


def main(out_path):
    import tomosar2height_amd as t2h
    from tomosar2height_amd import TomoSAR2Height
    from tomosar2height_amd.config import berlin_config
    from tomosar2height_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    if os.environ.get("T2H_HIP_CONV") == "0":
        t2h.allow_library_fallback(True).set()
    model = det_init_(TomoSAR2Height(berlin_config()), seed=41).to(dev)
    model.set_channels_last(True)
    cloud = synth_cloud(40000, seed=5).to(dev)
    w = torch.randn(512, 512, generator=torch.Generator().manual_seed(1)).to(dev)
    pa, _ = model(input_cloud=cloud)
    (pa.squeeze() * w).mean().backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}
    res = {"heights": pa.detach().cpu(), "grads": grads}
    # the Trainer's side of the switches: three tiles of one accumulation window (pipeline, side streams, bucket, compose cache)
    for p in model.parameters():
        p.grad = None
    tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=dev, optimize_every=8, use_cloud=True)
    dsm = (torch.rand(1, 512, 512, generator=torch.Generator().manual_seed(2)) * 30).to(dev)
    for i in range(3):
        tr.train_step({"inputs": synth_cloud(40000, seed=50 + i).to(dev), "dsm": dsm})
    tr.flush_gradients()
    torch.cuda.synchronize()
    res["trainer_loss"] = float(tr.accumulated_loss)
    res["trainer_grads"] = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
    torch.save(res, out_path)
    print(json.dumps({"fallbacks": sum(t2h.fallback_counts().values()), "n_grads": len(grads),
                      "finite": bool(all(torch.isfinite(g).all() for g in grads.values()) and torch.isfinite(res["heights"]).all())}))


if __name__ == "__main__":
    main(sys.argv[1])
