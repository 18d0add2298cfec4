"""Rank body of tests/test_ddp_gpu.py's gradient-exchange rehearsals (run under torch.distributed.run, gloo backend, the
ranks share GPU 0).  Trains a small CRNN (HIP BiGRU, fused tails, master-weight Adam, no dropout, deterministic
convolution solvers) for ITERS iterations on per-rank batches with the step the MODE argument selects and prints one
``RANKLINE {json}`` with every loss and a digest of the final weights:

  staged    seld_graph.GraphedTrainStep, backward pass cut into stages, buckets all-reduced asynchronously between them
  blocking  the same stepper with Config.OVERLAP_ALLREDUCE off: one graph, one blocking exchange, one update graph
  fp32wire  staged, gradients cast to fp32 for the exchange (Config.GRAD_REDUCE_DTYPE = "fp32")
  ddp       Config.GRAPH_STEP off: the eager loop under DistributedDataParallel (bucketed reducer)"""
import hashlib
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "sound-event-localization-detection_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ITERS = 16


def main():
    mode = sys.argv[1]
    kind = sys.argv[2] if len(sys.argv) > 2 else "crnn"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    import trainer
    cfg = trainer.config
    cfg.MODEL_TYPE, cfg.CRNN_CNN_CHANNELS = kind, [16, 16, 32, 32]
    cfg.RESNET_CONF_N_LAYERS = 1                  # (a one-block Conformer stack behind the ResNet-50 encoder keeps the run short)
    cfg.GRAPH_STEP = mode != "ddp"
    cfg.OVERLAP_ALLREDUCE = mode != "blocking"
    cfg.GRAD_REDUCE_DTYPE = "fp32" if mode == "fp32wire" else "param"
    torch.backends.cudnn.deterministic = True
    trainer.ensure_process_group(device)
    torch.manual_seed(100 + rank)                 # different initial weights: the replica broadcast has to level them
    model = trainer.prepare_model_for_device(trainer.build_model((18, 36)), device).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.GRU):
            m.dropout = 0.0
        if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), torch.nn.Dropout):
            m.dropout.p = 0.0
    trainer.enable_master_weights(model, device)
    graphed = trainer.graph_step_enabled(device, world)
    assert graphed == (mode != "ddp")
    model = trainer.broadcast_replica_state(model, world) if graphed else trainer.wrap_ddp(model, device, world)
    crit = trainer.SMRSELDLoss("mse", 1.0, grid_size=(18, 36))
    opt = trainer.make_optimizer(model, 1e-3, device, capturable=graphed)
    step = trainer.make_stepper(model, crit, opt, device, world)
    g = torch.Generator().manual_seed(7 + rank)   # per-rank batches
    losses = []
    batch = 4 if kind == "crnn" else 2
    for i in range(ITERS):
        x = (torch.randn(batch, 250, 4, 64, generator=g) * 20 - 30).to(device)
        m = ((torch.rand(batch, 250, 648, generator=g) < 0.02).to(torch.int32) << 3).to(torch.uint16).to(device)
        total, _ = step(x, m)
        losses.append(float(total.item()))
    stats = step.stats() if hasattr(step, "stats") else None
    if hasattr(step, "close"):
        step.close()
    sd = trainer.model_state_dict(model)
    names = {n for n, _ in trainer.unwrap(model).named_parameters()}
    digest = hashlib.sha256()
    for k in sorted(sd):
        if k in names:        # parameters (fp32 masters); BatchNorm's running statistics are per rank by design
            digest.update(sd[k].detach().float().cpu().numpy().tobytes())
    print("RANKLINE " + json.dumps({"rank": rank, "mode": mode, "losses": losses, "digest": digest.hexdigest(),
                                    "stats": stats}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
