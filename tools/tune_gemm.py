"""Developer experiment: PyTorch TunableOp (rocBLAS / hipBLASLt solution search) on the GEMMs of one CRNN optimiser
iteration.  Prints eager ms/iteration before and after tuning and writes the selection file.
usage: python tools/tune_gemm.py [out.csv]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import torch.cuda.tunable as tunable
import trainer
torch.backends.cudnn.benchmark = True
out = sys.argv[1] if len(sys.argv) > 1 else str(ROOT / "gpurun_out" / "tunableop_results.csv")
dev = torch.device("cuda:0")
trainer.config.MODEL_TYPE = "crnn"
torch.manual_seed(0)
model = trainer.prepare_model_for_device(trainer.build_model((18, 36)), dev).train()
crit = trainer.SMRSELDLoss("mse", 1.0, grid_size=(18, 36))
trainer.enable_master_weights(model, dev)
opt = trainer.make_optimizer(model, 1e-3, dev)
x = torch.randn(32, 250, 4, 64, device=dev) * 20 - 30
mask = torch.zeros(32, 250, 648, dtype=torch.uint16, device=dev)


def timeit(n=30):
    for _ in range(4):
        trainer.train_step(model, crit, opt, x, mask, dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        trainer.train_step(model, crit, opt, x, mask, dev)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"default GEMM selection: {timeit():.3f} ms / iteration", flush=True)
tunable.set_filename(out)
tunable.set_max_tuning_duration(30)
tunable.set_max_tuning_iterations(20)
tunable.enable(True)
tunable.tuning_enable(True)
t0 = time.perf_counter()
for _ in range(2):
    trainer.train_step(model, crit, opt, x, mask, dev)
torch.cuda.synchronize()
print(f"tuning pass took {time.perf_counter() - t0:.1f} s, {len(tunable.get_results())} entries", flush=True)
tunable.tuning_enable(False)
print(f"tuned GEMM selection:   {timeit():.3f} ms / iteration", flush=True)
tunable.write_file(out) if hasattr(tunable, "write_file") else None
for r in tunable.get_results()[:40]:
    print(r)
