"""Developer diagnostic: where do the captured training step and the eager run of the same stepper part ways?
Runs eager twice (run-to-run determinism) and graph once on identical batches; prints the first iteration whose loss
differs and, for that iteration, which parameter's update differs first."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd"), str(ROOT / "tests")]
import torch
import test_graph_gpu as T

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
batches = [b for b in T._batches(dev, n + 3, 8) if b[0].shape[0] == 8][:n]


def run(graphs, upto):
    losses, sd, _ = T._train("crnn", dev, graphs, batches[:upto])
    return losses, sd


e1, _ = run(False, n)
e2, _ = run(False, n)
g, _ = run(True, n)
print("eager vs eager max diff", (e1 - e2).abs().max().item())
d = (e1 - g).abs()
print("eager vs graph per-iteration diff", [f"{v:.1e}" for v in d.tolist()])
first = next((i for i, v in enumerate(d.tolist()) if v > 0), None)
print("first differing iteration:", first)
if first is not None:
    k = max(first, 4)                       # weights AFTER k iterations (k-1 is the last common loss)
    for upto in (first, first + 1):
        _, a = run(False, upto)
        _, b = run(True, upto)
        bad = [(name, (a[name] - b[name]).abs().max().item()) for name in a if not torch.equal(a[name], b[name])]
        print(f"after {upto} iterations: {len(bad)} tensors differ", bad[:12])
