import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import numpy as np, torch
from oracle import features as ofeat
import seld_native
dev = torch.device("cuda:0")
pcm = ofeat.synth_pcm(11, 4, 9600 + 5, "noise")
pcm[2] = 0.0
pcm[:, 4800:6000] = 0.0
feat = seld_native.spatial_features(pcm.to(dev), "logmel_gcc").cpu()
ref = ofeat.gcc_phat_f64(pcm.numpy())
got = feat[:, 4:].permute(1, 2, 0).numpy()
err = np.abs(got - ref)
print("max err per pair", err.max(axis=(1, 2)))
print("max err per frame", np.round(err.max(axis=(0, 1)), 4))
p, l, f = np.unravel_index(err.argmax(), err.shape)
print("worst", p, l, f, got[p, l, f], ref[p, l, f])
print("got pair1 frame", f, np.round(got[1, 28:36, f], 3), "ref", np.round(ref[1, 28:36, f], 3))
spec = seld_native.stft(pcm.to(dev)).cpu()
print("spec ch2 absmax", spec[2].abs().max().item(), "frame", f, "absmax per channel", [spec[c, f].abs().max().item() for c in range(4)],
      "exact zeros per channel", [(spec[c, f] == 0).sum().item() for c in range(4)])
