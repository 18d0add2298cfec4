"""Oracle (TEST INFRASTRUCTURE): the evaluation metrics of the reference's ``test_model`` restated on host arrays.

trainer.py:542-556  overall / non-background argmax accuracy over ALL collected predictions and labels
trainer.py:621-635  the frames that contain at least one non-background cell (pure-Python N x T scan upstream)

The product computes these batch by batch on the device from the compact uint16 label mask; here they are recomputed the
reference's way from dense [N, T, G, M] arrays."""
import numpy as np


def accuracies(predictions: np.ndarray, labels: np.ndarray, num_classes: int = 14):
    """-> (overall_accuracy %, non_bg_accuracy %, active cells, all cells)   trainer.py:542-556"""
    pred_classes = predictions.argmax(axis=-1)
    true_classes = labels.argmax(axis=-1)
    overall = float((pred_classes == true_classes).astype(np.float32).mean()) * 100
    non_bg = true_classes != num_classes - 1
    if non_bg.sum() > 0:
        non_bg_accuracy = float((pred_classes[non_bg] == true_classes[non_bg]).astype(np.float32).mean()) * 100
    else:
        non_bg_accuracy = 0.0
    return overall, non_bg_accuracy, int(non_bg.sum()), int(non_bg.size)


def frames_with_events(labels: np.ndarray, num_classes: int = 14):
    """-> list of (window_idx, time_idx, num_active) in the reference's scan order   trainer.py:621-635"""
    out = []
    n, t = labels.shape[:2]
    for window_idx in range(n):
        for time_idx in range(t):
            frame_classes = labels[window_idx, time_idx].argmax(axis=-1)
            num_active = int((frame_classes != num_classes - 1).sum())
            if num_active > 0:
                out.append((window_idx, time_idx, num_active))
    return out
