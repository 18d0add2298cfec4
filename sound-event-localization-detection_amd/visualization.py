"""Plots written by the trainer (drop-in for the reference's ``visualization.py`` names).

Out of the accelerated scope (plain matplotlib on the host, rank 0 only).  Three entry points:
``plot_loss_curves`` (visualization.py:262-306), ``visualize_grid_predictions`` (:308-394) and
``visualize_loss_components`` (:12-260).
"""
import logging
import os

import matplotlib

matplotlib.use("Agg")
import matplotlib.pyplot as plt  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from matplotlib.colors import ListedColormap  # noqa: E402

logger = logging.getLogger("SMR_SELD")
BACKGROUND = 13


def plot_loss_curves(train_losses, test_losses, save_path=None):
    """Train / test loss per epoch with the best epoch of each starred."""
    fig, ax = plt.subplots(figsize=(12, 6))
    epochs = np.arange(1, len(train_losses) + 1)
    for series, style, name in ((train_losses, "b-o", "Training"), (test_losses, "r-s", "Test")):
        ax.plot(epochs, series, style, linewidth=2, markersize=4, label=f"{name} Loss")
        best = int(np.argmin(series))
        ax.plot(best + 1, series[best], style[0] + "*", markersize=15, label=f"Best {name}: {series[best]:.4f}")
    ax.set_xlabel("Epoch", fontsize=12)
    ax.set_ylabel("Loss", fontsize=12)
    ax.set_title("Training and Test Loss Curves", fontsize=14, fontweight="bold")
    ax.grid(True, alpha=0.3)
    ax.legend(fontsize=10)
    fig.tight_layout()
    if save_path:
        fig.savefig(save_path, dpi=300, bbox_inches="tight")
        logger.info(f"Loss curve saved to {save_path}")
    plt.close(fig)
    return fig


def _grid_panel(ax, grid, title, cmap, vmax):
    im = ax.imshow(grid, cmap=cmap, vmin=0, vmax=vmax, aspect="auto")
    ax.set_title(title, fontsize=14, fontweight="bold")
    ax.set_xlabel("Azimuth bins (J)", fontsize=11)
    ax.set_ylabel("Elevation bins (I)", fontsize=11)
    ax.grid(True, alpha=0.3, color="gray", linewidth=0.5)
    return im


def visualize_grid_predictions(ground_truth, predictions, time_frame, grid_size, title_prefix="", save_path=None):
    """Ground truth / argmax prediction / agreement map of one frame on the I x J grid.
    ``ground_truth`` and ``predictions`` are [G, M] (labels and logits)."""
    rows, cols = grid_size
    gt = torch.argmax(ground_truth, dim=-1).cpu().numpy()
    pred = torch.argmax(predictions, dim=-1).cpu().numpy()
    fig, axes = plt.subplots(1, 3, figsize=(18, 5))
    classes = matplotlib.colormaps["tab20"].resampled(14)
    for ax, data, name in ((axes[0], gt, "Ground Truth"), (axes[1], pred, "Predictions")):
        im = _grid_panel(ax, data.reshape(rows, cols), f"{title_prefix}{name}\nFrame {time_frame}", classes, BACKGROUND)
        fig.colorbar(im, ax=ax, fraction=0.046, pad=0.04).set_label("Class ID", fontsize=10)
    is_bg = gt == BACKGROUND
    agree = (gt == pred).astype(int)
    agree[is_bg] = 2
    _grid_panel(axes[2], agree.reshape(rows, cols),
                f"{title_prefix}Comparison\nFrame {time_frame}\n(Green=Correct, Red=Wrong, Gray=Background)",
                ListedColormap(["red", "green", "lightgray"]), 2)
    events = ~is_bg
    if events.any():
        acc, bg_acc = (gt[events] == pred[events]).mean() * 100, (gt[is_bg] == pred[is_bg]).mean() * 100
    else:
        acc, bg_acc = 0.0, (gt == pred).mean() * 100
    fig.text(0.5, 0.02, f"Non-BG Accuracy: {acc:.1f}%\nBG Accuracy: {bg_acc:.1f}%\nActive Events: {events.sum()}/{gt.size}",
             ha="center", fontsize=12, bbox=dict(boxstyle="round", facecolor="wheat", alpha=0.5))
    fig.tight_layout(rect=[0, 0.08, 1, 1])
    if save_path:
        fig.savefig(save_path, dpi=200, bbox_inches="tight")
        logger.info(f"Visualization saved to {save_path}")
    plt.close(fig)
    return fig


def visualize_loss_components(y_pred, y_true, criterion, epoch, save_dir="loss_visualizations", frame_idx=None):
    """For the frame with the most events: target grid, predicted non-background probability, squared
    error per cell, and the converging-localisation attention map (what each loss term looks at)."""
    os.makedirs(save_dir, exist_ok=True)
    probs = F.softmax(y_pred.detach().float().cpu(), dim=-1)
    truth = y_true.detach().float().cpu()
    b, t, g, m = probs.shape
    rows, cols = criterion.I, criterion.J
    true_cls = torch.argmax(truth, dim=-1)
    counts = (true_cls != m - 1).float().sum(dim=-1)            # [B, T]
    if frame_idx is None:
        if counts.max() < 1:
            logger.warning("No frames with sufficient events found for visualization")
            return None
        flat = int(torch.argmax(counts))
        bi, ti = divmod(flat, t)
    else:
        bi, ti = 0, int(frame_idx)
    p, y = probs[bi, ti], truth[bi, ti]
    nonbg_true = y[:, :-1].sum(-1).reshape(rows, cols)
    nonbg_pred = p[:, :-1].sum(-1).reshape(rows, cols)
    sq_err = ((p - y) ** 2).sum(-1).reshape(rows, cols)
    n_non = (nonbg_true > 0.01).sum().clamp(min=1).float()
    n_bac = (nonbg_true < 0.01).sum().float()
    y_prime = torch.where(nonbg_true > 0.01, -(n_bac / n_non) * torch.ones_like(nonbg_true), torch.ones_like(nonbg_true))
    padded = F.pad(y_prime[None, None], (1, 1, 1, 1), mode="circular")[0, 0]
    neigh = sum(padded[1 + di:rows + 1 + di, 1 + dj:cols + 1 + dj] for di in (-1, 0, 1) for dj in (-1, 0, 1)
                if di or dj)
    attention = y_prime + (neigh - 8 * y_prime) / 8.0
    fig, axes = plt.subplots(1, 4, figsize=(22, 5))
    panels = ((nonbg_true, "Target events"), (nonbg_pred, "Predicted P(event)"),
              (sq_err, "Squared error (MSE term)"), (attention, "CL attention map"))
    for ax, (data, name) in zip(axes, panels):
        im = ax.imshow(data.numpy(), aspect="auto", cmap="viridis")
        ax.set_title(f"{name}\nepoch {epoch}, window {bi}, frame {ti}", fontsize=12)
        fig.colorbar(im, ax=ax, fraction=0.046, pad=0.04)
    fig.tight_layout()
    save_path = os.path.join(save_dir, f"loss_components_epoch_{epoch}.png")
    fig.savefig(save_path, dpi=150, bbox_inches="tight")
    logger.info(f"Loss visualization saved to {save_path}")
    plt.close(fig)
    return save_path
