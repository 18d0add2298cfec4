"""SMR-SELD loss (drop-in for the reference's ``loss.py``): ``SMRSELDLoss(loss_type, w_class, w_aiur,
w_cl, grid_size, class_weights)``; ``forward(y_pred, y_true) -> (loss tensor, {name: float})``.

Active term (loss.py:149-172): ``w_class * class_loss`` where class_loss is
  'mse' : mean((softmax(logits) - y)^2)             loss.py:43-54   -> fused HIP kernel on ROCm devices
  'ce'  : weighted CrossEntropy(argmax(y))          loss.py:27-41   -> stock ops
The AIUR and converging-localisation terms (loss.py:56-146) are, like upstream, not part of ``forward`` by default;
``three_term=True`` (Config.THREE_TERM_LOSS) sums all three on the probabilities as the reference's
``smrl_seld_gaussian.py:946-1072`` does -- on a ROCm device in one fused pass (csrc/loss3.hip).

``y_true`` may be the dense float tensor [B,T,G,M] the reference uses or the compact uint16 class
mask [B,T,G] produced by the label rasteriser (bit c = class c; mask 0 = background).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _FusedSoftmaxMSE(torch.autograd.Function):
    """loss and d(loss)/d(logits) from ONE pass of seld_softmax_mse (csrc/loss.hip)."""

    @staticmethod
    def forward(ctx, logits, labels):
        import seld_native
        n = logits.numel()
        loss, grad = seld_native.softmax_mse(logits, labels, grad_scale=(2.0 / n) if logits.requires_grad else None)
        ctx.save_for_backward(grad if grad is not None else torch.empty(0))
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        import seld_native
        (grad,) = ctx.saved_tensors
        # the saved gradient is consumed exactly once: scale it in place by the upstream gradient (a device scalar;
        # exactly 1 under loss.backward(), in which case the kernel returns without touching the 145 MB tensor)
        return seld_native.scale_by_device_scalar_(grad, grad_out), None


class _FusedThreeTerm(torch.autograd.Function):
    """total = w_class * MSE + w_aiur * AIUR + w_cl * CL and d(total)/d(logits) from one pass (csrc/loss3.hip);
    returns the four scalars (total, mse, aiur, cl) as one tensor -- only element 0 carries a gradient."""

    @staticmethod
    def forward(ctx, logits, labels, grid, w_class, w_aiur, w_cl):
        import seld_native
        terms, grad = seld_native.smr_loss(logits, labels, grid, w_class, w_aiur, w_cl, logits.requires_grad)
        ctx.save_for_backward(grad if grad is not None else torch.empty(0))
        return terms

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        import seld_native
        (grad,) = ctx.saved_tensors
        return seld_native.scale_by_device_scalar_(grad, grad_out[:1]), None, None, None, None, None


def mask_to_dense(mask, num_classes):
    """uint16 class mask [..., G] -> float [..., G, M] with the background rule (dataset.py:110-117)."""
    bits = (mask.to(torch.int32).unsqueeze(-1) >> torch.arange(num_classes, device=mask.device)) & 1
    dense = bits.to(torch.float32)
    dense[..., num_classes - 1] = torch.where(mask == 0, 1.0, dense[..., num_classes - 1])
    return dense


class SMRSELDLoss(nn.Module):
    fused_enabled = True     # flipped by the trainer from Config.FUSED_LOSS

    def __init__(self, loss_type="ce", w_class=1.0, w_aiur=0.5, w_cl=0.5, grid_size=None, class_weights=None,
                 three_term=False):
        """``three_term``: total = w_class * class + w_aiur * AIUR + w_cl * CL, the auxiliary terms on the probabilities,
        as smrl_seld_gaussian.py:1058-1072 sums them (upstream's modular loss.py keeps them commented out of ``forward``,
        loss.py:158-166: the default here).  Config.THREE_TERM_LOSS selects it in the trainer."""
        super().__init__()
        self.three_term = bool(three_term)
        self.last_terms = None               # device tensor (total, class, aiur, cl) of the last three-term evaluation
        self.loss_type = loss_type
        self.w_class = w_class
        self.w_aiur = w_aiur
        self.w_cl = w_cl
        self.eps = 1e-10
        self.I, self.J = grid_size if grid_size is not None else (None, None)
        self.ce_loss = nn.CrossEntropyLoss(weight=class_weights) if class_weights is not None else nn.CrossEntropyLoss()

    # ---- class terms ---------------------------------------------------------------------------
    def _dense(self, y_true, num_classes):
        return mask_to_dense(y_true, num_classes) if y_true.dtype == torch.uint16 else y_true

    def class_ce_loss(self, y_pred, y_true):
        m = y_pred.shape[-1]
        target = torch.argmax(self._dense(y_true, m), dim=-1)
        return self.ce_loss(y_pred.reshape(-1, m).float(), target.reshape(-1))

    def class_mse_loss(self, y_pred, y_true):
        if y_pred.is_cuda and self.fused_enabled and y_pred.shape[-1] == 14 \
                and y_pred.dtype in (torch.float32, torch.bfloat16):
            labels = y_true if y_true.dtype == torch.uint16 else y_true.to(torch.float32)
            return _FusedSoftmaxMSE.apply(y_pred, labels)
        probs = F.softmax(y_pred.float(), dim=-1)
        return F.mse_loss(probs, self._dense(y_true, y_pred.shape[-1]))

    # ---- auxiliary terms (loss.py:56-146), operate on probabilities ------------------------------
    def aiur_loss(self, y_pred, y_true):
        """1 - mean IoU between predicted and true event cells per (batch, frame); argmax based."""
        bg = y_pred.shape[-1] - 1
        pred_evt = (torch.argmax(y_pred, dim=-1) != bg).float()
        true_evt = (torch.argmax(y_true, dim=-1) != bg).float()
        inter = (pred_evt * true_evt).sum(dim=-1)
        union = pred_evt.sum(dim=-1) + true_evt.sum(dim=-1) - inter
        iou = torch.where(union > 0, inter / (union + 1e-8), torch.ones_like(inter))
        return 1.0 - iou.mean()

    def converging_localization_loss(self, y_pred, y_true):
        """Non-background probability weighted by an 8-neighbour 'attention' map of the targets
        (circular padding on the I x J grid), averaged over frames that contain events."""
        b, t, g, m = y_pred.shape
        rows, cols = (self.I, self.J) if self.I is not None and self.J is not None else (int(math.sqrt(g)),) * 2
        true_nonbg = y_true.view(b, t, rows, cols, m)[..., :-1].sum(dim=-1)
        pred_nonbg = y_pred.view(b, t, rows, cols, m)[..., :-1].sum(dim=-1)
        is_evt = true_nonbg > 0.01
        n_bac = (true_nonbg < 0.01).sum(dim=(2, 3), keepdim=True).float()
        n_non = is_evt.sum(dim=(2, 3), keepdim=True).float()
        y_prime = torch.where(is_evt, (-(n_bac / (n_non + self.eps))).expand_as(true_nonbg), torch.ones_like(true_nonbg))
        padded = F.pad(y_prime, (1, 1, 1, 1), mode="circular")
        diff_sum = torch.zeros_like(y_prime)
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                if di or dj:
                    diff_sum += padded[:, :, 1 + di:rows + 1 + di, 1 + dj:cols + 1 + dj] - y_prime
        y_at = y_prime + diff_sum / 8.0
        has_events = (n_non > 0).float()
        return ((pred_nonbg * y_at) * has_events).sum() / (has_events.sum() * rows * cols + self.eps)

    # ---- forward -------------------------------------------------------------------------------
    def three_term_tensor(self, y_pred, y_true):
        """(total, class term) of the three-term loss; ``self.last_terms`` keeps (total, class, aiur, cl) on the device."""
        rows, cols = (self.I, self.J) if self.I is not None else (int(math.sqrt(y_pred.shape[-2])),) * 2
        if y_pred.is_cuda and self.fused_enabled and self.loss_type == "mse" and y_pred.shape[-1] == 14 \
                and y_pred.dtype in (torch.float32, torch.bfloat16) and 3 <= rows and 3 <= cols and rows * cols <= 1024:
            labels = y_true if y_true.dtype == torch.uint16 else y_true.to(torch.float32)
            terms = _FusedThreeTerm.apply(y_pred, labels, (rows, cols), float(self.w_class), float(self.w_aiur),
                                          float(self.w_cl))
            self.last_terms = terms.detach()
            return terms[0], terms[1].detach()
        dense = self._dense(y_true, y_pred.shape[-1]).float()
        probs = F.softmax(y_pred.float(), dim=-1)
        term = F.mse_loss(probs, dense) if self.loss_type == "mse" else self.class_ce_loss(y_pred, y_true)
        aiur = self.aiur_loss(probs, dense)
        cl = self.converging_localization_loss(probs, dense)
        total = self.w_class * term + self.w_aiur * aiur + self.w_cl * cl
        self.last_terms = torch.stack((total.detach(), term.detach(), aiur.detach(), cl.detach()))
        return total, term.detach()

    def loss_tensor(self, y_pred, y_true):
        """The scalar the optimiser differentiates, without the host sync of ``forward``'s breakdown."""
        if self.three_term:
            return self.three_term_tensor(y_pred, y_true)
        term = self.class_mse_loss(y_pred, y_true) if self.loss_type == "mse" else self.class_ce_loss(y_pred, y_true)
        return (term if self.w_class == 1.0 else self.w_class * term), term      # no multiply (+ its backward) by 1

    def forward(self, y_pred: torch.Tensor, y_true: torch.Tensor):
        total, term = self.loss_tensor(y_pred, y_true)
        if self.three_term:                  # the breakdown of smrl_seld_gaussian.py:1064-1068
            _, cls, aiur, cl = self.last_terms.tolist()
            return total, {f"class_{self.loss_type}": cls, "aiur": aiur, "cl": cl}
        return total, {f"class_{self.loss_type}": float(term.item())}
