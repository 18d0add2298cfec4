"""GRU module with the reference's parameter names (``rnn.weight_ih_l0_reverse`` ...) whose
recurrence runs in the persistent HIP BiGRU kernel on ROCm devices.

``SeldGRU`` subclasses ``torch.nn.GRU`` so construction, initialisation order (identical RNG
consumption -> identical seeded weights) and ``state_dict`` keys are exactly those of the
``nn.GRU`` at model_crnn.py:65-72.  ``forward`` dispatches:
  * GPU tensor and the fused kernel applicable  -> ``seld_gru`` (HIP, see csrc/gru.hip)
  * CPU tensor (plumbing mode)                   -> stock ``nn.GRU.forward``
"""
import torch
import torch.nn as nn


class SeldGRU(nn.GRU):
    fused_enabled = True     # flipped by the trainer from Config.FUSED_GRU

    def forward(self, input, hx=None):
        if input.is_cuda and self.fused_enabled and hx is None:
            try:
                import seld_gru
            except ImportError:
                seld_gru = None
            if seld_gru is not None and seld_gru.applicable(self, input):
                return seld_gru.bigru_forward(self, input)
        return super().forward(input, hx)
