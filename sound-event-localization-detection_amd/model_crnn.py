"""CRNN for SELD (drop-in for the reference's ``model_crnn.py``; same constructor, same
``state_dict`` keys, same [B,T,C,F] -> [B,T,648,14] logits contract).

  4 x (Conv3x3 -> BatchNorm -> ReLU -> MaxPool over frequency)      model_crnn.py:5-17, :34-57
  2-layer bidirectional GRU(2048 -> 256)                            model_crnn.py:65-72
  Linear 512->512, LayerNorm, ReLU, Dropout, Linear 512->9072       model_crnn.py:77-83

On a ROCm device the convolutions run channels-last under bf16 autocast (MFMA via MIOpen /
hipBLASLt, set up by the trainer) and the recurrence runs in the persistent HIP BiGRU kernel
(``seld_rnn.SeldGRU``); on a CPU everything is stock PyTorch (the reference's own CPU case).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from seld_layernorm import run_head
from seld_linear import SeldLinear
from seld_rnn import SeldGRU


class _Conv3x3(torch.autograd.Function):
    """3x3 / stride 1 / pad 1 convolution whose DATA gradient is evaluated as a forward convolution of dy with the
    transposed, flipped weights.  Same numbers (a different summation order); on gfx950 MIOpen's forward solvers beat
    its backward-data solvers on the encoder's shapes (tools/bench_conv_bwd.py, bf16 channels-last, batch 32:
    256->512 channels 219 -> 170 us, 128->256 130 -> 115 us, 64->128 100 -> 87 us including the weight transform)."""

    enabled = True

    @staticmethod
    def forward(ctx, x, weight):
        cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
        with torch.autocast(device_type="cuda", enabled=False):
            xc, wc = x.to(cdt), weight.to(cdt)
            y = F.conv2d(xc, wc, padding=1)
        ctx.save_for_backward(xc, wc)
        ctx.dtypes = (x.dtype, weight.dtype)
        # the input is a backward-pass cut leaf (seld_cut.boundary): this convolution's backward ends a stage
        ctx.after_cut = bool(x.is_leaf and x.requires_grad)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        x_dtype, w_dtype = ctx.dtypes
        with torch.autocast(device_type="cuda", enabled=False):
            dy = dy.to(xc.dtype).contiguous(memory_format=torch.channels_last)
            dx = None
            if ctx.needs_input_grad[0]:
                if wc.is_contiguous(memory_format=torch.channels_last):
                    import seld_native
                    wt = seld_native.conv_weight_flip_transpose(wc)                 # one launch (flip + copy are two)
                else:
                    wt = wc.transpose(0, 1).flip(2, 3).contiguous(memory_format=torch.channels_last)
                import seld_overlap
                seld_overlap.release_held(dy.device)      # (wgrad_order "late": the previous block's weight gradient)
                dx = F.conv2d(dy, wt, padding=1)
                if dx.dtype != x_dtype:
                    dx = dx.to(x_dtype)
            import seld_overlap
            if seld_overlap.conv_wgrad_side and seld_overlap.enabled:
                # side stream, beside the rest of the data-gradient chain (seld_overlap.launch_now); the stepper joins
                dw = torch.empty_like(wc, dtype=w_dtype)

                def job():
                    dw.copy_(torch.ops.aten.convolution_backward(dy, xc, wc, None, (1, 1), (1, 1), (1, 1), False, (0, 0),
                                                                 1, (False, True, False))[1])
                seld_overlap.launch_now(dy.device, [dy, xc, wc, dw], job, last_of_stage=ctx.after_cut, outputs=[dw], hold=True)
                return dx, dw.view_as(dw)          # a fresh alias: autograd takes it over instead of cloning (seld_overlap)
            dw = torch.ops.aten.convolution_backward(dy, xc, wc, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                     (False, True, False))[1]
        return dx, dw if dw.dtype == w_dtype else dw.to(w_dtype)


def conv3x3(c, x):
    """``c(x)`` for an nn.Conv2d; 3x3 / stride 1 / pad 1 / bias-free convolutions on channels-last GPU activations go
    through ``_Conv3x3`` when gradients are being recorded."""
    if (x.is_cuda and _Conv3x3.enabled and c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1)
            and c.dilation == (1, 1) and c.groups == 1 and c.bias is None and c.padding_mode == "zeros"
            and x.is_contiguous(memory_format=torch.channels_last) and torch.is_grad_enabled()
            and (x.requires_grad or c.weight.requires_grad)):
        return _Conv3x3.apply(x, c.weight)
    return c(x)


class ConvBlock(nn.Module):
    """conv -> bn -> relu -> optional max-pool; attribute names fixed by the checkpoint format."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), pool_size=None):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = nn.ReLU(inplace=True)
        self.pool = nn.MaxPool2d(pool_size) if pool_size else None

    def forward(self, x):
        x = conv3x3(self.conv, x)
        if x.is_cuda:
            import seld_convtail
            if seld_convtail.applicable(self, x):           # fused BN -> ReLU -> pool (csrc/convtail.hip)
                return seld_convtail.conv_tail(self, x)
        x = self.act(self.bn(x))
        return x if self.pool is None else self.pool(x)


def build_cnn_encoder(n_channels, n_mels, cnn_channels, max_pools=4):
    """The shared CRNN / Conformer encoder: frequency is halved by each of the first four blocks,
    time is never pooled.  Returns (ModuleList, out_channels, out_freq)."""
    blocks = nn.ModuleList()
    channels, freq = n_channels, n_mels
    for depth, width in enumerate(cnn_channels):
        pool = (1, 2) if depth < max_pools else None
        blocks.append(ConvBlock(channels, width, pool_size=pool))
        channels = width
        if pool:
            freq //= pool[1]
    return blocks, channels, freq


def run_cnn_blocks(blocks, x, inner_cut=True):
    """[B, T, C, F] -> [B, C_out, T, F_out] (channels-last memory on a GPU).  ``inner_cut``: also mark the backward-pass
    cut before the last block (worth a stage only when something is handed over to it: the CRNN's GRU layer 0 weight
    gradients; for the Conformer that stage would complete no gradient bucket of its own)."""
    x = x.permute(0, 2, 1, 3)
    if x.is_cuda:
        # one copy does the layout change AND the cast the first convolution would otherwise do under autocast
        dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() and x.is_floating_point() else x.dtype
        x = x.to(dtype=dtype, memory_format=torch.channels_last)
        import seld_convtail
        import seld_cut
        with seld_convtail.batched_counters():          # one launch for the blocks' num_batches_tracked increments
            for depth, block in enumerate(blocks):
                x = block(x)
                # backward-pass cut points of the data-parallel captured step (identity otherwise): before the last
                # block -- GRU layer 0's weight gradients, carried over from the recurrent stage, are computed beside
                # that block's data gradient and travel under the rest -- and at the encoder's output
                if inner_cut and depth == len(blocks) - 2:
                    x = seld_cut.boundary(x, level=2)
        return seld_cut.boundary(x, level=1)
    for block in blocks:
        x = block(x)
    return x


def run_cnn_encoder(blocks, x):
    """[B, T, C, F] -> [B, T, C_out * F_out] (model_crnn.py:106-116)."""
    x = run_cnn_blocks(blocks, x).permute(0, 2, 1, 3)
    return x.reshape(x.shape[0], x.shape[1], -1)


class SELD_CRNN(nn.Module):
    def __init__(self, n_channels=4, n_mels=64, grid_size=(18, 36), num_classes=14,
                 cnn_channels=[64, 128, 256, 512], rnn_hidden=256, rnn_layers=2, dropout=0.3):
        super().__init__()
        self.I, self.J = grid_size
        self.grid_cells = self.I * self.J
        self.num_classes = num_classes
        self.cnn_blocks, self.cnn_out_channels, self.cnn_out_freq = build_cnn_encoder(n_channels, n_mels, cnn_channels)
        self.rnn_input_size = self.cnn_out_channels * self.cnn_out_freq
        self.rnn = SeldGRU(input_size=self.rnn_input_size, hidden_size=rnn_hidden, num_layers=rnn_layers,
                           batch_first=True, bidirectional=True, dropout=dropout if rnn_layers > 1 else 0)
        self.rnn_out_size = rnn_hidden * 2
        self.fnn = nn.Sequential(
            SeldLinear(self.rnn_out_size, 512),
            nn.LayerNorm(512),
            nn.ReLU(),
            nn.Dropout(dropout),
            SeldLinear(512, self.grid_cells * num_classes),
        )

    def forward(self, x):
        """x [B, T, C, F] -> logits [B, T, G, M]."""
        batch, frames = x.shape[0], x.shape[1]
        y = run_cnn_blocks(self.cnn_blocks, x)                                  # [B, C, T, F]
        if y.is_cuda and y.is_contiguous(memory_format=torch.channels_last) and SeldGRU.fused_enabled:
            # channels-last memory is [B][T][F][C]: the (frequency, channel)-ordered feature vector is a VIEW of it.
            # The HIP BiGRU path takes it as is and permutes the columns of layer 0's W_ih instead (6 MB of weights
            # rather than two 33 MB activation copies, forward and backward); same numbers.
            import seld_gru
            import seld_overlap
            feats_fc = y.permute(0, 2, 3, 1).reshape(batch, frames, -1)
            if seld_gru.applicable(self.rnn, feats_fc):
                # weight gradients of the head and of GRU layer 1 go to a side stream, under the backward recurrences
                # (seld_overlap.py); their identity nodes have to be created here, before GRU layer 0's
                overlap = seld_overlap.active(feats_fc) and self.training
                if overlap:
                    seld_overlap.defer_linear(self.fnn[0], self.fnn[4])
                feats, _ = seld_gru.bigru_forward(self.rnn, feats_fc, feature_cf=(y.shape[1], y.shape[3]),
                                                  overlap=overlap, need_hn=False)
                return run_head(self.fnn, feats).view(batch, frames, self.grid_cells, self.num_classes)
        feats = y.permute(0, 2, 1, 3).reshape(batch, frames, -1)                # (channel, frequency) order
        feats, _ = self.rnn(feats)
        return run_head(self.fnn, feats).view(batch, frames, self.grid_cells, self.num_classes)
