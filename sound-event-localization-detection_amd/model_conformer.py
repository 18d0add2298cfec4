"""Conformer for SELD (drop-in for the reference's ``model_conformer.py``: same class names,
constructor arguments and ``state_dict`` keys).

Block = half-step FFN -> multi-head self-attention -> convolution module -> half-step FFN ->
LayerNorm (model_conformer.py:98-113).  No mask and no positional encoding: the sequence is
always one 250-frame window.  The attention contraction goes through
``scaled_dot_product_attention`` (fused softmax(QK^T/sqrt(d))V on the matrix cores, never
materialising the [B, H, 250, 250] score tensor in HBM) -- numerically the computation at
model_conformer.py:58-62.
"""
import torch
import torch.nn as nn

from seld_layernorm import norm as layer_norm_of, run_head
from seld_linear import SeldLinear
import torch.nn.functional as F

from model_crnn import ConvBlock, build_cnn_encoder, run_cnn_encoder  # noqa: F401


class Swish(nn.Module):
    def forward(self, x):
        return F.silu(x)            # x * sigmoid(x), model_conformer.py:6-8


class _HalfStepResidual(torch.autograd.Function):
    """x + scale * dropout(z) (the half-step residual of model_conformer.py:27-29) with the framework's own dropout
    (``native_dropout``: the reference's random stream) but without the separate scaling kernels: forward = dropout +
    one ``add(alpha=scale)``, backward = ONE masked scaling with the residual's factor folded into the dropout's."""

    @staticmethod
    def forward(ctx, x, z, p, scale):
        d, mask = torch.ops.aten.native_dropout(z, p, True)
        ctx.save_for_backward(mask)
        ctx.factor = scale / (1.0 - p)
        return torch.add(x, d, alpha=scale)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g, torch.ops.aten.native_dropout_backward(g, mask, ctx.factor), None, None


class FeedForward(nn.Module):
    """Pre-norm position-wise FFN with a half-step residual (model_conformer.py:10-29)."""

    def __init__(self, d_model, d_ff=2048, dropout=0.1):
        super().__init__()
        self.linear1 = SeldLinear(d_model, d_ff)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = SeldLinear(d_ff, d_model)
        self.norm = nn.LayerNorm(d_model)
        self.swish = Swish()

    def forward(self, x):
        y = self.dropout(self.swish(self.linear1(layer_norm_of(self.norm, x))))
        z = self.linear2(y)
        if x.is_cuda and z.dtype == x.dtype:
            if self.training and torch.is_grad_enabled() and 0.0 < self.dropout.p < 1.0:
                return _HalfStepResidual.apply(x, z, self.dropout.p, 0.5)
            if not self.training or self.dropout.p == 0.0:
                return torch.add(x, z, alpha=0.5)
        return x + 0.5 * self.dropout(z)


class MultiHeadSelfAttention(nn.Module):
    """Pre-norm MHSA, separate biased q/k/v/o projections (model_conformer.py:31-69)."""

    def __init__(self, d_model, n_heads=4, dropout=0.1):
        super().__init__()
        self.d_model = d_model
        self.n_heads = n_heads
        self.head_dim = d_model // n_heads
        assert self.head_dim * n_heads == d_model, "d_model must be divisible by n_heads"
        self.w_q = SeldLinear(d_model, d_model)
        self.w_k = SeldLinear(d_model, d_model)
        self.w_v = SeldLinear(d_model, d_model)
        self.w_o = SeldLinear(d_model, d_model)
        self.dropout = nn.Dropout(dropout)
        self.norm = nn.LayerNorm(d_model)

    fused_qkv = True        # Config.FUSED_QKV via trainer.prepare_model_for_device

    def _heads(self, proj, x):
        b, t, _ = x.shape
        return proj(x).view(b, t, self.n_heads, self.head_dim).transpose(1, 2)      # [B, H, T, Dh]

    def pack_parameters(self):
        """q / k / v weights (and biases) as consecutive slices of one buffer: the three projections of the same input are
        then ONE GEMM on a view (seld_pack.py); names, values and ``state_dict`` unchanged."""
        import seld_pack
        seld_pack.pack((self.w_q.weight, self.w_k.weight, self.w_v.weight))
        seld_pack.pack((self.w_q.bias, self.w_k.bias, self.w_v.bias))

    def _qkv(self, y):
        """The three projections model_conformer.py:52-55 applies to the same normalised input, as one [3D, D] GEMM
        forward, one data-gradient GEMM and one weight-gradient product backward (six launches fewer per block and
        direction at sizes where a launch is all a GEMM costs: 8000 x 256 x 256)."""
        import seld_pack
        from seld_linear import _Linear
        b, t, d = y.shape
        w = seld_pack.join(self.w_q.weight, self.w_k.weight, self.w_v.weight)
        bias = seld_pack.join(self.w_q.bias, self.w_k.bias, self.w_v.bias)
        qkv = _Linear.apply(y, w, bias).view(b, t, 3, self.n_heads, self.head_dim)
        q, k, v = qkv.unbind(dim=2)
        return q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)              # [B, H, T, Dh] views

    def forward(self, x):
        b, t, d = x.shape
        y = layer_norm_of(self.norm, x)
        import seld_pack
        if (self.fused_qkv and y.is_cuda and y.dtype in (torch.float32, torch.bfloat16)
                and seld_pack.adjacent(self.w_q.weight, self.w_k.weight, self.w_v.weight)
                and seld_pack.adjacent(self.w_q.bias, self.w_k.bias, self.w_v.bias)):
            q, k, v = self._qkv(y)
        else:
            q, k, v = self._heads(self.w_q, y), self._heads(self.w_k, y), self._heads(self.w_v, y)
        ctx = F.scaled_dot_product_attention(q, k, v, dropout_p=self.dropout.p if self.training else 0.0)
        ctx = ctx.transpose(1, 2).reshape(b, t, d)
        return x + self.dropout(self.w_o(ctx))


class ConformerConvModule(nn.Module):
    """LN -> pointwise 2x -> GLU -> depthwise k -> BN -> Swish -> pointwise -> dropout (:71-96)."""

    def __init__(self, d_model, kernel_size=31, dropout=0.1):
        super().__init__()
        self.layer_norm = nn.LayerNorm(d_model)
        self.pointwise_conv1 = nn.Conv1d(d_model, d_model * 2, kernel_size=1, stride=1, padding=0, bias=True)
        self.glu = nn.GLU(dim=1)
        self.depthwise_conv = nn.Conv1d(d_model, d_model, kernel_size=kernel_size, stride=1,
                                        padding=(kernel_size - 1) // 2, groups=d_model, bias=True)
        self.batch_norm = nn.BatchNorm1d(d_model)
        self.swish = Swish()
        self.pointwise_conv2 = nn.Conv1d(d_model, d_model, kernel_size=1, stride=1, padding=0, bias=True)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        if x.is_cuda:
            import seld_dwconv
            if seld_dwconv.applicable(self, x):                       # channels-last evaluation (csrc/dwconv.hip)
                return seld_dwconv.conv_module_forward(self, x)
        y = layer_norm_of(self.layer_norm, x).transpose(1, 2)        # [B, D, T]
        y = self.glu(self.pointwise_conv1(y))
        y = self.swish(self.batch_norm(self.depthwise_conv(y)))
        y = self.dropout(self.pointwise_conv2(y))
        return x + y.transpose(1, 2)


class ConformerBlock(nn.Module):
    def __init__(self, d_model, n_heads=4, d_ff=1024, kernel_size=31, dropout=0.1):
        super().__init__()
        self.ff1 = FeedForward(d_model, d_ff, dropout)
        self.attn = MultiHeadSelfAttention(d_model, n_heads, dropout)
        self.conv = ConformerConvModule(d_model, kernel_size, dropout)
        self.ff2 = FeedForward(d_model, d_ff, dropout)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, x):
        return layer_norm_of(self.norm, self.ff2(self.conv(self.attn(self.ff1(x)))))


class SELD_Conformer(nn.Module):
    """CNN encoder (shared with the CRNN) -> Linear 2048->d_model -> N Conformer blocks -> grid head
    (model_conformer.py:115-215)."""

    def __init__(self, n_channels=4, n_mels=64, grid_size=(18, 36), num_classes=14,
                 cnn_channels=[64, 128, 256, 512],
                 conf_d_model=256, conf_n_heads=4, conf_n_layers=2, conf_kernel_size=31, dropout=0.3):
        super().__init__()
        self.I, self.J = grid_size
        self.grid_cells = self.I * self.J
        self.num_classes = num_classes
        self.cnn_blocks, self.cnn_out_channels, self.cnn_out_freq = build_cnn_encoder(n_channels, n_mels, cnn_channels)
        self.cnn_feat_size = self.cnn_out_channels * self.cnn_out_freq
        self.proj = SeldLinear(self.cnn_feat_size, conf_d_model)
        self.conformer_blocks = nn.ModuleList([
            ConformerBlock(d_model=conf_d_model, n_heads=conf_n_heads, d_ff=conf_d_model * 4,
                           kernel_size=conf_kernel_size, dropout=dropout)
            for _ in range(conf_n_layers)])
        self.fnn = nn.Sequential(
            SeldLinear(conf_d_model, 512),
            nn.LayerNorm(512),
            nn.ReLU(),
            nn.Dropout(dropout),
            SeldLinear(512, self.grid_cells * num_classes),
        )

    def forward(self, x):
        batch, frames = x.shape[0], x.shape[1]
        y = None
        if x.is_cuda:
            from model_crnn import run_cnn_blocks
            from seld_linear import linear_on_channels_last_features
            enc = run_cnn_blocks(self.cnn_blocks, x, inner_cut=False)      # [B, C, T, F], channels-last memory
            y = linear_on_channels_last_features(self.proj, enc)           # no feature copy: the weight's columns move
            if y is None:
                enc = enc.permute(0, 2, 1, 3)
                y = self.proj(enc.reshape(batch, frames, -1))
        if y is None:
            y = self.proj(run_cnn_encoder(self.cnn_blocks, x))
        for block in self.conformer_blocks:
            y = block(y)
        return run_head(self.fnn, y).view(batch, frames, self.grid_cells, self.num_classes)
