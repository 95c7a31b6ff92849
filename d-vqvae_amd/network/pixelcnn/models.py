"""Class-conditional GatedPixelCNN prior over the 3x3 grid of code indices.

Mirrors DVQ-VAE/network/pixelcnn/models.py:21-198 (same module tree, 140 state_dict keys
for GatedPixelCNN(512, 512, 15)).  This part of the path runs on stock PyTorch-ROCm
operators; it has no hand-written kernel yet.  Differences from the reference, none
visible at its only supported batch size of 1:
  * the mask-'A' zeroing (make_causal) is applied to a masked copy of the weights instead
    of overwriting the parameters on every forward;
  * sampling normalises each row by its own sum (the reference divides by the sum over the
    whole batch, models.py:194) and accepts a torch.Generator.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from dvq_hip.rng import sample_rows


def weights_init(m):
    classname = m.__class__.__name__
    if classname.find('Conv') != -1 and hasattr(m, 'weight') and isinstance(getattr(m, 'weight'), torch.Tensor):
        nn.init.xavier_uniform_(m.weight.data)
        m.bias.data.fill_(0)


class GatedActivation(nn.Module):
    def forward(self, x):
        a, b = x.chunk(2, dim=1)
        return torch.tanh(a) * torch.sigmoid(b)


class GatedMaskedConv2d(nn.Module):
    def __init__(self, mask_type, dim, kernel, residual=True, n_classes=128):
        super().__init__()
        assert kernel % 2 == 1, "Kernel size must be odd"
        self.mask_type = mask_type
        self.residual = residual
        self.kernel = kernel
        self.class_cond_embedding = nn.Embedding(n_classes, 2 * dim)
        self.vert_stack = nn.Conv2d(dim, dim * 2, (kernel // 2 + 1, kernel), 1, (kernel // 2, kernel // 2))
        self.vert_to_horiz = nn.Conv2d(2 * dim, 2 * dim, 1)
        self.horiz_stack = nn.Conv2d(dim, dim * 2, (1, kernel // 2 + 1), 1, (0, kernel // 2))
        self.horiz_resid = nn.Conv2d(dim, dim, 1)
        self.gate = GatedActivation()

    def make_causal(self):
        self.vert_stack.weight.data[:, :, -1].zero_()
        self.horiz_stack.weight.data[:, :, :, -1].zero_()

    def forward(self, x_v, x_h, h):
        wv, wh = self.vert_stack.weight, self.horiz_stack.weight
        if self.mask_type == 'A':
            mv = torch.ones_like(wv); mv[:, :, -1] = 0
            mh = torch.ones_like(wh); mh[:, :, :, -1] = 0
            wv, wh = wv * mv, wh * mh
        k = self.kernel
        h = self.class_cond_embedding(h)
        h_vert = F.conv2d(x_v, wv, self.vert_stack.bias, 1, (k // 2, k // 2))[:, :, :x_v.size(-1), :]
        out_v = self.gate(h_vert + h[:, :, None, None])
        h_horiz = F.conv2d(x_h, wh, self.horiz_stack.bias, 1, (0, k // 2))[:, :, :, :x_h.size(-2)]
        v2h = self.vert_to_horiz(h_vert)
        out = self.gate(v2h + h_horiz + h[:, :, None, None])
        out_h = self.horiz_resid(out) + x_h if self.residual else self.horiz_resid(out)
        return out_v, out_h


class GatedPixelCNN(nn.Module):
    def __init__(self, input_dim=256, dim=128, n_layers=15, n_classes=128):
        super().__init__()
        self.dim = dim
        self.embedding = nn.Embedding(input_dim, dim)
        self.layers = nn.ModuleList()
        for i in range(n_layers):
            self.layers.append(GatedMaskedConv2d('A' if i == 0 else 'B', dim, 5 if i == 0 else 3, i != 0, n_classes))
        self.output_conv = nn.Sequential(nn.Conv2d(dim, 2048, 1), nn.ReLU(True), nn.Conv2d(2048, input_dim, 1))
        self.apply(weights_init)

    def forward(self, x, label):
        shp = x.size() + (-1,)
        x = self.embedding(x.view(-1)).view(shp).permute(0, 3, 1, 2)
        x_v, x_h = x, x
        for layer in self.layers:
            x_v, x_h = layer(x_v, x_h, label)
        return self.output_conv(x_h)

    @torch.no_grad()
    def generate(self, x_start, label, shape=(3, 3), batch_size=64, generator=None, return_probs=False,
                 uniforms=None):
        """models.py:176-198: one full forward per grid cell, cell (i,j) drawn from
        softmax(logits[:, :, i, j]).  `x_start` is accepted and ignored, as in the reference."""
        param = next(self.parameters())
        x = torch.zeros((batch_size, *shape), dtype=torch.int64, device=param.device)
        probs_all = []
        for i in range(shape[0]):
            for j in range(shape[1]):
                logits = self.forward(x, label)
                probs = F.softmax(logits[:, :, i, j], -1)
                probs = probs / probs.sum(dim=1, keepdim=True)
                if return_probs:
                    probs_all.append(probs)
                if uniforms is not None:
                    x[:, i, j] = sample_rows(probs, uniforms[:, i * shape[1] + j])
                else:
                    x[:, i, j] = probs.multinomial(1, generator=generator).squeeze(1)
        if return_probs:
            return x, torch.stack(probs_all, 1)
        return x
