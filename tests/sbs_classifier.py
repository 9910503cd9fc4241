"""The reference's three-layer ConvSBS classifier (mnist.py:169-284: two layers of two 9-core snake strings with a
two-valued middle core, a final string with ten labels on its middle core), shared by the data-parallel tests."""
import torch

from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
from dctn_amd.conv_sbs_spec import SBSSpecCore
from dctn_amd.pos2d import Pos2D

SNAKE_A = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]   # mnist.py:190-199
SNAKE_B = [(0, 0), (1, 0), (2, 0), (2, 1), (1, 1), (0, 1), (0, 2), (1, 2), (2, 2)]   # mnist.py:201-210


def string(pos, mid):
    return tuple(SBSSpecCore(Pos2D(*p), mid if i == 4 else 1) for i, p in enumerate(pos))


class ConvSBSClassifier(torch.nn.Module):
    """reference_form=False (the data-parallel tests): tanh(scale * output) between the layers with fixed scales, so that
    27 random cores neither underflow nor need data to be usable.  reference_form=True (the timing tools): exactly the
    reference's model - `forward` is the chain of layers and the mean over positions (mnist.py:255-263), `calibrate` is its
    `scale_layers_using_batch` (mnist.py:265-283: every string divided by the std of its output on a batch)."""

    def __init__(self, bond: int = 4, ring: bool = False, labels: int = 10, reference_form: bool = False):
        super().__init__()
        init = DumbNormalInitialization((2 * bond) ** -0.5 * 1.3)
        two = (string(SNAKE_A, 2), string(SNAKE_B, 2))
        self.layers = torch.nn.ModuleList([
            ManyConvSBS(1, 2, bond, ring, two, (init,) * 2),
            ManyConvSBS(2, 2, bond, ring, two, (init,) * 2),
            ManyConvSBS(2, 2, bond, ring, (string(SNAKE_A, labels),), (init,)),
        ])
        self.reference_form = reference_form
        self.scales = [1.0, 1.0, 1.0]   # fixed per-layer output scales (see calibrate): 27 random cores would underflow

    def forward(self, x):   # x: (1, B, H, W, 2) -> (B, labels)
        inter = (x[0],)
        for layer, scale in zip(self.layers, self.scales):
            inter = layer(inter) if self.reference_form else tuple(torch.tanh(o * scale) for o in layer(inter))
        (out,) = inter
        return out.reshape(out.shape[0], -1, out.shape[-1]).mean(1)

    @torch.no_grad()
    def calibrate(self, x):
        inter = (x[0],)
        for k, layer in enumerate(self.layers):
            outs = layer(inter)
            if self.reference_form:
                for st, t in zip(layer.strings, outs):
                    std = float(t.std())
                    if std != 0.0:
                        st.multiply_by_scalar(1.0 / std)
                inter = layer(inter)
            else:
                self.scales[k] = 1.0 / float(torch.cat([o.reshape(-1) for o in outs]).abs().median())
                inter = tuple(torch.tanh(o * self.scales[k]) for o in outs)
