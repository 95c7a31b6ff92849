"""Drop-in for the `mano` package the reference vendors (DVQ-VAE-2/MANO-master/mano):
mano.load(...) / mano.MANO(...) -> nn.Module whose forward runs the fused HIP kernel."""
from .model import MANO, ModelOutput, load  # noqa: F401
