"""devis_amd - the DEviS evidential U-Net hot path on MI355X (gfx950) HIP kernels."""
from .unet3d import Unet, Unetdrop            # noqa: F401
from .attunet3d import AttUnet                # noqa: F401
from .vnet3d import VNet                      # noqa: F401
from .vnet2d import VNet2D                    # noqa: F401
from .unet2d import Unet2D, Unet2DDRIVE       # noqa: F401
from .trustworthy_seg import TMSU             # noqa: F401
from .predict import tailor_and_concat        # noqa: F401
from . import criterions, ops                 # noqa: F401
