from .knn import KNNBenchmarkModule, macro_metrics  # noqa: F401
from .resnet import ResNet18, create_model  # noqa: F401
from .simclr import SimCLR  # noqa: F401
from .vit import VisionTransformer, vit_base, vit_small, vit_tiny  # noqa: F401
from .dino import DINO, DINOViT  # noqa: F401
from .mae import MAE, MAEBackbone, MAEDecoder, SimMIM, masked_autoencoder, vit_b_32  # noqa: F401
from .moco import MoCo  # noqa: F401
from .siamese import BYOL, FastSiam, SimSiam  # noqa: F401
from .evals import LinearClassifier, MultilabelLinearClassifier, SupervisedR18, fit_linear_probe  # noqa: F401
from .dclw import DCLW  # noqa: F401
from .barlow import BarlowTwins  # noqa: F401
from .vicreg import VICReg  # noqa: F401
from .swav import SwaV  # noqa: F401
from .msn import MSN, PMSN  # noqa: F401
