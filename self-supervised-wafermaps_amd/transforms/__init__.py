from .augmentations import (  # noqa: F401
    DieNoise, DPWTransform, MedianFilter, RandomOneOf, ViewSpec, augment_views, get_base_transforms,
    get_inference_transforms, multicrop_view, sample_view_params,
)
from .utils import NORMALIZE_STATS  # noqa: F401
from .views import BaseViewTransform, InferenceTransform, MultiCropTransform, MultiViewTransform, Views  # noqa: F401
from .collate import (  # noqa: F401
    BaseCollateFunction, MultiViewCollateFunction, WaferDINOCOllateFunction, WaferImageCollateFunction,
    WaferMAECollateFunction2, WaferMSNCollateFunction, WaferSwaVCollateFunction, rgb_scale,
)
