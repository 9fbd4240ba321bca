from .resnet import ResNet18, create_model  # noqa: F401
