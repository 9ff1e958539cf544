from .graph import GraphConvolution  # noqa: F401
