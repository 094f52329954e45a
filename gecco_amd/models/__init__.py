from .activation import GaussianActivation
from .mlp import MLP
from .normalization import AdaGN
from .set_transformer import AttentionPool, Broadcast, BroadcastingLayer, SetTransformer
from .linear_lift import LinearLift
from .ray import GroupNormBNC, RayNetwork
from .feature_pyramid import ConvNeXtExtractor, FeaturePyramidContext, FeaturePyramidExtractor
