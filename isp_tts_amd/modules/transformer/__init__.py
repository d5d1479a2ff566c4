from .attend import Attend, AttentionIntermediates  # noqa: F401
from .attention import Attention, AttentionConfig, AttentionSharedIntermediates  # noqa: F401
from .feedforward import FeedForward, FeedForwardConfig  # noqa: F401
from .normalization import LayerNorm, AdaptiveLayerNorm  # noqa: F401
from .embeddings import ALiBiPositionalBias, LearnedALiBiPositionalBias, TimePositionalEmbedding  # noqa: F401
from .transformer import (TransformerLayerConfig, TransformerLayer, TransformerConfig, Transformer,  # noqa: F401
                          TransformerOutput, TransformerLayerOutput, TransformerLayerIntermediates)
