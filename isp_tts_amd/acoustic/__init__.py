from .alignment import Aligner, AlignerOutput, ConvAttention  # noqa: F401
from .temporal_adaptor import (FlowTemporalAdaptor, FlowTransformerTemporalModule, LengthRegulator,  # noqa: F401
                               TemporalAdaptorOutput, TemporalAverager, TransformerTemporalModule, generate_soft_path)
from .model import AcousticModel, AcousticModelOutput  # noqa: F401
