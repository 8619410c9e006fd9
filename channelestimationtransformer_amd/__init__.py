"""MI355X-native channel-predictor + LMMSE inference path.

Drop-in counterparts of the reference's `models.model` classes and helpers; the arithmetic
runs in hand-written HIP kernels for gfx950 (csrc/, C ABI in include/cet.h)."""
from .hostpipe import HostPipeline
from .lmmse import (LMMSEEqualizer, LSChannelEstimator, ResourceGrid, StreamManagement, lmmse,
                    lmmse_llr)
from .model import GRU, LSTM, RNN, Informer, InformerStack, InformerStack_e2e
from .utils import LoadBatch, decoder_input, load_batch_device, real2complex

__all__ = ["Informer", "InformerStack", "InformerStack_e2e", "GRU", "LSTM", "RNN", "LoadBatch",
           "real2complex", "load_batch_device", "decoder_input", "lmmse", "lmmse_llr", "LMMSEEqualizer", "LSChannelEstimator",
           "ResourceGrid", "StreamManagement", "HostPipeline"]
