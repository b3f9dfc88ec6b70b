"""hybridbackend_amd -- MI355X-native sharded-embedding engine behind HybridBackend's
custom-op surface.  Only the hot path of SURVEY.md section 8 lives here:
``csrc/`` (gfx950 HIP kernels + RCCL communicator + the C ABI of include/hbk.h) and the
host-side mirror of the reference's Python interface for that path."""
from hybridbackend_amd import data
from hybridbackend_amd import distribute
from hybridbackend_amd import embedding
from hybridbackend_amd import feature_column
from hybridbackend_amd import training
from hybridbackend_amd._lib import HbkError
from hybridbackend_amd._lib import InvalidArgumentError

__version__ = '0.1.0'
