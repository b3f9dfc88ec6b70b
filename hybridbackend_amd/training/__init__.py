"""Checkpoints of sharded embedding tables
(host mirror of ``hybridbackend/tensorflow/training/saver.py``)."""
from hybridbackend_amd.training.saver import Saver
from hybridbackend_amd.training.saver import ShardedSlice
from hybridbackend_amd.training.saver import load_full
from hybridbackend_amd.training.saver import export_reference
