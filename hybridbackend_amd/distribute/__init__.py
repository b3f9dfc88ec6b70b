"""Collective / partition ops of the sharded-embedding path
(host mirror of ``hybridbackend/tensorflow/distribute``)."""
from hybridbackend_amd.distribute.partition import partition_by_dual_modulo_n
from hybridbackend_amd.distribute.partition import partition_by_dual_modulo_stage_one
from hybridbackend_amd.distribute.partition import partition_by_dual_modulo_stage_two
from hybridbackend_amd.distribute.partition import partition_by_modulo
from hybridbackend_amd.distribute.partition import partition_by_modulo_n
